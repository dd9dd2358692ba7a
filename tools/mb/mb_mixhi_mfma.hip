// Round 6: how many wait states does gfx950 want between v_fma_mixhi_f16 (a write of the HIGH half of a register, op_sel:[1,0,0]) and
// a v_mfma_f32_16x16x16_f16 that reads the register as its A operand?  Variants of the gap: nothing | s_waitcnt (already satisfied) | s_nop 0 | s_nop 1 |
// one independent VALU instruction.  A = (1 | 2 | 0 | 0) per lane after the writes (the register held 0 before), B = ones: every element of the result
// must be 12 (4 with a stale high half).  All waves of a workgroup run it at once (two per SIMD, like k_chain16), mismatching ELEMENTS counted per variant.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/mb_mixhi_mfma.hip -o /tmp/mb_mixhi_mfma && /tmp/mb_mixhi_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define PROBE(GAP, IDX)                                                                                                   \
  {                                                                                                                       \
    float r0, r1, r2, r3;                                                                                                 \
    asm volatile(                                                                                                         \
        "v_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\tv_mov_b32 v52, 0x3c003c00\n\tv_mov_b32 v53, 0x3c003c00\n\t"              \
        "v_mov_b32 v54, 0x3c00\n\tv_mov_b32 v55, 0x4000\n\tv_mov_b32 v56, 0\n\tv_mov_b32 v57, 0\n\tv_mov_b32 v58, 0\n\tv_mov_b32 v59, 0\n\t" \
        "s_nop 7\n\t"                                                                                                     \
        "v_fma_mixlo_f16 v50, v54, 1.0, 0 op_sel_hi:[1,0,0]\n\t"                                                          \
        "s_nop 3\n\t"                                                                                                     \
        "v_fma_mixhi_f16 v50, v55, 1.0, 0 op_sel_hi:[1,0,0]\n\t" GAP                                                      \
        "v_mfma_f32_16x16x16_f16 v[56:59], v[50:51], v[52:53], v[56:59]\n\t"                                             \
        "s_nop 7\n\ts_nop 7\n\t"                                                                                          \
        "v_mov_b32 %0, v56\n\tv_mov_b32 %1, v57\n\tv_mov_b32 %2, v58\n\tv_mov_b32 %3, v59\n\t"                            \
        : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : : "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "memory"); \
    bad[IDX] += (r0 != 12.f) + (r1 != 12.f) + (r2 != 12.f) + (r3 != 12.f);                                                \
  }
// a FULL-register write (v_mov_b32) right in front of the MFMA, for comparison
#define PROBE_F(GAP, IDX)                                                                                                 \
  {                                                                                                                       \
    float r0, r1, r2, r3;                                                                                                 \
    asm volatile(                                                                                                         \
        "v_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\tv_mov_b32 v52, 0x3c003c00\n\tv_mov_b32 v53, 0x3c003c00\n\t"              \
        "v_mov_b32 v54, 0x40003c00\n\tv_mov_b32 v56, 0\n\tv_mov_b32 v57, 0\n\tv_mov_b32 v58, 0\n\tv_mov_b32 v59, 0\n\ts_nop 7\n\t" \
        "v_mov_b32 v50, v54\n\t" GAP                                                                                       \
        "v_mfma_f32_16x16x16_f16 v[56:59], v[50:51], v[52:53], v[56:59]\n\t"                                             \
        "s_nop 7\n\ts_nop 7\n\t"                                                                                          \
        "v_mov_b32 %0, v56\n\tv_mov_b32 %1, v57\n\tv_mov_b32 %2, v58\n\tv_mov_b32 %3, v59\n\t"                            \
        : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : : "v50", "v51", "v52", "v53", "v54", "v56", "v57", "v58", "v59", "memory"); \
    bad[IDX] += (r0 != 12.f) + (r1 != 12.f) + (r2 != 12.f) + (r3 != 12.f);                                                \
  }
// the same write read by a plain VALU instruction (v_mov_b32) or stored to LDS (ds_write_b32 + read back): 0x40003c00 = (2.0 | 1.0) expected
#define PROBE_V(GAP, IDX)                                                                                                 \
  {                                                                                                                       \
    unsigned r;                                                                                                           \
    asm volatile("v_mov_b32 v50, 0\n\tv_mov_b32 v54, 0x3c00\n\tv_mov_b32 v55, 0x4000\n\ts_nop 7\n\t"                        \
                 "v_fma_mixlo_f16 v50, v54, 1.0, 0 op_sel_hi:[1,0,0]\n\ts_nop 3\n\t"                                      \
                 "v_fma_mixhi_f16 v50, v55, 1.0, 0 op_sel_hi:[1,0,0]\n\t" GAP "v_mov_b32 %0, v50\n\t"                      \
                 : "=v"(r) : : "v50", "v54", "v55", "memory");                                                            \
    bad[IDX] += r != 0x40003c00u;                                                                                         \
  }
#define PROBE_D(GAP, IDX)                                                                                                 \
  {                                                                                                                       \
    unsigned r;                                                                                                           \
    asm volatile("v_mov_b32 v50, 0\n\tv_mov_b32 v54, 0x3c00\n\tv_mov_b32 v55, 0x4000\n\ts_nop 7\n\t"                        \
                 "v_fma_mixlo_f16 v50, v54, 1.0, 0 op_sel_hi:[1,0,0]\n\ts_nop 3\n\t"                                      \
                 "v_fma_mixhi_f16 v50, v55, 1.0, 0 op_sel_hi:[1,0,0]\n\t" GAP "ds_write_b32 %1, v50\n\ts_waitcnt lgkmcnt(0)\n\t" \
                 "ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\t"                                                         \
                 : "=&v"(r) : "v"(ldsa) : "v50", "v54", "v55", "memory");                                                 \
    bad[IDX] += r != 0x40003c00u;                                                                                         \
  }
__global__ __launch_bounds__(512) void k(unsigned long long* cnt, int iters) {
  __shared__ unsigned slot[512];
  const unsigned ldsa = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)(slot + threadIdx.x);
  unsigned long long bad[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    PROBE("", 0)
    PROBE("s_waitcnt lgkmcnt(0)\n\t", 1)
    PROBE("s_nop 0\n\t", 2)
    PROBE("s_nop 1\n\t", 3)
    PROBE("v_mov_b32 v51, 0\n\t", 4)
    PROBE_V("", 5)
    PROBE_V("s_nop 0\n\t", 6)
    PROBE_D("", 7)
    PROBE_D("s_nop 0\n\t", 8)
    PROBE_F("", 9)
    PROBE_F("s_nop 0\n\t", 10)
  }
  for (int j = 0; j < 11; ++j) if (bad[j]) atomicAdd(cnt + j, bad[j]);
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  unsigned long long *cnt, h[11];
  CHK(hipMalloc(&cnt, sizeof(h))); CHK(hipMemset(cnt, 0, sizeof(h)));
  hipLaunchKernelGGL(k, dim3(1024), dim3(512), 0, 0, cnt, iters);
  CHK(hipDeviceSynchronize());
  CHK(hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost));
  const char* names[11] = {"mfma: nothing", "mfma: s_waitcnt (satisfied)", "mfma: s_nop 0", "mfma: s_nop 1", "mfma: one independent VALU", "v_mov: nothing", "v_mov: s_nop 0", "ds_write: nothing", "ds_write: s_nop 0", "FULL write (v_mov) -> mfma: nothing", "FULL write -> mfma: s_nop 0"};
  printf("v_fma_mixhi_f16 v50 -> [gap] -> v_mfma_f32_16x16x16_f16 A = v[50:51]; %lld result elements per variant\n", 4ll * 1024 * 512 * iters);
  for (int j = 0; j < 11; ++j) printf("   consumer and gap %-38s wrong %llu\n", names[j], h[j]);
  return 0;
}
