// Round 6: v_bitop3_b32 (new in gfx950) in the forms hipcc emits for the LDS swizzles of k_chain16's edge phase -- among them two inline constants in
// src1 / src2, which appear when the staging area's base is not a compile-time constant -- against the truth table, alone and BESIDE another kernel's
// v_mfma_f32_16x16x32_f16 (the load under which packed-fp32 instructions with an op_sel bit return wrong low halves in lanes 48-63: mb_pksgpr3.hip).
// Counts by lane quarter.   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/mb_bitop3.hip -o /tmp/mb_bitop3 && /tmp/mb_bitop3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned tt(unsigned t, unsigned a, unsigned b, unsigned c) {   // bit i of the result = table[(a_i << 2) | (b_i << 1) | c_i]
  unsigned r = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const unsigned idx = (((a >> i) & 1) << 2) | (((b >> i) & 1) << 1) | ((c >> i) & 1);
    r |= ((t >> idx) & 1u) << i;
  }
  return r;
}
constexpr int NF = 8;
__global__ __launch_bounds__(256) void k_probe(int iters, unsigned long long* cnt) {
  const int lane = threadIdx.x & 63;
  unsigned bad[NF] = {};
  for (int it = 0; it < iters; ++it) {
    unsigned a = (unsigned)(threadIdx.x + 977 * it) * 2654435761u, b = (unsigned)(lane * 40503 + it) * 2246822519u, r[NF];
    asm volatile("" : "+v"(a), "+v"(b));
    asm volatile("v_bitop3_b32 %0, %1, 4, 12 bitop3:0x6c" : "=v"(r[0]) : "v"(a));
    asm volatile("v_bitop3_b32 %0, %1, 8, 12 bitop3:0x6c" : "=v"(r[1]) : "v"(a));
    asm volatile("v_bitop3_b32 %0, %1, 12, %1 bitop3:0xc" : "=v"(r[2]) : "v"(a));
    asm volatile("v_bitop3_b32 %0, %1, %2, 4 bitop3:0x36" : "=v"(r[3]) : "v"(a), "v"(b));
    asm volatile("v_bitop3_b32 %0, %1, %2, 3 bitop3:0x78" : "=v"(r[4]) : "v"(a), "v"(b));
    asm volatile("v_bitop3_b32 %0, %1, %2, 15 bitop3:0x78" : "=v"(r[5]) : "v"(a), "v"(b));
    asm volatile("v_bitop3_b32 %0, %1, 16, %2 bitop3:0x36" : "=v"(r[6]) : "v"(a), "v"(b));
    asm volatile("v_bitop3_b32 %0, %1, %2, %1 bitop3:0x96" : "=v"(r[7]) : "v"(a), "v"(b));
    const unsigned e[NF] = {tt(0x6c, a, 4, 12), tt(0x6c, a, 8, 12), tt(0x0c, a, 12, a), tt(0x36, a, b, 4), tt(0x78, a, b, 3), tt(0x78, a, b, 15), tt(0x36, a, 16, b), tt(0x96, a, b, a)};
#pragma unroll
    for (int j = 0; j < NF; ++j) bad[j] += r[j] != e[j];
  }
#pragma unroll
  for (int j = 0; j < NF; ++j) if (bad[j]) atomicAdd(cnt + 4 * j + (lane >> 4), (unsigned long long)bad[j]);
}
__global__ __launch_bounds__(256) void k_load(int iters, float* sink) {
  half8 x, y;
  for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(0.01f * (threadIdx.x + j)); y[j] = (_Float16)(0.02f * j); }
  floatx4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc, 0, 0, 0);
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  hipStream_t sp, sl;
  CHK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
  CHK(hipStreamCreateWithFlags(&sl, hipStreamNonBlocking));
  unsigned long long *cnt, h[4 * NF];
  float* sink;
  CHK(hipMalloc(&cnt, sizeof(h))); CHK(hipMalloc(&sink, 256));
  const char* names[NF] = {"a,4,12:0x6c", "a,8,12:0x6c", "a,12,a:0x0c", "a,b,4:0x36", "a,b,3:0x78", "a,b,15:0x78", "a,16,b:0x36", "a,b,a:0x96"};
  for (int L = 0; L < 2; ++L) {
    CHK(hipMemset(cnt, 0, sizeof(h)));
    CHK(hipDeviceSynchronize());
    if (L) hipLaunchKernelGGL(k_load, dim3(1024), dim3(256), 0, sl, 12000000, sink);
    for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(k_probe, dim3(512), dim3(256), 0, sp, iters, cnt);
    CHK(hipStreamSynchronize(sp));
    const bool still = L && hipStreamQuery(sl) == hipErrorNotReady;
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost));
    printf("load %s%s: %lld evaluations per form\n", L ? "MFMA 16x16x32 f16 on every CU" : "none", L ? (still ? " (still running when the probes ended)" : " (ENDED EARLY)") : "", 4ll * 512 * 256 * iters);
    for (int j = 0; j < NF; ++j) printf("   %-12s mismatches by lane quarter [%llu %llu %llu %llu]\n", names[j], h[4 * j], h[4 * j + 1], h[4 * j + 2], h[4 * j + 3]);
  }
  return 0;
}
