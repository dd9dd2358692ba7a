// Round 6: the ONE instruction form with an op_sel bit that the legalised library still carries -- ps_chain16.h's f16_lo_pk / f16_sel_pk:
//   v_fma_mixlo_f16 d, hi_pk, m, y0 op_sel_hi:[1,0,0]      v_fma_mixhi_f16 d, hi_pk, m, y1 op_sel:[1,0,0] op_sel_hi:[1,0,0]
// (op_sel there picks the HIGH 16 bits of ONE register, not the second register of a pair).  Does it hold beside another kernel's
// v_mfma_f32_16x16x32_f16, the load that breaks v_pk_*_f32 with op_sel (tools/mb/mb_pksgpr3.hip)?  Reference: the same values through
// v_cvt_f32_f16 / v_sub_f32 / v_cvt_f16_f32 on shifted copies.  Counts by lane quarter.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/mb_mixsel.hip -o tools/mb/mb_mixsel && tools/mb/mb_mixsel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_probe(int iters, float m, unsigned long long* cnt) {
  const int lane = threadIdx.x & 63;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    float y0 = 0.001f * (float)(lane + 1) + 0.0371f * (float)(it & 15), y1 = -0.0017f * (float)(lane + 3) + 0.0213f * (float)(it & 31);
    asm volatile("" : "+v"(y0), "+v"(y1));
    const half2v h = {(_Float16)y0, (_Float16)y1};
    unsigned hi_pk = __builtin_bit_cast(unsigned, h), t;
    asm volatile("" : "+v"(hi_pk));
    asm volatile("v_fma_mixlo_f16 %0, %1, %3, %2 op_sel_hi:[1,0,0]" : "=v"(t) : "v"(hi_pk), "v"(y0), "v"(m));
    asm volatile("v_fma_mixhi_f16 %0, %1, %3, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(t) : "v"(hi_pk), "v"(y1), "v"(m));
    // reference: fp16(float(h) * m + y) by scalar conversions (the product and the sum are exact in fp32 for m = -1 | -0)
    float f0, f1;
    asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(f0) : "v"(hi_pk));
    asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(f1) : "v"(hi_pk >> 16));
    volatile float e0 = f0 * m + y0, e1 = f1 * m + y1;
    const half2v eh = {(_Float16)(float)e0, (_Float16)(float)e1};
    bad += t != __builtin_bit_cast(unsigned, eh);
  }
  if (bad) atomicAdd(cnt + (lane >> 4), (unsigned long long)bad);
}
__global__ __launch_bounds__(256) void k_load(int iters, float* sink) {
  half8 x, y;
  for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(0.01f * (threadIdx.x + j)); y[j] = (_Float16)(0.02f * j); }
  floatx4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc, 0, 0, 0);
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 40000;
  hipStream_t sp, sl;
  CHK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
  CHK(hipStreamCreateWithFlags(&sl, hipStreamNonBlocking));
  unsigned long long *cnt, h[4];
  float* sink;
  CHK(hipMalloc(&cnt, 32)); CHK(hipMalloc(&sink, 256));
  for (int L = 0; L < 2; ++L)
    for (int mi = 0; mi < 2; ++mi) {
      CHK(hipMemset(cnt, 0, 32));
      CHK(hipDeviceSynchronize());
      if (L) hipLaunchKernelGGL(k_load, dim3(1024), dim3(256), 0, sl, 12000000, sink);
      for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(k_probe, dim3(512), dim3(256), 0, sp, iters, mi ? -1.f : -0.f, cnt);
      CHK(hipStreamSynchronize(sp));
      const bool still = L && hipStreamQuery(sl) == hipErrorNotReady;
      CHK(hipDeviceSynchronize());
      CHK(hipMemcpy(h, cnt, 32, hipMemcpyDeviceToHost));
      printf("load %-18s m = %-4s: mismatches by lane quarter [%llu %llu %llu %llu]%s\n", L ? "MFMA 16x16x32 f16" : "none", mi ? "-1" : "-0", h[0], h[1], h[2], h[3],
             L ? (still ? "  (load still running when the probes ended)" : "  (LOAD ENDED EARLY)") : "");
    }
  return 0;
}
