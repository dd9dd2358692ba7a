// Is a packed fp32 op (v_pk_fma_f32 / v_pk_mul_f32) one issue slot or two on gfx950?  Each wave runs a long chain-free stream of
// either 2N v_fma_f32 or N v_pk_fma_f32 (the same FLOPs) with 8 independent accumulator pairs; cycles per instruction by s_memtime.
// hipcc --offload-arch=gfx950 -O3 tools/mb/mb_pkfma.hip -o tools/mb/mb_pkfma && tools/mb/mb_pkfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters, long long* cyc) {
  v2f a[8];
  for (int i = 0; i < 8; ++i) a[i] = v2f{(float)threadIdx.x + i, 1.0f + i};
  const v2f b = {1.0000001f, 0.9999999f}, c = {1e-7f, -1e-7f};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) {
        asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %4, %5" : "+v"(a[i].x), "+v"(a[i].y) : "v"(b.x), "v"(c.x), "v"(b.y), "v"(c.y));
      } else if (MODE == 1) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      } else {
        asm volatile("v_sin_f32 %0, %0\n\tv_cos_f32 %1, %1" : "+v"(a[i].x), "+v"(a[i].y));
      }
    }
  }
  const long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* o; long long* c; hipMalloc(&o, 4 * 1024 * 1024); hipMalloc(&c, 8);
  const int iters = 4096;
  for (int waves = 1; waves <= 8; waves *= 2) {
    long long h[3];
    for (int m = 0; m < 3; ++m) {
      if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * 4 * waves), 0, 0, o, iters, c);
      if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * 4 * waves), 0, 0, o, iters, c);
      if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * 4 * waves), 0, 0, o, iters, c);
      hipDeviceSynchronize();
      hipMemcpy(&h[m], c, 8, hipMemcpyDeviceToHost);
    }
    const double n = (double)iters * 8;
    printf("%d wave(s) per SIMD: per wave, cycles per PAIR of results: 2 x v_fma_f32 %.2f | 1 x v_pk_fma_f32 %.2f | v_sin + v_cos %.2f\n", waves,
           h[0] / n, h[1] / n, h[2] / n);
  }
  return 0;
}
