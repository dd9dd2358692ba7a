// ds_read_b64_tr_b16 recipe check for the "one rel-PE image" plan:
// a 32-edge tile of the rel-PE rows sits in LDS ROW-MAJOR [edge][feature] (what the score pass wants: a lane's 8
// consecutive features of one edge are 16 contiguous bytes); the aggregation pass needs it as the MFMA B operand with
// k = EDGE, i.e. 8 consecutive edges of one feature per lane.  Two transposed LDS reads per fragment give exactly that:
// lane i of a 16-lane group points at 4 contiguous halfs (row i>>2, column quad i&3) of a [4 edges][16 features] block and
// receives column i of the block.  Checks C = P[16 x 32] * R[32 x 96] against the host and times plain vs transposed reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
constexpr int RS = 104;   // LDS row stride in halfs (96 features + pad; 208 B rows keep every 8-byte read aligned)

__global__ void k_check(const _Float16* __restrict__ P, const _Float16* __restrict__ R, float* __restrict__ C, int iters, long long* cyc) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[32 * RS];
  const int l = threadIdx.x;
  for (int i = l; i < 32 * 96; i += 64) lds[(i / 96) * RS + (i % 96)] = R[i];
  __syncthreads();
  const int m = l & 15, kq = l >> 4;
  half8 a;
  for (int j = 0; j < 8; ++j) a[j] = P[m * 32 + kq * 8 + j];
  floatx4 acc[6];
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int cb = 0; cb < 6; ++cb) {
      // block rows kq*8 + {0..3} and {4..7}; this lane points at row (l&15)>>2 of the block, column quad l&3
      const _Float16* p0 = lds + (kq * 8 + ((l & 15) >> 2)) * RS + cb * 16 + (l & 3) * 4;
      fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)p0);
      fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(p0 + 4 * RS));
      half8 b;
      for (int j = 0; j < 4; ++j) { b[j] = (_Float16)lo[j]; b[4 + j] = (_Float16)hi[j]; }
      floatx4 z = {0.f, 0.f, 0.f, 0.f};
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, it == 0 ? z : acc[cb], 0, 0, 0);
    }
  }
  long long t1 = clock64();
  if (l == 0) cyc[0] = t1 - t0;
  for (int cb = 0; cb < 6; ++cb)
    for (int j = 0; j < 4; ++j) C[(size_t)((l >> 4) * 4 + j) * 96 + cb * 16 + (l & 15)] = acc[cb][j];
}

int main() {
  std::vector<_Float16> P(16 * 32), R(32 * 96);
  srand(1);
  for (auto& v : P) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
  for (auto& v : R) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
  _Float16 *dP, *dR; float* dC; long long* dcyc;
  hipMalloc(&dP, P.size() * 2); hipMalloc(&dR, R.size() * 2); hipMalloc(&dC, 16 * 96 * 4); hipMalloc(&dcyc, 8);
  hipMemcpy(dP, P.data(), P.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dR, R.data(), R.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dP, dR, dC, 1, dcyc);
  std::vector<float> C(16 * 96);
  hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int h = 0; h < 16; ++h)
    for (int f = 0; f < 96; ++f) {
      double s = 0;
      for (int e = 0; e < 32; ++e) s += (double)(float)P[h * 32 + e] * (double)(float)R[e * 96 + f];
      worst = fmax(worst, fabs(s - C[h * 96 + f]));
    }
  printf("transposed-read B fragments: max |C - P*R| = %.3e (%s)\n", worst, worst < 1e-3 ? "OK" : "WRONG");
  hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dP, dR, dC, 1000, dcyc);
  long long cyc; hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
  printf("one wave: %.1f cycles per 32-edge tile (12 transposed reads + 6 MFMAs; row stride %d halfs)\n", cyc / 1000.0, RS);
  return worst < 1e-3 ? 0 : 1;
}
