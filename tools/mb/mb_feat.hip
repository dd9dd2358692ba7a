// Bit check of the packed Fourier-row arithmetic of the edge phase (ps_chain16.h: feat8, f16_sel_pk) against the scalar form it
// replaced in round 5 (x / d by reciprocal + correction, reduction to revolutions, v_sin / v_cos, normalise, hi = fp16(y),
// lo = fp16(y - float(hi))), and of the lane-selected halves against (lo ? f16_lo : f16_hi).  Random arguments over the ranges
// the rel-PE rows produce.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Iprosim_amd/csrc tools/mb/mb_feat.hip -o tools/mb/mb_feat
#include "ps_chain16.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
using namespace ps;

__device__ __noinline__ void feat8_scalar(float xs, float rstd, float nmr, const float* dv, const float* rdv, unsigned short* hi, unsigned short* lo) {
  for (int j = 0; j < 4; ++j) {
    float s, c;
    sincos_hw(fdiv16(xs, dv[j], rdv[j]), s, c);
    volatile float ys = fmaf(s, rstd, nmr), yc = fmaf(c, rstd, nmr);   // volatile: the fp32 value exists before the conversion
    const float a = ys, b = yc;
    const _Float16 h0 = (_Float16)a, h1 = (_Float16)b;
    volatile float d0 = a - (float)h0, d1 = b - (float)h1;
    const _Float16 l0 = (_Float16)(float)d0, l1 = (_Float16)(float)d1;
    hi[2 * j] = __builtin_bit_cast(unsigned short, h0); hi[2 * j + 1] = __builtin_bit_cast(unsigned short, h1);
    lo[2 * j] = __builtin_bit_cast(unsigned short, l0); lo[2 * j + 1] = __builtin_bit_cast(unsigned short, l1);
  }
}

__global__ void k_check(const float* xs, const float* rstd, const float* nmr, const float* div32, int n, unsigned long long* bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int kq = i & 3;
  float dv[4], rdv[4];
  for (int j = 0; j < 4; ++j) { dv[j] = div32[2 * (4 * kq + j)]; rdv[j] = 1.0f / dv[j]; }
  const f32x2 dvp[2] = {{dv[0], dv[1]}, {dv[2], dv[3]}}, rdvp[2] = {{rdv[0], rdv[1]}, {rdv[2], rdv[3]}};
  half8 h, l;
  feat8(xs[i], rstd[i], nmr[i], dvp, rdvp, h, l);
  unsigned short rh[8], rl[8];
  feat8_scalar(xs[i], rstd[i], nmr[i], dv, rdv, rh, rl);
  int nb = 0;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 hu = __builtin_bit_cast(u32x4, h), lu = __builtin_bit_cast(u32x4, l);
  const unsigned hw[4] = {hu.x, hu.y, hu.z, hu.w}, lw[4] = {lu.x, lu.y, lu.z, lu.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const unsigned short gh = (unsigned short)(hw[j >> 1] >> (16 * (j & 1))), gl = (unsigned short)(lw[j >> 1] >> (16 * (j & 1)));
    if ((gh != rh[j] || gl != rl[j]) && atomicAdd(bad + 1, 1ull) < 24) printf("lane %d j %d: hi got %04x want %04x | lo got %04x want %04x\n", i, j, gh, rh[j], gl, rl[j]);
    nb += gh != rh[j];
    nb += gl != rl[j];
  }
  // lane-selected halves (probabilities in [0, 1], q values of either sign)
  volatile float p0v = fabsf(nmr[i]) * 0.3f, p1v = xs[i] * 1e-3f;   // (volatile: hipcc folds fp16(a * b) into ONE v_fma_mixlo_f16 -- a single rounding, not the value of the fp32 product)
  const float p0 = p0v, p1 = p1v;
  for (int sel = 0; sel < 2; ++sel) {
    const unsigned r = f16_sel_pk(f16_hi_pk(p0, p1), p0, p1, sel ? -1.f : -0.f);
    volatile float e0 = p0 - (float)(_Float16)p0, e1 = p1 - (float)(_Float16)p1;
    const _Float16 w0 = sel ? (_Float16)(float)e0 : (_Float16)p0, w1 = sel ? (_Float16)(float)e1 : (_Float16)p1;
    const unsigned want = (unsigned)__builtin_bit_cast(unsigned short, w0) | ((unsigned)__builtin_bit_cast(unsigned short, w1) << 16);
    if (r != want && atomicAdd(bad + 2, 1ull) < 12) printf("lane %d sel %d: got %08x want %08x (p0 %.9g p1 %.9g)\n", i, sel, r, want, p0, p1);
    nb += (r != want) && !(p1 == 0.f);
  }
  if (nb) atomicAdd(bad, (unsigned long long)nb);
}

// the VALU xor butterflies of ps_device.h (round 5) against the __shfl_xor forms they replace, bit for bit
__global__ void k_check_xor(const float* x, int n, unsigned long long* bad) {   // bad[4 + k]: mismatches of check k
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const float v = x[i % n] * (1.f + 0.37f * (threadIdx.x & 63));
  auto chk = [&](int k, float a, float b) { if (__float_as_uint(a) != __float_as_uint(b)) atomicAdd(bad + 4 + k, 1ull); };
  chk(0, xor_add<32>(v), v + __shfl_xor(v, 32)); chk(1, xor_add<16>(v), v + __shfl_xor(v, 16)); chk(2, xor_add<8>(v), v + __shfl_xor(v, 8));
  chk(3, xor_add<4>(v), v + __shfl_xor(v, 4)); chk(4, xor_add<2>(v), v + __shfl_xor(v, 2)); chk(5, xor_add<1>(v), v + __shfl_xor(v, 1));
  chk(6, xor_max<32>(v), fmaxf(v, __shfl_xor(v, 32))); chk(7, xor_max<16>(v), fmaxf(v, __shfl_xor(v, 16)));
  float s = v, m = v;
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); m = fmaxf(m, __shfl_xor(m, o)); }
  chk(8, wave_sum(v), s); chk(9, wave_max(v), m);
  float acc[1][4] = {{v, -v * 0.5f, v * v, 1.f / (1.f + fabsf(v))}}, ref[4];
  for (int j = 0; j < 4; ++j) { ref[j] = acc[0][j]; for (int o = 8; o < 64; o <<= 1) ref[j] += __shfl_xor(ref[j], o); }
  fold_kgroups<1, 8>(acc);
  for (int j = 0; j < 4; ++j) chk(10, acc[0][j], ref[j]);
  float acc4[1][4] = {{v, v + 1.f, v - 2.f, 3.f * v}}, ref4[4];
  for (int j = 0; j < 4; ++j) { ref4[j] = acc4[0][j]; for (int o = 4; o < 64; o <<= 1) ref4[j] += __shfl_xor(ref4[j], o); }
  fold_kgroups<1, 4>(acc4);
  for (int j = 0; j < 4; ++j) chk(11, acc4[0][j], ref4[j]);
}

int main() {
  const int n = 1 << 22;
  float *hx = (float*)malloc(4 * n), *hr = (float*)malloc(4 * n), *hn = (float*)malloc(4 * n), hd[32];
  for (int k = 0; k < 32; ++k) hd[k] = powf(10000.f, (float)(2 * (k / 2)) / 32.f);
  srand(5);
  for (int i = 0; i < n; ++i) {
    const double u = rand() / (double)RAND_MAX, v = rand() / (double)RAND_MAX, w = rand() / (double)RAND_MAX;
    const int kind = i % 3;   // 2 pi x (distance up to 300 m | angle in [-pi, pi) | tiny values)
    hx[i] = (float)(6.283185307179586 * (kind == 0 ? 300.0 * u : kind == 1 ? (2 * u - 1) * 3.14159265 : (u - 0.5) * 1e-3));
    hr[i] = (float)(1.2 + 0.6 * v);
    hn[i] = (float)(-0.8 + 1.2 * w);
  }
  float *dx, *dr, *dn, *dd; unsigned long long *db, hb = 0;
  hipMalloc(&dx, 4 * n); hipMalloc(&dr, 4 * n); hipMalloc(&dn, 4 * n); hipMalloc(&dd, 128); hipMalloc(&db, 256);
  hipMemcpy(dx, hx, 4 * n, hipMemcpyHostToDevice); hipMemcpy(dr, hr, 4 * n, hipMemcpyHostToDevice); hipMemcpy(dn, hn, 4 * n, hipMemcpyHostToDevice);
  hipMemcpy(dd, hd, 128, hipMemcpyHostToDevice); hipMemset(db, 0, 256);
  hipLaunchKernelGGL(k_check, dim3(n / 256), dim3(256), 0, 0, dx, dr, dn, dd, n, db);
  hipDeviceSynchronize();
  hipMemcpy(&hb, db, 8, hipMemcpyDeviceToHost);
  printf("mb_feat: %d lanes x (16 feature halves + 2 selected pairs), mismatching values: %llu\n", n, hb);
  hipLaunchKernelGGL(k_check_xor, dim3(4096), dim3(256), 0, 0, dx, n, db);
  hipDeviceSynchronize();
  unsigned long long h4[32] = {0};
  hipMemcpy(h4, db, 256, hipMemcpyDeviceToHost);
  unsigned long long tot = 0;
  static const char* nm[12] = {"xor_add<32>", "xor_add<16>", "xor_add<8>", "xor_add<4>", "xor_add<2>", "xor_add<1>", "xor_max<32>", "xor_max<16>", "wave_sum", "wave_max", "fold_kgroups<1,8>", "fold_kgroups<1,4>"};
  printf("mb_feat: xor butterflies against __shfl_xor, mismatching values:");
  for (int k = 0; k < 12; ++k) { printf(" %s %llu", nm[k], h4[4 + k]); tot += h4[4 + k]; }
  printf("\n");
  h4[3] = tot;
  return hb != 0 || h4[3] != 0;
}
