// Row-tile kernels (round 4): the dense per-row stacks of the path -- PointNet polyline encoders, the node half of a split
// attention layer, the k | v projection -- as CHAINED matrix-core GEMMs whose activations never leave one wave's registers.
//
// Why.  Round 3's k_pointnet_mfma / k_node ran every Linear as "GEMM -> fp32 tile to LDS -> barrier -> epilogue by 256
// threads -> split planes to LDS -> barrier": 64 (16) rows per 4-wave workgroup, 4 % of the dense f16 peak, bound by that
// dependent stage chain.  Here a wave owns 16 * MT rows for the WHOLE
// stack and there is no stage chain at all:
//   * the GEMMs are computed TRANSPOSED, D^T[out feature][row] = W[out][k] . X^T[k][row]: the weight fragment is the MFMA's A
//     operand, the activations are its B operand (lane = row n + 16 kq holds 8 k values of row n);
//   * the C / D layout of v_mfma_f32_16x16x32_f16 then leaves lane (n, kq) with features 16 t + 4 kq + j (t = 16-feature tile,
//     j < 4) of ITS OWN row n -- which is already a legal B operand of the next GEMM if that GEMM's K index is read in the
//     order  k-block ks, element i  <->  feature 32 ks + 16 (i >> 2) + 4 kq + (i & 3)  (C tiles 2 ks and 2 ks + 1).  The host
//     packs the next layer's weight fragments with the same K permutation (Builder::fragments, perm = true), so a result
//     becomes an operand with NO data movement: bias / LayerNorm / ReLU / hi-lo split run on the accumulator registers
//     (a row's 128 features sit in its 4 kq lanes: LayerNorm sums are 32 in-lane adds + two permlane swaps);
//   * weights stream from L2 / L1 through a small register ring (every wave of a CU reads the same fragments);
//   * no __syncthreads anywhere: waves are independent, LDS is only a wave-private scratch for the max-pools.
// Split-fp16 operands as everywhere in this library (x = hi + lo; w.hi x.hi + w.lo x.hi + w.hi x.lo in fp32).
//
// Reference math: PointNetPolylineEncoder.forward (prosim/models/scene_encoder/pointnet_encoder.py:24-62),
// AttentionLayer.forward's per-row half (prosim/models/layers/attention_layer.py:56-79, :100-121).
#pragma once
#include "ps_kernels.h"
#include "ps_attn.h"

namespace ps {

constexpr int RT_DEPTH = 4;    // weight-fragment pairs (hi | lo) in flight per wave
constexpr int RT_PS = 132;     // row stride (floats) of the wave-private [16][128] LDS rows

// acc[t][mt] += W[16 t .. 16 t + 15][k] . X[k][16 mt + n]  for t < NT; F: [t][k-block of KT][hi|lo][lane 64][8] (Builder::fragments),
// k-blocks k0 .. k0 + K32 - 1 of it against the operand's blocks 0 .. K32 - 1.
template <int MT, int K32, int KT, int NT>
__device__ __forceinline__ void rt_gemm(floatx4 (&acc)[NT][MT], const half8 (&xh)[MT][4], const half8 (&xl)[MT][4],
                                        const _Float16* __restrict__ F, int k0, int lane) {
  constexpr int NG = NT * K32;
  half8 fh[RT_DEPTH], fl[RT_DEPTH];
  const _Float16* f0 = F + (size_t)k0 * 1024 + lane * 8;
#pragma unroll
  for (int d = 0; d < RT_DEPTH; ++d)
    if (d < NG) {
      const _Float16* f = f0 + (size_t)((d / K32) * KT + (d % K32)) * 1024;
      fh[d] = ldgh8(f);
      fl[d] = ldgh8(f + 512);
    }
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int s = g % RT_DEPTH;
    const half8 ah = fh[s], al = fl[s];
    if (g + RT_DEPTH < NG) {
      const int g2 = g + RT_DEPTH;
      const _Float16* f = f0 + (size_t)((g2 / K32) * KT + (g2 % K32)) * 1024;
      fh[s] = ldgh8(f);
      fl[s] = ldgh8(f + 512);
    }
    const int t = g / K32, ks = g % K32;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh[mt][ks], acc[t][mt], 0, 0, 0);
      acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh[mt][ks], acc[t][mt], 0, 0, 0);
      acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl[mt][ks], acc[t][mt], 0, 0, 0);
    }
  }
}
template <int MT, int NT>
__device__ __forceinline__ void rt_zero(floatx4 (&acc)[NT][MT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = floatx4{0.f, 0.f, 0.f, 0.f};
}
// the result tiles of this lane's rows as the next GEMM's operand (K in the permuted order of the header)
template <int MT>
__device__ __forceinline__ void rt_to_operand(const floatx4 (&a)[8][MT], half8 (&xh)[MT][4], half8 (&xl)[MT][4]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = a[2 * ks + (i >> 2)][mt][i & 3];
        xh[mt][ks][i] = f16_hi(v);
        xl[mt][ks][i] = f16_lo(v);
      }
}
// + bias[feature] (global, 128 floats)
template <int MT>
__device__ __forceinline__ void rt_bias(floatx4 (&a)[8][MT], const float* __restrict__ bias, int kq) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float4 b = ldg4(bias + 16 * t + 4 * kq);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { a[t][mt][0] += b.x; a[t][mt][1] += b.y; a[t][mt][2] += b.z; a[t][mt][3] += b.w; }
  }
}
// LayerNorm over the 128 features of every row (torch.nn.LayerNorm: biased variance, eps inside the sqrt; two passes like
// torch's), affine from global memory; a row's features live in this lane and its three kq partners
template <int MT>
__device__ __forceinline__ void rt_ln(floatx4 (&a)[8][MT], const float* __restrict__ w, const float* __restrict__ b, float eps, int kq) {
  float rstd[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float sm = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) sm += (a[t][mt][0] + a[t][mt][1]) + (a[t][mt][2] + a[t][mt][3]);
    const float mean = kq_sum(sm) * (1.f / 128.f);
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a[t][mt][j] -= mean;
        sq = fmaf(a[t][mt][j], a[t][mt][j], sq);
      }
    rstd[mt] = 1.f / sqrtf(kq_sum(sq) * (1.f / 128.f) + eps);
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float4 wv = ldg4(w + 16 * t + 4 * kq), bv = ldg4(b + 16 * t + 4 * kq);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      a[t][mt][0] = fmaf(a[t][mt][0] * rstd[mt], wv.x, bv.x);
      a[t][mt][1] = fmaf(a[t][mt][1] * rstd[mt], wv.y, bv.y);
      a[t][mt][2] = fmaf(a[t][mt][2] * rstd[mt], wv.z, bv.z);
      a[t][mt][3] = fmaf(a[t][mt][3] * rstd[mt], wv.w, bv.w);
    }
  }
}
template <int MT>
__device__ __forceinline__ void rt_relu(floatx4 (&a)[8][MT]) {
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) a[t][mt][j] = fmaxf(a[t][mt][j], 0.f);
}
// a [16][128] fp32 LDS row block as a one-tile operand (column n = row n of the block), K in the permuted order
__device__ __forceinline__ void rt_operand_from_lds(const float* __restrict__ rows, int n, int kq, int nlive, half8 (&xh)[1][4], half8 (&xl)[1][4]) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (n < nlive) {
      v0 = *reinterpret_cast<const float4*>(rows + n * RT_PS + 32 * ks + 4 * kq);
      v1 = *reinterpret_cast<const float4*>(rows + n * RT_PS + 32 * ks + 16 + 4 * kq);
    }
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      xh[0][ks][i] = f16_hi(v[i]);
      xl[0][ks][i] = f16_lo(v[i]);
    }
  }
}

// ---- max-pool over the points of each polyline (pointnet_encoder.py:47, :53: max over the zero-filled feature buffer, so a
// masked point counts as 0).  Per 16-feature tile the wave's rows go through a wave-private [16 features][rows] LDS stage
// (row stride RS = 16 MT + 4: the two kq groups of a 32-lane store land on disjoint banks), then lane (g, f) folds polyline
// g's P points of feature f.  LDS operations of one wave execute in order: no barrier between the store and the fold.
template <int MT>
__device__ __forceinline__ void rt_pool(const floatx4 (&a)[8][MT], const bool (&vld)[MT], float* __restrict__ S, float* __restrict__ pooled,
                                        int P, int G, int lane) {
  constexpr int RS = 16 * MT + 4;
  const int n = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) S[(4 * kq + j) * RS + 16 * mt + n] = vld[mt] ? a[t][mt][j] : 0.f;
    __builtin_amdgcn_wave_barrier();
    for (int gb = 0; gb < G; gb += 4) {
      const int g = gb + kq;
      if (g < G) {
        const float* s = S + n * RS + g * P;
        float m = s[0];
        for (int p = 1; p < P; ++p) m = fmaxf(m, s[p]);
        pooled[g * RT_PS + 16 * t + n] = m;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int MT>
constexpr size_t rt_pn_wave_bytes() { return ((size_t)16 * (16 * MT + 4) + 2 * 16 * RT_PS + 16) * 4; }

// PointNetPolylineEncoder on row tiles: a wave takes G = min(16, 16 MT / P) whole polylines (their points are its rows, in order).
template <int MT>
__global__ __launch_bounds__(256, (MT <= 2 ? 2 : 1)) void k_pointnet_rt(PointNetW w, const float* __restrict__ pts, const uint8_t* __restrict__ pmask,
                                                                       const int* __restrict__ rows, int n_rows, int P, int feat_mask_dim,
                                                                       float* __restrict__ out, float eps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rt_smem[];
  constexpr int RS = 16 * MT + 4;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* S = reinterpret_cast<float*>(rt_smem + (size_t)wave * rt_pn_wave_bytes<MT>());
  float* pooled = S + 16 * RS;            // [16][RT_PS]
  float* pbuf = pooled + 16 * RT_PS;      // [16][RT_PS] per-polyline bias of mlps[0] / scratch
  int* anyf = reinterpret_cast<int*>(pbuf + 16 * RT_PS);   // [16] polyline has a valid point
  const int G = min(16, (16 * MT) / P);
  const int g0 = (blockIdx.x * 4 + wave) * G;
  if (g0 >= n_rows) return;   // (no workgroup barrier in this kernel)
  const int n = lane & 15, kq = lane >> 4;
  const int Cin = w.in_dim;
  if (lane < 16) anyf[lane] = 0;
  bool vld[MT];
  int gl[MT];
  half8 xh[MT][4], xl[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int r = 16 * mt + n;
    const int g = r / P, p = r - g * P;
    const bool ok = g < G && g0 + g < n_rows;
    const int row = ok ? (rows ? rows[g0 + g] : g0 + g) : 0;
    bool v = ok;
    if (ok && pmask) {
      if (feat_mask_dim == 0) v = pmask[(size_t)row * P + p] != 0;
      else
        for (int f = 0; f < feat_mask_dim; ++f) v = v && pmask[((size_t)row * P + p) * feat_mask_dim + f] != 0;
    }
    vld[mt] = v;
    gl[mt] = min(g, 15);
    if (v) anyf[gl[mt]] = 1;   // (every writer stores the same value)
    const float* px = pts + ((size_t)row * P + p) * Cin;
#pragma unroll
    for (int j = 0; j < 8; ++j) {   // layer 0 reads K in the natural order: lane group kq holds features 8 kq .. 8 kq + 7 (K padded to 32)
      const int k = 8 * kq + j;
      const float xv = (v && k < Cin) ? px[k] : 0.f;
      xh[mt][0][j] = f16_hi(xv);
      xl[mt][0][j] = f16_lo(xv);
    }
  }
  floatx4 acc[8][MT];
  // ---- pre_mlps: Linear, LayerNorm, ReLU (the last one: Linear, ReLU)   (:33-38 through layers/mlp.py)
  rt_zero<MT, 8>(acc);
  rt_gemm<MT, 1, 1, 8>(acc, xh, xl, w.pre_F[0], 0, lane);
  rt_bias<MT>(acc, w.pre_b[0], kq);
  if (w.pre_lnw[0]) rt_ln<MT>(acc, w.pre_lnw[0], w.pre_lnb[0], eps, kq);
  rt_relu<MT>(acc);
  for (int l = 1; l < w.n_pre; ++l) {
    rt_to_operand<MT>(acc, xh, xl);
    rt_zero<MT, 8>(acc);
    rt_gemm<MT, 4, 4, 8>(acc, xh, xl, w.pre_Q[l], 0, lane);
    rt_bias<MT>(acc, w.pre_b[l], kq);
    if (w.pre_lnw[l]) rt_ln<MT>(acc, w.pre_lnw[l], w.pre_lnb[l], eps, kq);
    rt_relu<MT>(acc);
  }
  rt_pool<MT>(acc, vld, S, pooled, P, G, lane);
  // ---- mlps: layer 0 consumes cat(point feature, pooled): the pooled half is a per-polyline bias   (:48-50)
  for (int l = 0; l < w.n_mid; ++l) {
    rt_to_operand<MT>(acc, xh, xl);
    if (l == 0) {
      half8 ph[1][4], pl[1][4];
      rt_operand_from_lds(pooled, n, kq, G, ph, pl);
      floatx4 pacc[8][1];
      rt_zero<1, 8>(pacc);
      rt_gemm<1, 4, 4, 8>(pacc, ph, pl, w.mid_PQ, 0, lane);
      rt_bias<1>(pacc, w.mid_b[0], kq);
#pragma unroll
      for (int t = 0; t < 8; ++t)
        *reinterpret_cast<float4*>(pbuf + n * RT_PS + 16 * t + 4 * kq) = make_float4(pacc[t][0][0], pacc[t][0][1], pacc[t][0][2], pacc[t][0][3]);
      __builtin_amdgcn_wave_barrier();
    }
    rt_zero<MT, 8>(acc);
    rt_gemm<MT, 4, 4, 8>(acc, xh, xl, w.mid_Q[l], 0, lane);
    if (l == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float4 b = *reinterpret_cast<const float4*>(pbuf + gl[mt] * RT_PS + 16 * t + 4 * kq);
          acc[t][mt][0] += b.x; acc[t][mt][1] += b.y; acc[t][mt][2] += b.z; acc[t][mt][3] += b.w;
        }
    } else {
      rt_bias<MT>(acc, w.mid_b[l], kq);
    }
    if (w.mid_lnw[l]) rt_ln<MT>(acc, w.mid_lnw[l], w.mid_lnb[l], eps, kq);
    rt_relu<MT>(acc);
  }
  // ---- max-pool (:53), out_mlps (:57): Linear, ReLU, Linear on the pooled rows; a polyline without a valid point keeps a zero feature (:56-60)
  rt_pool<MT>(acc, vld, S, pooled, P, G, lane);
  {
    half8 ph[1][4], pl[1][4];
    rt_operand_from_lds(pooled, n, kq, G, ph, pl);
    floatx4 o[8][1];
    rt_zero<1, 8>(o);
    rt_gemm<1, 4, 4, 8>(o, ph, pl, w.out_Q0, 0, lane);
    rt_bias<1>(o, w.out_b0, kq);
    rt_relu<1>(o);
    rt_to_operand<1>(o, ph, pl);
    rt_zero<1, 8>(o);
    rt_gemm<1, 4, 4, 8>(o, ph, pl, w.out_Q1, 0, lane);
    rt_bias<1>(o, w.out_b1, kq);
    if (n < G && g0 + n < n_rows) {
      const bool any = anyf[n] != 0;
#pragma unroll
      for (int t = 0; t < 8; ++t)
        *reinterpret_cast<float4*>(out + (size_t)(g0 + n) * 128 + 16 * t + 4 * kq) =
            any ? make_float4(o[t][0][0], o[t][0][1], o[t][0][2], o[t][0][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

}  // namespace ps
