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

constexpr int RT_PS = 132;     // row stride (floats) of the wave-private [16][128] LDS rows

// Weight fragments are read through a raw buffer descriptor (SGPRs) with ONE per-lane byte offset (16 * lane) and the group's
// offset as the instruction's scalar operand: with flat 64-bit addresses hipcc materialised a VGPR pair per 4 KB window of
// every GEMM of the kernel and hoisted them all out of the layer loops (128 registers of addresses, spills at MT = 1).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rt_rsrc(const void* p) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ half8 rt_ldfrag(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// The ring of weight-fragment pairs a wave keeps in flight.  A GEMM consumes group g from slot g % DEPTH and requests group
// g + DEPTH into it; rt_prefetch() requests the first DEPTH groups of the NEXT GEMM as soon as a GEMM's last group is consumed, so
// the epilogue between two GEMMs (bias / LayerNorm / ReLU / split: a few hundred VALU instructions) covers their L2 round trip.
template <int DEPTH>
struct RtRing {
  half8 h[DEPTH], l[DEPTH];
};
template <int MT>
constexpr int rt_depth() { return MT == 1 ? 8 : (MT <= 3 ? 6 : 4); }
__device__ __forceinline__ unsigned rt_goff(int g, int K32, int KT, int k0) { return (unsigned)(((g / K32) * KT + (g % K32) + k0) * 2048); }
template <int DEPTH, int K32, int KT, int NT>
__device__ __forceinline__ void rt_prefetch(RtRing<DEPTH>& R, const _Float16* __restrict__ F, int k0, int lane) {
  const __amdgpu_buffer_rsrc_t rs = rt_rsrc(F);
  const unsigned voff = 16u * (unsigned)lane;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d < NT * K32) {
      const unsigned so = rt_goff(d, K32, KT, k0);
      R.h[d] = rt_ldfrag(rs, voff, so);
      R.l[d] = rt_ldfrag(rs, voff, so + 1024u);
    }
}
// acc[t][mt] += W[16 t .. 16 t + 15][k] . X[k][16 mt + n]  for t < NT; F: [t][k-block of KT][hi|lo][lane 64][8] (Builder::fragments),
// k-blocks k0 .. k0 + K32 - 1 of it against the operand's blocks 0 .. K32 - 1.  The ring must hold the GEMM's first groups
// (rt_prefetch with the same F / k0) on entry and is empty on return.
template <int MT, int K32, int KT, int NT, int DEPTH>
__device__ __forceinline__ void rt_gemm(RtRing<DEPTH>& R, floatx4 (&acc)[NT][MT], const half8 (&xh)[MT][4], const half8 (&xl)[MT][4],
                                        const _Float16* __restrict__ F, int k0, int lane) {
  constexpr int NG = NT * K32;
  const __amdgpu_buffer_rsrc_t rs = rt_rsrc(F);
  const unsigned voff = 16u * (unsigned)lane;
  floatx4 p1[MT], p2[MT];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int s = g % DEPTH;
    const half8 ah = R.h[s], al = R.l[s];
    if (g + DEPTH < NG) {
      const unsigned so = rt_goff(g + DEPTH, K32, KT, k0);
      R.h[s] = rt_ldfrag(rs, voff, so);
      R.l[s] = rt_ldfrag(rs, voff, so + 1024u);
    }
    // (left alone, the scheduler sinks every fragment load next to the MFMA that consumes it -- one exposed L2 round trip per
    // group: 10x the MFMA time; the barrier pins the request DEPTH groups ahead of its use)
    __builtin_amdgcn_sched_barrier(0);
    // A dependent MFMA waits ~45 cycles for its accumulator (measured: SQ_WAIT_INST_ANY of the first version, three products
    // back to back on one accumulator), an independent one issues every 16: the group's MFMAs go product-major over the
    // wave's row tiles, and with a single tile the three products use separate partial accumulators.
    const int t = g / K32, ks = g % K32;
    if (MT >= 2) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh[mt][ks], acc[t][mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh[mt][ks], acc[t][mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl[mt][ks], acc[t][mt], 0, 0, 0);
    } else {
      if (ks == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { p1[mt] = floatx4{0.f, 0.f, 0.f, 0.f}; p2[mt] = p1[mt]; }
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh[mt][ks], acc[t][mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) p1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh[mt][ks], p1[mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) p2[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl[mt][ks], p2[mt], 0, 0, 0);
      if (ks == K32 - 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] += p1[mt] + p2[mt];
      }
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
// accumulators that start from bias[feature] (global, 128 floats) instead of zero: the Linear's bias costs no instruction
template <int MT>
__device__ __forceinline__ void rt_bias(floatx4 (&a)[8][MT], const float* __restrict__ bias, int kq) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float4 b = ldg4(bias + 16 * t + 4 * kq);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[t][mt] = floatx4{b.x, b.y, b.z, b.w};
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

// ---- PointNet row layout.  A polyline's P points take L lanes (a power of two, consecutive lanes of the wave's 16 columns)
// x MT row tiles: point p sits in tile p / L, lane g L + p % L.  A wave then carries G = 16 / L polylines, and both max-pools
// (pointnet_encoder.py:47, :53: max over the zero-filled feature buffer, so a masked point counts as 0) are MT - 1 in-lane
// maxima + log2(L) DPP steps per value -- no LDS, and the pooled row comes out REPLICATED in the polyline's own lanes, which
// is exactly the operand column layout of the pooled-row GEMMs (mlps[0]'s pooled half, out_mlps): their results land in the
// lanes whose rows need them.
template <int L>
__device__ __forceinline__ float rt_group_max(float v) {
  if (L >= 2) v = fmaxf(v, dpp_xor1(v));
  if (L >= 4) v = fmaxf(v, dpp_xor2(v));
  if (L >= 8) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true)));    // row_half_mirror
  if (L >= 16) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true)));   // row_mirror
  return v;
}
template <int MT, int L>
__device__ __forceinline__ void rt_pool(const floatx4 (&a)[8][MT], const bool (&vld)[MT], floatx4 (&pooled)[8][1]) {
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float m = vld[0] ? a[t][0][j] : 0.f;
#pragma unroll
      for (int mt = 1; mt < MT; ++mt) m = fmaxf(m, vld[mt] ? a[t][mt][j] : 0.f);
      pooled[t][0][j] = rt_group_max<L>(m);
    }
}

// phase clocks (tools only: -DPS_RT_PROF): wave 0 of block 0 charges the cycles since the previous mark to slot i
#ifdef PS_RT_PROF
__device__ unsigned long long g_rt_prof[32];
#define RT_MARK(i)                                                                   \
  do {                                                                               \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                                       \
      const long long now_ = clock64();                                              \
      g_rt_prof[i] += (unsigned long long)(now_ - rt_t0);                            \
      rt_t0 = now_;                                                                  \
    }                                                                                \
  } while (0)
#else
#define RT_MARK(i) do { (void)rt_t0; } while (0)
#endif

// PointNetPolylineEncoder on row tiles: MT * L >= P slots per polyline, G = 16 / L polylines per wave, 4 waves per workgroup.
template <int MT, int L>
__global__ __launch_bounds__(256, (MT <= 2 ? 2 : 1)) void k_pointnet_rt(PointNetW w, const float* __restrict__ pts, const uint8_t* __restrict__ pmask,
                                                                       const int* __restrict__ rows, int n_rows, int P, int feat_mask_dim,
                                                                       float* __restrict__ out, float eps) {
  constexpr int G = 16 / L;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g0 = (blockIdx.x * 4 + wave) * G;
  if (g0 >= n_rows) return;   // (no barrier in this kernel)
  const int n = lane & 15, kq = lane >> 4;
  const int g = n / L, q = n - g * L;
  const int Cin = w.in_dim;
  long long rt_t0 = clock64();
  const bool ok = g0 + g < n_rows;
  const int row = ok ? (rows ? rows[g0 + g] : g0 + g) : 0;
  floatx4 acc[8][MT];
  constexpr int DP = rt_depth<MT>();
  RtRing<DP> R;
  rt_prefetch<DP, 1, 1, 8>(R, w.pre_F[0], 0, lane);   // (weights do not depend on the rows: requested before the rows are)
  bool vld[MT];
  half8 xh[MT][4], xl[MT][4];
  bool any = false;
  {
    // every load of the wave's rows leaves before the first is used: the point features unconditionally (a masked or absent
    // point reads the polyline's point 0 and is zeroed afterwards), the masks beside them
    float xv[MT][8];
    unsigned mk[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int p = mt * L + q;
      const int pc = p < P ? p : 0;
      const float* px = pts + ((size_t)row * P + pc) * Cin;
#pragma unroll
      for (int j = 0; j < 8; ++j) {   // layer 0 reads K in the natural order: lane group kq holds features 8 kq .. 8 kq + 7 (K padded to 32)
        const int k = 8 * kq + j;
        xv[mt][j] = k < Cin ? px[k] : 0.f;
      }
      mk[mt] = 1u;
      if (pmask) {
        if (feat_mask_dim == 0) mk[mt] = pmask[(size_t)row * P + pc];
        else
          for (int f = 0; f < feat_mask_dim; ++f) mk[mt] &= (unsigned)(pmask[((size_t)row * P + pc) * feat_mask_dim + f] != 0);
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bool v = ok && mt * L + q < P && mk[mt] != 0;
      vld[mt] = v;
      any = any || v;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = v ? xv[mt][j] : 0.f;
        xh[mt][0][j] = f16_hi(x);
        xl[mt][0][j] = f16_lo(x);
      }
    }
  }
  any = rt_group_max<L>(any ? 1.f : 0.f) != 0.f;   // the polyline has a valid point
  RT_MARK(0);
  // ---- pre_mlps: Linear, LayerNorm, ReLU (the last one: Linear, ReLU)   (:33-38 through layers/mlp.py)
  rt_bias<MT>(acc, w.pre_b[0], kq);
  rt_gemm<MT, 1, 1, 8>(R, acc, xh, xl, w.pre_F[0], 0, lane);
  RT_MARK(1);
  // the next GEMM's fragments leave before this one's epilogue.  (A second, deeper ring for the pooled-row GEMMs -- one column
  // tile, 96 MFMAs for 64 KB of fragments: pure streaming -- was tried: inside the run-time layer loops its 96 registers stay
  // live across every iteration and the kernel spills 0.2 - 0.9 KB per lane.)
  rt_prefetch<DP, 4, 4, 8>(R, w.n_pre > 1 ? w.pre_Q[1] : w.mid_PQ, 0, lane);
  if (w.pre_lnw[0]) rt_ln<MT>(acc, w.pre_lnw[0], w.pre_lnb[0], eps, kq);
  rt_relu<MT>(acc);
  RT_MARK(2);
  for (int l = 1; l < w.n_pre; ++l) {
    rt_to_operand<MT>(acc, xh, xl);
    rt_bias<MT>(acc, w.pre_b[l], kq);
    rt_gemm<MT, 4, 4, 8>(R, acc, xh, xl, w.pre_Q[l], 0, lane);
    rt_prefetch<DP, 4, 4, 8>(R, l + 1 < w.n_pre ? w.pre_Q[l + 1] : w.mid_PQ, 0, lane);
    if (w.pre_lnw[l]) rt_ln<MT>(acc, w.pre_lnw[l], w.pre_lnb[l], eps, kq);
    rt_relu<MT>(acc);
  }
  RT_MARK(3);
  floatx4 pooled[8][1];
  half8 ph[1][4], pl[1][4];
  // ---- mlps: layer 0 consumes cat(point feature, pooled): the pooled half is a per-polyline bias   (:48-50)
  for (int l = 0; l < w.n_mid; ++l) {
    if (l == 0) {
      rt_pool<MT, L>(acc, vld, pooled);
      RT_MARK(4);
    }
    rt_to_operand<MT>(acc, xh, xl);
    if (l == 0) {   // accumulators start from the row's polyline bias: pooled W[:, 128:256]^T + b, computed in the polyline's own lanes
      rt_to_operand<1>(pooled, ph, pl);
      rt_bias<1>(pooled, w.mid_b[0], kq);
      rt_gemm<1, 4, 4, 8>(R, pooled, ph, pl, w.mid_PQ, 0, lane);
      rt_prefetch<DP, 4, 4, 8>(R, w.mid_Q[0], 0, lane);
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = pooled[t][0];
      RT_MARK(5);
    } else {
      rt_bias<MT>(acc, w.mid_b[l], kq);
    }
    rt_gemm<MT, 4, 4, 8>(R, acc, xh, xl, w.mid_Q[l], 0, lane);
    RT_MARK(6);
    rt_prefetch<DP, 4, 4, 8>(R, l + 1 < w.n_mid ? w.mid_Q[l + 1] : w.out_Q0, 0, lane);
    if (w.mid_lnw[l]) rt_ln<MT>(acc, w.mid_lnw[l], w.mid_lnb[l], eps, kq);
    rt_relu<MT>(acc);
    RT_MARK(7);
  }
  // ---- max-pool (:53), out_mlps (:57): Linear, ReLU, Linear on the pooled rows; a polyline without a valid point keeps a zero feature (:56-60)
  rt_pool<MT, L>(acc, vld, pooled);
  RT_MARK(8);
  {
    rt_to_operand<1>(pooled, ph, pl);
    rt_bias<1>(pooled, w.out_b0, kq);
    rt_gemm<1, 4, 4, 8>(R, pooled, ph, pl, w.out_Q0, 0, lane);
    rt_prefetch<DP, 4, 4, 8>(R, w.out_Q1, 0, lane);
    rt_relu<1>(pooled);
    rt_to_operand<1>(pooled, ph, pl);
    rt_bias<1>(pooled, w.out_b1, kq);
    rt_gemm<1, 4, 4, 8>(R, pooled, ph, pl, w.out_Q1, 0, lane);
    RT_MARK(9);
    if (ok && q == 0) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
        *reinterpret_cast<float4*>(out + (size_t)(g0 + g) * 128 + 16 * t + 4 * kq) =
            any ? make_float4(pooled[t][0][0], pooled[t][0][1], pooled[t][0][2], pooled[t][0][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

}  // namespace ps
