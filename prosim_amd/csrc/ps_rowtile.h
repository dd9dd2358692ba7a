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
//   * weights cross the CU ONCE per workgroup and GEMM: the workgroup's waves (each with its own rows) share every Linear, so a GEMM's
//     fragments (<= 64 KB) are copied L2 -> LDS by DMA (buffer_load ... lds: no registers), each wave a share, into one of two
//     stage buffers while the previous GEMM computes; one workgroup barrier per GEMM; the MFMA loop reads A fragments from LDS
//     (conflict-free ds_read_b128).  (First version: every wave streamed its own copy through a register ring -- four waves x 64 KB
//     per GEMM saturate the CU's 64 B/clk L1 path: 140 cycles per 2 KB fragment pair against 48 of MFMA work at one tile per wave.)
//   * no LDS round trip for activations anywhere: the max-pools are DPP folds inside a polyline's lane group.
// Split-fp16 operands (x = hi + lo; w.hi x.hi + w.lo x.hi + w.hi x.lo in fp32), the lo halves scaled by 2^11 (ps_device.h f16_los:
// a weight's lo part would otherwise be a subnormal fp16): the two cross products of a (tile, k-sweep) accumulate in their own
// fp32 tile, which joins the hi.hi tile as acc += 2^-11 x once per tile.
//
// Reference math: PointNetPolylineEncoder.forward (prosim/models/scene_encoder/pointnet_encoder.py:24-62),
// AttentionLayer.forward's per-row half (prosim/models/layers/attention_layer.py:56-79, :100-121).
#pragma once
#include "ps_kernels.h"
#include "ps_attn.h"

namespace ps {

constexpr float RT_LO_INV = PS_LO_INV;
__device__ __forceinline__ _Float16 rt_lo(float x) { return f16_los(x); }

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
// ---- weight stages in LDS.  A stage = the fragments of one GEMM as [piece][lane 64][16 B], piece = 1 KB = one DMA instruction
// (a [hi] or [lo] fragment of a (tile, k-block) group: group g sits at 2048 g, its lo half 1024 further -- the global layout of
// Builder::fragments, so a contiguous weight array is copied as it lies).  Two stage buffers of RT_STAGE_BYTES; protocol per GEMM:
//   every wave: s_waitcnt vmcnt(0) (its share of THIS stage has landed) -> __syncthreads (everybody's has, and everybody is done
//   reading the other buffer) -> issue its share of the NEXT stage into the other buffer -> MFMA loop over this buffer.
// Waves that own no rows still copy their share and meet the barriers.
constexpr int RT_STAGE_BYTES = 64 * 1024;
struct RtStage {
  unsigned lds0;          // LDS byte address of stage buffer 0
  const unsigned char* p0;   // the same as a pointer
  int cb, nb;             // buffer the next GEMM reads / the next fill writes
  int wave, lane, nw;
};
__device__ __forceinline__ RtStage rt_stage_init(unsigned char* smem, int wave, int lane, int nw) {
  RtStage S;
  S.lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);
  S.p0 = smem;
  S.cb = 0; S.nb = 0;
  S.wave = wave; S.lane = lane; S.nw = nw;
  return S;
}
// one DMA instruction: 64 lanes x 16 B from (rsrc base + voff per lane + soff) to LDS [lds_dst + 16 lane] (M0 carries the LDS base;
// inline asm: hipcc neither counts it in vmcnt nor waits for it -- rt_gemm's explicit wait does)
__device__ __forceinline__ void rt_dma1k(unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned rt_goff(int g, int K32, int KT, int k0) { return (unsigned)(((g / K32) * KT + (g % K32) + k0) * 2048); }
// this wave's share of a stage of NT x K32 groups of F (k-blocks k0 .. k0 + K32 - 1 of KT per tile), pieces wave, wave + nw, ...
template <int K32, int KT, int NT>
__device__ __forceinline__ void rt_fill(RtStage& S, const _Float16* __restrict__ F, int k0) {
  const __amdgpu_buffer_rsrc_t rs = rt_rsrc(F);
  const unsigned voff = 16u * (unsigned)S.lane;
  const unsigned dst = S.lds0 + (unsigned)S.nb * RT_STAGE_BYTES;
  for (int p = S.wave; p < 2 * NT * K32; p += S.nw) rt_dma1k(voff, rs, rt_goff(p >> 1, K32, KT, k0) + 1024u * (unsigned)(p & 1), dst + 1024u * (unsigned)p);
  S.nb ^= 1;
}
// ... of `pieces` contiguous KB
__device__ __forceinline__ void rt_fill_linear(RtStage& S, const void* __restrict__ F, int pieces) {
  const __amdgpu_buffer_rsrc_t rs = rt_rsrc(F);
  const unsigned voff = 16u * (unsigned)S.lane;
  const unsigned dst = S.lds0 + (unsigned)S.nb * RT_STAGE_BYTES;
  for (int p = S.wave; p < pieces; p += S.nw) rt_dma1k(voff, rs, 1024u * (unsigned)p, dst + 1024u * (unsigned)p);
  S.nb ^= 1;
}
// the stage the next GEMM reads is complete and the other buffer is free: returns the stage's LDS pointer.  `next` issues the
// following stage's fill (a lambda; may do nothing)
template <class NextFill>
__device__ __forceinline__ const unsigned char* rt_stage_ready(RtStage& S, NextFill next) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned char* buf = S.p0 + (size_t)S.cb * RT_STAGE_BYTES;
  S.cb ^= 1;
  next();
  return buf;
}
// acc[t][mt] += W[16 t .. 16 t + 15][k] . X[k][16 mt + n]  for t < NT over K32 k-blocks, fragments from the current stage (group
// g = t K32 + ks at 2048 g).  A dependent MFMA waits ~45 cycles for its accumulator, an independent one issues every 16: the
// group's MFMAs go product-major over the wave's row tiles, and with a single tile the three products use separate partial
// accumulators.
template <int MT, int K32, int NT, class NextFill>
__device__ __forceinline__ void rt_gemm(RtStage& S, floatx4 (&acc)[NT][MT], const half8 (&xh)[MT][4], const half8 (&xl)[MT][4], NextFill next) {
  constexpr int NG = NT * K32;
  constexpr int DEPTH = 3;
  const half8* fb = reinterpret_cast<const half8*>(rt_stage_ready(S, next)) + S.lane;
  half8 fh[DEPTH], fl[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d < NG) { fh[d] = fb[d * 128]; fl[d] = fb[d * 128 + 64]; }
  floatx4 p1[MT], p2[MT];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int s = g % DEPTH;
    const half8 ah = fh[s], al = fl[s];
    if (g + DEPTH < NG) { fh[s] = fb[(g + DEPTH) * 128]; fl[s] = fb[(g + DEPTH) * 128 + 64]; }
    __builtin_amdgcn_sched_barrier(0);   // (pins the LDS reads DEPTH groups ahead of their use: left alone the scheduler sinks them next to it)
    const int t = g / K32, ks = g % K32;
    if (ks == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) p1[mt] = p2[mt] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh[mt][ks], acc[t][mt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) p1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh[mt][ks], p1[mt], 0, 0, 0);
    // (the two cross products on their OWN accumulators whatever the tile count: a row's bits must not depend on how many tiles
    // its wave carries -- the shapes differ between latency and throughput mode -- and at a single tile the second product would
    // otherwise wait ~45 cycles for the first)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) p2[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl[mt][ks], p2[mt], 0, 0, 0);
    if (ks == K32 - 1) {   // the cross products carry the lo halves' scale
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        p1[mt] += p2[mt];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][mt][j] = fmaf(p1[mt][j], RT_LO_INV, acc[t][mt][j]);
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
        xl[mt][ks][i] = rt_lo(v);
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
constexpr size_t RT_LDS_BYTES = 2 * (size_t)RT_STAGE_BYTES;
template <int MT, int L>
__global__ __launch_bounds__(256, 1) void k_pointnet_rt(PointNetW w, const float* __restrict__ pts, const uint8_t* __restrict__ pmask,
                                                       const int* __restrict__ rows, int n_rows, int P, int feat_mask_dim,
                                                       float* __restrict__ out, float eps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rt_smem[];
  constexpr int G = 16 / L;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  RtStage S = rt_stage_init(rt_smem, wave, lane, 4);
  rt_fill<1, 1, 8>(S, w.pre_Q[0], 0);   // (weights do not depend on the rows: requested before the rows are)
  const int g0 = (blockIdx.x * 4 + wave) * G;   // (a wave without polylines still copies its share of every stage and meets the barriers)
  const int n = lane & 15, kq = lane >> 4;
  const int g = n / L, q = n - g * L;
  const int Cin = w.in_dim;
  long long rt_t0 = clock64();
  const bool ok = g0 + g < n_rows;
  const int row = ok ? (rows ? rows[g0 + g] : g0 + g) : 0;
  floatx4 acc[8][MT];
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
        xl[mt][0][j] = rt_lo(x);
      }
    }
  }
  any = rt_group_max<L>(any ? 1.f : 0.f) != 0.f;   // the polyline has a valid point
  RT_MARK(0);
  // ---- pre_mlps: Linear, LayerNorm, ReLU (the last one: Linear, ReLU)   (:33-38 through layers/mlp.py)
  rt_bias<MT>(acc, w.pre_b[0], kq);
  rt_gemm<MT, 1, 8>(S, acc, xh, xl, [&] { rt_fill<4, 4, 8>(S, w.n_pre > 1 ? w.pre_Q[1] : w.mid_PQ, 0); });
  RT_MARK(1);
  if (w.pre_lnw[0]) rt_ln<MT>(acc, w.pre_lnw[0], w.pre_lnb[0], eps, kq);
  rt_relu<MT>(acc);
  RT_MARK(2);
  for (int l = 1; l < w.n_pre; ++l) {
    rt_to_operand<MT>(acc, xh, xl);
    rt_bias<MT>(acc, w.pre_b[l], kq);
    rt_gemm<MT, 4, 8>(S, acc, xh, xl, [&] { rt_fill<4, 4, 8>(S, l + 1 < w.n_pre ? w.pre_Q[l + 1] : w.mid_PQ, 0); });
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
      rt_gemm<1, 4, 8>(S, pooled, ph, pl, [&] { rt_fill<4, 4, 8>(S, w.mid_Q[0], 0); });
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = pooled[t][0];
      RT_MARK(5);
    } else {
      rt_bias<MT>(acc, w.mid_b[l], kq);
    }
    rt_gemm<MT, 4, 8>(S, acc, xh, xl, [&] { rt_fill<4, 4, 8>(S, l + 1 < w.n_mid ? w.mid_Q[l + 1] : w.out_Q0, 0); });
    RT_MARK(6);
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
    rt_gemm<1, 4, 8>(S, pooled, ph, pl, [&] { rt_fill<4, 4, 8>(S, w.out_Q1, 0); });
    rt_relu<1>(pooled);
    rt_to_operand<1>(pooled, ph, pl);
    rt_bias<1>(pooled, w.out_b1, kq);
    rt_gemm<1, 4, 8>(S, pooled, ph, pl, [] {});
    RT_MARK(9);
    if (ok && q == 0) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
        *reinterpret_cast<float4*>(out + (size_t)(g0 + g) * 128 + 16 * t + 4 * kq) =
            any ? make_float4(pooled[t][0][0], pooled[t][0][1], pooled[t][0][2], pooled[t][0][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// The node half of a split attention layer on row tiles (replaces k_node for geometric edge sets; the edge half stays
// k_edge_small, the per-destination exchange stays EdgeIO in global memory).  A wave owns 16 MT destination rows.
//   PRE  (attention_layer.py:61-69, :106-107, :114): LN_dst, [k | v of the rows themselves for self-attention], q (+ q~, <q, kb>), s, g
//   POST (:76-77, :89, :100-107): to_v_r fold, agg, gate, u, to_out, LN_post + residual, LN_ffpre, FFN in four 128-wide hidden chunks,
//        LN_ffpost + residual
typedef _Float16 rt_half4 __attribute__((ext_vector_type(4)));

// rows of a [.][128] fp32 array into the C layout (lane (n, kq): features 16 t + 4 kq + j of row n); absent rows read as zero
template <int MT>
__device__ __forceinline__ void rt_load_rows(floatx4 (&a)[8][MT], const float* __restrict__ src, const long (&row)[MT], const bool (&live)[MT], int kq) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live[mt]) v = ldg4(src + row[mt] * 128 + 16 * t + 4 * kq);
      a[t][mt] = floatx4{v.x, v.y, v.z, v.w};
    }
}
template <int MT>
__device__ __forceinline__ void rt_store_rows(const floatx4 (&a)[8][MT], float* __restrict__ dst, long stride, long off, const long (&row)[MT],
                                              const bool (&live)[MT], int kq) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
    if (live[mt]) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
        *reinterpret_cast<float4*>(dst + row[mt] * stride + off + 16 * t + 4 * kq) = make_float4(a[t][mt][0], a[t][mt][1], a[t][mt][2], a[t][mt][3]);
    }
}

template <int MT>
__global__ __launch_bounds__(256, 1) void k_node_pre_rt(const float* __restrict__ x, int Nd, const ChainStep* __restrict__ pre, EdgeIO io,
                                                       float eps, float* __restrict__ kv_out, _Float16* __restrict__ khl_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rt_smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile0 = (blockIdx.x * 4 + wave) * MT;   // (a wave without rows still copies its share of every stage and meets the barriers)
  const int n = lane & 15, kq = lane >> 4;
  const AttnW& w = pre->w;
  RtStage S = rt_stage_init(rt_smem, wave, lane, 4);
  if (kv_out) rt_fill<4, 4, 8>(S, w.Wkv_Q, 0);
  else rt_fill<4, 4, 8>(S, w.Fqsg_Q, 0);
  long row[MT];
  bool live[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    row[mt] = 16L * (tile0 + mt) + n;
    live[mt] = row[mt] < Nd;
  }
  floatx4 acc[8][MT];
  half8 xh[MT][4], xl[MT][4];
  rt_load_rows<MT>(acc, x, row, live, kq);
  rt_ln<MT>(acc, w.ln_dst_w, w.ln_dst_b, eps, kq);   // xn = LN_dst(x)
  rt_to_operand<MT>(acc, xh, xl);
  if (kv_out) {   // the rows are also this self-attention layer's sources (LN_src == LN_dst): k | v (:61, :65, :115-116)
    rt_zero<MT, 8>(acc);
    rt_gemm<MT, 4, 8>(S, acc, xh, xl, [&] { rt_fill<4, 4, 8>(S, w.Wkv_Q + (size_t)8 * 4 * 1024, 0); });
    rt_store_rows<MT>(acc, kv_out, 256, 0, row, live, kq);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      if (live[mt]) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          rt_half4 hh, ll;
#pragma unroll
          for (int j = 0; j < 4; ++j) { hh[j] = f16_hi(acc[t][mt][j]); ll[j] = f16_lo(acc[t][mt][j]); }
          *reinterpret_cast<rt_half4*>(khl_out + row[mt] * 256 + 16 * t + 4 * kq) = hh;
          *reinterpret_cast<rt_half4*>(khl_out + row[mt] * 256 + 128 + 16 * t + 4 * kq) = ll;
        }
      }
    rt_bias<MT>(acc, w.bkv + 128, kq);
    rt_gemm<MT, 4, 8>(S, acc, xh, xl, [&] { rt_fill<4, 4, 8>(S, w.Fqsg_Q, 0); });
    rt_store_rows<MT>(acc, kv_out, 256, 128, row, live, kq);
  }
  // ---- q = to_q(xn) + bq; q~[h][c] = sum_d q[16 h + d] Wkr_g3[16 h + d][c]; cq[h] = <q_h, kb_h>
  rt_bias<MT>(acc, w.bq, kq);
  rt_gemm<MT, 4, 8>(S, acc, xh, xl, [&] { rt_fill_linear(S, w.Fkr3_x, 48); });
  rt_store_rows<MT>(acc, io.q, 128, 0, row, live, kq);
  {
    // tile t of q IS head t, and this lane's four values of it are a K = 16 MFMA's B operand as they are: per (head, 16-column
    // tile of c) one v_mfma_f32_16x16x16_f16 per split product, A = the K = 16 fragments of Wkr_g3 (stage piece h 6 + ct: hi | lo halves)
    const unsigned char* kb_ = rt_stage_ready(S, [&] { rt_fill<4, 4, 8>(S, w.Fqsg_Q + (size_t)8 * 4 * 1024, 0); });
    const rt_half4* fk = reinterpret_cast<const rt_half4*>(kb_) + lane;
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      rt_half4 ah[6], al[6];
#pragma unroll
      for (int ct = 0; ct < 6; ++ct) {
        ah[ct] = fk[(h * 6 + ct) * 128];
        al[ct] = fk[(h * 6 + ct) * 128 + 64];
      }
      const float4 kb = ldg4(w.kb + 16 * h + 4 * kq);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        rt_half4 bh, bl;
#pragma unroll
        for (int j = 0; j < 4; ++j) { bh[j] = f16_hi(acc[h][mt][j]); bl[j] = rt_lo(acc[h][mt][j]); }
        float c = 0.f;
        c = fmaf(acc[h][mt][0], kb.x, c); c = fmaf(acc[h][mt][1], kb.y, c); c = fmaf(acc[h][mt][2], kb.z, c); c = fmaf(acc[h][mt][3], kb.w, c);
        c = kq_sum(c);
        if (live[mt] && kq == 0) io.cq[row[mt] * 8 + h] = c;
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) {
          floatx4 o = {0.f, 0.f, 0.f, 0.f}, ox = o;
          o = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[ct], bh, o, 0, 0, 0);
          ox = __builtin_amdgcn_mfma_f32_16x16x16f16(al[ct], bh, ox, 0, 0, 0);
          ox = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[ct], bl, ox, 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = fmaf(ox[j], RT_LO_INV, o[j]);
          if (live[mt]) *reinterpret_cast<float4*>(io.qt + row[mt] * 1024 + h * 128 + 16 * ct + 4 * kq) = make_float4(o[0], o[1], o[2], o[3]);
        }
      }
    }
  }
  // ---- s = to_s(xn) + bs, g_x = to_g's x_dst half + bg   (:106-107; consumed by the POST half)
  rt_bias<MT>(acc, w.bs, kq);
  rt_gemm<MT, 4, 8>(S, acc, xh, xl, [&] { rt_fill<4, 4, 8>(S, w.Fqsg_Q + (size_t)16 * 4 * 1024, 0); });
  rt_store_rows<MT>(acc, io.s, 128, 0, row, live, kq);
  rt_bias<MT>(acc, w.bg, kq);
  rt_gemm<MT, 4, 8>(S, acc, xh, xl, [] {});
  rt_store_rows<MT>(acc, io.g, 128, 0, row, live, kq);
}

template <int MT>
__global__ __launch_bounds__(256, 1) void k_node_post_rt(float* x, int Nd, const ChainStep* __restrict__ post, EdgeIO io, float eps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rt_smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile0 = (blockIdx.x * 4 + wave) * MT;   // (a wave without rows still copies its share of every stage and meets the barriers)
  const int n = lane & 15, kq = lane >> 4;
  const AttnW& w = post->w;
  RtStage S = rt_stage_init(rt_smem, wave, lane, 4);
  rt_fill_linear(S, w.Fvr3_Q, 48);
  long row[MT];
  bool live[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    row[mt] = 16L * (tile0 + mt) + n;
    live[mt] = row[mt] < Nd;
  }
  floatx4 acc[8][MT];
  half8 xh[MT][4], xl[MT][4];
  // ---- to_v_r fold: fold[row][16 h + d] = sum_c a_r[row][h][c] Wvr_g3[c][16 h + d], c < 96: per head one 16-row tile of outputs,
  //      K = 96 in the NATURAL order (the operand comes straight from the edge kernel's fp32 rows: Fvr3 as it is)
  {
    float4 cur[MT][3][2], nxt[MT][3][2];
    auto ld_ar = [&](int h, float4 (&v)[MT][3][2]) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          v[mt][ks][0] = make_float4(0.f, 0.f, 0.f, 0.f);
          v[mt][ks][1] = v[mt][ks][0];
          if (live[mt]) {
            const float* ap = io.ar + row[mt] * 1024 + h * 128 + 32 * ks + 8 * kq;
            v[mt][ks][0] = ldg4(ap);
            v[mt][ks][1] = ldg4(ap + 4);
          }
        }
    };
    ld_ar(0, cur);
    const half8* fv = reinterpret_cast<const half8*>(rt_stage_ready(S, [&] { rt_fill<4, 4, 8>(S, w.Fga_Q, 0); })) + lane;
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      half8 fh[3], fl[3];
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        fh[ks] = fv[(h * 3 + ks) * 128];
        fl[ks] = fv[(h * 3 + ks) * 128 + 64];
      }
      if (h < 7) ld_ar(h + 1, nxt);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        floatx4 o = {0.f, 0.f, 0.f, 0.f}, ox = o;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const float v[8] = {cur[mt][ks][0].x, cur[mt][ks][0].y, cur[mt][ks][0].z, cur[mt][ks][0].w,
                              cur[mt][ks][1].x, cur[mt][ks][1].y, cur[mt][ks][1].z, cur[mt][ks][1].w};
          half8 bh, bl;
#pragma unroll
          for (int i = 0; i < 8; ++i) { bh[i] = f16_hi(v[i]); bl[i] = rt_lo(v[i]); }
          o = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[ks], bh, o, 0, 0, 0);
          ox = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[ks], bh, ox, 0, 0, 0);
          ox = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[ks], bl, ox, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaf(ox[j], RT_LO_INV, o[j]);
        acc[h][mt] = o;
      }
      if (h < 7) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) { cur[mt][ks][0] = nxt[mt][ks][0]; cur[mt][ks][1] = nxt[mt][ks][1]; }
      }
    }
  }
  // ---- agg = (a_v + fold + l vb) / (l + 1e-16)   (:89, :100); tile t is head t
  floatx4 agg[8][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float l = 0.f;
      float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live[mt]) {
        l = ldg1(io.l + row[mt] * 8 + t);
        av = ldg4(io.av + row[mt] * 128 + 16 * t + 4 * kq);
      }
      const float4 vb = ldg4(w.vb + 16 * t + 4 * kq);
      const float inv = 1.f / (l + 1e-16f);
      agg[t][mt][0] = (av.x + acc[t][mt][0] + l * vb.x) * inv;
      agg[t][mt][1] = (av.y + acc[t][mt][1] + l * vb.y) * inv;
      agg[t][mt][2] = (av.z + acc[t][mt][2] + l * vb.z) * inv;
      agg[t][mt][3] = (av.w + acc[t][mt][3] + l * vb.w) * inv;
    }
  // ---- gated update (:106-107): g = sigmoid(Wg [agg | x_dst] + bg) (the x_dst half + bias: the PRE half's io.g); u = agg + g (s - agg)
  rt_to_operand<MT>(agg, xh, xl);
  rt_load_rows<MT>(acc, io.g, row, live, kq);
  rt_gemm<MT, 4, 8>(S, acc, xh, xl, [&] { rt_fill<4, 4, 8>(S, w.Fout_Q, 0); });
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live[mt]) sv = ldg4(io.s + row[mt] * 128 + 16 * t + 4 * kq);
      const float s4[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g = 1.f / (1.f + expf(-acc[t][mt][j]));
        agg[t][mt][j] = agg[t][mt][j] + g * (s4[j] - agg[t][mt][j]);
      }
    }
  // ---- x = x + LN_post(to_out(u))  (:76), xn = LN_ffpre(x)  (:77)
  rt_to_operand<MT>(agg, xh, xl);
  rt_bias<MT>(acc, w.bout, kq);
  rt_gemm<MT, 4, 8>(S, acc, xh, xl, [&] { rt_fill<4, 4, 8>(S, w.F1_Q, 0); });
  rt_ln<MT>(acc, w.ln_post_w, w.ln_post_b, eps, kq);
  {
    floatx4 xr[8][MT];
    rt_load_rows<MT>(xr, x, row, live, kq);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[t][mt] += xr[t][mt];
  }
  // (the rows go back to x now and are read again for the last residual: five 32-register-per-tile sets would be live otherwise)
  rt_store_rows<MT>(acc, x, 128, 0, row, live, kq);
  rt_ln<MT>(acc, w.ln_ffpre_w, w.ln_ffpre_b, eps, kq);
  rt_to_operand<MT>(acc, xh, xl);
  // ---- FFN: relu(W1 xn + b1) in four 128-wide chunks, each straight into its quarter of W2's K; x = x + LN_ffpost(. + b2)
  floatx4 y[8][MT];
  rt_bias<MT>(y, w.b2, kq);
#pragma unroll 1
  for (int c = 0; c < 4; ++c) {
    half8 hh[MT][4], hl[MT][4];
    rt_bias<MT>(acc, w.b1 + 128 * c, kq);
    rt_gemm<MT, 4, 8>(S, acc, xh, xl, [&] { rt_fill<4, 16, 8>(S, w.F2_Q, 4 * c); });
    rt_relu<MT>(acc);
    rt_to_operand<MT>(acc, hh, hl);
    rt_gemm<MT, 4, 8>(S, y, hh, hl, [&] { if (c < 3) rt_fill<4, 4, 8>(S, w.F1_Q + (size_t)(c + 1) * 8 * 4 * 1024, 0); });
  }
  rt_ln<MT>(y, w.ln_ffpost_w, w.ln_ffpost_b, eps, kq);
  {
    const float* xr_src = x;
    asm volatile("" : "+s"(xr_src));   // (opaque: the compiler must reload the rows instead of keeping the stored values live)
    rt_load_rows<MT>(acc, xr_src, row, live, kq);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) y[t][mt] += acc[t][mt];
  }
  rt_store_rows<MT>(y, x, 128, 0, row, live, kq);
}

// ------------------------------------------------------------------------------------------------------------
// ActDecoder._compute_traj + step_agent_traj on row tiles (K = 1 motion mode, anchor mode: every released config; act_decoder.py:78-140,
// CG_stacked mlp.py:207-241, traj_sam.py:300-347).  The staged k_policy_head_mfma takes 39 us per replan for 128 agents -- six
// dependent 16-row GEMM stages with their LDS round trips; here a wave carries 16 agents through CG_decode (3 blocks) and the motion
// head (128 -> 128 -> 64 -> steps * state) in registers, 64 agents per workgroup, and the trajectory tail (cumulative sums, heading
// wrap, rotation into the agent-init frame, append) runs on the last GEMM's rows exactly as in the staged kernel.
template <int MT, int NT>
__device__ __forceinline__ void rt_bias_n(floatx4 (&a)[NT][MT], const float* __restrict__ bias, int kq) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float4 b = ldg4(bias + 16 * t + 4 * kq);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[t][mt] = floatx4{b.x, b.y, b.z, b.w};
  }
}
// LayerNorm over the 16 NT features of every row
template <int MT, int NT>
__device__ __forceinline__ void rt_ln_n(floatx4 (&a)[NT][MT], const float* __restrict__ w, const float* __restrict__ b, float eps, int kq) {
  float rstd[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float sm = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) sm += (a[t][mt][0] + a[t][mt][1]) + (a[t][mt][2] + a[t][mt][3]);
    const float mean = kq_sum(sm) * (1.f / (16.f * NT));
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a[t][mt][j] -= mean;
        sq = fmaf(a[t][mt][j], a[t][mt][j], sq);
      }
    rstd[mt] = 1.f / sqrtf(kq_sum(sq) * (1.f / (16.f * NT)) + eps);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
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
__global__ __launch_bounds__(256, 1) void k_policy_head_rt(HeadW w, const float* __restrict__ fused, const int* __restrict__ agent_type, int n_agents,
                                                          int steps, int sdim, float* __restrict__ motion_pred, float* __restrict__ traj,
                                                          float* __restrict__ vel, int stride_steps, int last, int replan, float eps,
                                                          const float* __restrict__ noise, int vcol) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rt_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  RtStage S = rt_stage_init(rt_smem, wave, lane, 4);
  rt_fill<4, 4, 8>(S, w.cgQ[0], 0);
  const int n = lane & 15, kq = lane >> 4;
  const int ag_w = blockIdx.x * 64 + wave * 16 + n;
  const int agc = ag_w < n_agents ? ag_w : n_agents - 1;
  floatx4 inp[8][1], ctx[8][1], acc[8][1];
  {
    const float* an = w.anchors + (size_t)(agent_type[agc] - 1) * 128;   // anchor of (type, mode 0)   (act_decoder.py:66-68)
    const float* fu = fused + (size_t)agc * 128;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float4 a = ldg4(an + 16 * t + 4 * kq), c = ldg4(fu + 16 * t + 4 * kq);
      inp[t][0] = floatx4{a.x, a.y, a.z, a.w};
      ctx[t][0] = floatx4{c.x, c.y, c.z, c.w};
    }
  }
  half8 xh[1][4], xl[1][4];
  rt_to_operand<1>(inp, xh, xl);
  // CG_stacked(3): block i: y = relu(LN(W inp + b)) * context; with one mode the context's maximum over the modes is y itself
#pragma unroll 1
  for (int i = 0; i < 3; ++i) {
    rt_bias<1>(acc, w.cgb[i], kq);
    rt_gemm<1, 4, 8>(S, acc, xh, xl, [&] { rt_fill<4, 4, 8>(S, i < 2 ? w.cgQ[i + 1] : w.m0Q, 0); });
    rt_ln<1>(acc, w.cglnw[i], w.cglnb[i], eps, kq);
    rt_relu<1>(acc);
    const float fi = (float)i, fd = (float)(i + 1);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float y = acc[t][0][j] * ctx[t][0][j];
        if (i == 0) {
          inp[t][0][j] = y;
          ctx[t][0][j] = y;
        } else {
          inp[t][0][j] = (inp[t][0][j] * fi + y) / fd;
          ctx[t][0][j] = (ctx[t][0][j] * fi + y) / fd;
        }
      }
    rt_to_operand<1>(inp, xh, xl);
  }
  // motion_head: 128 -> 128 (LN, ReLU) -> 64 (LN, ReLU) -> steps * sdim (zero-padded to 128 columns; its bias joins in the tail)
  rt_bias<1>(acc, w.m0b, kq);
  rt_gemm<1, 4, 8>(S, acc, xh, xl, [&] { rt_fill<4, 4, 4>(S, w.m1Q, 0); });
  rt_ln<1>(acc, w.m0lnw, w.m0lnb, eps, kq);
  rt_relu<1>(acc);
  rt_to_operand<1>(acc, xh, xl);
  floatx4 h[4][1];
  rt_bias_n<1, 4>(h, w.m1b, kq);
  rt_gemm<1, 4, 4>(S, h, xh, xl, [&] { rt_fill<2, 2, 8>(S, w.m2Q, 0); });
  rt_ln_n<1, 4>(h, w.m1lnw, w.m1lnb, eps, kq);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float v = fmaxf(h[2 * ks + (i >> 2)][0][i & 3], 0.f);
      xh[0][ks][i] = f16_hi(v);
      xl[0][ks][i] = rt_lo(v);
    }
  rt_zero<1, 8>(acc);
  rt_gemm<1, 2, 8>(S, acc, xh, xl, [] {});
  // the rows of the last Linear meet in LDS (the stage buffer nobody reads any more) for the per-(agent, step) tail
  float* C = reinterpret_cast<float*>(rt_smem + (size_t)S.cb * RT_STAGE_BYTES);
#pragma unroll
  for (int t = 0; t < 8; ++t)
    *reinterpret_cast<float4*>(C + (wave * 16 + n) * PN_CS + 16 * t + 4 * kq) = make_float4(acc[t][0][0], acc[t][0][1], acc[t][0][2], acc[t][0][3]);
  __syncthreads();
  // cumsum over steps of (dx, dy, dtheta); wrap theta (act_decoder.py:117-121).  One thread per (agent, step): it re-adds the prefix in
  // step order, so the sums round exactly like the sequential scan (the staged kernel's tail, K = 1)
  const int ag0 = blockIdx.x * 64;
  for (int i = tid; i < 64 * steps; i += 256) {
    const int row = i / steps, s = i - row * steps;
    const int ag = ag0 + row;
    if (ag >= n_agents) continue;
    const float* o = C + row * PN_CS;
    const float* ob = w.m2b;
    float cx = 0.f, cy = 0.f, ch = 0.f;
    const float* nz = noise ? noise + (size_t)ag * steps * 2 : nullptr;
    for (int j = 0; j <= s; ++j) {
      const float dx = o[j * sdim] + ob[j * sdim], dy = o[j * sdim + 1] + ob[j * sdim + 1];
      cx += nz ? dx + nz[2 * j] : dx;
      cy += nz ? dy + nz[2 * j + 1] : dy;
      ch += o[j * sdim + 2] + ob[j * sdim + 2];
    }
    const float hh = wrap_angle(ch);
    float* mp = motion_pred + (size_t)ag * steps * sdim + s * sdim;
    mp[0] = cx;
    mp[1] = cy;
    mp[2] = hh;
    for (int f = 3; f < sdim; ++f) mp[f] = o[s * sdim + f] + ob[s * sdim + f];
    if (s < replan) {
      // step_agent_traj (traj_sam.py:322-347): rotate into the agent-init frame, append
      const float* cur = traj + ((size_t)ag * stride_steps + last - 1) * 4;
      const float c0 = cur[0], c1 = cur[1];
      const float lth = atan2f(cur[2], cur[3]);
      const float cl = cosf(lth), sl = sinf(lth);
      float* t = traj + ((size_t)ag * stride_steps + last + s) * 4;
      float* v = vel + ((size_t)ag * stride_steps + last + s) * 2;
      t[0] = (cx * cl - cy * sl) + c0;
      t[1] = (cy * cl + cx * sl) + c1;
      const float pth = wrap_angle(lth + hh);
      t[2] = sinf(pth);
      t[3] = cosf(pth);
      if (vcol >= 0) {
        const float vx = o[s * sdim + vcol] + ob[s * sdim + vcol], vy = o[s * sdim + vcol + 1] + ob[s * sdim + vcol + 1];
        v[0] = vx * cl - vy * sl;
        v[1] = vy * cl + vx * sl;
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------
// The same head for FEW agents (round 5; K = 1): one 4-wave workgroup per agent, every Linear a wave-local fp32 GEMV whose
// weights stream through registers one stage ahead (the machinery of k_attn_chain's node stages, ps_attn.h).  The row-tile
// kernel above needs a workgroup per 64 agents and is bound by its stage fills (six 64 KB weight stages by LDS-DMA at ~15 B/clk per
// CU: 35 us per replan on the TWO workgroups of a 128-agent scene, 6 % of that scene's rollout); here 128 workgroups pull the
// 320 KB of head weights at the per-CU register-streaming rate each (~42 B/clk).  Chosen by the agent count alone (<= 128 rows:
// ps_policy_step), so a scene's bits do not depend on the engine mode.  CG_stacked / motion head / tail: the operations of
// k_policy_head_rt, fp32 FMAs instead of split-fp16 MFMAs.
__global__ __launch_bounds__(256) void k_policy_head_row(HeadW w, const float* __restrict__ fused, const int* __restrict__ agent_type, int n_agents,
                                                         int steps, int sdim, float* __restrict__ motion_pred, float* __restrict__ traj,
                                                         float* __restrict__ vel, int stride_steps, int last, int replan, float eps,
                                                         const float* __restrict__ noise, int vcol, StepNext nx) {
  __shared__ __attribute__((aligned(16))) float xin[128], ctx[128], y[128], h64[64], o[128];
  const int ag = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // N = 128 GEMVs (K = 128 or 64): wave -> 32 output columns, lane -> (column quad c8, k-group kgl); N = 64: wave -> 16 columns
  const int c8 = lane & 7, kgl = lane >> 3, ncol = wave * 32 + 4 * c8;
  const int c4 = lane & 3, kg4 = lane >> 2, ncol4 = wave * 16 + 4 * c4;
  const size_t woff = (size_t)(kgl * 16) * 128 + ncol;
  WC<16> wa, wb;
  wload(wa, w.cgWt[0] + woff, 128);
  if (tid < 128) {
    xin[tid] = ldg1(w.anchors + (size_t)(agent_type[ag] - 1) * 128 + tid);   // anchor of (type, mode 0)   (act_decoder.py:66-68)
    ctx[tid] = ldg1(fused + (size_t)ag * 128 + tid);
  }
  __syncthreads();
  auto stage128 = [&](const WC<16>& cur, const float* __restrict__ bias) {   // y = W x + b over K = 128
    float acc[1][4];
    zero_acc<1>(acc);
    wfma<1, 16>(cur, xin + kgl * 16, 128, acc);
    fold_kgroups<1, 8>(acc);
    if (kgl == 0) *reinterpret_cast<float4*>(y + ncol) = make_float4(acc[0][0] + bias[ncol], acc[0][1] + bias[ncol + 1], acc[0][2] + bias[ncol + 2], acc[0][3] + bias[ncol + 3]);
    __syncthreads();
  };
  // CG_stacked(3): block i: y = relu(LN(W inp + b)) * context; with one mode the context's maximum over the modes is y itself
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    WC<16>& cur = (i & 1) ? wb : wa;
    WC<16>& nxt = (i & 1) ? wa : wb;
    wload(nxt, (i < 2 ? w.cgWt[i + 1] : w.m0t) + woff, 128);   // (weights do not depend on activations: one stage ahead)
    stage128(cur, w.cgb[i]);
    if (wave == 0) ln_row_wave(y, y, w.cglnw[i], w.cglnb[i], eps, lane, true);
    __syncthreads();
    if (tid < 128) {
      const float yy = y[tid] * ctx[tid];
      const float fi = (float)i, fd = (float)(i + 1);
      const float v = i == 0 ? yy : (xin[tid] * fi + yy) / fd;
      const float cv = i == 0 ? yy : (ctx[tid] * fi + yy) / fd;
      xin[tid] = v;
      ctx[tid] = cv;
    }
    __syncthreads();
  }
  // motion_head: 128 -> 128 (LN, ReLU) -> 64 (LN, ReLU) -> steps * sdim (zero-padded to 128 columns; its bias joins in the tail)
  WC<8> wc, wd;
  wload(wc, w.m1t + (size_t)(kg4 * 8) * 64 + ncol4, 64);
  stage128(wb, w.m0b);   // (after three blocks the motion head's first Linear sits in wb)
  if (wave == 0) ln_row_wave(y, y, w.m0lnw, w.m0lnb, eps, lane, true);
  __syncthreads();
  wload(wd, w.m2t + (size_t)(kgl * 8) * 128 + ncol, 128);
  {
    float acc[1][4];
    zero_acc<1>(acc);
    wfma<1, 8>(wc, y + kg4 * 8, 128, acc);
    fold_kgroups<1, 4>(acc);
    if (kg4 == 0) *reinterpret_cast<float4*>(h64 + ncol4) = make_float4(acc[0][0] + w.m1b[ncol4], acc[0][1] + w.m1b[ncol4 + 1], acc[0][2] + w.m1b[ncol4 + 2], acc[0][3] + w.m1b[ncol4 + 3]);
  }
  __syncthreads();
  if (wave == 0) {   // LayerNorm over 64 + ReLU: one element per lane
    const float a = h64[lane];
    const float mean = wave_sum(a) * (1.f / 64.f);
    const float d = a - mean;
    const float rstd = 1.f / sqrtf(wave_sum(d * d) * (1.f / 64.f) + eps);
    h64[lane] = fmaxf(fmaf(d * rstd, w.m1lnw[lane], w.m1lnb[lane]), 0.f);
  }
  __syncthreads();
  {
    float acc[1][4];
    zero_acc<1>(acc);
    wfma<1, 8>(wd, h64 + kgl * 8, 64, acc);
    fold_kgroups<1, 8>(acc);
    if (kgl == 0) *reinterpret_cast<float4*>(o + ncol) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
  }
  __syncthreads();
  // cumsum over steps of (dx, dy, dtheta); wrap theta (act_decoder.py:117-121): thread s re-adds the prefix in step order (k_policy_head_rt's tail)
  if (tid < steps) {
    const int s = tid;
    const float* ob = w.m2b;
    float cx = 0.f, cy = 0.f, ch = 0.f;
    const float* nz = noise ? noise + (size_t)ag * steps * 2 : nullptr;
    for (int j = 0; j <= s; ++j) {
      const float dx = o[j * sdim] + ob[j * sdim], dy = o[j * sdim + 1] + ob[j * sdim + 1];
      cx += nz ? dx + nz[2 * j] : dx;
      cy += nz ? dy + nz[2 * j + 1] : dy;
      ch += o[j * sdim + 2] + ob[j * sdim + 2];
    }
    const float hh = wrap_angle(ch);
    float* mp = motion_pred + (size_t)ag * steps * sdim + s * sdim;
    mp[0] = cx;
    mp[1] = cy;
    mp[2] = hh;
    for (int f = 3; f < sdim; ++f) mp[f] = o[s * sdim + f] + ob[s * sdim + f];
    if (s < replan) {
      // step_agent_traj (traj_sam.py:322-347): rotate into the agent-init frame, append
      const float* cur = traj + ((size_t)ag * stride_steps + last - 1) * 4;
      const float c0 = cur[0], c1 = cur[1];
      const float lth = atan2f(cur[2], cur[3]);
      const float cl = cosf(lth), sl = sinf(lth);
      float* t = traj + ((size_t)ag * stride_steps + last + s) * 4;
      float* v = vel + ((size_t)ag * stride_steps + last + s) * 2;
      t[0] = (cx * cl - cy * sl) + c0;
      t[1] = (cy * cl + cx * sl) + c1;
      const float pth = wrap_angle(lth + hh);
      t[2] = sinf(pth);
      t[3] = cosf(pth);
      if (vcol >= 0) {
        const float vx = o[s * sdim + vcol] + ob[s * sdim + vcol], vy = o[s * sdim + vcol + 1] + ob[s * sdim + vcol + 1];
        v[0] = vx * cl - vy * sl;
        v[1] = vy * cl + vx * sl;
      }
    }
  }
  // Round 5: the NEXT replan's step_env of this agent (traj_sam.py:205-274) -- it reads nothing but this agent's own states, the last `replan`
  // of them written just above.  Workgroup-scope release / acquire around the barrier: writers and readers share the CU's vector cache (an
  // agent-scope fence pair here writes the XCD's L2 back and costs 12 us per launch: measured).
  if (nx.on) {
    __threadfence_block();
    __syncthreads();
    __threadfence_block();
    step_env_body(ag, tid, traj, vel, stride_steps, nx.last, nx.hist, nx.dt, nx.init_pos, nx.init_head, nx.static_in, nx.obs_dim, nx.obs_in,
                  nx.cur_pos, nx.cur_ori, 1, nx.tok_pos, nx.tok_ori, nx.lg, nx.fd_vel);
  }
}

}  // namespace ps
