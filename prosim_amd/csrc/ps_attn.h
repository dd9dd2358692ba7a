// Fused graph-attention layer chain (K5-K8 of SURVEY.md section 2a) for gfx950.
//
// Reference math: AttentionLayer.forward (prosim/models/layers/attention_layer.py:56-121).
// One workgroup owns T destination rows and carries them through `nsteps` layers without
// leaving the CU: pre-norm, q/s/gate projections, the per-edge score + softmax + aggregation,
// the gated update, to_out, post-norm, and the 128->512->128 FFN.  A chain is legal whenever the
// source side (k, v) of every step is already materialised: self-attention chains have one step
// (k/v change every layer -> kernel boundary), the policy's 12 bipartite layers are ONE launch
// because agent and map tokens do not change inside a replan (act_decoder.py:267-277).
//
// Per-edge projections are factored (SURVEY.md section 7 "algebraic shortcut"):
//   <q_i, Wkr r^_e>_h = <Wkr_h^T q_i,h , r^_e>          -> q~[h][128] once per destination
//   sum_e a_e (Wvr r^_e + bvr)  = Wvr (sum_e a_e r^_e) + bvr (sum_e a_e)
// with the LayerNorm affine of r folded into the weights on the host (r~ = normalised r without
// affine is shared by all layers of an edge set).  16x fewer FLOPs than the reference's per-edge
// 128x128 projections; identical in exact arithmetic.
#pragma once
#include "ps_device.h"

namespace ps {

struct AttnW {
  const float *ln_src_w, *ln_src_b, *ln_dst_w, *ln_dst_b;
  const float *Wq_t, *bq, *Ws_t, *bs, *Wgx_t, *bg;  // [128][128] K-major
  const float* Wkr_g;                               // [hd=128][c=128]: to_k_r.weight * gamma_r[c]
  const float* kb;                                  // [128]: to_k_r.weight @ beta_r
  const float* Wvr_gt;                              // [c=128][hd=128]: (to_v_r.weight * gamma_r)^T
  // The geometric rel-PE row is fourier([dist, rel_ori, angle, angle]) (act_decoder.py:217 and twins): features
  // 96..127 repeat 64..95.  "Folded" variants add those weight columns / rows onto 64..95, so the kernels read
  // and multiply only 96 of the 128 rel-PE columns (ChainStep::kr == 3).
  const float* Wkr_g3;                              // Wkr_g with columns 96..127 added onto 64..95
  const float* Wvr_gt3;                             // Wvr_gt with rows 96..127 added onto 64..95
  const float* vb;                                  // [128]: to_v_r.weight @ beta_r + to_v_r.bias
  const float *Wga_t, *Wout_t, *bout;
  const float *ln_post_w, *ln_post_b, *ln_ffpre_w, *ln_ffpre_b;
  const float *W1_t, *b1, *W2_t, *b2, *ln_ffpost_w, *ln_ffpost_b;
  const float *Wkv_t, *bkv;                         // [128][256] K-major (k | v), [256]
  const _Float16* Wkv_F;                            // the same as split-fp16 MFMA B fragments [n-tile 16][k-block 4][hi|lo][64][8]
  const float* sp;                                  // packed small vectors, SP_* offsets below (2432 floats)
};
// offsets (floats) inside AttnW::sp -- one coalesced load per layer stages them in LDS, so no
// bias / LayerNorm-parameter load ever sits on the layer's dependency chain
enum : int { SP_LN_DST_W = 0, SP_LN_DST_B = 128, SP_BQ = 256, SP_BS = 384, SP_BG = 512, SP_KB = 640, SP_VB = 768,
             SP_BOUT = 896, SP_LN_POST_W = 1024, SP_LN_POST_B = 1152, SP_LN_FFPRE_W = 1280, SP_LN_FFPRE_B = 1408,
             SP_B1 = 1536, SP_B2 = 2048, SP_LN_FFPOST_W = 2176, SP_LN_FFPOST_B = 2304, SP_SIZE = 2432 };

struct ChainStep {
  AttnW w;
  const float* kv;    // [Ns][256] projected sources (k | v) for this layer
  const int* eoff;    // [Nd+1] CSR offsets by destination
  const int* esrc;    // [E] source row in kv
  const int* toff;    // [Nd+1] offsets in 32-edge tiles (sum of ceil(deg/32)) into rtT
  // normalised relative-PE rows (no affine) as split fp16 (hi | lo), cut into 32-edge tiles per destination and
  // stored twice, once per MFMA operand shape (layouts: k_tile_transpose in ps_kernels.h)
  const _Float16* rtA;    // [tiles][8192]: score pass A operand (edge-major fragments, 1 KB contiguous per load)
  const _Float16* rtT;    // [tiles][8192]: aggregation pass B operand (edge-minor)
  const _Float16* khl;    // [Ns][256]: the k rows of kv as split fp16 (hi | lo)
  int kr;                 // rel-PE column blocks of 32 that are distinct: 3 for geometric edge sets (columns 96..127
                          // repeat 64..95 and are neither stored nor read), 4 for condition rows
};

// Edge lists are walked in chunks of CH edges per destination with an online (running max / sum)
// softmax, so the LDS score tile is 8*CH*T floats whatever the degree.
// The chunk is 256 edges (128 at T = 4, so that two 4-row workgroups fit the 160 KB of a CU).
template <int T>
__host__ __device__ constexpr int chunk_edges() { return T >= 4 ? 128 : 256; }
// LDS plan (floats), NW = waves per workgroup: rows 6*128*T + 512*T | big NW*8*QP (q~ image, then per-wave
// partial a_r) | sc 8*CH*T scores | avp NW*128 | ml NW*16 | cq 8*T | sp 2*SP_SIZE | esl CH*T (source rows of the chunk)
template <int T, int NW = 4>
__host__ __device__ constexpr size_t attn_lds_floats(int /*maxdeg*/) {
  return (size_t)(6 * 128 + 512) * T + NW * 8 * QP + (size_t)8 * chunk_edges<T>() * T + NW * 128 + NW * 16 + 8 * T + 64 +
         2 * SP_SIZE + (size_t)chunk_edges<T>() * T;
}

// 8 partial sums (one per head) held by each of 8 consecutive lanes -> lane cc ends with the
// total of head cc (transpose-reduce, 7 shuffles instead of 24).
__device__ __forceinline__ float reduce8_to_lane(const float (&a)[8], int cc) {
  float b[4], c[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float lo = a[i], hi = a[i + 4];
    const float send = (cc & 4) ? lo : hi, keep = (cc & 4) ? hi : lo;
    b[i] = keep + __shfl_xor(send, 4);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float lo = b[i], hi = b[i + 2];
    const float send = (cc & 2) ? lo : hi, keep = (cc & 2) ? hi : lo;
    c[i] = keep + __shfl_xor(send, 2);
  }
  const float send = (cc & 1) ? c[0] : c[1], keep = (cc & 1) ? c[1] : c[0];
  return keep + __shfl_xor(send, 1);
}

// ---- weight streaming.  With one destination row per workgroup a layer is GEMV work: 960 KB of
// fp32 weights stream through each CU per layer and nothing is reused, so the kernel is bound by
// how many bytes it keeps in flight and by the length of its dependency chain.
//  * Weights do not depend on activations: every 16-row weight chunk (16 x float4 per thread =
//    64 KB per workgroup) is requested one full chunk AHEAD of the chunk being multiplied, into the
//    other of two register sets (plain global loads survive s_barrier; the compiler's in-order
//    vmcnt lets the older set complete while the newer flies).
//  * Every GEMV is WAVE-LOCAL: a wave owns a block of output columns and all of K, its lanes split
//    K, and the partial sums meet by shuffles -- no LDS partial buffer, one barrier per stage.
template <int R>   // R weight rows x 4 columns per lane: 16 with 4 waves per workgroup, 8 with 8
struct WC {
  float4 w[R];
};
template <int R>
__device__ __forceinline__ void wload(WC<R>& c, const float* __restrict__ p, int N) {
#pragma unroll
  for (int i = 0; i < R; ++i) c.w[i] = ldg4(p + (size_t)i * N);
}
template <int T, int R>
__device__ __forceinline__ void wfma(const WC<R>& c, const float* x, int xs, float (&acc)[T][4]) {
  // row-outer: R x-values (ds_read_b128s) live at a time, not R*T
#pragma unroll
  for (int t = 0; t < T; ++t) {
    float xv[R];
#pragma unroll
    for (int i = 0; i < R / 4; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(x + t * xs + 4 * i);
      xv[4 * i] = v.x; xv[4 * i + 1] = v.y; xv[4 * i + 2] = v.z; xv[4 * i + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      acc[t][0] = fmaf(xv[i], c.w[i].x, acc[t][0]);
      acc[t][1] = fmaf(xv[i], c.w[i].y, acc[t][1]);
      acc[t][2] = fmaf(xv[i], c.w[i].z, acc[t][2]);
      acc[t][3] = fmaf(xv[i], c.w[i].w, acc[t][3]);
    }
  }
}
template <int T>
__device__ __forceinline__ void zero_acc(float (&acc)[T][4]) {
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f;
}
// sum the partials of the lanes that share output columns: lane bits >= LOWBITS index the k-group
template <int T, int LOW>
__device__ __forceinline__ void fold_kgroups(float (&acc)[T][4]) {
#pragma unroll
  for (int o = LOW; o < 64; o <<= 1) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t][j] += __shfl_xor(acc[t][j], o);
    }
  }
}

// stage helper.  Weight chunks do not depend on activations, so every chunk is requested before the stage that
// multiplies it.  EARLY (T == 1, full register file): the next chunk is requested BEFORE this chunk's FMAs, two
// register sets live.  Otherwise (two workgroups per CU, <= 256 registers): it is requested right AFTER this
// chunk's FMAs, into registers that just died -- one set live, and the request still flies under the fold, the
// LDS write and the barrier that close the stage.  CURP/CURN are kept for readability only.
#define PS_STAGE(CUR, CURP, CURN, NXT, NXTP, NXTN, X, XS) \
  do {                                                    \
    if (PF) wload(NXT, NXTP, NXTN);                       \
    wfma<T, RK>(CUR, X, XS, acc);                         \
    if (!PF) {                                            \
      __builtin_amdgcn_sched_barrier(0);                  \
      wload(NXT, NXTP, NXTN);                             \
    }                                                     \
  } while (0)

// KR: distinct rel-PE column blocks of 32 (3 for geometric edge sets, 4 for condition rows / the test hook); a
// launch only ever chains steps of one kind, so it is a compile-time parameter (ChainStep::kr must agree).
// BIG: compiled for ONE workgroup per CU with the whole register file (512 per lane): weight chunks and edge
// rows are software-prefetched into second register sets.  !BIG (T >= 2 default): two workgroups per CU, <= 256
// registers, the co-resident workgroup hides latency instead.
template <int T, int NW = 4, int KR = 3, bool BIG = (T == 1)>
__global__ __launch_bounds__(64 * NW, ((!BIG && NW == 4) ? 2 : 1)) void k_attn_chain(float* __restrict__ x, const float* __restrict__ x_in, int Nd,
                                                     const ChainStep* __restrict__ steps, int nsteps, int maxdeg, float eps, int flags,
                                                     unsigned long long* __restrict__ prof) {
  // phase clocks for tools/gpu_phase.py (prof == nullptr in every product launch): thread 0 of each
  // workgroup charges the core-clock cycles since the previous mark to phase i
  long long tprev = prof ? clock64() : 0;
#define PS_MARK(i)                                                        \
  do {                                                                    \
    if (prof && threadIdx.x == 0) {                                       \
      const long long now_ = clock64();                                   \
      atomicAdd(prof + (i), (unsigned long long)(now_ - tprev));          \
      tprev = now_;                                                       \
    }                                                                     \
  } while (0)
  constexpr int NT = 64 * NW;   // threads
  constexpr int W = NW / T;     // waves per destination in the edge phase
  // N = 128 GEMVs: wave -> CW output columns; lane -> (LQ column quads) x (KG k-groups of RK weight rows)
  constexpr int CW = 128 / NW, LQ = CW / 4, KG = 64 / LQ, RK = 128 / KG;
  // N = 512 (FFN up): wave -> 512/NW columns; lane -> (L5 column quads) x (64/L5 k-groups of 4*RK rows)
  constexpr int L5 = 128 / NW;
  static_assert(KG * RK == 128 && (64 / L5) * 4 * RK == 128, "GEMV tiling");
  constexpr int CH = chunk_edges<T>();
  if ((flags >> 8) && blockIdx.x >= gridDim.x / 2) {   // experiment: de-phase the two workgroups of a CU
    const long long t0 = clock64(), dl = (long long)(flags >> 8) << 10;
    while (clock64() - t0 < dl) __builtin_amdgcn_s_sleep(32);
  }
  // T == 1: one workgroup per CU with the full register file -> software prefetch (weights one chunk ahead,
  // edge rows one tile ahead).  T >= 2: compiled for 2 workgroups per CU (__launch_bounds__(256, 2), <= 256
  // registers): the co-resident workgroup hides the latency instead and nothing is double-buffered.
  constexpr bool PF = BIG;                     // weight chunks one stage ahead (second register set)
  constexpr bool PFE = BIG;                    // edge rows one tile ahead
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                 // [T][128] residual stream
  float* xn = xs + 128 * T;         // [T][128] normed / scratch row
  float* qb = xn + 128 * T;         // [T][128] q
  float* sb = qb + 128 * T;         // [T][128] to_s(x_dst)
  float* gb = sb + 128 * T;         // [T][128] to_g's x_dst half (+bias)
  float* ag = gb + 128 * T;         // [T][128] aggregated message, then the gated update u
  float* f1 = ag + 128 * T;         // [T][512]
  float* big = f1 + 512 * T;        // [NW][8][QP]
  float* un = big + NW * 8 * QP;    // [T][CH][8] score tile
  float* avp = un + (size_t)8 * CH * T;  // [NW][128]
  float* ml = avp + NW * 128;       // [NW][16]: per wave (max[8] | sum[8])
  float* cq = ml + NW * 16;         // [T][8]
  float* spb = cq + 8 * T + 56;     // [2][SP_SIZE] small per-layer vectors, double-buffered
  int* esl = reinterpret_cast<int*>(spb + 2 * SP_SIZE);  // [T][CH] source rows of the current chunk of each edge list

  const int tid_o = threadIdx.x, wave_o = tid_o >> 6, lane_o = tid_o & 63;
  const int c8_o = lane_o % LQ, kgl_o = lane_o / LQ;
  const int ncol_o = wave_o * CW + 4 * c8_o;
  const size_t woff_o = (size_t)(kgl_o * RK) * 128 + ncol_o;
  const int row0 = blockIdx.x * T;
  // two register sets, one chunk in flight behind the one being multiplied.  (Three sets / two
  // chunks in flight measured no faster -- a CU's 4 waves already pull ~100 GB/s, the per-CU
  // L2->register ceiling measured by ps_test_stream -- and spill at T = 4.)
  WC<RK> wA, wB;
  if (PF) wload(wA, steps[0].w.Wq_t + woff_o, 128);
  // small vectors of layer 0 -> LDS buffer 0 (608 float4: threads take float4 tid, tid+NT, ...)
  constexpr int NSP = (SP_SIZE / 4 + NT - 1) / NT;
  float4 spr[NSP];
  auto sp_load = [&](const float* __restrict__ sp) {
#pragma unroll
    for (int i = 0; i < NSP; ++i)
      if (tid_o + i * NT < SP_SIZE / 4) spr[i] = ldg4(sp + 4 * (tid_o + i * NT));
  };
  auto sp_store = [&](float* dst) {
#pragma unroll
    for (int i = 0; i < NSP; ++i)
      if (tid_o + i * NT < SP_SIZE / 4) *reinterpret_cast<float4*>(dst + 4 * (tid_o + i * NT)) = spr[i];
  };
  sp_load(steps[0].w.sp);
  // load the T residual rows (rows past Nd are zero-filled and never stored)
  for (int i = tid_o; i < T * 128; i += NT) {
    const int t = i >> 7, r = row0 + t;
    xs[i] = (r < Nd) ? ldg1(x_in + (size_t)r * 128 + (i & 127)) : 0.f;   // rows come from x_in, leave to x
  }
  sp_store(spb);
  __syncthreads();
  for (int tt = wave_o; tt < T; tt += NW) ln_row_wave(xs + tt * 128, xn + tt * 128, spb + SP_LN_DST_W, spb + SP_LN_DST_B, eps, lane_o, false);
  __syncthreads();

  for (int s = 0; s < nsteps; ++s) {
    const ChainStep& st = steps[s];
    const AttnW& w = st.w;
    // Lane-derived indices are re-materialised behind an opaque asm every layer: otherwise LICM hoists
    // hundreds of loop-invariant LDS/global addresses out of the layer loop and they spill (T >= 2).
    int tid_v = threadIdx.x;
    asm volatile("" : "+v"(tid_v));
    const int tid = tid_v, lane = tid_v & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid_v >> 6);   // wave-uniform: pointer math on the SALU
    const int c8 = lane % LQ, kgl = lane / LQ;
    const int ncol = wave * CW + 4 * c8;
    const size_t woff = (size_t)(kgl * RK) * 128 + ncol;
    const int c32 = lane & 31, k2 = lane >> 5;                 // q~ stage: 4 columns x (head | K half)
    const int c5 = lane % L5, k5 = lane / L5;                  // FFN up: 4 columns x k-group
    const int ncol5 = wave * (512 / NW) + 4 * c5;
    const float* sp = spb + (s & 1) * SP_SIZE;          // this layer's small vectors (LDS)
    float* sp_next = spb + ((s + 1) & 1) * SP_SIZE;     // filled mid-layer for the next one
    if (s + 1 < nsteps) sp_load(steps[s + 1].w.sp);
    // ---- edge list of each destination -> LDS (source rows), so the row gathers below never wait on an index
    const int t = wave / W, wi = wave % W;
    const int r = row0 + t;
    const int e_beg = (r < Nd) ? ldgi(st.eoff + r) : 0;
    const int deg = ((r < Nd) && !(flags & 1)) ? (ldgi(st.eoff + r + 1) - e_beg) : 0;
    int* el = esl + t * CH;
    for (int e = wi * 64 + lane; e < deg && e < CH; e += 64 * W) el[e] = ldgi(st.esrc + e_beg + e);   // first chunk
    // ---- q / s / gate(x) projections of the pre-normed rows (:61-69, :106-107, :114); xn = LN_dst(x)
    {
      float acc[T][4];
      zero_acc<T>(acc);
      if (!PF) wload(wA, w.Wq_t + woff, 128);   // (late mode keeps no weights in registers across the layer boundary)
      PS_STAGE(wA, w.Wq_t + woff, 128, wB, w.Ws_t + woff, 128, xn + kgl * RK, 128);    // Wq
      fold_kgroups<T, LQ>(acc);
      if (kgl == 0) {
#pragma unroll
        for (int tt = 0; tt < T; ++tt)
          *reinterpret_cast<float4*>(qb + tt * 128 + ncol) =
              make_float4(acc[tt][0] + sp[SP_BQ + ncol], acc[tt][1] + sp[SP_BQ + ncol + 1], acc[tt][2] + sp[SP_BQ + ncol + 2],
                          acc[tt][3] + sp[SP_BQ + ncol + 3]);
      }
      zero_acc<T>(acc);
      PS_STAGE(wB, w.Ws_t + woff, 128, wA, w.Wgx_t + woff, 128, xn + kgl * RK, 128);   // Ws
      fold_kgroups<T, LQ>(acc);
      if (kgl == 0) {
#pragma unroll
        for (int tt = 0; tt < T; ++tt)
          *reinterpret_cast<float4*>(sb + tt * 128 + ncol) =
              make_float4(acc[tt][0] + sp[SP_BS + ncol], acc[tt][1] + sp[SP_BS + ncol + 1], acc[tt][2] + sp[SP_BS + ncol + 2],
                          acc[tt][3] + sp[SP_BS + ncol + 3]);
      }
      zero_acc<T>(acc);
      // q~ chunk: NW = 4: wave -> heads 2*wave + k2, all 16 rows of the head; NW = 8: wave -> head `wave`,
      // k2 -> its 8-row half.  lane & 31 -> 4 columns.
      const float* wkr = (KR == 3 ? w.Wkr_g3 : w.Wkr_g) + (size_t)(NW == 4 ? (2 * wave + k2) * 16 : wave * 16 + 8 * k2) * 128 + 4 * c32;
      PS_STAGE(wA, w.Wgx_t + woff, 128, wB, wkr, 128, xn + kgl * RK, 128);   // Wgx
      fold_kgroups<T, LQ>(acc);
      if (kgl == 0) {
#pragma unroll
        for (int tt = 0; tt < T; ++tt)
          *reinterpret_cast<float4*>(gb + tt * 128 + ncol) =
              make_float4(acc[tt][0] + sp[SP_BG + ncol], acc[tt][1] + sp[SP_BG + ncol + 1], acc[tt][2] + sp[SP_BG + ncol + 2],
                          acc[tt][3] + sp[SP_BG + ncol + 3]);
      }
    }
    __syncthreads();
    PS_MARK(0);
    // ---- q~[t][h][c] = sum_d q[t][16h+d] * Wkr_g[16h+d][c];  cq[t][h] = <q_h, kb_h>
    {
      const int h = NW == 4 ? 2 * wave + k2 : wave;
      const int hr = NW == 4 ? h * 16 : h * 16 + 8 * k2;   // first of this lane's RK rows of Wkr_g / elements of q_h
      float acc[T][4];
      zero_acc<T>(acc);
      wfma<T, RK>(wB, qb + hr, 128, acc);                                               // Wkr_g
      if (NW == 8) fold_kgroups<T, 32>(acc);
      if (NW == 4 || k2 == 0) {
#pragma unroll
        for (int tt = 0; tt < T; ++tt)
          *reinterpret_cast<float4*>(big + (size_t)(tt * 8 + h) * QP + 4 * c32) =
              make_float4(acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]);
      }
      if (tid < 8 * T) {
        const int tt = tid >> 3, hh = tid & 7;
        float a = 0.f;
        for (int d = 0; d < DH; ++d) a = fmaf(qb[tt * 128 + hh * DH + d], sp[SP_KB + hh * DH + d], a);
        cq[tid] = a;
      }
    }
    __syncthreads();
    PS_MARK(1);

    // ---- edge phase: wave -> (destination t, sub-wave wi); lane -> columns (2*lane, 2*lane+1)
    // Chunks of CH edges; per chunk: scores (pass 1) -> running max / rescale -> exp -> weighted sums (pass 2).
    // Every row read is a fully coalesced 512-byte line per wave (r~ rows, k rows, v rows).
    {
      float* sc = un + (size_t)t * CH * 8;
      const int t_beg = (r < Nd) ? ldgi(st.toff + r) : 0;
      constexpr bool k4 = KR != 3;   // the fourth rel-PE column block exists (condition rows)
      // B operands of the score MFMAs, built once per destination and layer: lane -> column n = lane & 15
      // (head n & 7, hi half for n < 8 / lo half for n >= 8), k-block lane >> 4 (8 consecutive columns)
      half8 bq[4], bk[4];
      {
        const int n = lane & 15, hB = n & 7, kqB = lane >> 4;
        const bool lo = n >= 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const float* qp = big + (size_t)(t * 8 + hB) * QP + 32 * ks + 8 * kqB;
          const float4 v0 = *reinterpret_cast<const float4*>(qp), v1 = *reinterpret_cast<const float4*>(qp + 4);
          const float qv_[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          // k rows: column block 32*ks + 8*kq lies inside head 2*ks + (kq >> 1); only that head's q is non-zero
          const bool mine = (2 * ks + (kqB >> 1)) == hB;
          const float* kp = qb + t * 128 + 32 * ks + 8 * kqB;
          const float4 w0 = *reinterpret_cast<const float4*>(kp), w1 = *reinterpret_cast<const float4*>(kp + 4);
          const float kv_[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            bq[ks][j] = lo ? f16_lo(qv_[j]) : f16_hi(qv_[j]);
            const float kk = mine ? kv_[j] : 0.f;
            bk[ks][j] = lo ? f16_lo(kk) : f16_hi(kk);
          }
        }
      }
      const float cqm = cq[t * 8 + (lane & 7)];
      __syncthreads();   // q~ (in `big`) is in registers now: the k staging area [f1, big) may overwrite it
      float4 av = make_float4(0.f, 0.f, 0.f, 0.f);   // a_v partial: columns 4*(lane & 31)..+3, edges of parity lane >> 5
      const int hv = (lane & 31) >> 2;               // ... which belong to head hv
      float m_run[8], l_run[8];
      floatx4 ar[8];   // a_r as MFMA accumulators: [column block][row 4*(lane>>4)+r = (p hi | p lo) x head]
#pragma unroll
      for (int h = 0; h < 8; ++h) { ar[h] = floatx4{0.f, 0.f, 0.f, 0.f}; m_run[h] = -INFINITY; l_run[h] = 0.f; }
      // the chunk count must be uniform over the workgroup (barriers inside): max degree of its T rows
      int dmax = 0;
#pragma unroll
      for (int tt = 0; tt < T; ++tt) {
        const int rr_ = row0 + tt;
        const int d_ = ((rr_ < Nd) && !(flags & 1)) ? (ldgi(st.eoff + rr_ + 1) - ldgi(st.eoff + rr_)) : 0;
        dmax = d_ > dmax ? d_ : dmax;
      }
      PS_MARK(2);
      for (int c0 = 0; c0 < dmax; c0 += CH) {
        const int cn = (deg - c0) < CH ? (deg - c0) : CH;   // edges of this destination in the chunk (may be <= 0)
        if (c0 > 0) {   // source rows of a later chunk (the barrier that closed the previous chunk freed `el`)
          for (int e = wi * 64 + lane; e < cn; e += 64 * W) el[e] = ldgi(st.esrc + e_beg + c0 + e);
          __syncthreads();
        }
        // pass 1 on the matrix cores: S[16 edges][8 heads] = R~[16 x 128] Q~^T + K[16 x 128] blockdiag(q)   (:88-90)
        // v_mfma_f32_16x16x32_f16 with split-fp16 operands: the 16-wide N carries (q hi | q lo) for the 8
        // heads and A runs over (r~ hi, r~ lo, k hi, k lo), so all four hi/lo cross terms are summed in the
        // fp32 accumulator: 16 MFMAs + 16 16-byte loads per 16 edges, no conversion work in the loop.
        {
          const int mi = lane & 15, kq = lane >> 4;
          // k rows are gathered by source.  Loaded straight into the A-fragment shape (lane = edge + 16*kq) the
          // 4 lanes of a quad would read 4 different rows and the load issues at a quarter of the rate; so the
          // gather uses lane = 4*row + piece (each quad reads 64 contiguous bytes) and the 16 x 256 B half rows
          // turn into fragments through a wave-private LDS staging area (row stride 272 B).
          const int rq = lane >> 2, pq = lane & 3;
          half8* stw = reinterpret_cast<half8*>(f1 + wave * 1088) + rq * 17 + pq;
          const half8* str = reinterpret_cast<const half8*>(f1 + wave * 1088) + mi * 17 + kq;
          half8 nrh[4], nrl[4], nkh[4], nkl[4];
          auto gather1 = [&](int eb) {
            const int blk = (c0 + eb) >> 4;
            const _Float16* ra = (flags & 32) ? st.rtA + (size_t)((t_beg + (blk >> 1)) & 7) * 8192 + (blk & 1) * 4096 + lane * 8
                                              : st.rtA + (size_t)(t_beg + (blk >> 1)) * 8192 + (blk & 1) * 4096 + lane * 8;
            const int e = (eb + rq < cn) ? eb + rq : cn - 1;
            const _Float16* kp = st.khl + (size_t)((flags & 64) ? (e & 15) : el[e]) * 256 + 8 * pq;   // 64: ablation, cached rows
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
              if (ks < 3 || k4) {
                nrh[ks] = ldgh8(ra + 512 * ks);
                nrl[ks] = ldgh8(ra + 2048 + 512 * ks);
              }
              nkh[ks] = ldgh8(kp + 32 * ks);
              nkl[ks] = ldgh8(kp + 128 + 32 * ks);
            }
          };
          if (PFE && wi * 16 < cn && !(flags & 16)) gather1(wi * 16);
          for (int eb = wi * 16; eb < cn && !(flags & 16); eb += 16 * W) {
            if (!PFE) gather1(eb);   // two workgroups per CU hide the latency instead of a second register set
            // all 16 row loads are in flight before anything waits (left alone, the scheduler sinks each load
            // next to its use: one round trip per MFMA)
            __builtin_amdgcn_sched_barrier(0);
            half8 arh[4], arl[4], akh[4], akl[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { arh[ks] = nrh[ks]; arl[ks] = nrl[ks]; }
            if (flags & 128) {   // ablation: no LDS staging (wrong fragments, timing only)
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) { akh[ks] = nkh[ks]; akl[ks] = nkl[ks]; }
            } else {
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) stw[4 * ks] = nkh[ks];
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) akh[ks] = str[4 * ks];
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) stw[4 * ks] = nkl[ks];
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) akl[ks] = str[4 * ks];
            }
            if (PFE && eb + 16 * W < cn) gather1(eb + 16 * W);   // the next tile's rows fly under this tile's MFMAs
            floatx4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
              if (ks < 3 || k4) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(arh[ks], bq[ks], acc, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(akh[ks], bk[ks], acc2, 0, 0, 0);
              if (ks < 3 || k4) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(arl[ks], bq[ks], acc, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(akl[ks], bk[ks], acc2, 0, 0, 0);
            }
            acc += acc2;
            // D[row = 4*(lane>>4) + r][col = lane & 15]: columns h and h + 8 (the lo half of q) meet by row_ror:8
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              const float v = acc[r4] + dpp_xor8(acc[r4]);
              const int er = eb + 4 * kq + r4;
              if (mi < 8 && er < cn) sc[(size_t)er * 8 + mi] = (v + cqm) * 0.25f;
            }
          }
        }
        PS_MARK(3);
        __syncthreads();
        PS_MARK(4);
        // online softmax over the destination's edges, per head (torch_geometric.utils.softmax:
        // max-shift, exp, / (sum + 1e-16)); every wave of the destination finds the chunk max
        float mc[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) mc[h] = m_run[h];
        for (int e = lane; e < cn; e += 64) {
          const float4 a = *reinterpret_cast<const float4*>(sc + (size_t)e * 8);
          const float4 b = *reinterpret_cast<const float4*>(sc + (size_t)e * 8 + 4);
          mc[0] = fmaxf(mc[0], a.x); mc[1] = fmaxf(mc[1], a.y); mc[2] = fmaxf(mc[2], a.z); mc[3] = fmaxf(mc[3], a.w);
          mc[4] = fmaxf(mc[4], b.x); mc[5] = fmaxf(mc[5], b.y); mc[6] = fmaxf(mc[6], b.z); mc[7] = fmaxf(mc[7], b.w);
        }
        float scl[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          mc[h] = wave_max(mc[h]);
          scl[h] = (m_run[h] == -INFINITY) ? 0.f : expf(m_run[h] - mc[h]);   // rescale of what is already accumulated
          m_run[h] = mc[h];
        }
        __syncthreads();  // all waves have read the raw scores before any wave overwrites them
        float ls[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) ls[h] = 0.f;
        for (int e = wi * 64 + lane; e < cn; e += 64 * W) {
          float4 a = *reinterpret_cast<const float4*>(sc + (size_t)e * 8);
          float4 b = *reinterpret_cast<const float4*>(sc + (size_t)e * 8 + 4);
          a.x = expf(a.x - mc[0]); a.y = expf(a.y - mc[1]); a.z = expf(a.z - mc[2]); a.w = expf(a.w - mc[3]);
          b.x = expf(b.x - mc[4]); b.y = expf(b.y - mc[5]); b.z = expf(b.z - mc[6]); b.w = expf(b.w - mc[7]);
          ls[0] += a.x; ls[1] += a.y; ls[2] += a.z; ls[3] += a.w; ls[4] += b.x; ls[5] += b.y; ls[6] += b.z; ls[7] += b.w;
          *reinterpret_cast<float4*>(sc + (size_t)e * 8) = a;
          *reinterpret_cast<float4*>(sc + (size_t)e * 8 + 4) = b;
        }
#pragma unroll
        for (int h = 0; h < 8; ++h) l_run[h] = l_run[h] * scl[h] + wave_sum(ls[h]);
        {
          const bool up = (lane >> 4) & 1;   // accumulator row 4*(lane>>4)+r belongs to head 4*((lane>>4)&1)+r
          const float s0 = up ? scl[4] : scl[0], s1 = up ? scl[5] : scl[1], s2 = up ? scl[6] : scl[2], s3 = up ? scl[7] : scl[3];
#pragma unroll
          for (int cb = 0; cb < 8; ++cb) { ar[cb][0] *= s0; ar[cb][1] *= s1; ar[cb][2] *= s2; ar[cb][3] *= s3; }
        }
        {
          float sh = scl[0];
#pragma unroll
          for (int h = 1; h < 8; ++h) sh = (hv == h) ? scl[h] : sh;
          av.x *= sh; av.y *= sh; av.z *= sh; av.w *= sh;
        }
        __syncthreads();
        PS_MARK(5);
        // pass 2a on the matrix cores: a_r[h][c] += sum_e p_e,h r~_e[c]   (:100, aggr='add').  Per 32-edge tile:
        // A[m][k] = (p hi | p lo)[head m & 7][edge k] from the LDS score tile, B[k][n] = r~ (hi, then lo) of 16
        // columns from the edge-minor image (one 16-byte load per fragment, 1 KB contiguous per wave);
        // 8 column blocks x (hi, lo) = 16 MFMAs per tile, all four hi/lo cross terms land in the fp32 accumulators.
        if (!(flags & 8)) {
          const int mA = lane & 15, kqA = lane >> 4;
          const bool loA = mA >= 8;
          const _Float16* tb = st.rtT + (size_t)(t_beg + (c0 >> 5)) * 8192 + mA * 32 + kqA * 8;
          for (int eb = wi * 32; eb < cn; eb += 32 * W) {
            const _Float16* tp = (flags & 32) ? st.rtT + (size_t)((t_beg + ((c0 + eb) >> 5)) & 7) * 8192 + mA * 32 + kqA * 8
                                              : tb + (size_t)(eb >> 5) * 8192;
            half8 bh[8], bl[8];
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
              if (cb < 6 || k4) {
                bh[cb] = ldgh8(tp + cb * 512);
                bl[cb] = ldgh8(tp + 4096 + cb * 512);
              }
            }
            half8 ap;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int ee = eb + 8 * kqA + j;
              const float pv = (ee < cn) ? sc[(size_t)ee * 8 + (mA & 7)] : 0.f;
              ap[j] = loA ? f16_lo(pv) : f16_hi(pv);
            }
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
              if (cb < 6 || k4) {
                ar[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ap, bh[cb], ar[cb], 0, 0, 0);
                ar[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ap, bl[cb], ar[cb], 0, 0, 0);
              }
            }
          }
        }
        PS_MARK(6);
        // pass 2b: a_v[hd] += sum_e p_e,h v_src[hd]: v rows are gathered by SOURCE (no edge-minor image).  Lane ->
        // (edge parity lane >> 5, columns 4*(lane & 31)..+3): a load instruction reads two whole rows as 16-byte
        // pieces, 8 instructions = 16 edges in flight per batch (each batch is one exposed round trip).
        const float* vbase = st.kv + 128 + 4 * (lane & 31);
        const int eh = lane >> 5;
        float4 vvn[8];
        auto gather2 = [&](int eb) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int ee = (eb + 2 * j + eh < cn) ? eb + 2 * j + eh : cn - 1;
            vvn[j] = ldg4(vbase + (size_t)el[ee] * 256);
          }
        };
        if (PFE && wi * 16 < cn) gather2(wi * 16);
        for (int eb = wi * 16; eb < cn && !(flags & 8); eb += 16 * W) {
          if (!PFE) gather2(eb);
          float4 vv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) vv[j] = vvn[j];
          if (PFE && eb + 16 * W < cn) gather2(eb + 16 * W);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int ee = eb + 2 * j + eh;
            const float ph = (ee < cn) ? sc[(size_t)ee * 8 + hv] : 0.f;
            av.x = fmaf(ph, vv[j].x, av.x);
            av.y = fmaf(ph, vv[j].y, av.y);
            av.z = fmaf(ph, vv[j].z, av.z);
            av.w = fmaf(ph, vv[j].w, av.w);
          }
        }
        PS_MARK(7);
        if (c0 + CH < dmax) __syncthreads();   // the next chunk's pass 1 overwrites the score tile
      }
      // the to_v_r fold's weights leave now and land while the partials are published
      wload(wA, (KR == 3 ? w.Wvr_gt3 : w.Wvr_gt) + woff, 128);
      // rows h (p hi) and h + 8 (p lo) sit 32 lanes apart: one half-wave swap folds two column blocks at a time
#pragma unroll
      for (int cb = 0; cb < 8; cb += 2) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const float v = swap_add32(ar[cb][r4], ar[cb + 1][r4]);   // lanes < 32: block cb, lanes >= 32: block cb + 1
          big[(size_t)(wave * 8 + 4 * ((lane >> 4) & 1) + r4) * QP + (cb + (lane >> 5)) * 16 + (lane & 15)] = v;
        }
      }
      {   // even-edge and odd-edge halves meet; lanes 0-31 publish the wave's a_v partial
        av.x += __shfl_xor(av.x, 32); av.y += __shfl_xor(av.y, 32); av.z += __shfl_xor(av.z, 32); av.w += __shfl_xor(av.w, 32);
        if (lane < 32) *reinterpret_cast<float4*>(avp + wave * 128 + 4 * lane) = av;
      }
      if (lane < 8) {
        float v = l_run[0];
#pragma unroll
        for (int h = 1; h < 8; ++h) v = (lane == h) ? l_run[h] : v;
        ml[wave * 16 + 8 + lane] = v;
      }
    }
    if (s + 1 < nsteps) sp_store(sp_next);   // layer s-1's buffer is dead: park the next layer's vectors there
    PS_MARK(8);
    __syncthreads();
    PS_MARK(9);
    if (W > 1) {  // sum the W sub-wave partials of each destination into its first slot
      for (int i = tid; i < T * 8 * 128; i += NT) {
        const int tt = i / 1024, hc = i % 1024, h = hc >> 7, c = hc & 127;
        float a = 0.f;
        for (int j = 0; j < W; ++j) a += big[(size_t)((tt * W + j) * 8 + h) * QP + c];
        big[(size_t)((tt * W) * 8 + h) * QP + c] = a;
      }
      for (int i = tid; i < T * 128; i += NT) {
        const int tt = i >> 7, c = i & 127;
        float a = 0.f;
        for (int j = 0; j < W; ++j) a += avp[(tt * W + j) * 128 + c];
        avp[(tt * W) * 128 + c] = a;
      }
      __syncthreads();
    }
    PS_MARK(10);
    // ---- agg = (a_v + Wvr_g^T a_r + l * vb) / (l + 1e-16)   (to_v_r fold; :89, :100)
    //      columns ncol.. belong to head ncol/16
    {
      float acc[T][4];
      zero_acc<T>(acc);
      PS_STAGE(wA, (KR == 3 ? w.Wvr_gt3 : w.Wvr_gt) + woff, 128, wB, w.Wga_t + woff, 128, big + (size_t)(ncol >> 4) * QP + kgl * RK, W * 8 * QP);   // Wvr
      fold_kgroups<T, LQ>(acc);
      if (kgl == 0) {
        const int h = ncol >> 4;
#pragma unroll
        for (int tt = 0; tt < T; ++tt) {
          float l = 0.f;
          for (int j = 0; j < W; ++j) l += ml[(tt * W + j) * 16 + 8 + h];
          const float inv = 1.f / (l + 1e-16f);
          float o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            o[j] = (avp[(tt * W) * 128 + ncol + j] + acc[tt][j] + l * sp[SP_VB + ncol + j]) * inv;
          *reinterpret_cast<float4*>(ag + tt * 128 + ncol) = make_float4(o[0], o[1], o[2], o[3]);
        }
      }
    }
    __syncthreads();
    // ---- gated update (:106-107): g = sigmoid(Wg [agg | x_dst] + bg); u = agg + g * (to_s(x_dst) - agg)
    {
      float acc[T][4];
      zero_acc<T>(acc);
      PS_STAGE(wB, w.Wga_t + woff, 128, wA, w.Wout_t + woff, 128, ag + kgl * RK, 128);   // Wga
      fold_kgroups<T, LQ>(acc);
      if (kgl == 0) {
#pragma unroll
        for (int tt = 0; tt < T; ++tt) {
          float o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = tt * 128 + ncol + j;
            const float g = 1.f / (1.f + expf(-(acc[tt][j] + gb[i])));
            const float a = ag[i];
            o[j] = a + g * (sb[i] - a);
          }
          *reinterpret_cast<float4*>(f1 + tt * 128 + ncol) = make_float4(o[0], o[1], o[2], o[3]);   // u parked in f1
        }
      }
    }
    __syncthreads();
    PS_MARK(11);
    // ---- x = x + LN_post(to_out(u))  (:76), then xn = LN_ffpre(x)  (:77)
    {
      float acc[T][4];
      zero_acc<T>(acc);
      PS_STAGE(wA, w.Wout_t + woff, 128, wB, w.W1_t + (size_t)(k5 * 4 * RK) * 512 + ncol5, 512, f1 + kgl * RK, 128);   // Wout
      fold_kgroups<T, LQ>(acc);
      if (kgl == 0) {
#pragma unroll
        for (int tt = 0; tt < T; ++tt)
          *reinterpret_cast<float4*>(xn + tt * 128 + ncol) =
              make_float4(acc[tt][0] + sp[SP_BOUT + ncol], acc[tt][1] + sp[SP_BOUT + ncol + 1],
                          acc[tt][2] + sp[SP_BOUT + ncol + 2], acc[tt][3] + sp[SP_BOUT + ncol + 3]);
      }
    }
    __syncthreads();
    for (int tt = wave; tt < T; tt += NW) {
      // one wave per row: LN_post, residual add, LN_ffpre, all in registers
      float* xr = xs + tt * 128;
      float* nr = xn + tt * 128;
      ln_row_wave(nr, nr, sp + SP_LN_POST_W, sp + SP_LN_POST_B, eps, lane, false);
      const float x0 = xr[lane] + nr[lane], x1 = xr[lane + 64] + nr[lane + 64];
      xr[lane] = x0;
      xr[lane + 64] = x1;
      ln_row_wave(xr, nr, sp + SP_LN_FFPRE_W, sp + SP_LN_FFPRE_B, eps, lane, false);
    }
    __syncthreads();
    PS_MARK(12);
    // ---- FFN up: f1 = relu(W1 xn + b1)   N = 512: wave -> 512/NW columns, lane (c5, k5), 4 chunks of RK rows
    {
      float acc[T][4];
      zero_acc<T>(acc);
      const float* w1 = w.W1_t + (size_t)(k5 * 4 * RK) * 512 + ncol5;
      const float* w2 = w.W2_t + (size_t)(kgl * 4 * RK) * 128 + ncol;
      const float* xk = xn + k5 * 4 * RK;
      PS_STAGE(wB, w1, 512, wA, w1 + (size_t)RK * 512, 512, xk, 128);
      PS_STAGE(wA, w1 + (size_t)RK * 512, 512, wB, w1 + (size_t)2 * RK * 512, 512, xk + RK, 128);
      PS_STAGE(wB, w1 + (size_t)2 * RK * 512, 512, wA, w1 + (size_t)3 * RK * 512, 512, xk + 2 * RK, 128);
      PS_STAGE(wA, w1 + (size_t)3 * RK * 512, 512, wB, w2, 128, xk + 3 * RK, 128);
      fold_kgroups<T, L5>(acc);
      if (k5 == 0) {
#pragma unroll
        for (int tt = 0; tt < T; ++tt)
          *reinterpret_cast<float4*>(f1 + tt * 512 + ncol5) =
              make_float4(fmaxf(acc[tt][0] + sp[SP_B1 + ncol5], 0.f), fmaxf(acc[tt][1] + sp[SP_B1 + ncol5 + 1], 0.f),
                          fmaxf(acc[tt][2] + sp[SP_B1 + ncol5 + 2], 0.f), fmaxf(acc[tt][3] + sp[SP_B1 + ncol5 + 3], 0.f));
      }
    }
    __syncthreads();
    // ---- FFN down: xn = W2 f1 + b2   K = 512: 4 chunks of RK rows per k-group
    {
      float acc[T][4];
      zero_acc<T>(acc);
      const float* w2 = w.W2_t + (size_t)(kgl * 4 * RK) * 128 + ncol;
      const float* fk = f1 + kgl * 4 * RK;
      PS_STAGE(wB, w2, 128, wA, w2 + (size_t)RK * 128, 128, fk, 512);
      PS_STAGE(wA, w2 + (size_t)RK * 128, 128, wB, w2 + (size_t)2 * RK * 128, 128, fk + RK, 512);
      PS_STAGE(wB, w2 + (size_t)2 * RK * 128, 128, wA, w2 + (size_t)3 * RK * 128, 128, fk + 2 * RK, 512);
      wfma<T, RK>(wA, fk + 3 * RK, 512, acc);
      // the next layer's first chunk leaves now; it lands during the fold and the two norms
      if (PF && s + 1 < nsteps) wload(wA, steps[s + 1].w.Wq_t + woff, 128);
      fold_kgroups<T, LQ>(acc);
      if (kgl == 0) {
#pragma unroll
        for (int tt = 0; tt < T; ++tt)
          *reinterpret_cast<float4*>(xn + tt * 128 + ncol) =
              make_float4(acc[tt][0] + sp[SP_B2 + ncol], acc[tt][1] + sp[SP_B2 + ncol + 1], acc[tt][2] + sp[SP_B2 + ncol + 2],
                          acc[tt][3] + sp[SP_B2 + ncol + 3]);
      }
    }
    __syncthreads();
    for (int tt = wave; tt < T; tt += NW) {
      // x = x + LN_ffpost(ffn); then the NEXT layer's pre-norm, in registers
      float* xr = xs + tt * 128;
      float* nr = xn + tt * 128;
      ln_row_wave(nr, nr, sp + SP_LN_FFPOST_W, sp + SP_LN_FFPOST_B, eps, lane, false);
      xr[lane] += nr[lane];
      xr[lane + 64] += nr[lane + 64];
      if (s + 1 < nsteps) ln_row_wave(xr, nr, sp_next + SP_LN_DST_W, sp_next + SP_LN_DST_B, eps, lane, false);
    }
    __syncthreads();
    PS_MARK(13);
  }
  for (int i = tid_o; i < T * 128; i += NT) {
    const int t = i >> 7, r = row0 + t;
    if (r < Nd) x[(size_t)r * 128 + (i & 127)] = xs[i];
  }
#undef PS_MARK
}

// k | v projection of source tokens for L layers: kv[l][n][0:128] = Wk LN_src(x_n),
// kv[l][n][128:256] = Wv LN_src(x_n) + bv   (attention_layer.py:61,65,115-116).  grid (ceil(Ns/64), L).
// 64 rows per workgroup: LayerNorm with a lane quad per row, then two 64 x 128 x 128 GEMMs on the matrix cores
// (k half, v half; pn_gemm, split-fp16 operands), each written out as whole 512-byte rows; the k half also
// leaves as split fp16 (hi | lo) for the score MFMAs of k_attn_chain.
constexpr size_t KV_LDS_BYTES = (size_t)2 * PN_ROWS * PN_AS * 2 + (size_t)PN_ROWS * PN_CS * 4;
__global__ __launch_bounds__(WG) void k_kv_proj(const float* __restrict__ x, int Ns, const AttnW* __restrict__ layers,
                                                float* __restrict__ kv, _Float16* __restrict__ khl, size_t layer_stride, float eps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kv_smem[];
  _Float16* Ah = reinterpret_cast<_Float16*>(kv_smem);
  _Float16* Al = Ah + PN_ROWS * PN_AS;
  float* C = reinterpret_cast<float*>(Al + PN_ROWS * PN_AS);
  const AttnW& w = layers[blockIdx.y];
  const int row0 = blockIdx.x * PN_ROWS, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {   // LN_src: thread -> (row tid >> 2, 32-column quarter tid & 3)
    const int r = tid >> 2, c0 = (tid & 3) * 32;
    float a[32];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + r < Ns) v = ldg4(x + (size_t)(row0 + r) * 128 + c0 + 4 * i);
      a[4 * i] = v.x; a[4 * i + 1] = v.y; a[4 * i + 2] = v.z; a[4 * i + 3] = v.w;
    }
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) sm += a[i];
    sm += dpp_xor1(sm);
    sm += dpp_xor2(sm);
    const float mean = sm * (1.f / 128.f);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      a[i] -= mean;
      sq = fmaf(a[i], a[i], sq);
    }
    sq += dpp_xor1(sq);
    sq += dpp_xor2(sq);
    const float rstd = 1.f / sqrtf(sq * (1.f / 128.f) + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      half8 hh, ll;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = fmaf(a[8 * i + j] * rstd, w.ln_src_w[c0 + 8 * i + j], w.ln_src_b[c0 + 8 * i + j]);
        hh[j] = f16_hi(v);
        ll[j] = f16_lo(v);
      }
      *reinterpret_cast<half8*>(Ah + r * PN_AS + c0 + 8 * i) = hh;
      *reinterpret_cast<half8*>(Al + r * PN_AS + c0 + 8 * i) = ll;
    }
  }
  __syncthreads();
  float* out = kv + blockIdx.y * layer_stride;
  _Float16* outh = khl + blockIdx.y * layer_stride;
  for (int half = 0; half < 2; ++half) {
    pn_gemm<4>(Ah, Al, 4, w.Wkv_F + (size_t)half * 8 * 4 * 1024, C, PN_CS, PN_ROWS, wave, lane);
    __syncthreads();
    for (int i = tid; i < PN_ROWS * 32; i += WG) {   // 64 rows x 32 float4
      const int r = i >> 5, c = (i & 31) * 4;
      if (row0 + r >= Ns) continue;
      float4 v = *reinterpret_cast<const float4*>(C + r * PN_CS + c);
      if (half == 1) {
        v.x += w.bkv[128 + c]; v.y += w.bkv[128 + c + 1]; v.z += w.bkv[128 + c + 2]; v.w += w.bkv[128 + c + 3];
      }
      *reinterpret_cast<float4*>(out + (size_t)(row0 + r) * 256 + half * 128 + c) = v;
      if (half == 0) {
        typedef _Float16 half4v __attribute__((ext_vector_type(4)));
        *reinterpret_cast<half4v*>(outh + (size_t)(row0 + r) * 256 + c) = half4v{f16_hi(v.x), f16_hi(v.y), f16_hi(v.z), f16_hi(v.w)};
        *reinterpret_cast<half4v*>(outh + (size_t)(row0 + r) * 256 + 128 + c) = half4v{f16_lo(v.x), f16_lo(v.y), f16_lo(v.z), f16_lo(v.w)};
      }
    }
    __syncthreads();
  }
}

}  // namespace ps
