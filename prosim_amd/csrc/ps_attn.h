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
  // split path (k_node): the node Linears as split-fp16 B fragments [n-tile][k-block][hi|lo][lane 64][8]
  const _Float16* Fqsg;                             // [to_q ; to_s ; to_g's x half]: 24 n-tiles x 4 k-blocks
  const _Float16 *Fkr, *Fkr3;                       // q~: [head 8][n-tile 8 | 6][1 k-block = the 32 q columns holding the head, other head zero]
  const _Float16 *Fvr, *Fvr3;                       // to_v_r fold: [head 8 = n-tile][k-block 4 | 3]
  const _Float16 *Fga, *Fout, *F1, *F2;             // gate (agg half), to_out, FFN up (32 n-tiles), FFN down (16 k-blocks)
  const float* sp;                                  // packed small vectors, SP_* offsets below (2432 floats)
  // row-tile node kernels (ps_rowtile.h): the same Linears with the K index of every 32-wide k-block in the order a result tile hands
  // its features on (Builder::fragments, perm = true), and q~'s per-head 16 x 96 blocks as K = 16 fragments [head 8][c-tile 6][hi|lo][64][4]
  const _Float16 *Wkv_Q, *Fqsg_Q, *Fga_Q, *Fout_Q, *F1_Q, *F2_Q, *Fkr3_x, *Fvr3_Q;   // (lo halves scaled by 2^11, ps_rowtile.h)
};
// offsets (floats) inside AttnW::sp -- one coalesced load per layer stages them in LDS, so no
// bias / LayerNorm-parameter load ever sits on the layer's dependency chain
enum : int { SP_LN_DST_W = 0, SP_LN_DST_B = 128, SP_BQ = 256, SP_BS = 384, SP_BG = 512, SP_KB = 640, SP_VB = 768,
             SP_BOUT = 896, SP_LN_POST_W = 1024, SP_LN_POST_B = 1152, SP_LN_FFPRE_W = 1280, SP_LN_FFPRE_B = 1408,
             SP_B1 = 1536, SP_B2 = 2048, SP_LN_FFPOST_W = 2176, SP_LN_FFPOST_B = 2304, SP_SIZE = 2432 };

// Per-destination vectors exchanged between the split kernels of a layer (k_node <-> k_edge_small):
// the node kernel leaves q, q~, <q, kb> for the edge kernel, which leaves the softmax-weighted sums.
struct EdgeIO {
  float* q;    // [Nd][128]      to_q(LN_dst(x)) + bias
  float* qt;   // [Nd][8][128]   q~[h] = Wkr_g,h^T q_h  (columns >= 32*KR unused)
  float* cq;   // [Nd][8]        <q_h, kb_h>
  float* ar;   // [Nd][8][128]   sum_e p_e,h r~_e
  float* av;   // [Nd][128]      sum_e p_e,h v_src
  float* l;    // [Nd][8]        sum_e p_e,h (softmax denominators, running-max scaled like ar / av)
  float* s;    // [Nd][128]      to_s(LN_dst(x)) + bias      (node kernel only)
  float* g;    // [Nd][128]      to_g's x_dst half + bias     (node kernel only)
  float* m;    // [W][Nd][8]     running maxima of the W partial edge sums of a destination (k_chain16 with W > 1 waves per row)
};

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
  const void* geo;        // [E] EdgeGeo records (ps_chain16.h): what k_chain16 rebuilds the rel-PE rows from (geometric sets)
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
// (round 5: the xor steps as VALU instructions -- xor_add, ps_device.h -- instead of ds_bpermute round trips; the same butterfly, the same bits)
template <int T, int LOW>
__device__ __forceinline__ void fold_kgroups(float (&acc)[T][4]) {
#pragma unroll
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = acc[t][j];
      if (LOW <= 4) v = xor_add<4>(v);
      if (LOW <= 8) v = xor_add<8>(v);
      if (LOW <= 16) v = xor_add<16>(v);
      if (LOW <= 32) v = xor_add<32>(v);
      acc[t][j] = v;
    }
  }
}

// stage helper.  Weight chunks do not depend on activations, so every chunk is requested before the stage that
// multiplies it.  EARLY (T == 1, full register file): the next chunk is requested BEFORE this chunk's FMAs, two
// register sets live.  Otherwise (two workgroups per CU, <= 256 registers): it is requested right AFTER this
// chunk's FMAs, into registers that just died -- one set live, and the request still flies under the fold, the
// LDS write and the barrier that close the stage.  CURP/CURN are kept for readability only.
#define PS_STAGE_(EARLY, CUR, CURP, CURN, NXT, NXTP, NXTN, X, XS) \
  do {                                                    \
    if (EARLY) wload(NXT, NXTP, NXTN);                    \
    wfma<T, RK>(CUR, X, XS, acc);                         \
    if (!(EARLY)) {                                       \
      __builtin_amdgcn_sched_barrier(0);                  \
      wload(NXT, NXTP, NXTN);                             \
    }                                                     \
  } while (0)
#define PS_STAGE(CUR, CURP, CURN, NXT, NXTP, NXTN, X, XS) PS_STAGE_(PF, CUR, CURP, CURN, NXT, NXTP, NXTN, X, XS)
// the one-chunk stages (q, s, g, the fold, the gate, to_out): GEO requests their next chunk early too (PFN)
#define PS_STAGE_N(CUR, CURP, CURN, NXT, NXTP, NXTN, X, XS) PS_STAGE_(PFN, CUR, CURP, CURN, NXT, NXTP, NXTN, X, XS)

// KR: distinct rel-PE column blocks of 32 (3 for geometric edge sets, 4 for condition rows / the test hook); a
// launch only ever chains steps of one kind, so it is a compile-time parameter (ChainStep::kr must agree).
// BIG: compiled for ONE workgroup per CU with the whole register file (512 per lane): weight chunks and edge
// rows are software-prefetched into second register sets.  !BIG (T >= 2 default): two workgroups per CU, <= 256
// registers, the co-resident workgroup hides latency instead.
// POLICY: the per-replan policy launch (12 layers) gets its own symbol, so a kernel trace lists the dominant launch
// apart from the 1-2-layer launches of the same build (profiles/, bench.py's roofline) -- and the XCD-aware block ->
// row mapping (xcd_block): the rows of one scene share an XCD, whose L2 then holds that scene's a2p / m2p k|v rows
// (measured at 8 scenes x 128 agents: 546 -> 525 us).  The generator's launches are left on the plain mapping: with
// 512-neighbour s2p rows the same mapping costs 6 % (every workgroup of an XCD gathers the same rows at once).
// GEO (round 5; T = 1, KR = 3 only): the edge phase works on the 32-byte GEOMETRY RECORDS of k_edge_geo like k_chain16's -- the rel-PE rows are
// rebuilt in registers instead of read from two 384-byte operand images per edge (k_relpe_tiles is not launched for such a set), the k rows of a
// tile arrive by LDS-DMA, every wave runs its own online softmax over the tiles 4 j + wave with no workgroup barrier inside the edge loop, and a
// tile's loads all leave one tile (the first tile's: a whole node stage) ahead: c16_lat_* in ps_chain16.h.  (Phase clocks of the image form on one
// 128-agent scene: the edge phase was 42 % of the launch, one exposed L2 round trip per pass and 16-edge block, and the 1.4 MB of images per layer
// and XCD pushed the layer's weights out of L2.)  The node stages stay the fp32 GEMVs on register-streamed weights; the four waves' partial sums
// are merged like k_chain16's POST half merges them (common maximum, rescale, add).  One workgroup per CU (122 KB of LDS, up to 512 registers).
// LDS behind the kernel's own buffers: [4 wave areas: k staging hi | lo, probability tile, feature tile, source rows][5 slots: the waves' a_r
// (+ l, m), q~][5 rows: q, the waves' a_v].
constexpr int G1_WAVE_FLOATS = 3104, G1_QSL = 808, G1_QH = 100, G1_AGS = 132;
constexpr size_t G1_FLOATS = (size_t)4 * G1_WAVE_FLOATS + 5 * G1_QSL + 5 * G1_AGS;
struct GeoRec { float4 g; float nn; int src; };          // an edge's record as loaded: (a0, a1, a2, rstd), -mean rstd, source row
struct C16LatState { GeoRec r0, r1; float4 vv[8]; float dv[4], rdv[4]; };     // a wave's requests in flight between the calls below; this lane's four Fourier divisors and their reciprocals (made once per launch)
template <int TAG> __device__ void c16_lat_request(const ChainStep* __restrict__ stp, float* g1, int e_beg, int deg, C16LatState& S);
template <int TAG> __device__ void c16_lat_pre(const ChainStep* __restrict__ stp, float* g1, int e_beg, int deg, C16LatState& S);
template <int TAG> __device__ void c16_lat_main(const ChainStep* __restrict__ stp, float* g1, const float* cq, const float* __restrict__ div32, int e_beg, int deg,
                                                C16LatState& S);

template <int T, int NW = 4, int KR = 3, bool BIG = (T == 1), bool POLICY = false, bool GEO = false>
__global__ __launch_bounds__(64 * NW, ((!BIG && NW == 4 && !GEO) ? 2 : 1)) void k_attn_chain(float* __restrict__ x, const float* __restrict__ x_in, int Nd,
                                                     const ChainStep* __restrict__ steps, int nsteps, int maxdeg, float eps, int flags,
                                                     unsigned long long* __restrict__ prof, const float* __restrict__ div32) {
  static_assert(!GEO || (T == 1 && NW == 4 && KR == 3), "the geometry-record edge phase is built for one row on four waves");
  // flags: ablation switches of tools/gpu_ablate.py / gpu_profile.sh (0 in every product launch; results are wrong
  // when set): 1 no edges, 8 skip the aggregation pass, 16 skip the score pass, 32 read the rel-PE images from a
  // cache-resident region, 128 no k staging; 512 plain blockIdx -> row mapping in the policy launch
  // instead of the XCD-aware one (results stay correct).
  // phase clocks (PS_CHAIN_PROF=1; prof == nullptr in every product launch): thread 0 of each
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
  // T == 1: one workgroup per CU with the full register file -> software prefetch (weights one chunk ahead,
  // edge rows one tile ahead).  T >= 2: compiled for 2 workgroups per CU (__launch_bounds__(256, 2), <= 256
  // registers): the co-resident workgroup hides the latency instead and nothing is double-buffered.
  constexpr bool PF = BIG;                     // weight chunks one stage ahead (second register set)
  // ... in the one-chunk stages (q, s, g, fold, gate, to_out).  Not for GEO either: a lane addresses 256 registers, what the launch bound adds
  // beyond them are AGPRs the allocator spills into -- with early requests anywhere (all stages: 0.227 ms, these stages: 0.224) the FFN loop
  // pays v_accvgpr moves per weight register (its phase 137 k -> 160 k cycles) and the launch loses against 0.221 ms
  constexpr bool PFN = BIG;
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
  float* g1 = smem + attn_lds_floats<T, NW>(0);          // GEO: k_chain16's edge-phase areas (G1_FLOATS)
  float* g1qa = g1 + 4 * G1_WAVE_FLOATS;                 //   [5][G1_QSL] slots 0-3: the waves' a_r (+ l, m), 4: q~
  float* g1ag = g1qa + 5 * G1_QSL;                       //   [5][G1_AGS] row 0: q, rows 1-4: the waves' a_v

  const int tid_o = threadIdx.x, wave_o = tid_o >> 6, lane_o = tid_o & 63;
  const int c8_o = lane_o % LQ, kgl_o = lane_o / LQ;
  const int ncol_o = wave_o * CW + 4 * c8_o;
  const size_t woff_o = (size_t)(kgl_o * RK) * 128 + ncol_o;
  const int row0 = xcd_block(blockIdx.x, gridDim.x, !(flags & 1024) || (flags & 512)) * T;
  // two register sets, one chunk in flight behind the one being multiplied.  (Three sets / two
  // chunks in flight measured no faster -- a CU's 4 waves already pull ~100 GB/s, the per-CU
  // L2->register ceiling measured by ps_test_stream -- and spill at T = 4.)
  WC<RK> wA, wB;
  // GEO: this row's edge range in the current layer's set (and the next layer's, loaded a layer early), the edge phase's requests in flight
  C16LatState lat;
  int g_eb = 0, g_dg = 0, g_ebn = 0, g_dgn = 0;
  if constexpr (GEO) {
    if (row0 < Nd && !(flags & 1)) {
      g_eb = ldgi(steps[0].eoff + row0);
      g_dg = ldgi(steps[0].eoff + row0 + 1) - g_eb;
    }
    c16_lat_request<POLICY ? 1 : 2>(steps, g1, g_eb, g_dg, lat);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      lat.dv[j] = ldg1(div32 + 2 * (4 * (lane_o >> 4) + j));
      lat.rdv[j] = 1.0f / lat.dv[j];
    }
  }
  if (PFN) wload(wA, steps[0].w.Wq_t + woff_o, 128);
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
    const int e_beg = GEO ? g_eb : ((r < Nd) ? ldgi(st.eoff + r) : 0);
    const int deg = GEO ? g_dg : (((r < Nd) && !(flags & 1)) ? (ldgi(st.eoff + r + 1) - e_beg) : 0);
    if constexpr (GEO) {
      c16_lat_pre<POLICY ? 1 : 2>(&st, g1, e_beg, deg, lat);   // the first tile's k / v rows and the second tile's records leave now
      g_ebn = g_dgn = 0;
      if (s + 1 < nsteps && r < Nd && !(flags & 1)) {
        g_ebn = ldgi(steps[s + 1].eoff + r);
        g_dgn = ldgi(steps[s + 1].eoff + r + 1) - g_ebn;
      }
    }
    int* el = esl + t * CH;
    if (!GEO)
      for (int e = wi * 64 + lane; e < deg && e < CH; e += 64 * W) el[e] = ldgi(st.esrc + e_beg + e);   // first chunk

    // ---- q / s / gate(x) projections of the pre-normed rows (:61-69, :106-107, :114); xn = LN_dst(x)
    {
      float acc[T][4];
      zero_acc<T>(acc);
      if (!PFN) wload(wA, w.Wq_t + woff, 128);   // (late mode keeps no weights in registers across the layer boundary)
      PS_STAGE_N(wA, w.Wq_t + woff, 128, wB, w.Ws_t + woff, 128, xn + kgl * RK, 128);    // Wq
      fold_kgroups<T, LQ>(acc);
      if (kgl == 0) {
#pragma unroll
        for (int tt = 0; tt < T; ++tt)
        {
          const float4 qv = make_float4(acc[tt][0] + sp[SP_BQ + ncol], acc[tt][1] + sp[SP_BQ + ncol + 1], acc[tt][2] + sp[SP_BQ + ncol + 2],
                                        acc[tt][3] + sp[SP_BQ + ncol + 3]);
          *reinterpret_cast<float4*>(qb + tt * 128 + ncol) = qv;
          if (GEO) *reinterpret_cast<float4*>(g1ag + ncol) = qv;
        }
      }
      zero_acc<T>(acc);
      PS_STAGE_N(wB, w.Ws_t + woff, 128, wA, w.Wgx_t + woff, 128, xn + kgl * RK, 128);   // Ws
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
      PS_STAGE_N(wA, w.Wgx_t + woff, 128, wB, wkr, 128, xn + kgl * RK, 128);   // Wgx
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
        for (int tt = 0; tt < T; ++tt) {
          if (GEO) {
            if (c32 < 24) *reinterpret_cast<float4*>(g1qa + 4 * G1_QSL + h * G1_QH + 4 * c32) = make_float4(acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]);
          } else {
            *reinterpret_cast<float4*>(big + (size_t)(tt * 8 + h) * QP + 4 * c32) =
                make_float4(acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]);
          }
        }
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
    if constexpr (GEO) {
      c16_lat_main<POLICY ? 1 : 2>(&st, g1, cq, div32, e_beg, deg, lat);
      wload(wA, w.Wvr_gt3 + woff, 128);
      if (s + 1 < nsteps) c16_lat_request<POLICY ? 1 : 2>(steps + s + 1, g1, g_ebn, g_dgn, lat);   // the next layer's first records
      g_eb = g_ebn;
      g_dg = g_dgn;
    } else {
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
          // turn into fragments through a wave-private LDS staging area (4 KB, swizzled: below).
          const int rq = lane >> 2, pq = lane & 3;
          // staging layout: 16-byte unit (row, 4 ks + piece) sits at row*16 + ((4 ks + piece) ^ ((row & 3) << 2) ^ (row >> 2)):
          // the quad-major writes (16 lanes = 4 rows x 4 pieces) and the fragment reads (16 lanes = 16 rows, one piece)
          // both touch 16 distinct bank groups (a plain padded row stride serves only one of the two)
          half8* stw = reinterpret_cast<half8*>(f1 + wave * 1024) + rq * 16 + (pq ^ (rq >> 2));
          const half8* str = reinterpret_cast<const half8*>(f1 + wave * 1024) + mi * 16 + (kq ^ (mi >> 2));
          const int swa = rq & 3, sra = mi & 3;
          half8 nrh[4], nrl[4], nkh[4], nkl[4];
          auto gather1 = [&](int eb) {
            const int blk = (c0 + eb) >> 4;
            const _Float16* ra = (flags & 32) ? st.rtA + (size_t)((t_beg + (blk >> 1)) & 7) * 8192 + (blk & 1) * 4096 + lane * 8
                                              : st.rtA + (size_t)(t_beg + (blk >> 1)) * 8192 + (blk & 1) * 4096 + lane * 8;
            const int e = (eb + rq < cn) ? eb + rq : cn - 1;
            const _Float16* kp = st.khl + (size_t)el[e] * 256 + 8 * pq;
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
              for (int ks = 0; ks < 4; ++ks) stw[4 * (ks ^ swa)] = nkh[ks];
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) akh[ks] = str[4 * (ks ^ sra)];
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) stw[4 * (ks ^ swa)] = nkl[ks];
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) akl[ks] = str[4 * (ks ^ sra)];
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
        av.x = xor_add<32>(av.x); av.y = xor_add<32>(av.y); av.z = xor_add<32>(av.z); av.w = xor_add<32>(av.w);
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
    if constexpr (GEO) {   // the four waves' partial softmax sums: common maximum, rescale, add (c16_node_phase's merge).  32 threads per head.
      const int h = tid >> 5, l5 = tid & 31;
      float sc[4];
      {
        float mp[4], mm = -INFINITY;
#pragma unroll
        for (int p = 0; p < 4; ++p) { mp[p] = g1qa[p * G1_QSL + h * G1_QH + 97]; mm = fmaxf(mm, mp[p]); }
#pragma unroll
        for (int p = 0; p < 4; ++p) sc[p] = (mp[p] == -INFINITY) ? 0.f : exp2f(mp[p] - mm);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = l5 + 32 * k;
        float a = 0.f;
        if (k < 3) {
#pragma unroll
          for (int p = 0; p < 4; ++p) a = fmaf(g1qa[p * G1_QSL + h * G1_QH + c], sc[p], a);
        }
        big[(size_t)h * QP + c] = a;   // (columns 96 .. 127: the fold's weights there are folded onto 64 .. 95)
      }
      if (l5 <= 16) {   // a_v columns of head h, and its l
        float a = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p) a = fmaf(l5 < 16 ? g1ag[(1 + p) * G1_AGS + h * 16 + l5] : g1qa[p * G1_QSL + h * G1_QH + 96], sc[p], a);
        if (l5 < 16) avp[h * 16 + l5] = a;
        else { ml[8 + h] = a; ml[16 + 8 + h] = 0.f; ml[32 + 8 + h] = 0.f; ml[48 + 8 + h] = 0.f; }
      }
      __syncthreads();
    } else if (W > 1) {  // sum the W sub-wave partials of each destination into its first slot
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
      PS_STAGE_N(wA, (KR == 3 ? w.Wvr_gt3 : w.Wvr_gt) + woff, 128, wB, w.Wga_t + woff, 128, big + (size_t)(ncol >> 4) * QP + kgl * RK, W * 8 * QP);   // Wvr
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
      PS_STAGE_N(wB, w.Wga_t + woff, 128, wA, w.Wout_t + woff, 128, ag + kgl * RK, 128);   // Wga
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
      PS_STAGE_N(wA, w.Wout_t + woff, 128, wB, w.W1_t + (size_t)(k5 * 4 * RK) * 512 + ncol5, 512, f1 + kgl * RK, 128);   // Wout
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
      if (PFN && s + 1 < nsteps) wload(wA, steps[s + 1].w.Wq_t + woff, 128);
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
        ll[j] = f16_los(v);
      }
      *reinterpret_cast<half8*>(Ah + r * PN_AS + c0 + 8 * i) = hh;
      *reinterpret_cast<half8*>(Al + r * PN_AS + c0 + 8 * i) = ll;
    }
  }
  __syncthreads();
  float* out = kv + blockIdx.y * layer_stride;
  _Float16* outh = khl + blockIdx.y * layer_stride;
  // (round 5: a launch with few workgroups -- one scene's 128 rows are two -- puts the k half and the v half on workgroups of their own,
  // gridDim.z = 2: the same arithmetic, half the dependent work per workgroup; 12.5 -> ~9 us for the 25 such launches of a single-scene rollout)
  const int half0 = gridDim.z == 2 ? (int)blockIdx.z : 0, half1 = gridDim.z == 2 ? half0 + 1 : 2;
  for (int half = half0; half < half1; ++half) {
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

// ------------------------------------------------------------------------------------------------------------
// Split path, node side.  k_node carries 16 destination rows per workgroup through everything of a layer that is
// not per-edge, as split-fp16 MFMA GEMMs against pre-split weight fragments (pn_gemm's operand scheme):
//   POST(layer l):  to_v_r fold, agg, gate, u, to_out, LN_post + residual, LN_ffpre, FFN, LN_ffpost + residual
//   PRE(layer l+1): LN_dst, q | s | g projections, q~[h] = Wkr_g,h^T q_h, <q_h, kb_h>
// Between two k_node launches the edge kernel (k_edge_small) turns (q, q~, cq) into (ar, av, l).
// Weights cross the CU once per 16 rows (the fused chain: once per 1-4), the K reductions happen inside the MFMA
// (no shuffle folds), and with ~100 registers the fragment loads are software-pipelined one group ahead.
// ND_CS: the 128-column GEMM results; ND_CW: the PRE half's wide results (q | s | g = 384 columns, k | v = 256), which
// live in the FFN planes' memory (dead outside the FFN) -- 75 KB in all, so two workgroups (or one and a 78 KB
// attention-chain workgroup of another pipelined rollout) share a CU.
constexpr int ND_ROWS = 16, ND_AS = 136, ND_AS5 = 520, ND_CS = 132, ND_CW = 388, ND_XS = 132;
static_assert((size_t)ND_ROWS * ND_CW * 4 <= (size_t)2 * ND_ROWS * ND_AS5 * 2, "wide results must fit the FFN planes");
constexpr size_t ND_LDS_BYTES = (size_t)2 * ND_ROWS * ND_AS * 2 + (size_t)2 * ND_ROWS * ND_AS5 * 2 + (size_t)ND_ROWS * ND_CS * 4 +
                                (size_t)2 * ND_ROWS * ND_XS * 4 + (size_t)SP_SIZE * 4;

// C[16 x 16*ntiles] = A[16 x 32*K32] * W.  A: LDS planes (row stride `as` halfs); F: [n-tile][k-block K32][hi|lo][64][8].
// Wave w makes tiles w, w+4, ...; its work is a list of (tile, 4-k-block) groups of <= 8 fragment loads + <= 12 MFMAs.
// Weights do not depend on activations, so a ring of ND_DEPTH groups stays in flight ACROSS the epilogues and
// barriers between GEMMs: frag_prefetch() requests the first groups of the next GEMM as soon as the ring is free,
// gemm16() consumes group g and requests group g + ND_DEPTH.  (One group ahead hid ~10 % of a 1-2 us round trip: a group's
// MFMAs take 0.1 us.)  Round 3: with every fragment load made an L1 hit the node phases are only 10 % faster -- the
// dependent stage chain (GEMM -> LDS -> barrier -> epilogue -> LDS -> barrier) bounds them, not the weight fetch -- so
// k_node keeps a ring of 2 (214 registers) and TWO workgroups share a CU (2 x 75 KB of LDS): one's stage waits are the
// other's issue slots.  Same bits, +1.8 % on the pipelined benchmark over a ring of 4 at one workgroup per CU.
constexpr int ND_DEPTH = 2;
template <int DEPTH_>
struct FragRingT {
  static constexpr int DEPTH = DEPTH_;
  half8 h[DEPTH_][4], l[DEPTH_][4];
};
typedef FragRingT<ND_DEPTH> FragRing;
template <int K32, class RingT, int NWV = 4>
__device__ __forceinline__ void frag_issue(RingT& R, int slot, const _Float16* __restrict__ F, int g, int wave, int lane) {
  constexpr int KB = K32 < 4 ? K32 : 4, KG = (K32 + KB - 1) / KB;
#ifdef PS_ABL_FRAG0   // timing ablation: every group of every wave reads the GEMM's first 8 KB (L1 hits)
  const _Float16* f = F + lane * 8;
#else
  const _Float16* f = F + ((size_t)(wave + NWV * (g / KG)) * K32 + (g % KG) * KB) * 1024 + lane * 8;
#endif
#ifdef PS_ABL_NOFRAG   // timing ablation: no fragment loads at all (the ring's registers keep what they hold)
  (void)f;
#pragma unroll
  for (int j = 0; j < KB; ++j) asm volatile("" : "+v"(R.h[slot][j]), "+v"(R.l[slot][j]));
#else
#pragma unroll
  for (int j = 0; j < KB; ++j) {
    R.h[slot][j] = ldgh8(f + j * 1024);
    R.l[slot][j] = ldgh8(f + j * 1024 + 512);
  }
#endif
}
template <int K32, class RingT, int NWV = 4>
__device__ __forceinline__ void frag_prefetch(RingT& R, const _Float16* __restrict__ F, int ntiles, int wave, int lane) {
  constexpr int KB = K32 < 4 ? K32 : 4, KG = (K32 + KB - 1) / KB;
  const int G = ((ntiles - wave + NWV - 1) / NWV) * KG;
#pragma unroll
  for (int d = 0; d < RingT::DEPTH; ++d)
    if (d < G) frag_issue<K32, RingT, NWV>(R, d, F, d, wave, lane);
}
// ph / pl (optional): instead of C, a finished tile leaves as relu(acc + bias) in split-fp16 planes (row stride ps) -- the
// FFN-up result goes straight into the A operand of the FFN-down GEMM.
template <int K32, class RingT, int NWV = 4>
__device__ __forceinline__ void gemm16(RingT& R, const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al, int as,
                                       const _Float16* __restrict__ F, int ntiles, float* __restrict__ C, int cs, int wave, int lane,
                                       _Float16* __restrict__ ph = nullptr, _Float16* __restrict__ pl = nullptr, int ps = 0,
                                       const float* __restrict__ bias = nullptr) {
  constexpr int KB = K32 < 4 ? K32 : 4, KG = (K32 + KB - 1) / KB;
  const int mi = lane & 15, kq = lane >> 4;
  const int G = ((ntiles - wave + NWV - 1) / NWV) * KG;
  // three accumulators (hi*hi, hi*lo, lo*hi): the MFMAs of a k-block do not wait for one another
  floatx4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = acc, acc2 = acc;
  constexpr int DEPTH = RingT::DEPTH;
  for (int g0 = 0; g0 < G; g0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int g = g0 + d;
      if (g < G) {
        half8 ch[KB], cl[KB];
#pragma unroll
        for (int j = 0; j < KB; ++j) { ch[j] = R.h[d][j]; cl[j] = R.l[d][j]; }
        if (g + DEPTH < G) frag_issue<K32, RingT, NWV>(R, d, F, g + DEPTH, wave, lane);
        const int kg = g % KG;
        if (kg == 0) { acc = floatx4{0.f, 0.f, 0.f, 0.f}; acc1 = acc; acc2 = acc; }
#pragma unroll
        for (int j = 0; j < KB; ++j) {
          const half8 ah = *reinterpret_cast<const half8*>(Ah + mi * as + (kg * KB + j) * 32 + kq * 8);
          const half8 al = *reinterpret_cast<const half8*>(Al + mi * as + (kg * KB + j) * 32 + kq * 8);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ch[j], acc, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, cl[j], acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, ch[j], acc2, 0, 0, 0);
        }
        if (kg == KG - 1) acc += (acc1 + acc2) * PS_LO_INV;   // (the cross products carry the lo halves' 2^11)
        if (kg == KG - 1) {
          const int nt = wave + NWV * (g / KG);
          if (ph) {
            const float bv = bias[nt * 16 + mi];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = fmaxf(acc[r] + bv, 0.f);
              ph[(4 * kq + r) * ps + nt * 16 + mi] = f16_hi(v);
              pl[(4 * kq + r) * ps + nt * 16 + mi] = f16_los(v);
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(4 * kq + r) * cs + nt * 16 + mi] = acc[r];
          }
        }
      }
    }
  }
}

// ---- round 6: the same GEMM with everything about the fragment traffic known at compile time.  The node phases of k_chain16 see a layer's
// weights as ONE stream of fragment groups ("items": <= 8 loads of 16 B per lane each) through a ring of DEPTH slots -- item i lives in slot
// i % DEPTH, and consuming item i requests item i + DEPTH whatever GEMM that belongs to (issue(integral_constant<i + DEPTH>)).  With the
// group count a constant the loops unroll into straight-line code, and hipcc's waits become exact (`s_waitcnt vmcnt(16)`: this group is in,
// the two behind it stay in flight).  With a run-time group count every consumption sat behind `s_waitcnt vmcnt(0)`: the ring drained in
// bursts of DEPTH groups and a burst's last request had the whole L2 round trip exposed.  Same MFMAs in the same order: same bits as gemm16.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// tile(integral_constant<j>, nt, acc): the wave's j-th finished tile (columns 16 nt .. 16 nt + 15; acc[r] = row 4 kq + r, column 16 nt + mi) -- the
// elementwise epilogues of the node phases work on the accumulators where they are (no C round trip, no barrier)
template <int K32, int NTC, int NWV, int ITEM0, class RingT, class Issue, class Tile>
__device__ __forceinline__ void gemm16t(RingT& R, const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al, int as, int wave, int lane,
                                        Issue&& issue, Tile&& tile) {
  constexpr int KB = K32 < 4 ? K32 : 4, KG = (K32 + KB - 1) / KB, G = NTC / NWV * KG, DEPTH = RingT::DEPTH;
  static_assert(NTC % NWV == 0, "every wave makes the same number of tiles");
  const int mi = lane & 15, kq = lane >> 4;
  floatx4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = acc, acc2 = acc;
  static_for<0, G>([&](auto gc) {
    constexpr int g = decltype(gc)::value, slot = (ITEM0 + g) % DEPTH, kg = g % KG;
    half8 ch[KB], cl[KB];
#pragma unroll
    for (int j = 0; j < KB; ++j) { ch[j] = R.h[slot][j]; cl[j] = R.l[slot][j]; }
    issue(std::integral_constant<int, ITEM0 + g + DEPTH>{});
    if (kg == 0) { acc = floatx4{0.f, 0.f, 0.f, 0.f}; acc1 = acc; acc2 = acc; }
#pragma unroll
    for (int j = 0; j < KB; ++j) {
      const half8 ah = *reinterpret_cast<const half8*>(Ah + mi * as + (kg * KB + j) * 32 + kq * 8);
      const half8 al = *reinterpret_cast<const half8*>(Al + mi * as + (kg * KB + j) * 32 + kq * 8);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ch[j], acc, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, cl[j], acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, ch[j], acc2, 0, 0, 0);
    }
    if constexpr (kg == KG - 1) {
      acc += (acc1 + acc2) * PS_LO_INV;   // (the cross products carry the lo halves' 2^11)
      tile(std::integral_constant<int, g / KG>{}, wave + NWV * (g / KG), acc);
    }
  });
}
template <int K32, int NTC, int NWV, int ITEM0, class RingT, class Issue>
__device__ __forceinline__ void gemm16s(RingT& R, const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al, int as, float* __restrict__ C, int cs,
                                        int wave, int lane, Issue&& issue, _Float16* __restrict__ ph = nullptr, _Float16* __restrict__ pl = nullptr,
                                        int ps = 0, const float* __restrict__ bias = nullptr) {
  const int mi = lane & 15, kq = lane >> 4;
  gemm16t<K32, NTC, NWV, ITEM0>(R, Ah, Al, as, wave, lane, issue, [&](auto, int nt, const floatx4& acc) {
    if (ph) {   // relu(acc + bias) as split-fp16 planes: the FFN-up result is the FFN-down GEMM's A operand
      const float bv = bias[nt * 16 + mi];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = fmaxf(acc[r] + bv, 0.f);
        ph[(4 * kq + r) * ps + nt * 16 + mi] = f16_hi(v);
        pl[(4 * kq + r) * ps + nt * 16 + mi] = f16_los(v);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) C[(4 * kq + r) * cs + nt * 16 + mi] = acc[r];
    }
  });
}

// sum over the 16 lanes that share an epilogue row (lanes tid & 15 vary)
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_xor1(v);
  v += dpp_xor2(v);
  v = xor_add<4>(v);
  v = xor_add<8>(v);
  return v;
}
// y = LayerNorm(a[0..8)) over a 128-wide row held by 16 lanes x 8 columns (columns c0..c0+7)
__device__ __forceinline__ void row16_ln(float (&a)[8], const float* __restrict__ w, const float* __restrict__ b, int c0, float eps) {
  float sm = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) sm += a[i];
  const float mean = row16_sum(sm) * (1.f / 128.f);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] -= mean;
    sq = fmaf(a[i], a[i], sq);
  }
  const float rstd = 1.f / sqrtf(row16_sum(sq) * (1.f / 128.f) + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = fmaf(a[i] * rstd, w[c0 + i], b[c0 + i]);
}
__device__ __forceinline__ void planes_store8(_Float16* __restrict__ Ph, _Float16* __restrict__ Pl, const float (&a)[8]) {
  half8 h, l;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    h[i] = f16_hi(a[i]);
    l[i] = f16_los(a[i]);
  }
  *reinterpret_cast<half8*>(Ph) = h;
  *reinterpret_cast<half8*>(Pl) = l;
}

// kv_out / khl_out (optional, PRE of a SELF-attention layer, where LN_src == LN_dst): the rows are also the
// layer's sources, so their k | v projection (k_kv_proj's job) is one more GEMM on the normed rows already in LDS.
template <int KR>
__global__ __launch_bounds__(256, 2) void k_node(float* __restrict__ x, int Nd, const ChainStep* __restrict__ post,
                                              const ChainStep* __restrict__ pre, EdgeIO io, float eps,
                                              float* __restrict__ kv_out, _Float16* __restrict__ khl_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char nd_smem[];
  _Float16* P0h = reinterpret_cast<_Float16*>(nd_smem);
  _Float16* P0l = P0h + ND_ROWS * ND_AS;
  _Float16* P1h = P0l + ND_ROWS * ND_AS;
  _Float16* P1l = P1h + ND_ROWS * ND_AS5;
  float* C = reinterpret_cast<float*>(P1l + ND_ROWS * ND_AS5);   // [16][132] results of the 128-column GEMMs
  float* Cw = reinterpret_cast<float*>(P1h);                     // [16][388] wide results of the PRE half (FFN planes' memory)
  float* X = C + ND_ROWS * ND_CS;        // [16][132] residual stream
  float* AG = X + ND_ROWS * ND_XS;       // [16][132] agg (POST) / q (PRE)
  float* sp = AG + ND_ROWS * ND_XS;      // [SP_SIZE] the layer's small vectors
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mi = lane & 15, kq = lane >> 4;
  const int row0 = blockIdx.x * ND_ROWS;
  const int er = tid >> 4, ec = (tid & 15) * 8;   // epilogue mapping: row, 8 columns
  const int grow = row0 + er;
  const bool live = grow < Nd;
  auto stage_sp = [&](const float* __restrict__ src) {
    for (int i = tid; i < SP_SIZE / 4; i += 256) *reinterpret_cast<float4*>(sp + 4 * i) = ldg4(src + 4 * i);
  };
  {
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (live) { v0 = ldg4(x + (size_t)grow * 128 + ec); v1 = ldg4(x + (size_t)grow * 128 + ec + 4); }
    *reinterpret_cast<float4*>(X + er * ND_XS + ec) = v0;
    *reinterpret_cast<float4*>(X + er * ND_XS + ec + 4) = v1;
  }
  FragRing R;
  if (post) {
    const AttnW& w = post->w;
    stage_sp(w.sp);
    frag_prefetch<4>(R, w.Fga, 8, wave, lane);
    // this thread's slice of the edge kernel's / previous node kernel's rows: requested now, used after the fold
    float4 in_av0 = make_float4(0.f, 0.f, 0.f, 0.f), in_av1 = in_av0, in_g0 = in_av0, in_g1 = in_av0, in_s0 = in_av0, in_s1 = in_av0;
    float in_l = 0.f;
    if (live) {
      in_l = ldg1(io.l + (size_t)grow * 8 + (ec >> 4));
      in_av0 = ldg4(io.av + (size_t)grow * 128 + ec); in_av1 = ldg4(io.av + (size_t)grow * 128 + ec + 4);
      in_g0 = ldg4(io.g + (size_t)grow * 128 + ec); in_g1 = ldg4(io.g + (size_t)grow * 128 + ec + 4);
      in_s0 = ldg4(io.s + (size_t)grow * 128 + ec); in_s1 = ldg4(io.s + (size_t)grow * 128 + ec + 4);
    }
    // ---- to_v_r fold: C[row][16h + d] = sum_c a_r[row][h][c] * Wvr_g[c][16h + d]; wave -> heads 2w, 2w+1; the A
    //      fragments come straight from the edge kernel's fp32 rows
    {
      const _Float16* Fv = KR == 3 ? w.Fvr3 : w.Fvr;
      float4 a0[2][KR], a1[2][KR];
      half8 bh[2][KR], bl[2][KR];
#pragma unroll
      for (int t = 0; t < 2; ++t) {   // every load of both heads is in flight before the first MFMA
        const int h = 2 * wave + t;
#pragma unroll
        for (int ks = 0; ks < KR; ++ks) {
          a0[t][ks] = make_float4(0.f, 0.f, 0.f, 0.f);
          a1[t][ks] = a0[t][ks];
          if (row0 + mi < Nd) {
            const float* ap = io.ar + (size_t)(row0 + mi) * 1024 + h * 128 + ks * 32 + kq * 8;
            a0[t][ks] = ldg4(ap);
            a1[t][ks] = ldg4(ap + 4);
          }
          const _Float16* f = Fv + ((size_t)(h * KR + ks) * 2) * 512 + lane * 8;
          bh[t][ks] = ldgh8(f);
          bl[t][ks] = ldgh8(f + 512);
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int h = 2 * wave + t;
        floatx4 acc = {0.f, 0.f, 0.f, 0.f}, acx = acc;
#pragma unroll
        for (int ks = 0; ks < KR; ++ks) {
          const float av_[8] = {a0[t][ks].x, a0[t][ks].y, a0[t][ks].z, a0[t][ks].w, a1[t][ks].x, a1[t][ks].y, a1[t][ks].z, a1[t][ks].w};
          half8 ah, al;
#pragma unroll
          for (int j = 0; j < 8; ++j) { ah[j] = f16_hi(av_[j]); al[j] = f16_los(av_[j]); }
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[t][ks], acc, 0, 0, 0);
          acx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[t][ks], acx, 0, 0, 0);
          acx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[t][ks], acx, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) C[(4 * kq + r) * ND_CS + h * 16 + mi] = fmaf(acx[r], PS_LO_INV, acc[r]);
      }
    }
    __syncthreads();
    // ---- agg = (a_v + fold + l * vb) / (l + 1e-16)   (:89, :100)
    float agg[8];
    {
      const float l = in_l;
      const float inv = 1.f / (l + 1e-16f);
      const float avv[8] = {in_av0.x, in_av0.y, in_av0.z, in_av0.w, in_av1.x, in_av1.y, in_av1.z, in_av1.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) agg[i] = (avv[i] + C[er * ND_CS + ec + i] + l * sp[SP_VB + ec + i]) * inv;
      planes_store8(P0h + er * ND_AS + ec, P0l + er * ND_AS + ec, agg);
    }
    __syncthreads();
    // ---- gated update (:106-107): g = sigmoid(Wg [agg | x_dst] + bg); u = agg + g * (to_s(x_dst) - agg)
    gemm16<4>(R, P0h, P0l, ND_AS, w.Fga, 8, C, ND_CS, wave, lane);
    frag_prefetch<4>(R, w.Fout, 8, wave, lane);
    __syncthreads();
    {
      const float gv[8] = {in_g0.x, in_g0.y, in_g0.z, in_g0.w, in_g1.x, in_g1.y, in_g1.z, in_g1.w};
      const float sv[8] = {in_s0.x, in_s0.y, in_s0.z, in_s0.w, in_s1.x, in_s1.y, in_s1.z, in_s1.w};
      float u[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float g = 1.f / (1.f + expf(-(C[er * ND_CS + ec + i] + gv[i])));
        u[i] = agg[i] + g * (sv[i] - agg[i]);
      }
      __syncthreads();   // every thread has read its gate columns of C and P0 is no longer an operand
      planes_store8(P0h + er * ND_AS + ec, P0l + er * ND_AS + ec, u);
    }
    __syncthreads();
    // ---- x = x + LN_post(to_out(u))  (:76), then LN_ffpre(x)  (:77)
    gemm16<4>(R, P0h, P0l, ND_AS, w.Fout, 8, C, ND_CS, wave, lane);
    frag_prefetch<4>(R, w.F1, 32, wave, lane);
    __syncthreads();
    {
      float o[8], xv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = C[er * ND_CS + ec + i] + sp[SP_BOUT + ec + i];
      row16_ln(o, sp + SP_LN_POST_W, sp + SP_LN_POST_B, ec, eps);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        xv[i] = X[er * ND_XS + ec + i] + o[i];
        X[er * ND_XS + ec + i] = xv[i];
      }
      row16_ln(xv, sp + SP_LN_FFPRE_W, sp + SP_LN_FFPRE_B, ec, eps);
      __syncthreads();
      planes_store8(P0h + er * ND_AS + ec, P0l + er * ND_AS + ec, xv);
    }
    __syncthreads();
    // ---- FFN: relu(W1 . + b1), W2 . + b2, x = x + LN_ffpost(.)
    gemm16<4>(R, P0h, P0l, ND_AS, w.F1, 32, nullptr, 0, wave, lane, P1h, P1l, ND_AS5, sp + SP_B1);
    frag_prefetch<16>(R, w.F2, 8, wave, lane);
    __syncthreads();
    gemm16<16>(R, P1h, P1l, ND_AS5, w.F2, 8, C, ND_CS, wave, lane);
    if (pre && !kv_out) frag_prefetch<4>(R, pre->w.Fqsg, 24, wave, lane);
    __syncthreads();
    {
      float y[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) y[i] = C[er * ND_CS + ec + i] + sp[SP_B2 + ec + i];
      row16_ln(y, sp + SP_LN_FFPOST_W, sp + SP_LN_FFPOST_B, ec, eps);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        y[i] += X[er * ND_XS + ec + i];
        X[er * ND_XS + ec + i] = y[i];
      }
      if (live) {
        *reinterpret_cast<float4*>(x + (size_t)grow * 128 + ec) = make_float4(y[0], y[1], y[2], y[3]);
        *reinterpret_cast<float4*>(x + (size_t)grow * 128 + ec + 4) = make_float4(y[4], y[5], y[6], y[7]);
      }
    }
    __syncthreads();
  }
  if (pre) {
    const AttnW& w = pre->w;
    stage_sp(w.sp);
    if (!post && !kv_out) frag_prefetch<4>(R, w.Fqsg, 24, wave, lane);
    if (kv_out) frag_prefetch<4>(R, w.Wkv_F, 16, wave, lane);   // (a POST in the same launch left the ring empty)
    __syncthreads();
    // ---- xn = LN_dst(x); q | s | g = [Wq ; Ws ; Wg_x] xn + bias  (:61-69, :106-107, :114)
    {
      float xn[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) xn[i] = X[er * ND_XS + ec + i];
      row16_ln(xn, sp + SP_LN_DST_W, sp + SP_LN_DST_B, ec, eps);
      planes_store8(P0h + er * ND_AS + ec, P0l + er * ND_AS + ec, xn);
    }
    __syncthreads();
    if (kv_out) {   // k | v of these rows as sources (attention_layer.py:61,65,115-116)
      gemm16<4>(R, P0h, P0l, ND_AS, w.Wkv_F, 16, Cw, ND_CW, wave, lane);
      frag_prefetch<4>(R, w.Fqsg, 24, wave, lane);
      __syncthreads();
      if (live) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          float v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = Cw[er * ND_CW + half * 128 + ec + i] + (half ? w.bkv[128 + ec + i] : 0.f);
          float* o = kv_out + (size_t)grow * 256 + half * 128 + ec;
          *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
          if (half == 0) {
            half8 hh, ll;
#pragma unroll
            for (int i = 0; i < 8; ++i) { hh[i] = f16_hi(v[i]); ll[i] = f16_lo(v[i]); }
            *reinterpret_cast<half8*>(khl_out + (size_t)grow * 256 + ec) = hh;
            *reinterpret_cast<half8*>(khl_out + (size_t)grow * 256 + 128 + ec) = ll;
          }
        }
      }
      __syncthreads();
    }
    gemm16<4>(R, P0h, P0l, ND_AS, w.Fqsg, 24, Cw, ND_CW, wave, lane);
    frag_prefetch<1>(R, KR == 3 ? w.Fkr3 : w.Fkr, 8 * 2 * KR, wave, lane);
    __syncthreads();
    {
      float q[8], sv[8], gv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        q[i] = Cw[er * ND_CW + ec + i] + sp[SP_BQ + ec + i];
        sv[i] = Cw[er * ND_CW + 128 + ec + i] + sp[SP_BS + ec + i];
        gv[i] = Cw[er * ND_CW + 256 + ec + i] + sp[SP_BG + ec + i];
        AG[er * ND_XS + ec + i] = q[i];
      }
      planes_store8(P0h + er * ND_AS + ec, P0l + er * ND_AS + ec, q);
      if (live) {
        *reinterpret_cast<float4*>(io.q + (size_t)grow * 128 + ec) = make_float4(q[0], q[1], q[2], q[3]);
        *reinterpret_cast<float4*>(io.q + (size_t)grow * 128 + ec + 4) = make_float4(q[4], q[5], q[6], q[7]);
        *reinterpret_cast<float4*>(io.s + (size_t)grow * 128 + ec) = make_float4(sv[0], sv[1], sv[2], sv[3]);
        *reinterpret_cast<float4*>(io.s + (size_t)grow * 128 + ec + 4) = make_float4(sv[4], sv[5], sv[6], sv[7]);
        *reinterpret_cast<float4*>(io.g + (size_t)grow * 128 + ec) = make_float4(gv[0], gv[1], gv[2], gv[3]);
        *reinterpret_cast<float4*>(io.g + (size_t)grow * 128 + ec + 4) = make_float4(gv[4], gv[5], gv[6], gv[7]);
      }
    }
    __syncthreads();
    // ---- q~[row][h][c] = sum_d q[row][16h + d] * Wkr_g[16h + d][c]: (head, 16-column tile) pairs over the waves;
    //      A = the 32-column block of q that holds the head (the fragment zeroes the other head's 16 rows)
    {
      constexpr int NTQ = 2 * KR;
      const _Float16* Fk = KR == 3 ? w.Fkr3 : w.Fkr;
      const int G = (8 * NTQ - wave + 3) / 4;   // one k-block per tile: group g = tile wave + 4 g
      for (int g0 = 0; g0 < G; g0 += ND_DEPTH) {
#pragma unroll
        for (int d = 0; d < ND_DEPTH; ++d) {
          const int g = g0 + d;
          if (g < G) {
            const half8 bh = R.h[d][0], bl = R.l[d][0];
            if (g + ND_DEPTH < G) frag_issue<1>(R, d, Fk, g + ND_DEPTH, wave, lane);
            const int t = wave + 4 * g, h = t / NTQ, nt = t - h * NTQ;
            const half8 ah = *reinterpret_cast<const half8*>(P0h + mi * ND_AS + (h >> 1) * 32 + kq * 8);
            const half8 al = *reinterpret_cast<const half8*>(P0l + mi * ND_AS + (h >> 1) * 32 + kq * 8);
            floatx4 acc = {0.f, 0.f, 0.f, 0.f}, acx = acc;
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
            acx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acx, 0, 0, 0);
            acx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acx, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (row0 + 4 * kq + r < Nd) io.qt[(size_t)(row0 + 4 * kq + r) * 1024 + h * 128 + nt * 16 + mi] = fmaf(acx[r], PS_LO_INV, acc[r]);
          }
        }
      }
      if (tid < 128) {   // cq[row][h] = <q_h, kb_h>
        const int r = tid >> 3, h = tid & 7;
        float a = 0.f;
        for (int d = 0; d < DH; ++d) a = fmaf(AG[r * ND_XS + h * DH + d], sp[SP_KB + h * DH + d], a);
        if (row0 + r < Nd) io.cq[(size_t)(row0 + r) * 8 + h] = a;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Split path, edge side, for SMALL neighbourhoods (kNN graphs: degree <= 128 = 4 tiles; the scene encoder's s2s
// layers have exactly 32).  One WAVE per destination, four destinations per workgroup, no cross-wave barrier: all
// scores of the destination sit in registers (<= 8 blocks of 16 edges), the softmax is a few cross-lane steps, the
// probabilities go through 4 KB of wave-private LDS to become MFMA A fragments, and the sums leave to global memory
// for k_node.  Same operand scheme as k_attn_chain's edge phase (rtA / rtT images, quad-contiguous k gather).
// MAXB: score blocks of 16 edges a destination can have (2 for the 32-neighbour s2s graphs -> 4 waves per SIMD).
constexpr int ES_MAXDEG = 128;
template <int MAXB>
constexpr size_t es_lds_bytes() { return (size_t)4 * (MAXB * 16 * 8 * 4 + 16 * 16 * 16); }

template <int KR, int MAXB>
__global__ __launch_bounds__(256, (MAXB <= 2 ? 4 : 2)) void k_edge_small(int Nd, const ChainStep* __restrict__ step, EdgeIO io) {
  constexpr int ES_DEG = MAXB * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char es_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* P = reinterpret_cast<float*>(es_smem) + wave * (ES_DEG * 8);                       // [deg][8] probabilities
  half8* stg = reinterpret_cast<half8*>(es_smem + 4 * ES_DEG * 8 * 4) + wave * (16 * 16);   // k staging (swizzled, as in k_attn_chain)
  const ChainStep& st = *step;
  const int r = blockIdx.x * 4 + wave;
  if (r >= Nd) return;   // (no barriers in this kernel)
  const int mi = lane & 15, kq = lane >> 4;
  const int e_beg = ldgi(st.eoff + r);
  const int deg = min(ldgi(st.eoff + r + 1) - e_beg, ES_DEG);
  const int t_beg = ldgi(st.toff + r);
  // B operands of the score MFMAs (as in k_attn_chain): lane -> column n = lane & 15 (head n & 7, hi | lo half), k-block kq
  half8 bq[KR], bk[4];
  float cqm;
  {
    const int hB = mi & 7;
    const bool lo = mi >= 8;
    const float* qtp = io.qt + (size_t)r * 1024 + hB * 128 + 8 * kq;
    const float* qp = io.q + (size_t)r * 128 + 8 * kq;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < KR) {
        const float4 v0 = ldg4(qtp + 32 * ks), v1 = ldg4(qtp + 32 * ks + 4);
        const float qv_[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) bq[ks < KR ? ks : 0][j] = lo ? f16_lo(qv_[j]) : f16_hi(qv_[j]);
      }
      const bool mine = (2 * ks + (kq >> 1)) == hB;   // the 8 columns 32 ks + 8 kq lie inside head 2 ks + (kq >> 1)
      const float4 w0 = ldg4(qp + 32 * ks), w1 = ldg4(qp + 32 * ks + 4);
      const float kv_[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float kk = mine ? kv_[j] : 0.f;
        bk[ks][j] = lo ? f16_lo(kk) : f16_hi(kk);
      }
    }
    cqm = ldg1(io.cq + (size_t)r * 8 + hB);
  }
  // ---- scores: block b = edges 16 b ..+15; lane (mi < 8, kq) ends with the scores of head mi, edges 16 b + 4 kq + r4
  const int nb = (deg + 15) >> 4;
  float sreg[MAXB][4];
  const int rq = lane >> 2, pq = lane & 3;
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) sreg[b][r4] = -INFINITY;
    if (b < nb) {
      const _Float16* ra = st.rtA + (size_t)(t_beg + (b >> 1)) * 8192 + (b & 1) * 4096 + lane * 8;
      const int e = min(16 * b + rq, deg - 1);
      const _Float16* kp = st.khl + (size_t)ldgi(st.esrc + e_beg + e) * 256 + 8 * pq;
      half8 arh[KR], arl[KR], nkh[4], nkl[4], akh[4], akl[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks < KR) {
          arh[ks < KR ? ks : 0] = ldgh8(ra + 512 * ks);
          arl[ks < KR ? ks : 0] = ldgh8(ra + 2048 + 512 * ks);
        }
        nkh[ks] = ldgh8(kp + 32 * ks);
        nkl[ks] = ldgh8(kp + 128 + 32 * ks);
      }
      __builtin_amdgcn_sched_barrier(0);
      half8* stw = stg + rq * 16 + (pq ^ (rq >> 2));
      const half8* str = stg + mi * 16 + (kq ^ (mi >> 2));
      const int swa = rq & 3, sra = mi & 3;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) stw[4 * (ks ^ swa)] = nkh[ks];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) akh[ks] = str[4 * (ks ^ sra)];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) stw[4 * (ks ^ swa)] = nkl[ks];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) akl[ks] = str[4 * (ks ^ sra)];
      floatx4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks < KR) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(arh[ks < KR ? ks : 0], bq[ks < KR ? ks : 0], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(akh[ks], bk[ks], acc2, 0, 0, 0);
        if (ks < KR) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(arl[ks < KR ? ks : 0], bq[ks < KR ? ks : 0], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(akl[ks], bk[ks], acc2, 0, 0, 0);
      }
      acc += acc2;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float v = acc[r4] + dpp_xor8(acc[r4]);   // columns h and h + 8 (q hi | q lo)
        if (16 * b + 4 * kq + r4 < deg) sreg[b][r4] = (v + cqm) * 0.25f;
      }
    }
  }
  // ---- softmax per head (torch_geometric.utils.softmax: max-shift, exp, / (sum + 1e-16)): the head's scores live in the
  //      four lanes mi = h (+ the duplicates mi = h + 8), kq = 0..3
  float m = -INFINITY;
#pragma unroll
  for (int b = 0; b < MAXB; ++b)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) m = fmaxf(m, sreg[b][r4]);
  m = xor_max<16>(m);
  m = xor_max<32>(m);
  float lsum = 0.f;
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    if (b < nb) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int e = 16 * b + 4 * kq + r4;
        const float p = (e < deg) ? expf(sreg[b][r4] - m) : 0.f;
        lsum += p;
        if (mi < 8 && e < deg) P[e * 8 + mi] = p;
      }
    }
  }
  lsum = xor_add<16>(lsum);
  lsum = xor_add<32>(lsum);
  if (lane < 8) io.l[(size_t)r * 8 + lane] = lsum;
  // ---- aggregation: a_r on the matrix cores per 32-edge tile (A = p hi | lo x head, B = rtT), a_v on the VALU
  floatx4 ar[2 * KR];
#pragma unroll
  for (int cb = 0; cb < 2 * KR; ++cb) ar[cb] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int ntile = (deg + 31) >> 5;
  const bool loA = mi >= 8;
  for (int t = 0; t < ntile; ++t) {
    const _Float16* tp = st.rtT + (size_t)(t_beg + t) * 8192 + mi * 32 + kq * 8;
    half8 bh[2 * KR], bl[2 * KR];
#pragma unroll
    for (int cb = 0; cb < 2 * KR; ++cb) {
      bh[cb] = ldgh8(tp + cb * 512);
      bl[cb] = ldgh8(tp + 4096 + cb * 512);
    }
    half8 ap;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ee = 32 * t + 8 * kq + j;
      const float pv = (ee < deg) ? P[ee * 8 + (mi & 7)] : 0.f;
      ap[j] = loA ? f16_lo(pv) : f16_hi(pv);
    }
#pragma unroll
    for (int cb = 0; cb < 2 * KR; ++cb) {
      ar[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ap, bh[cb], ar[cb], 0, 0, 0);
      ar[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ap, bl[cb], ar[cb], 0, 0, 0);
    }
  }
  float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    const float* vbase = st.kv + 128 + 4 * (lane & 31);
    const int eh = lane >> 5, hv = (lane & 31) >> 2;
    for (int eb = 0; eb < deg; eb += 16) {
      float4 vv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ee = min(eb + 2 * j + eh, deg - 1);
        vv[j] = ldg4(vbase + (size_t)ldgi(st.esrc + e_beg + ee) * 256);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ee = eb + 2 * j + eh;
        const float ph = (ee < deg) ? P[ee * 8 + hv] : 0.f;
        av.x = fmaf(ph, vv[j].x, av.x);
        av.y = fmaf(ph, vv[j].y, av.y);
        av.z = fmaf(ph, vv[j].z, av.z);
        av.w = fmaf(ph, vv[j].w, av.w);
      }
    }
    av.x = xor_add<32>(av.x); av.y = xor_add<32>(av.y); av.z = xor_add<32>(av.z); av.w = xor_add<32>(av.w);
    if (lane < 32) *reinterpret_cast<float4*>(io.av + (size_t)r * 128 + 4 * lane) = av;
  }
  // rows h (p hi) and h + 8 (p lo) sit 32 lanes apart: one half-wave swap folds two column blocks at a time
#pragma unroll
  for (int cb = 0; cb < 2 * KR; cb += 2) {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const float v = swap_add32(ar[cb][r4], ar[cb + 1][r4]);   // lanes < 32: block cb, lanes >= 32: block cb + 1
      io.ar[(size_t)r * 1024 + (4 * ((lane >> 4) & 1) + r4) * 128 + (cb + (lane >> 5)) * 16 + (lane & 15)] = v;
    }
  }
}

}  // namespace ps
