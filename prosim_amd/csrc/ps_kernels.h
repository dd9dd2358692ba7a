// PointNet polyline encoder, neighbour search, relative-PE, prompt/condition encoders and the
// per-replan state kernels (K1-K5, K9-K12 of SURVEY.md section 2a) for gfx950.
#pragma once
#include "ps_device.h"

namespace ps {

// ------------------------------------------------------------------------------------------
// K1  PointNetPolylineEncoder.forward (prosim/models/scene_encoder/pointnet_encoder.py:24-62)
// Masked points contribute 0 to both max-pools (:40-44, :49-53), exactly as the reference's zero-filled
// scatter buffers do.
struct PointNetW {
  int in_dim, n_pre, n_mid;             // n_pre = NUM_PRE_LAYERS, n_mid = NUM_MLP_LAYERS - NUM_PRE_LAYERS
  const float* pre_Wt[4];               // [K][128] K-major; K = in_dim for layer 0, else 128
  const float* pre_b[4];
  const float* pre_lnw[4];              // nullptr on the last pre layer (Linear + ReLU only)
  const float* pre_lnb[4];
  const float* mid_Wt[4];               // layer 0: [256][128] (rows 0..127 point feature, 128..255 pooled)
  const float* mid_b[4];
  const float* mid_lnw[4];
  const float* mid_lnb[4];
  const float *out_W0t, *out_b0, *out_W1t, *out_b1;  // out_mlps: Linear, ReLU, Linear (no norm)
  // the per-point Linear weights once more as split-fp16 MFMA B fragments: [n-tile 8][k-block K/32][hi|lo][lane 64][8]
  // (lane = column n + 16*kq holds k = 32*ks + 8*kq ..+8; K padded to 32 with zeros; mid layer 0: rows 0..127 only)
  const _Float16* pre_F[4];
  const _Float16* mid_F[4];
  const _Float16 *mid_P, *out_F0, *out_F1;   // mlps[0] rows 128..255 (pooled half), out_mlps Linear 0 / 1
  // the same Linears for the row-tile kernel (ps_rowtile.h): fragments whose K index runs in the order a result tile hands its
  // features on (k-block ks, lane group kq, element i <-> input feature 32 ks + 16 (i >> 2) + 4 kq + (i & 3)); layer 0 of pre_mlps
  // reads the input rows in the natural order and keeps pre_F[0]
  const _Float16* pre_Q[4];
  const _Float16* mid_Q[4];
  const _Float16 *mid_PQ, *out_Q0, *out_Q1;
};

// The encoder on the matrix cores.  A workgroup (4 waves) takes G = 64 / P polylines = up to 64
// point rows.  Every per-point Linear is a [64 x K] x [K x 128] GEMM: the rows live in LDS as split-fp16 planes
// (A operand, hi | lo), the weights stream from L2 as pre-split B fragments (1 KB contiguous per wave load),
// and hi*hi + hi*lo + lo*hi accumulate in fp32 (v_mfma_f32_16x16x32_f16; the dropped lo*lo term is ~2^-22).
// Wave w owns output columns 16*w..+15 and 16*(w+4)..+15 for all four 16-row tiles.  Bias / LayerNorm / ReLU
// and the two max-pools run on the fp32 result in LDS; the pooled half of mlps[0] and out_mlps are per-polyline
// 16-row GEMMs of the same kind.  (The first version -- one 128-thread workgroup per polyline, VALU, all weights
// re-streamed for 20 rows -- took 800 us for 8192 map polylines; this one takes 275 us.)
constexpr int PN_G = 8;   // (PN_ROWS / PN_AS / PN_CS: ps_device.h, shared with the k|v projection)
constexpr size_t PN_LDS_BYTES = (size_t)2 * PN_ROWS * PN_AS * 2 + (size_t)PN_ROWS * PN_CS * 4 + (size_t)2 * PN_G * 128 * 4 + PN_ROWS * 4;

// bias (per column, or per polyline and column) + LayerNorm + ReLU (or ReLU alone) on the fp32 rows, which
// also become the next GEMM's A planes.  All 64 rows at once: thread -> (row tid >> 2, 32-column quarter
// tid & 3); the LayerNorm sums meet inside the lane quad (two DPP steps, no wave-wide reduction chain).
__device__ __forceinline__ void pn_epilogue(float* __restrict__ C, _Float16* __restrict__ Ah, _Float16* __restrict__ Al,
                                            const float* __restrict__ bias, const float* __restrict__ pbias, int P,
                                            const float* __restrict__ lnw, const float* __restrict__ lnb, float eps) {
  const int r = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * 32;
  const float* bb = pbias ? pbias + (r / P < PN_G ? r / P : 0) * 128 : bias;
  float a[32];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float4 v = *reinterpret_cast<const float4*>(C + r * PN_CS + c0 + 4 * i);
    a[4 * i] = v.x + bb[c0 + 4 * i];
    a[4 * i + 1] = v.y + bb[c0 + 4 * i + 1];
    a[4 * i + 2] = v.z + bb[c0 + 4 * i + 2];
    a[4 * i + 3] = v.w + bb[c0 + 4 * i + 3];
  }
  if (lnw) {
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
    for (int i = 0; i < 32; ++i) a[i] = fmaf(a[i] * rstd, lnw[c0 + i], lnb[c0 + i]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float4 v = make_float4(fmaxf(a[4 * i], 0.f), fmaxf(a[4 * i + 1], 0.f), fmaxf(a[4 * i + 2], 0.f), fmaxf(a[4 * i + 3], 0.f));
    *reinterpret_cast<float4*>(C + r * PN_CS + c0 + 4 * i) = v;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    half8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = fmaxf(a[8 * i + j], 0.f);
      h[j] = f16_hi(v);
      l[j] = f16_los(v);
    }
    *reinterpret_cast<half8*>(Ah + r * PN_AS + c0 + 8 * i) = h;
    *reinterpret_cast<half8*>(Al + r * PN_AS + c0 + 8 * i) = l;
  }
}

__global__ __launch_bounds__(256) void k_pointnet_mfma(PointNetW w, const float* __restrict__ pts, const uint8_t* __restrict__ pmask,
                                                      const int* __restrict__ rows, int n_rows, int P, int feat_mask_dim,
                                                      float* __restrict__ out, float eps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pn_smem[];
  _Float16* Ah = reinterpret_cast<_Float16*>(pn_smem);
  _Float16* Al = Ah + PN_ROWS * PN_AS;
  float* C = reinterpret_cast<float*>(Al + PN_ROWS * PN_AS);
  float* pooled = C + PN_ROWS * PN_CS;    // [PN_G][128]
  float* pb = pooled + PN_G * 128;        // [PN_G][128]
  int* valid = reinterpret_cast<int*>(pb + PN_G * 128);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = min(PN_G, PN_ROWS / P);
  const int g0 = blockIdx.x * G;
  const int Cin = w.in_dim;
  if (tid < PN_ROWS) {
    const int g = tid / P, p = tid - g * P;
    int v = 0;
    if (g < G && g0 + g < n_rows) {
      const int row = rows ? rows[g0 + g] : g0 + g;
      if (feat_mask_dim == 0) v = pmask[(size_t)row * P + p];
      else {
        v = 1;
        for (int f = 0; f < feat_mask_dim; ++f) v &= pmask[((size_t)row * P + p) * feat_mask_dim + f];
      }
    }
    valid[tid] = v;
  }
  __syncthreads();
  for (int i = tid; i < PN_ROWS * 32; i += 256) {   // input rows, K padded to 32, masked points zero-filled
    const int r = i >> 5, k = i & 31;
    const int g = r / P, p = r - g * P;
    float xv = 0.f;
    if (k < Cin && valid[r]) {
      const int row = rows ? rows[g0 + g] : g0 + g;
      xv = pts[((size_t)row * P + p) * Cin + k];
    }
    Ah[r * PN_AS + k] = f16_hi(xv);
    Al[r * PN_AS + k] = f16_los(xv);
  }
  __syncthreads();
  // ---- pre_mlps
  for (int l = 0; l < w.n_pre; ++l) {
    pn_gemm<4>(Ah, Al, l == 0 ? 1 : 4, w.pre_F[l], C, PN_CS, PN_ROWS, wave, lane);
    __syncthreads();
    pn_epilogue(C, Ah, Al, w.pre_b[l], nullptr, P, w.pre_lnw[l], w.pre_lnb[l], eps);
    __syncthreads();
  }
  // per-polyline rows (pooled features, then out_mlps' hidden row) as A planes of a 16-row GEMM; they live in
  // C's memory, which is dead between a max-pool and the next full GEMM
  _Float16* Ph = reinterpret_cast<_Float16*>(C);
  _Float16* Pl = Ph + 16 * PN_AS;
  auto pool_to_planes = [&]() {   // max over the polyline's points of the zero-filled feature buffer (:47, :53)
    float m[8];   // 16 x 128 values over 256 threads
    int cnt = 0;
#pragma unroll
    for (int i = tid; i < 16 * 128; i += 256, ++cnt) {
      const int g = i >> 7, col = i & 127;
      float mm = 0.f;
      if (g < G)
        for (int p = 0; p < P; ++p) {
          const float v = valid[g * P + p] ? C[(g * P + p) * PN_CS + col] : 0.f;
          mm = p == 0 ? v : fmaxf(mm, v);
        }
      m[cnt] = mm;
    }
    __syncthreads();   // every thread has read its C values: the planes may overwrite them
    cnt = 0;
#pragma unroll
    for (int i = tid; i < 16 * 128; i += 256, ++cnt) {
      const int g = i >> 7, col = i & 127;
      Ph[g * PN_AS + col] = f16_hi(m[cnt]);
      Pl[g * PN_AS + col] = f16_los(m[cnt]);
    }
    __syncthreads();
  };
  pool_to_planes();
  // ---- mlps: layer 0 consumes cat(point feature, pooled): the pooled half is a per-polyline bias  (:48-50)
  for (int l = 0; l < w.n_mid; ++l) {
    if (l == 0) {
      pn_gemm<1>(Ph, Pl, 4, w.mid_P, pb, 128, PN_G, wave, lane);   // pb[g] = pooled[g] W[128:256]  (+ bias below)
      __syncthreads();
      for (int i = tid; i < PN_G * 128; i += 256) pb[i] += w.mid_b[0][i & 127];
    }
    pn_gemm<4>(Ah, Al, 4, w.mid_F[l], C, PN_CS, PN_ROWS, wave, lane);
    __syncthreads();
    pn_epilogue(C, Ah, Al, w.mid_b[l], l == 0 ? pb : nullptr, P, w.mid_lnw[l], w.mid_lnb[l], eps);
    __syncthreads();
  }
  // max-pool (:53), then out_mlps (:57): Linear, ReLU, Linear on the pooled rows
  pool_to_planes();
  pn_gemm<1>(Ph, Pl, 4, w.out_F0, pb, 128, PN_G, wave, lane);
  __syncthreads();
  for (int i = tid; i < 16 * 128; i += 256) {
    const int g = i >> 7, col = i & 127;
    const float a = g < PN_G ? fmaxf(pb[g * 128 + col] + w.out_b0[col], 0.f) : 0.f;
    Ph[g * PN_AS + col] = f16_hi(a);
    Pl[g * PN_AS + col] = f16_los(a);
  }
  __syncthreads();
  pn_gemm<1>(Ph, Pl, 4, w.out_F1, pb, 128, PN_G, wave, lane);
  __syncthreads();
  for (int i = tid; i < G * 128; i += 256) {
    const int g = i >> 7, col = i & 127;
    if (g0 + g < n_rows) {
      bool any = false;   // a polyline without a valid point keeps a zero feature (pointnet_encoder.py:56-60)
      for (int p = 0; p < P; ++p) any |= valid[g * P + p] != 0;
      out[(size_t)(g0 + g) * 128 + col] = any ? pb[i] + w.out_b1[col] : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------
// K2-K4 neighbour search.  torch_cluster (third-party, absent from /root/reference, unpinned:
// install_local_env.sh:4) -- semantics of its CUDA kernels restated: per query scan the
// candidates of the same scene in index order; d2 = dx*dx + dy*dy rounded per op (the file is
// built with -ffp-contract=off); radius keeps the first `cap` with d2 < r*r; knn keeps the k
// smallest by (d2, index).  Candidates of scene b = tokens [r1[2b], r1[2b+1]) then [r2[2b], r2[2b+1]): begin / end pairs,
// so that the replicas of one scene (ps_set_replicas) can all name the SAME map token range.
struct CandSet {
  const float* pos;   // [n_tokens][2]
  const int* r1;      // [B][2] first range per scene (global token indices)
  const int* r2;      // [B][2] second range per scene or nullptr
};

__device__ __forceinline__ float dist2(float ax, float ay, float bx, float by) {
  const float dx = ax - bx, dy = ay - by;
  return dx * dx + dy * dy;  // unfused: -ffp-contract=off
}

// One wave per query.  mode 0: count only -> cnt[q];  mode 1: fill esrc/edst at eoff[q].
// self_base >= 0: candidates and queries are the same set (radius_graph, loop=False): query q
// is token self_base + q, the scan keeps cap+1 matches and then drops the self match.
// Up to two edge sets over the SAME queries go out in one launch (blockIdx.y = set): the policy's a2p and m2p
// sets are built from the same agent poses every replan, and each tiny launch costs ~5 us.
struct RadSet {
  CandSet cs;
  float r2;
  int cap, self_base;
  int* cnt;
  int *eoff, *toff, *tdst, *esrc, *edst;
  // optional candidate filter: candidate token i takes part iff cand_ok[i - cand_base] != 0 (tokens below cand_base
  // always do).  p2p: only POLICY agents are prompts; a2p: log-replay agents that dropped out of the log are no tokens.
  const int* cand_ok;
  int cand_base;
};
struct RadSets {
  RadSet s[2];
  int scanned;   // != 0: k_exclusive_scan already made eoff / toff (large query counts, CSR_PREFIX_MAX_Q); the fill pass reads them
};
// up to this many queries the fill pass sweeps cnt[0..q) itself (one graph node less per search); beyond, the sweep's O(nq^2) loads
// (3e7 at 8192 queries, 5e9 at the 100 k rows check_c16_offsets still admits: ADVICE round 4) cost more than the scan launch
constexpr int CSR_PREFIX_MAX_Q = 16384;
// A query's CSR and tile offsets from the counts of the queries before it (round 4: the fill pass computes its own exclusive prefix --
// one strided sweep of cnt[0..q) per wave, a few thousand integers at most -- and writes eoff / toff itself; the separate one-workgroup
// scan launch between the count and the fill pass is gone: one dependent graph node less per neighbour search).
__device__ __forceinline__ void csr_prefix(const int* __restrict__ cnt, int q, int nq, int lane, int* __restrict__ eoff, int* __restrict__ toff,
                                           int& e0, int& t0, int& mine, int scanned = 0) {
  if (scanned) {   // (k_exclusive_scan ran between the count and the fill pass)
    mine = cnt[q];
    e0 = eoff[q];
    t0 = toff[q];
    return;
  }
  int se = 0, stl = 0;
  // (eight independent loads per trip: rolled one by one, the sweep was up to sixteen exposed L2 round trips per wave at 1024 queries)
  for (int i0 = lane; i0 < q; i0 += 512) {
    int c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = i0 + 64 * u < q ? cnt[i0 + 64 * u] : 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      se += c[u];
      stl += (c[u] + 31) >> 5;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    se += __shfl_xor(se, o);
    stl += __shfl_xor(stl, o);
  }
  mine = cnt[q];
  e0 = se;
  t0 = stl;
  if (lane == 0) {
    eoff[q] = se;
    toff[q] = stl;
    if (q == nq - 1) {
      eoff[nq] = se + mine;
      toff[nq] = stl + ((mine + 31) >> 5);
    }
  }
}
constexpr int RAD_U = 4;
template <int MODE>
__global__ void k_radius(RadSets sets, const float* __restrict__ qpos, const int* __restrict__ qscene, int nq) {
  const RadSet& S = sets.s[blockIdx.y];
  const CandSet cs = S.cs;
  const float r2 = S.r2;
  const int cap = S.cap, self_base = S.self_base;
  const int* __restrict__ cand_ok = S.cand_ok;
  const int cand_base = S.cand_base;
  int* __restrict__ cnt = S.cnt;
  int* __restrict__ tdst = S.tdst;
  int* __restrict__ esrc = S.esrc;
  int* __restrict__ edst = S.edst;
  const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (q >= nq) return;
  int base_e = 0;
  if (MODE == 1) {   // this query's offsets, and the tile -> destination map of the rel-PE operand images (<= 25 tiles per destination)
    int t0, mine;
    csr_prefix(cnt, q, nq, lane, S.eoff, S.toff, base_e, t0, mine, sets.scanned);
    const int nt = (mine + 31) >> 5;
    if (lane < nt) tdst[t0 + lane] = q;
  }
  const float qx = qpos[2 * q], qy = qpos[2 * q + 1];
  const int b = qscene[q];
  const int capx = cap + (self_base >= 0 ? 1 : 0);
  // (a query that is filtered out as a candidate has no self match to drop)
  const int self = (self_base >= 0 && (!cand_ok || cand_ok[self_base + q - cand_base])) ? self_base + q : -1;
  int run = 0;
  const int base_out = base_e;
  for (int rg = 0; rg < 2 && run < capx; ++rg) {
    const int* rr = rg == 0 ? cs.r1 : cs.r2;
    if (!rr) break;
    const int beg = rr[2 * b], end = rr[2 * b + 1];
    // RAD_U 64-candidate chunks per trip: their position loads leave together (the loop was one dependent load -> ballot round trip
    // per chunk; round 5: sixteen chunks per trip measured slower -- the ballots of chunks past a short range are not free), the chunks are then ranked in index order as before
    for (int i0 = beg; i0 < end && run < capx; i0 += 64 * RAD_U) {
      bool okv[RAD_U];
#pragma unroll
      for (int u = 0; u < RAD_U; ++u) {
        const int i = i0 + 64 * u + lane;
        okv[u] = false;
        if (i < end) okv[u] = dist2(cs.pos[2 * i], cs.pos[2 * i + 1], qx, qy) < r2 && (!cand_ok || i < cand_base || cand_ok[i - cand_base]);
      }
#pragma unroll
      for (int u = 0; u < RAD_U; ++u) {
        const int i = i0 + 64 * u + lane;
        const bool ok = okv[u] && run < capx;   // (a chunk behind a full list takes nobody, as when the loop stopped there)
        const unsigned long long m = __ballot(ok);
        const int rank = run + __popcll(m & ((1ull << lane) - 1ull));
        if (MODE == 1 && ok && rank < capx && i != self) {
          const int o = base_out + rank - ((self >= 0 && self < i) ? 1 : 0);
          esrc[o] = i;
          edst[o] = q;
        }
        run += __popcll(m);
      }
    }
  }
  if (MODE == 0 && lane == 0) {
    int total = run < capx ? run : capx;
    cnt[q] = total;
  }
}
// self handling for the count: the self match is always inside r (d2 = 0) -- it occupies a rank
// slot iff its rank < cap+1.  Done in a second tiny pass to keep k_radius simple.
__global__ void k_radius_selfrank(CandSet cs, const float* __restrict__ qpos, const int* __restrict__ qscene, int nq,
                                  float r2, int cap, int self_base, int* __restrict__ cnt, const int* __restrict__ cand_ok,
                                  int cand_base) {
  const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (q >= nq) return;
  if (cand_ok && !cand_ok[self_base + q - cand_base]) return;   // not a candidate itself: nothing to drop
  const float qx = qpos[2 * q], qy = qpos[2 * q + 1];
  const int b = qscene[q];
  const int self = self_base + q;
  int before = 0;
  for (int rg = 0; rg < 2; ++rg) {
    const int* rr = rg == 0 ? cs.r1 : cs.r2;
    if (!rr) break;
    const int beg = rr[2 * b], end = rr[2 * b + 1] < self ? rr[2 * b + 1] : self;
    for (int i0 = beg; i0 < end; i0 += 64) {
      const int i = i0 + lane;
      bool ok = false;
      if (i < end) ok = dist2(cs.pos[2 * i], cs.pos[2 * i + 1], qx, qy) < r2 && (!cand_ok || i < cand_base || cand_ok[i - cand_base]);
      before += __popcll(__ballot(ok));
    }
  }
  if (lane == 0 && before < cap + 1) cnt[q] -= 1;
}

// exclusive scan of cnt[0..n) -> off[0..n] and of ceil(cnt/32) -> toff[0..n] (32-edge tiles of the rel-PE
// operand images), one workgroup per edge set (n is a few thousand at most).  Declared after RadSets.
__global__ void k_exclusive_scan(RadSets sets, int n) {
  const int* __restrict__ cnt = sets.s[blockIdx.x].cnt;
  int* __restrict__ off = sets.s[blockIdx.x].eoff;
  int* __restrict__ toff = sets.s[blockIdx.x].toff;
  __shared__ int wsum[16], wsum2[16];
  __shared__ int carry, carry2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  if (tid == 0) { carry = 0; carry2 = 0; }
  __syncthreads();
  for (int base = 0; base < n; base += blockDim.x) {
    const int i = base + tid;
    const int v = i < n ? cnt[i] : 0;
    const int v2 = (v + 31) >> 5;
    int s = v, s2 = v2;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(s, o), y2 = __shfl_up(s2, o);
      if (lane >= o) { s += y; s2 += y2; }
    }
    if (lane == 63) { wsum[wave] = s; wsum2[wave] = s2; }
    __syncthreads();
    int woff = 0, woff2 = 0;
    for (int j = 0; j < wave; ++j) { woff += wsum[j]; woff2 += wsum2[j]; }
    const int c = carry, c2 = carry2;
    if (i < n) { off[i] = c + woff + s - v; toff[i] = c2 + woff2 + s2 - v2; }
    __syncthreads();
    if (tid == blockDim.x - 1) { carry = c + woff + s; carry2 = c2 + woff2 + s2; }
    __syncthreads();
  }
  if (tid == 0) { off[n] = carry; toff[n] = carry2; }
}

// MFMA operand images of the split-fp16 rel-PE rows.  The edges of a destination are cut into tiles of 32
// (the last one zero-padded); tile tau owns 8192 halfs in each image:
//  * rtA (score pass, A operand: lane = edge m + 16*kq holds 8 consecutive columns):
//      [sub 2 (edges 0-15 | 16-31)][part 2 (hi | lo)][ks 4][lane 64][8] -- one wave load instruction reads
//      1 KB CONTIGUOUS.  (Read from row-major rows the same fragment is 16 rows x 64 B: adjacent lanes hit
//      different cache lines and the texture-address unit issues it 4x slower -- tools/mb/mb_gather.)
//  * rtT (aggregation pass, B operand: lane = column n + 16*kq holds 8 consecutive EDGES of one column):
//      [part 2][column 128][edge 32].
// One 256-thread workgroup per tile.  (The engine makes both images straight from the geometry, in
// the fused rel-PE kernel further down; this row-major -> images variant serves the ps_test_attn hook.)
__global__ __launch_bounds__(256) void k_tile_transpose(const int* __restrict__ eoff, const int* __restrict__ toff, int nq,
                                                       const _Float16* __restrict__ rthl, _Float16* __restrict__ rtA,
                                                       _Float16* __restrict__ rtT) {
  __shared__ __attribute__((aligned(16))) _Float16 buf[32][264];
  const int tid = threadIdx.x;
  const int ntiles = toff[nq];
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int lo = 0, hi = nq - 1;   // largest d with toff[d] <= tile (destinations without edges share a toff value)
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (toff[mid] <= tile) lo = mid; else hi = mid - 1;
    }
    while (toff[lo + 1] <= tile) ++lo;   // skip empty destinations (cannot run past nq: tile < toff[nq])
    const int e0 = eoff[lo] + (tile - toff[lo]) * 32;
    const int n = min(32, eoff[lo + 1] - e0);
    __syncthreads();
    for (int i = tid; i < 32 * 32; i += 256) {   // 32 rows x 32 chunks of 8 halfs
      const int r = i >> 5, ch = i & 31;
      half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (r < n) v = ldgh8(rthl + (size_t)(e0 + r) * 256 + 8 * ch);
      *reinterpret_cast<half8*>(&buf[r][8 * ch]) = v;
    }
    __syncthreads();
    // A image: piece P = ((sub*2 + part)*4 + ks)*64 + kq*16 + m  <-  row sub*16 + m, halfs part*128 + ks*32 + kq*8 ..+8
    for (int P = tid; P < 1024; P += 256) {
      const int m = P & 15, kq = (P >> 4) & 3, ks = (P >> 6) & 3, part_ = (P >> 8) & 1, sub = P >> 9;
      *reinterpret_cast<half8*>(rtA + (size_t)tile * 8192 + (size_t)P * 8) =
          *reinterpret_cast<const half8*>(&buf[sub * 16 + m][part_ * 128 + ks * 32 + kq * 8]);
    }
    const int part = tid >> 7, c = tid & 127;
    _Float16* o = rtT + (size_t)tile * 8192 + part * 4096 + c * 32;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      half8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = buf[8 * g + j][part * 128 + c];
      *reinterpret_cast<half8*>(o + 8 * g) = v;
    }
  }
}

// knn (loop=True): one WAVE per query.  Each lane keeps up to KNN_SLOTS candidates' d2 in registers
// (non-negative floats order like their bit patterns); the k-th smallest key v_k is found by a
// 32-step bitwise bisection (count(key < trial) via ballots), then every key < v_k is kept and the
// ties at v_k are kept in index order until k are out -- i.e. the k smallest by (d2, index).
// Edges of a destination come out in candidate-index order.  eoff is closed-form (host).
constexpr int KNN_SLOTS = 40;   // up to 64*40 = 2560 candidates per scene (2048 polylines + 256 agents fits)
// cand_ok (optional): candidate token i takes part iff i < cand_base or cand_ok[i - cand_base] != 0 -- agent rows that
// are not in the scene at the initial step (they enter with a later fut_obs frame) are no tokens yet.
// (SLOTS: 64-candidate slots a lane holds -- the smallest of 4 / 20 / KNN_SLOTS that covers the query's scene, chosen by a wave-uniform
// branch: every counting pass runs over the slots, and a fully unrolled pass over 20 is half the work of one over 40.  A per-slot
// guard inside ONE 40-slot body was slower than no guard at all: the branches break the batches of compares and ballots.)
template <int SLOTS>
__device__ __forceinline__ void knn_select(const CandSet& cs, float qx, float qy, int b1, int n1, int b2, int n, int q, int lane, int k,
                                           const int* __restrict__ eoff, int* __restrict__ esrc, int* __restrict__ edst,
                                           const int* __restrict__ cand_ok, int cand_base) {
  unsigned key[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int j = s * 64 + lane;   // candidate order = global index order (range 1 then range 2)
    unsigned kk = 0xffffffffu;     // padding sorts last (real keys are finite floats < 0x7f800000)
    if (j < n) {
      const int i = j < n1 ? b1 + j : b2 + (j - n1);
      if (!cand_ok || i < cand_base || cand_ok[i - cand_base])
        kk = __float_as_uint(dist2(cs.pos[2 * i], cs.pos[2 * i + 1], qx, qy));
    }
    key[s] = kk;
  }
  int n_ok = n;
  if (cand_ok) {
    n_ok = 0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) n_ok += __popcll(__ballot(key[s] != 0xffffffffu));
  }
  const int kk_ = k < n_ok ? k : n_ok;
  // v_k = largest x with count(key < x) < k
  unsigned vk = 0;
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned trial = vk | (1u << bit);
    int c = 0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) c += __popcll(__ballot(key[s] < trial));
    if (c < kk_) vk = trial;
  }
  int c_lt = 0;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) c_lt += __popcll(__ballot(key[s] < vk));
  int ties_left = kk_ - c_lt;   // how many keys == v_k still go out, in index order
  int out = eoff[q];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int j = s * 64 + lane;
    const bool lt = key[s] < vk, eq = key[s] == vk && j < n && key[s] != 0xffffffffu;
    const unsigned long long meq = __ballot(eq);
    const int eq_rank = __popcll(meq & ((1ull << lane) - 1ull));
    const bool take = lt || (eq && eq_rank < ties_left);
    const unsigned long long mt = __ballot(take);
    if (take) {
      const int o = out + __popcll(mt & ((1ull << lane) - 1ull));
      esrc[o] = j < n1 ? b1 + j : b2 + (j - n1);
      edst[o] = q;
    }
    out += __popcll(mt);
    const int neq = __popcll(meq);
    ties_left -= neq < ties_left ? neq : ties_left;
  }
}
__global__ void k_knn(CandSet cs, const float* __restrict__ qpos, const int* __restrict__ qscene, int nq, int k,
                      const int* __restrict__ eoff, int* __restrict__ esrc, int* __restrict__ edst,
                      const int* __restrict__ cand_ok, int cand_base) {
  const int q = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave-uniform: so is n below)
  const int lane = threadIdx.x & 63;
  if (q >= nq) return;
  const float qx = qpos[2 * q], qy = qpos[2 * q + 1];
  const int b = qscene[q];
  const int b1 = cs.r1[2 * b], n1 = cs.r1[2 * b + 1] - b1;
  const int b2 = cs.r2 ? cs.r2[2 * b] : 0, n2 = cs.r2 ? cs.r2[2 * b + 1] - b2 : 0;
  const int n = n1 + n2;
  if (n <= 64 * 4) knn_select<4>(cs, qx, qy, b1, n1, b2, n, q, lane, k, eoff, esrc, edst, cand_ok, cand_base);
  else if (n <= 64 * 20) knn_select<20>(cs, qx, qy, b1, n1, b2, n, q, lane, k, eoff, esrc, edst, cand_ok, cand_base);
  else knn_select<KNN_SLOTS>(cs, qx, qy, b1, n1, b2, n, q, lane, k, eoff, esrc, edst, cand_ok, cand_base);
}

// MODEL.REL_POS_EDGE_FUNC 'knn' (decoder/sym_coord.py:85-96, policy/act_decoder.py:249-261): the generator's and the policy's edge
// sets from the `cap` NEAREST candidates of the query's scene (torch_cluster.knn; knn_graph(loop = False) for the prompt graph: the
// cap + 1 nearest, then without the query itself) instead of the first `cap` inside a radius.  Same RadSets plumbing as k_radius --
// MODE 0 counts (cnt), MODE 1 makes its own CSR / tile offsets (csr_prefix; k_exclusive_scan beyond CSR_PREFIX_MAX_Q queries), selects and fills (k_knn's bisection: the k-th
// smallest (d2, index) key, ties in index order) -- so everything downstream of the edge lists is shared.
template <int MODE>
__global__ void k_knn_sets(RadSets sets, const float* __restrict__ qpos, const int* __restrict__ qscene, int nq) {
  const RadSet& S = sets.s[blockIdx.y];
  const CandSet cs = S.cs;
  const int q = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  if (q >= nq) return;
  const int* __restrict__ cand_ok = S.cand_ok;
  const int cand_base = S.cand_base;
  const float qx = qpos[2 * q], qy = qpos[2 * q + 1];
  const int b = qscene[q];
  const int b1 = cs.r1[2 * b], n1 = cs.r1[2 * b + 1] - b1;
  const int b2 = cs.r2 ? cs.r2[2 * b] : 0, n2 = cs.r2 ? cs.r2[2 * b + 1] - b2 : 0;
  const int n = n1 + n2;
  // (a query that is filtered out as a candidate has no self match to drop)
  const int self = (S.self_base >= 0 && (!cand_ok || cand_ok[S.self_base + q - cand_base])) ? S.self_base + q : -1;
  unsigned key[KNN_SLOTS];
  int n_ok = 0;
#pragma unroll
  for (int s = 0; s < KNN_SLOTS; ++s) {
    const int j = s * 64 + lane;
    unsigned kk = 0xffffffffu;
    if (j < n) {
      const int i = j < n1 ? b1 + j : b2 + (j - n1);
      if (!cand_ok || i < cand_base || cand_ok[i - cand_base])
        kk = __float_as_uint(dist2(cs.pos[2 * i], cs.pos[2 * i + 1], qx, qy));
    }
    key[s] = kk;
    n_ok += __popcll(__ballot(kk != 0xffffffffu));
  }
  const int want = S.cap + (S.self_base >= 0 ? 1 : 0);
  const int kk_ = want < n_ok ? want : n_ok;
  // Is the query itself among the kk_ selected?  Its key is 0, the smallest -- but ties at distance 0 (candidates at the query's exact
  // position) are broken by index like every other tie, so with >= kk_ of them ahead of it the query is NOT selected, and
  // knn_graph(loop = False) then keeps all cap + 1 neighbours (MODE 0 assumed "always selected" and sized the CSR
  // range one short of what MODE 1 filled).  Both modes decide it with the selection's own (key, index) ranking.
  bool self_sel = false;
  if (self >= 0) {
    const int js = (self >= b1 && self < b1 + n1) ? self - b1 : n1 + (self - b2);
    int ahead = 0;
#pragma unroll
    for (int s = 0; s < KNN_SLOTS; ++s) ahead += __popcll(__ballot(key[s] == 0u && s * 64 + lane < js));
    self_sel = ahead < kk_;
  }
  if (MODE == 0) {
    if (lane == 0) S.cnt[q] = kk_ - (self_sel ? 1 : 0);
    return;
  }
  int out;
  {
    int t0, mine;
    csr_prefix(S.cnt, q, nq, lane, S.eoff, S.toff, out, t0, mine, sets.scanned);
    const int nt = (mine + 31) >> 5;
    for (int t = lane; t < nt; t += 64) S.tdst[t0 + t] = q;
  }
  unsigned vk = 0;
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned trial = vk | (1u << bit);
    int c = 0;
#pragma unroll
    for (int s = 0; s < KNN_SLOTS; ++s) c += __popcll(__ballot(key[s] < trial));
    if (c < kk_) vk = trial;
  }
  int c_lt = 0;
#pragma unroll
  for (int s = 0; s < KNN_SLOTS; ++s) c_lt += __popcll(__ballot(key[s] < vk));
  int ties_left = kk_ - c_lt;
#pragma unroll
  for (int s = 0; s < KNN_SLOTS; ++s) {
    const int j = s * 64 + lane;
    const int i = j < n1 ? b1 + j : b2 + (j - n1);
    const bool lt = key[s] < vk, eq = key[s] == vk && j < n && key[s] != 0xffffffffu;
    const unsigned long long meq = __ballot(eq);
    const int eq_rank = __popcll(meq & ((1ull << lane) - 1ull));
    const bool sel = lt || (eq && eq_rank < ties_left);
    const bool take = sel && i != self;
    const unsigned long long mt = __ballot(take);
    if (take) {
      const int o = out + __popcll(mt & ((1ull << lane) - 1ull));
      S.esrc[o] = i;
      S.edst[o] = q;
    }
    out += __popcll(mt);
    const int neq = __popcll(meq);
    ties_left -= neq < ties_left ? neq : ties_left;
  }
}

// ------------------------------------------------------------------------------------------
// K5  relative positional encoding of an edge (act_decoder.py:203-221 and twins) through
// FourierEmbeddingFix(32) (fourier_embedding.py:63-78), then the affine-free LayerNorm that
// every layer's attn_prenorm_r shares.  Feature 0..31: dist, 32..63: rel_ori, 64..127: angle twice;
// feature c uses div32[c & 31], sin for even c, cos for odd c.
__device__ __forceinline__ float fourier_feat(float x, int slot, const float* __restrict__ div) {
  const float v = (x * PS_TWO_PI_F) / div[slot];
  return (slot & 1) ? cosf(v) : sinf(v);
}

// One 256-thread workgroup per 32-edge tile (grid-stride; the tile count toff[nq] is read on the device and
// tdst maps a tile to its destination): each wave makes 8 of the tile's edges, the rows meet in LDS and
// leave as the two MFMA operand images (layouts: k_tile_transpose), written as contiguous 16-byte pieces.
struct PeSet {
  const int *esrc, *eoff, *toff, *tdst;
  int nq;
  const float *src_ori, *dst_pos, *dst_ori;
  _Float16 *rtA, *rtT;
};
struct PeSets {
  PeSet s[2];
};
__global__ __launch_bounds__(256) void k_relpe_tiles(PeSets sets, const float* __restrict__ src_pos,
                                                    const float* __restrict__ div32, float eps) {
  const PeSet& S = sets.s[blockIdx.y];
  const int* __restrict__ esrc = S.esrc;
  const int* __restrict__ eoff = S.eoff;
  const int* __restrict__ toff = S.toff;
  const int* __restrict__ tdst = S.tdst;
  const int nq = S.nq;
  const float* __restrict__ src_ori = S.src_ori;
  const float* __restrict__ dst_pos = S.dst_pos;
  const float* __restrict__ dst_ori = S.dst_ori;
  _Float16* __restrict__ rtA = S.rtA;
  _Float16* __restrict__ rtT = S.rtT;
  __shared__ __attribute__((aligned(16))) _Float16 buf[32][264];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntiles = toff[nq];
  // The 128 features are 64 (sin, cos) pairs: input dist | rel_ori | angle | angle again, 16 frequencies each
  // (div32[2i] == div32[2i+1], fourier_embedding.py:68-69); a pair is ONE sincosf (one argument reduction; a
  // per-feature sinf/cosf select would run both branches in every lane).
  const int sub = lane & 7;
  float dvv[8];   // this lane's 8 frequencies: pairs 8*(sub & 1) ..+7 of its input block
#pragma unroll
  for (int j = 0; j < 8; ++j) dvv[j] = div32[2 * ((sub & 1) * 8 + j)];
  __shared__ float geo[32][4];   // (dist, rel_ori, angle) of the tile's edges
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int d = tdst[tile];
    const int e0 = eoff[d] + (tile - toff[d]) * 32;
    const int n = min(32, eoff[d + 1] - e0);
    __syncthreads();   // the previous tile's image writes are done with buf / geo
    // the 4 scalars of every edge of the tile at once (one lane per edge: ONE dependent gather chain per tile,
    // not one per edge), act_decoder.py:203-217
    if (tid < n) {
      const float px = dst_pos[2 * d], py = dst_pos[2 * d + 1], od = dst_ori[d];
      const float cx = cosf(od), cy = sinf(od);
      const int s = esrc[e0 + tid];
      const float dx = src_pos[2 * s] - px, dy = src_pos[2 * s + 1] - py;
      geo[tid][0] = sqrtf(dx * dx + dy * dy);
      geo[tid][1] = wrap_angle(src_ori[s] - od);
      // torch's .sum(dim=-1) accumulates from +0, so a dot of (-0, -0) is +0 there: keep the explicit
      // 0.f + ... (IEEE forbids folding it) or self-loop edges would see atan2(+-0, -0) = +-pi.
      const float dot = (0.f + cx * dx) + cy * dy;
      geo[tid][2] = atan2f(cx * dy - cy * dx, dot);
    }
    __syncthreads();
    // features: lane -> (edge lane >> 3 of the wave's 8, pair group lane & 7 = 8 consecutive (sin, cos) pairs = 16
    // features); the 8 edges of a wave advance together and the LayerNorm sums meet inside 8 lanes
    {
      const int r = wave * 8 + (lane >> 3);
      float f[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) f[j] = 0.f;
      if (r < n) {
        const float xin = geo[r][sub < 4 ? (sub >> 1) : 2];
#pragma unroll
        for (int j = 0; j < 8; ++j) sincosf((xin * PS_TWO_PI_F) / dvv[j], &f[2 * j], &f[2 * j + 1]);
      }
      float sm = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) sm += f[j];
      sm += dpp_xor1(sm);
      sm += dpp_xor2(sm);
      sm = xor_add<4>(sm);
      const float mean = sm * (1.f / 128.f);
      float sq = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        f[j] -= mean;
        sq = fmaf(f[j], f[j], sq);
      }
      sq += dpp_xor1(sq);
      sq += dpp_xor2(sq);
      sq = xor_add<4>(sq);
      const float rstd = (r < n) ? 1.f / sqrtf(sq * (1.f / 128.f) + eps) : 0.f;
      half8 h0, h1, l0, l1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float y0 = f[j] * rstd, y1 = f[8 + j] * rstd;
        h0[j] = f16_hi(y0); l0[j] = f16_lo(y0);
        h1[j] = f16_hi(y1); l1[j] = f16_lo(y1);
      }
      _Float16* br = &buf[r][0];
      *reinterpret_cast<half8*>(br + 16 * sub) = h0;
      *reinterpret_cast<half8*>(br + 16 * sub + 8) = h1;
      *reinterpret_cast<half8*>(br + 128 + 16 * sub) = l0;
      *reinterpret_cast<half8*>(br + 128 + 16 * sub + 8) = l1;
    }
    __syncthreads();
    // columns 96..127 repeat 64..95 (the angle enters the embedding twice): neither image stores them, the
    // consumers fold the matching weights instead (AttnW::Wkr_g3 / Wvr_gt3)
    for (int P = tid; P < 1024; P += 256) {
      const int m = P & 15, kq = (P >> 4) & 3, ks = (P >> 6) & 3, part_ = (P >> 8) & 1, sub = P >> 9;
      if (ks < 3)
        *reinterpret_cast<half8*>(rtA + (size_t)tile * 8192 + (size_t)P * 8) =
            *reinterpret_cast<const half8*>(&buf[sub * 16 + m][part_ * 128 + ks * 32 + kq * 8]);
    }
    const int part = tid >> 7, c = tid & 127;
    _Float16* o = rtT + (size_t)tile * 8192 + part * 4096 + c * 32;
    if (c < 96) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = buf[8 * g + j][part * 128 + c];
        *reinterpret_cast<half8*>(o + 8 * g) = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Prompt encoder (prompt_encoder/base.py:36-46): MLP 7 -> 128 (LN, ReLU) -> 128, one WG per agent.
struct Mlp3W {  // up to three Linear layers with optional LayerNorm+ReLU between (reference MLP)
  int dims[4];
  int n;
  const float* W[3];   // torch layout [out][in]
  const float* b[3];
  const float* lnw[3];  // after layer i (i < n-1) or nullptr (then plain ReLU)
  const float* lnb[3];
};

__device__ __forceinline__ void mlp3_rows1(const Mlp3W& m, float* bufA, float* bufB, float eps) {
  // input row in bufA (dims[0] floats); result in bufA; 128 threads
  for (int l = 0; l < m.n; ++l) {
    gemv_small<1>(bufA, 0, m.dims[l], m.W[l], m.dims[l + 1], m.b[l], bufB, 0, false);
    if (l < m.n - 1) {
      const int nn = m.dims[l + 1];
      if (m.lnw[l]) {
        // LayerNorm over nn (64 or 128) features by wave 0
        if (threadIdx.x < 64) {
          const int lane = threadIdx.x;
          float a0 = lane < nn ? bufB[lane] : 0.f, a1 = (lane + 64) < nn ? bufB[lane + 64] : 0.f;
          const float mean = wave_sum(a0 + a1) / (float)nn;
          const float d0 = lane < nn ? a0 - mean : 0.f, d1 = (lane + 64) < nn ? a1 - mean : 0.f;
          const float var = wave_sum(d0 * d0 + d1 * d1) / (float)nn;
          const float rstd = 1.f / sqrtf(var + eps);
          if (lane < nn) bufB[lane] = fmaxf(fmaf(d0 * rstd, m.lnw[l][lane], m.lnb[l][lane]), 0.f);
          if (lane + 64 < nn) bufB[lane + 64] = fmaxf(fmaf(d1 * rstd, m.lnw[l][lane + 64], m.lnb[l][lane + 64]), 0.f);
        }
      } else {
        for (int i = threadIdx.x; i < nn; i += blockDim.x) bufB[i] = fmaxf(bufB[i], 0.f);
      }
      __syncthreads();
    }
    for (int i = threadIdx.x; i < m.dims[l + 1]; i += blockDim.x) bufA[i] = bufB[i];
    __syncthreads();
  }
}

// MLP_R rows per workgroup (round 4; one row per workgroup before): thread n still owns output n and walks k in order -- per row the
// same fma chain, the same LayerNorm reduction (one wave per row), so the results are bit-identical -- but a weight element is loaded
// once for MLP_R rows instead of once per row (1024 prompt rows: 34 -> 24 us; staging the weights through LDS as well was slower: the loop is bound by its LDS reads of the rows).  n_rows: rows of `out`; dims <= 128.
// (round 5: MLP_R = 1 again for up to 256 rows -- a single scene's 128 prompts are 16 workgroups of 8 rows on 16 of 256 CUs, 31 us per launch,
// against 128 one-row workgroups; the same fma chain per output either way)
template <int MLP_R>
__global__ __launch_bounds__(128) void k_mlp_rows(Mlp3W m, const float* __restrict__ in, const int* __restrict__ rows,
                                                  int in_stride, float* __restrict__ out, int out_stride, float eps, int n_rows) {
  __shared__ float a[MLP_R][128], b[MLP_R][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * MLP_R, nr = min(MLP_R, n_rows - row0);
  for (int r = 0; r < MLP_R; ++r) {
    const int src = r < nr ? (rows ? rows[row0 + r] : row0 + r) : 0;
    for (int i = tid; i < m.dims[0]; i += 128) a[r][i] = r < nr ? in[(size_t)src * in_stride + i] : 0.f;
  }
  __syncthreads();
  for (int l = 0; l < m.n; ++l) {
    const int K = m.dims[l], N = m.dims[l + 1];
    for (int n = tid; n < N; n += 128) {
      float s[MLP_R];
      const float b0 = m.b[l] ? m.b[l][n] : 0.f;
#pragma unroll
      for (int r = 0; r < MLP_R; ++r) s[r] = b0;
      const float* __restrict__ wr = m.W[l] + (size_t)n * K;
      for (int k = 0; k < K; ++k) {
        const float w = wr[k];
#pragma unroll
        for (int r = 0; r < MLP_R; ++r) s[r] = fmaf(a[r][k], w, s[r]);
      }
#pragma unroll
      for (int r = 0; r < MLP_R; ++r) b[r][n] = s[r];
    }
    __syncthreads();
    if (l < m.n - 1) {
      if (m.lnw[l]) {   // LayerNorm over N (64 or 128) features, one wave per row
        for (int r = wave; r < MLP_R; r += 2) {
          float a0 = lane < N ? b[r][lane] : 0.f, a1 = (lane + 64) < N ? b[r][lane + 64] : 0.f;
          const float mean = wave_sum(a0 + a1) / (float)N;
          const float d0 = lane < N ? a0 - mean : 0.f, d1 = (lane + 64) < N ? a1 - mean : 0.f;
          const float var = wave_sum(d0 * d0 + d1 * d1) / (float)N;
          const float rstd = 1.f / sqrtf(var + eps);
          if (lane < N) b[r][lane] = fmaxf(fmaf(d0 * rstd, m.lnw[l][lane], m.lnb[l][lane]), 0.f);
          if (lane + 64 < N) b[r][lane + 64] = fmaxf(fmaf(d1 * rstd, m.lnw[l][lane + 64], m.lnb[l][lane + 64]), 0.f);
        }
      } else {
        for (int r = 0; r < MLP_R; ++r)
          for (int i = tid; i < N; i += 128) b[r][i] = fmaxf(b[r][i], 0.f);
      }
      __syncthreads();
    }
    for (int r = 0; r < MLP_R; ++r)
      for (int i = tid; i < N; i += 128) a[r][i] = b[r][i];
    __syncthreads();
  }
  for (int r = 0; r < nr; ++r)
    for (int i = tid; i < m.dims[m.n]; i += 128) out[(size_t)(row0 + r) * out_stride + i] = a[r][i];
}

// _autoregressive_obs_fusion (attn_fusion.py:175-203, MODEL.OBS_UPDATE.FUSION 'mlp'): the agent token of this replan =
// obs_update_mlp(cat(token of the previous replan, re-encoded observation)).  An agent that is no scene token at
// this replan (a log-replay agent outside the log: live == 0) keeps a ZERO row, which is also what the reference
// feeds as "previous token" when the agent comes back (:193-195).  One WG (128 threads) per agent.
__global__ __launch_bounds__(128) void k_obs_fuse(Mlp3W m, float* __restrict__ tok, const float* __restrict__ new_emb,
                                                 const int* __restrict__ live, float eps) {
  __shared__ float a[256], b[256];
  const int row = blockIdx.x, tid = threadIdx.x;
  if (live && !live[row]) {
    tok[(size_t)row * 128 + tid] = 0.f;
    return;
  }
  a[tid] = tok[(size_t)row * 128 + tid];
  a[128 + tid] = new_emb[(size_t)row * 128 + tid];
  __syncthreads();
  mlp3_rows1(m, a, b, eps);
  tok[(size_t)row * 128 + tid] = a[tid];
}
__global__ void k_zero_dead_rows(float* __restrict__ tok, const int* __restrict__ live, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * 128 && !live[i >> 7]) tok[i] = 0.f;
}

// M-replica fan-out (replica_batch_for_parallel_rollout, rollout/gpu_utils.py:59-123: every per-agent tensor .repeat(M, ...)):
// rows [np, n) of a row-major [n][w] buffer become copies of rows [0, np), replica after replica.
__global__ void k_fan_out_rows(float* __restrict__ p, int np, int n, int w) {
  const size_t first = (size_t)np * w, total = (size_t)n * w;
  for (size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    p[i] = p[i % first];
}

// Rollout trajectories in the world frame (obtain_rollout_trajs_in_world, rollout/gpu_utils.py:255-267, over
// batch_rotate_2D / wrap_angle models/utils/geometry.py:13-22 and batch_nd_transform_points_pt / _angles_pt / angle_wrap
// rollout/utils.py:272-283, :347-392): the agent-init-frame steps are rotated by the initial heading and moved to the
// initial position (the scene-centre frame), then taken through the centre -> world matrix.  fp32, in the reference's
// order of operations.  One thread per (agent row, step); out [A][T][3] = (x, y, heading).
struct WorldTf { float m[9]; };   // row-major 3 x 3 homogeneous 2-D transform
__device__ __forceinline__ float floor_mod(float a, float m) {   // torch.remainder / numpy % for m > 0
  float r = fmodf(a, m);
  if (r != 0.f && r < 0.f) r += m;
  return r;
}
__global__ void k_world_traj(const float* __restrict__ traj, int stride_steps, int hist, int T, int A,
                             const float* __restrict__ init_pos, const float* __restrict__ init_head, WorldTf tf,
                             float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A * T) return;
  const int a = i / T, t = i - a * T;
  const float4 s = *(const float4*)(traj + ((size_t)a * stride_steps + hist + t) * 4);   // x, y, sin, cos
  const float th = init_head[a];
  const float c = cosf(th), sn = sinf(th);
  const float xc = (s.x * c - s.y * sn) + init_pos[2 * a];
  const float yc = (s.y * c + s.x * sn) + init_pos[2 * a + 1];
  const float PI = 3.14159265358979323846f, TWO_PI = 6.28318530717958647692f;
  const float hc = -PI + floor_mod((atan2f(s.z, s.w) + th) + PI, TWO_PI);
  const float xw = (xc * tf.m[0] + yc * tf.m[1]) + tf.m[2];
  const float yw = (xc * tf.m[3] + yc * tf.m[4]) + tf.m[5];
  const float hw = floor_mod((hc + atan2f(tf.m[3], tf.m[0])) + PI, TWO_PI) - PI;
  out[(size_t)i * 3] = xw;
  out[(size_t)i * 3 + 1] = yw;
  out[(size_t)i * 3 + 2] = hw;
}

// Condition encoders + mean pooling over the condition entries attached to one agent
// (condition_encoders.py:21-51, :76-141, :148-150; condition_attns.py:114-188), then
// r = pooled + relPE(edge) and the affine-free LayerNorm.  One WG (128 threads) per EDGE of the condition graph;
// entries are a host-built CSR: type (0 goal, 1 unary tag, 2 drag points, 3 binary tag), tag id | row of the drag-point
// embeddings (DragPointEncoder :152-191 = k_pointnet_mfma over the [x, y] points) | 2 * tag + half, 3 floats.  The
// operand images are zeroed by the caller; an edge writes its own slot.
struct CondW {
  Mlp3W goal;                 // MLP 2 -> 128 (ReLU) -> 128, no norm
  const float* tag_emb;       // [11][128] indexed by V_Action_MotionTag value
  const float* v2v_emb;       // [5][256] indexed by V2V_MotionTag value: source half | target half (binary tags; may be null)
  const float *div32, *div64, *div128;
};
// one edge of the condition layers' graph: unary conditions make self loops, binary ones the edges s -> t and t -> s
struct CondEdge {
  int src, dst;               // agent rows
  int img;                    // tile * 32 + slot of its row in the operand images
};
__global__ __launch_bounds__(128) void k_cond_edges(CondW w, const int* __restrict__ ent_off, const int* __restrict__ ent_type,
                                                   const float* __restrict__ ent_val, const float* __restrict__ drag_emd, int n_nodes,
                                                   const CondEdge* __restrict__ edges, const float* __restrict__ ppos,
                                                   const float* __restrict__ pori, _Float16* __restrict__ rtA,
                                                   _Float16* __restrict__ rtT, float eps) {
  __shared__ float a[128], b[128], accum[128];
  const int node = blockIdx.x, tid = threadIdx.x;
  accum[tid] = 0.f;
  __syncthreads();
  const int e0 = ent_off[node], e1 = ent_off[node + 1];
  for (int e = e0; e < e1; ++e) {
    const float v0 = ent_val[3 * e], v1 = ent_val[3 * e + 1], v2 = ent_val[3 * e + 2];
    if (ent_type[2 * e] == 0) {
      if (tid < 2) a[tid] = tid == 0 ? v0 : v1;
      __syncthreads();
      mlp3_rows1(w.goal, a, b, eps);
      // + FourierEmbeddingFix(128)(t): one channel, 128 slots
      const float v = (v2 * PS_TWO_PI_F) / w.div128[tid];
      accum[tid] += a[tid] + ((tid & 1) ? cosf(v) : sinf(v));
    } else if (ent_type[2 * e] == 2) {
      accum[tid] += drag_emd[(size_t)ent_type[2 * e + 1] * 128 + tid];
    } else if (ent_type[2 * e] == 3) {
      // binary tag: this edge's half of the [2 D] parameter + the same temporal embedding on both halves (:129-133)
      const int slot = tid & 63;
      const float x = tid < 64 ? v1 : v2;
      const float v = (x * PS_TWO_PI_F) / w.div64[slot];
      accum[tid] += w.v2v_emb[(size_t)ent_type[2 * e + 1] * 128 + tid] + ((slot & 1) ? cosf(v) : sinf(v));
    } else {
      // tag parameter + FourierEmbeddingFix(64)([t0, t1]): two channels x 64 slots
      const int slot = tid & 63;
      const float x = tid < 64 ? v1 : v2;
      const float v = (x * PS_TWO_PI_F) / w.div64[slot];
      accum[tid] += w.tag_emb[ent_type[2 * e + 1] * 128 + tid] + ((slot & 1) ? cosf(v) : sinf(v));
    }
    __syncthreads();
  }
  // mean pool, + relPE of the edge (condition_attns.py:95-112, the rows of k_relpe_tiles; a self loop: dist 0, rel_ori 0,
  // angle atan2(+-0, +0) = +-0)
  const CondEdge ce = edges[node];
  float pe;
  {
    const float px = ppos[2 * ce.dst], py = ppos[2 * ce.dst + 1], od = pori[ce.dst];
    const float cx = cosf(od), cy = sinf(od);
    const float dx = ppos[2 * ce.src] - px, dy = ppos[2 * ce.src + 1] - py;
    const int blk = tid >> 5;
    float x;
    if (blk == 0) x = sqrtf(dx * dx + dy * dy);
    else if (blk == 1) x = wrap_angle(pori[ce.src] - od);
    else x = atan2f(cx * dy - cy * dx, (0.f + cx * dx) + cy * dy);
    pe = fourier_feat(x, tid & 31, w.div32);
  }
  a[tid] = accum[tid] / (float)(e1 - e0) + pe;
  __syncthreads();
  if (tid < 64) ln_row_wave(a, b, nullptr, nullptr, eps, tid, false);
  __syncthreads();
  const float y = b[tid];
  // slot `sl` of tile `tile` of both operand images (layouts: k_relpe_tiles); the images were zeroed before the launch
  const int tile = ce.img >> 5, sl = ce.img & 31;
  {
    const int ks = tid >> 5, kq = (tid >> 3) & 3, j = tid & 7;
    _Float16* ta = rtA + (size_t)tile * 8192;
    const int P0 = (sl & 15) + 16 * kq + 64 * ks + 512 * (sl >> 4);
    ta[(size_t)P0 * 8 + j] = f16_hi(y);            // part hi
    ta[(size_t)(P0 + 256) * 8 + j] = f16_lo(y);    // part lo
  }
  rtT[(size_t)tile * 8192 + tid * 32 + sl] = f16_hi(y);
  rtT[(size_t)tile * 8192 + 4096 + tid * 32 + sl] = f16_lo(y);
}

// fp32 rows [n][128] -> split fp16 rows [n][256] (hi | lo); used by the test hooks
__global__ void k_split_rows(const float* __restrict__ in, int n, _Float16* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 128) return;
  const int r = i >> 7, c = i & 127;
  out[(size_t)r * 256 + c] = f16_hi(in[i]);
  out[(size_t)r * 256 + 128 + c] = f16_lo(in[i]);
}

// ------------------------------------------------------------------------------------------
// K12  step_env (traj_sam.py:205-274 + models/utils/geometry.py:24-58 + _get_rel_vel_acc :552):
// the last hist+2 states -> ego-relative history features written into obs_in[:, :hist, :8];
// also the agents' current pose (a_pos, :211-215).  One 64-thread WG per agent.
struct StepLog {                 // all null when every observed agent is a policy agent
  const int* is_policy;          // [A]
  const float* frame_in;         // [A][hist][obs_dim] logged observation of this replan (null at replan 0)
  const uint8_t* frame_mask;     // [A][hist][obs_dim]
  const float* frame_pos;        // [A][2] logged pose (null: the initial pose)
  const float* frame_head;       // [A]
  const uint8_t* init_mask;      // [A][hist][obs_dim] mask of the initial observation
  uint8_t* obs_mask;             // [A][hist][obs_dim] OUT: validity of obs_in for the agent encoder
  int* live;                     // [A] OUT: the agent is a scene token at this replan
};
// (the body: called by k_step_env with one 64-thread workgroup per agent, and by k_policy_head_row's tail -- 256 threads, the first wave
// works, every thread meets the barriers -- for the NEXT replan's step_env of the same agent: round 5, one launch less per replan)
__device__ __forceinline__ void step_env_body(const int a, const int tid, const float* traj, const float* vel, int stride_steps,
                                              int last, int hist, float dt, const float* __restrict__ init_pos,
                                              const float* __restrict__ init_head, const float* __restrict__ static_in,
                                              int obs_dim, float* __restrict__ obs_in, float* __restrict__ cur_pos,
                                              float* __restrict__ cur_ori, int write_obs, float* __restrict__ tok_pos,
                                              float* __restrict__ tok_ori, const StepLog& lg, int fd_vel) {
  const bool on = tid < 64;   // (the first wave works)
  if (lg.is_policy && !lg.is_policy[a]) {
    // log-replay agent (observed, not policy-controlled): observation, validity and pose of this replan come from
    // the log (batch.extras['fut_obs'][t], traj_sam.py:221-270), nothing from the simulated trajectories
    const int n = hist * obs_dim;
    int any_valid = 0;
    for (int i = on ? tid : n; i < n; i += 64) {
      const size_t o = (size_t)a * n + i;
      const bool ok = lg.frame_mask ? lg.frame_mask[o] != 0 : lg.init_mask[o] != 0;
      if (write_obs) {
        const float v = lg.frame_in ? lg.frame_in[o] : static_in[o];
        obs_in[o] = ok ? v : 0.f;
      }
      lg.obs_mask[o] = ok ? 1 : 0;
    }
    // a token exists iff some history step is fully valid (obs_encoder.py:84, mask.any)
    for (int s = on ? tid : hist; s < hist; s += 64) {
      bool all = true;
      for (int f = 0; f < obs_dim; ++f) {
        const size_t o = ((size_t)a * hist + s) * obs_dim + f;
        all &= lg.frame_mask ? lg.frame_mask[o] != 0 : lg.init_mask[o] != 0;
      }
      any_valid |= all ? 1 : 0;
    }
    any_valid = __any(any_valid);
    if (tid == 0) {
      const float px = lg.frame_pos ? lg.frame_pos[2 * a] : init_pos[2 * a], py = lg.frame_pos ? lg.frame_pos[2 * a + 1] : init_pos[2 * a + 1];
      const float hd = lg.frame_head ? lg.frame_head[a] : init_head[a];
      cur_pos[2 * a] = px;
      cur_pos[2 * a + 1] = py;
      cur_ori[a] = hd;
      if (tok_pos) {
        tok_pos[2 * a] = px;
        tok_pos[2 * a + 1] = py;
        tok_ori[a] = hd;
      }
      lg.live[a] = any_valid;
    }
    return;
  }
  if (lg.obs_mask) {   // policy agent in a scene that also has log-replay agents: its simulated history is all valid
    for (int i = on ? tid : hist * obs_dim; i < hist * obs_dim; i += 64) lg.obs_mask[(size_t)a * hist * obs_dim + i] = 1;
    if (tid == 0) lg.live[a] = 1;
  }
  const float* tr = traj + (size_t)a * stride_steps * 4;
  const float* vl = vel + (size_t)a * stride_steps * 2;
  const float lx = tr[(last - 1) * 4], ly = tr[(last - 1) * 4 + 1];
  const float th_last = atan2f(tr[(last - 1) * 4 + 2], tr[(last - 1) * 4 + 3]);
  if (tid == 0) {
    const float cpx = init_pos[2 * a] + lx, cpy = init_pos[2 * a + 1] + ly, cor = wrap_angle(th_last + init_head[a]);
    cur_pos[2 * a] = cpx;
    cur_pos[2 * a + 1] = cpy;
    cur_ori[a] = cor;
    if (tok_pos) {   // update_scene_emb (attn_fusion.py:205-250): the agents' scene tokens move with them
      tok_pos[2 * a] = cpx;
      tok_pos[2 * a + 1] = cpy;
      tok_ori[a] = cor;
    }
  }
  if (!write_obs) return;
  __shared__ float rvx[16], rvy[16], rpx[18], rpy[18];
  const float ct = cosf(-th_last), st = sinf(-th_last);
  const int nv = hist + 1;  // rel_vel over the last hist+1 steps
  if (fd_vel) {   // positions of the last hist + 2 steps in the last step's frame, then their differences / dt
    if (on && tid < hist + 2) {
      const int s = last - hist - 2 + tid;
      const float dx = tr[s * 4] - lx, dy = tr[s * 4 + 1] - ly;
      rpx[tid] = dx * ct - dy * st;
      rpy[tid] = dy * ct + dx * st;
    }
    __syncthreads();
    if (on && tid < nv) {
      rvx[tid] = (rpx[tid + 1] - rpx[tid]) / dt;
      rvy[tid] = (rpy[tid + 1] - rpy[tid]) / dt;
    }
  } else if (on && tid < nv) {
    const int s = last - nv + tid;
    const float vx = vl[s * 2], vy = vl[s * 2 + 1];
    rvx[tid] = vx * ct - vy * st;
    rvy[tid] = vy * ct + vx * st;
  }
  __syncthreads();
  if (on && tid < hist) {
    const int s = last - hist + tid;
    const float dx = tr[s * 4] - lx, dy = tr[s * 4 + 1] - ly;
    const float th = atan2f(tr[s * 4 + 2], tr[s * 4 + 3]);
    const float d = wrap_angle(th - th_last);
    float* o = obs_in + ((size_t)a * hist + tid) * obs_dim;
    o[0] = dx * ct - dy * st;
    o[1] = dy * ct + dx * st;
    o[2] = sinf(d);
    o[3] = cosf(d);
    o[4] = rvx[tid + 1];
    o[5] = rvy[tid + 1];
    o[6] = (rvx[tid + 1] - rvx[tid]) / dt;
    o[7] = (rvy[tid + 1] - rvy[tid]) / dt;
    const float* si = static_in + ((size_t)a * hist + tid) * obs_dim;
    for (int f = 8; f < obs_dim; ++f) o[f] = si[f];
  }
}
__global__ __launch_bounds__(64) void k_step_env(const float* __restrict__ traj, const float* __restrict__ vel, int stride_steps,
                                                 int last, int hist, float dt, const float* __restrict__ init_pos,
                                                 const float* __restrict__ init_head, const float* __restrict__ static_in,
                                                 int obs_dim, float* __restrict__ obs_in, float* __restrict__ cur_pos,
                                                 float* __restrict__ cur_ori, int write_obs, float* __restrict__ tok_pos,
                                                 float* __restrict__ tok_ori, StepLog lg,
                                                 int fd_vel /*PRED_VEL False: velocities from position differences (traj_sam.py:259, :553-554)*/) {
  step_env_body(blockIdx.x, threadIdx.x, traj, vel, stride_steps, last, hist, dt, init_pos, init_head, static_in, obs_dim, obs_in, cur_pos, cur_ori,
                write_obs, tok_pos, tok_ori, lg, fd_vel);
}
// what k_policy_head_row needs to run the next replan's step_env in its tail (on == 0: not this launch)
struct StepNext {
  int on, last, hist, obs_dim, fd_vel;
  float dt;
  const float *init_pos, *init_head, *static_in;
  float *obs_in, *cur_pos, *cur_ori, *tok_pos, *tok_ori;
  StepLog lg;
};

// init_agent_trajs (traj_sam.py:597-633): history -> state buffers (NaN -> 0).
__global__ void k_init_state(const float* __restrict__ obs_input, const int* __restrict__ rows, int n_agents, int hist,
                             int obs_dim, int stride_steps, float* __restrict__ traj, float* __restrict__ vel) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_agents * hist) return;
  const int a = i / hist, s = i % hist;
  const float* o = obs_input + ((size_t)rows[a] * hist + s) * obs_dim;
  float* t = traj + ((size_t)a * stride_steps + s) * 4;
  float* v = vel + ((size_t)a * stride_steps + s) * 2;
  for (int f = 0; f < 4; ++f) { const float x = o[f]; t[f] = (x != x) ? 0.f : x; }
  for (int f = 0; f < 2; ++f) { const float x = o[4 + f]; v[f] = (x != x) ? 0.f : x; }
}

// ------------------------------------------------------------------------------------------
// K9-K11  ActDecoder._compute_traj, anchor mode (act_decoder.py:78-140; CG_stacked mlp.py:207-241)
// fused with step_agent_traj (traj_sam.py:276-349, TOP_K = 1 -> mode 0).
struct HeadW {
  const float* anchors;                                   // [K*types][128]
  const float *cgWt[3], *cgb[3], *cglnw[3], *cglnb[3];    // CG_decode.CGs[i].MLP: Linear K-major [128][128] + LN
  const float *m0t, *m0b, *m0lnw, *m0lnb;                 // motion_head: 128 -> 128 (LN, ReLU)
  const float *m1t, *m1b, *m1lnw, *m1lnb;                 //              128 -> 64  (LN, ReLU), K-major [128][64]
  const float *m2t, *m2b;                                 //              64 -> out, K-major [64][128] (zero-padded)
  Mlp3W motion;                                           // (torch layout, kept for reference/tests)
  // the six Linears as split-fp16 MFMA B fragments (layout: pn_gemm); m2's output columns are zero-padded to 64
  const _Float16 *cgF[3], *m0F, *m1F, *m2F;
  const _Float16 *cgQ[3], *m0Q, *m1Q, *m2Q;               // the same with the row-tile kernels' K order (ps_rowtile.h k_policy_head_rt)
};

// relu(LayerNorm(C[r][0:N] + bias)) for 16 rows, a lane quad per row: thread t < 64 -> row t >> 2, columns
// (t & 3) * N/4 ..+N/4 returned in a[]
template <int N>
__device__ __forceinline__ void head_ln16(const float* __restrict__ C, const float* __restrict__ bias, const float* __restrict__ lw,
                                          const float* __restrict__ lb, float eps, float (&a)[N / 4]) {
  const int r = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * (N / 4);
#pragma unroll
  for (int i = 0; i < N / 4; ++i) a[i] = C[r * PN_CS + c0 + i] + bias[c0 + i];
  float sm = 0.f;
#pragma unroll
  for (int i = 0; i < N / 4; ++i) sm += a[i];
  sm += dpp_xor1(sm);
  sm += dpp_xor2(sm);
  const float mean = sm * (1.f / (float)N);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < N / 4; ++i) {
    a[i] -= mean;
    sq = fmaf(a[i], a[i], sq);
  }
  sq += dpp_xor1(sq);
  sq += dpp_xor2(sq);
  const float rstd = 1.f / sqrtf(sq * (1.f / (float)N) + eps);
#pragma unroll
  for (int i = 0; i < N / 4; ++i) a[i] = fmaxf(fmaf(a[i] * rstd, lw[c0 + i], lb[c0 + i]), 0.f);
}

// On the matrix cores: G = 16 / K agents per 256-thread workgroup (K = motion modes; one row per (agent, mode)), every Linear
// a 16-row split-fp16 GEMM (pn_gemm<1>) against pre-split weight fragments; the LayerNorms take a lane quad per row
// (64 threads).  CG_stacked's context of an agent is the maximum over its K mode rows (mlp.py:207-241, all-true mask).
// step_agent_traj appends the mode choice[agent] of this replan (traj_sam.py:300-313: an index among the top-k modes,
// drawn on the host -- motion_prob is all ones, so the draw does not depend on the model's output); nullptr = mode 0.
__global__ __launch_bounds__(256) void k_policy_head_mfma(HeadW w, const float* __restrict__ fused, const int* __restrict__ agent_type,
                                                         int n_agents, int motion_k, int steps, int sdim,
                                                         float* __restrict__ motion_pred, float* __restrict__ traj,
                                                         float* __restrict__ vel, int stride_steps, int last, int replan, float eps,
                                                         const int* __restrict__ choice, const float* __restrict__ noise /*[A][K][steps][2] or null*/,
                                                         int vcol /*first velocity column: 3, or 6 with PRED_GMM; -1: no velocity (PRED_VEL False)*/,
                                                         int mlp_mode /*TRAJ.PRED_MODE 'mlp' (act_decoder.py:90-91): one row per AGENT, no
                                                                        anchors and no CG_decode, the K modes are column blocks of the last Linear*/) {
  __shared__ __attribute__((aligned(16))) _Float16 Ah[16 * PN_AS], Al[16 * PN_AS];
  __shared__ __attribute__((aligned(16))) float C[16 * PN_CS], ctx[16 * PN_CS], inp[16 * PN_CS], Y[16 * PN_CS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = motion_k, G = mlp_mode ? 16 : 16 / K;   // agents per workgroup; rows g * K + k (mlp: row g), the other rows idle
  const int KR = mlp_mode ? 1 : K;                      // rows per agent
  const int ag0 = blockIdx.x * G;
  PnFrags fr;   // the next GEMM's weight fragments always leave before the barrier / epilogue in front of it
  pn_load(fr, mlp_mode ? w.m0F : w.cgF[0], 4, wave, lane);
  for (int i = tid; i < 16 * 128; i += 256) {
    const int row = i >> 7, c = i & 127;
    const int g = row / KR, k = row - g * KR;
    const int ag = (g < G && ag0 + g < n_agents) ? ag0 + g : n_agents - 1;
    const float cv = fused[(size_t)ag * 128 + c];
    ctx[row * PN_CS + c] = cv;      // (every mode row carries its agent's context)
    const float av = mlp_mode ? cv : w.anchors[(size_t)((agent_type[ag] - 1) * K + (g < G ? k : 0)) * 128 + c];   // anchor of (type, mode)
    Ah[row * PN_AS + c] = f16_hi(av);
    Al[row * PN_AS + c] = f16_los(av);
  }
  __syncthreads();
  // CG_stacked(3): block i: y = relu(LN(W inp + b)) * context; context' = max over the agent's modes of y
  for (int i = 0; i < (mlp_mode ? 0 : 3); ++i) {
    pn_mma<1>(fr, Ah, Al, 4, C, PN_CS, 16, wave, lane);
    pn_load(fr, i < 2 ? w.cgF[i + 1] : w.m0F, 4, wave, lane);
    __syncthreads();
    float a[32];
    const int r = tid >> 2, c0 = (tid & 3) * 32;
    if (tid < 64) {
      head_ln16<128>(C, w.cgb[i], w.cglnw[i], w.cglnb[i], eps, a);
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        a[j] *= ctx[r * PN_CS + c0 + j];
        Y[r * PN_CS + c0 + j] = a[j];
      }
    }
    __syncthreads();
    if (tid < 64) {
      const int g = r / K;
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const float y = a[j];
        float ymax = y;
        if (K > 1 && g < G) {
          ymax = Y[(g * K) * PN_CS + c0 + j];
          for (int k = 1; k < K; ++k) ymax = fmaxf(ymax, Y[(g * K + k) * PN_CS + c0 + j]);
        }
        float ni, nc;
        if (i == 0) {
          ni = y;
          nc = ymax;
        } else {
          ni = (inp[r * PN_CS + c0 + j] * (float)i + y) / (float)(i + 1);
          nc = (ctx[r * PN_CS + c0 + j] * (float)i + ymax) / (float)(i + 1);
        }
        inp[r * PN_CS + c0 + j] = ni;
        ctx[r * PN_CS + c0 + j] = nc;
        Ah[r * PN_AS + c0 + j] = f16_hi(ni);
        Al[r * PN_AS + c0 + j] = f16_los(ni);
      }
    }
    __syncthreads();
  }
  // motion_head: 128 -> 128 (LN, ReLU) -> 64 (LN, ReLU) -> steps*sdim
  pn_mma<1>(fr, Ah, Al, 4, C, PN_CS, 16, wave, lane);
  pn_load(fr, w.m1F, 4, wave, lane, 4);
  __syncthreads();
  if (tid < 64) {
    float a[32];
    const int r = tid >> 2, c0 = (tid & 3) * 32;
    head_ln16<128>(C, w.m0b, w.m0lnw, w.m0lnb, eps, a);
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      Ah[r * PN_AS + c0 + j] = f16_hi(a[j]);
      Al[r * PN_AS + c0 + j] = f16_los(a[j]);
    }
  }
  __syncthreads();
  pn_mma<1>(fr, Ah, Al, 4, C, PN_CS, 16, wave, lane, 4);
  const int nt2 = (steps * sdim * (mlp_mode ? K : 1) + 15) / 16;   // n-tiles of the last Linear (<= 8)
  pn_load(fr, w.m2F, 2, wave, lane, nt2);
  __syncthreads();
  if (tid < 64) {
    float a[16];
    const int r = tid >> 2, c0 = (tid & 3) * 16;
    head_ln16<64>(C, w.m1b, w.m1lnw, w.m1lnb, eps, a);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      Ah[r * PN_AS + c0 + j] = f16_hi(a[j]);
      Al[r * PN_AS + c0 + j] = f16_los(a[j]);
    }
  }
  __syncthreads();
  pn_mma<1>(fr, Ah, Al, 2, C, PN_CS, 16, wave, lane, nt2);
  __syncthreads();
  // cumsum over steps of (dx, dy, dtheta); wrap theta (act_decoder.py:117-121).  One thread per (row, step):
  // it re-adds the prefix in step order, so the sums round exactly like the sequential scan.
  for (int i = tid; i < G * K * steps; i += 256) {
    const int row = i / steps, s = i - row * steps;
    const int g = row / K, k = row - g * K, ag = ag0 + g;
    if (ag >= n_agents) continue;
    // the (agent, mode)'s steps * sdim outputs: its own row, or (mlp) the mode's column block of the agent's row
    const float* o = mlp_mode ? C + g * PN_CS + k * steps * sdim : C + row * PN_CS;
    const float* ob = mlp_mode ? w.m2b + k * steps * sdim : w.m2b;
    float cx = 0.f, cy = 0.f, ch = 0.f;
    const float* nz = noise ? noise + ((size_t)ag * K + k) * steps * 2 : nullptr;
    for (int j = 0; j <= s; ++j) {
      // (act_decoder.py:113-117: the noise joins the step before the cumulative sum)
      const float dx = o[j * sdim] + ob[j * sdim], dy = o[j * sdim + 1] + ob[j * sdim + 1];
      cx += nz ? dx + nz[2 * j] : dx;
      cy += nz ? dy + nz[2 * j + 1] : dy;
      ch += o[j * sdim + 2] + ob[j * sdim + 2];
    }
    const float hh = wrap_angle(ch);
    float* mp = motion_pred + ((size_t)ag * K + k) * steps * sdim + s * sdim;
    mp[0] = cx;
    mp[1] = cy;
    mp[2] = hh;
    for (int f = 3; f < sdim; ++f) mp[f] = o[s * sdim + f] + ob[s * sdim + f];
    const int pick = choice ? choice[ag] : 0;
    if (s < replan && k == pick) {
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
      if (vcol >= 0) {   // (PRED_VEL; without it the rollout keeps no velocity track, traj_sam.py:337)
        const float vx = o[s * sdim + vcol] + ob[s * sdim + vcol], vy = o[s * sdim + vcol + 1] + ob[s * sdim + vcol + 1];
        v[0] = vx * cl - vy * sl;
        v[1] = vy * cl + vx * sl;
      }
    }
  }
}

// Closed-loop displacement of the rolled-out xy from a ground-truth future in the agent-init frame (the quantity the
// Sim-Agents style evaluation of rollout/ works on): per agent the mean over the steps whose ground truth is finite and
// the displacement at the LAST such step -- the NaN-masked target / last-valid-index conventions of
// metrics/motion_pred.py:31-76.  gt [A][steps][2] (NaN = no ground truth at that step), or nullptr: the path length
// from the origin of the agent-init frame.  out [A][2]; NaN for log-replay agents and for agents without a valid step.
__global__ void k_rollout_metric(const float* __restrict__ traj, int stride_steps, int hist, int steps,
                                 const float* __restrict__ gt /*[A][steps][2] or null*/, int n_agents,
                                 float* __restrict__ out, const int* __restrict__ is_policy) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n_agents) return;
  const float nan = __int_as_float(0x7fc00000);
  if (is_policy && !is_policy[a]) {   // log-replay agents are not simulated: no metric row
    out[2 * a] = nan;
    out[2 * a + 1] = nan;
    return;
  }
  float sum = 0.f, last = nan;
  int cnt = 0;
  for (int s = 0; s < steps; ++s) {
    const float* t = traj + ((size_t)a * stride_steps + hist + s) * 4;
    const float gx = gt ? gt[((size_t)a * steps + s) * 2] : 0.f, gy = gt ? gt[((size_t)a * steps + s) * 2 + 1] : 0.f;
    if (gx != gx || gy != gy) continue;
    const float dx = t[0] - gx, dy = t[1] - gy;
    last = sqrtf(dx * dx + dy * dy);
    sum += last;
    ++cnt;
  }
  out[2 * a] = cnt ? sum / (float)cnt : nan;
  out[2 * a + 1] = last;
}

// The reference's validation metric PairMotionPred (metrics/motion_pred.py:111-199) on the device, one thread per
// agent row.  Pairs = (replan r, agent a) with pair_mask[r][a]; per pair the K-mode errors of _update_traj_error
// (:31-76): a step counts unless BOTH target coordinates are NaN, ade_k = sum / count (NaN without a valid step),
// fde_k = the masked distance at the last valid step (0 without one: index -1 picks the masked last step),
// ade / fde of the arg-max-probability mode, min over the modes (a NaN mode makes the minimum NaN, as torch.min does).
// Then the chained per-replan predictions against the chained targets (loss_func.py:215-313 rollout_traj /
// rollout_temp_traj_preds, PRED_GMM False): every replan's first `rs` steps, rotated by the wrapped cumulative heading
// of the replans before it, summed up; mean distance over the steps whose target x and y are finite (:125-143).
// tgt [R][A][S][5] (NaN = missing), pair_mask [R][A], pred [R][A][K][S][sd], prob [R][A][K] or nullptr (mode 0).
// out [A][10] = (sum ade, sum fde, sum min_ade, sum min_fde, the four counts of finite entries behind those sums -- what
// torchmetrics.MeanMetric keeps of an update --, rollout ade of the agent, 1 if the agent has a valid rollout step);
// NaN row for agents that are not policy agents.
__global__ void k_pair_metric(const float* __restrict__ pred, const float* __restrict__ prob, const float* __restrict__ tgt,
                              const uint8_t* __restrict__ pair_mask, int R, int A, int K, int S, int sd, int rs,
                              const int* __restrict__ is_policy, float* __restrict__ out) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= A) return;
  const float nan = __int_as_float(0x7fc00000);
  float* o = out + (size_t)a * 10;
  if (is_policy && !is_policy[a]) {
    for (int i = 0; i < 10; ++i) o[i] = nan;
    return;
  }
  float s_ade = 0.f, s_fde = 0.f, s_made = 0.f, s_mfde = 0.f;
  int n_ade = 0, n_fde = 0, n_made = 0, n_mfde = 0;
  // chained trajectories: wrapped cumulative heading and running position of target and prediction
  float th_t = 0.f, th_p = 0.f;          // unwrapped sums of the replans' heading steps so far
  float ptx = 0.f, pty = 0.f, ppx = 0.f, ppy = 0.f;
  float r_sum = 0.f;
  int r_cnt = 0;
  for (int r = 0; r < R; ++r) {
    const bool pm = pair_mask[(size_t)r * A + a] != 0;
    const float* tg = tgt + ((size_t)r * A + a) * S * 5;
    int kidx = 0;
    if (pm) {
      if (prob) {
        float best = prob[((size_t)r * A + a) * K];
        for (int k = 1; k < K; ++k) {
          const float v = prob[((size_t)r * A + a) * K + k];
          if (v > best) { best = v; kidx = k; }
        }
      }
      int last = -1, cnt = 0;
      for (int s = 0; s < S; ++s) {
        const float gx = tg[s * 5], gy = tg[s * 5 + 1];
        if (!(gx != gx && gy != gy)) { last = s; ++cnt; }
      }
      float ade_sel = 0.f, fde_sel = 0.f, ade_min = 0.f, fde_min = 0.f;
      bool ade_min_nan = false, fde_min_nan = false;
      for (int k = 0; k < K; ++k) {
        const float* pp = pred + (((size_t)r * A + a) * K + k) * S * sd;
        float sum = 0.f, fde = 0.f;
        for (int s = 0; s < S; ++s) {
          const float gx = tg[s * 5], gy = tg[s * 5 + 1];
          if (gx != gx && gy != gy) continue;                     // masked step: distance filled with 0
          const float dx = gx - pp[s * sd], dy = gy - pp[s * sd + 1];
          const float d = sqrtf(dx * dx + dy * dy);               // NaN when only one coordinate is missing (as the reference)
          sum += d;
          if (s == last) fde = d;
        }
        const float ade = sum / (float)cnt;                        // 0 / 0 = NaN without a valid step
        if (k == kidx) { ade_sel = ade; fde_sel = fde; }
        if (k == 0) { ade_min = ade; fde_min = fde; }
        ade_min_nan |= ade != ade;
        fde_min_nan |= fde != fde;
        ade_min = fminf(ade_min, ade);
        fde_min = fminf(fde_min, fde);
      }
      if (ade_min_nan) ade_min = nan;
      if (fde_min_nan) fde_min = nan;
      // MeanMetric drops the NaN entries of each metric separately
      if (ade_sel == ade_sel) { s_ade += ade_sel; ++n_ade; }
      if (fde_sel == fde_sel) { s_fde += fde_sel; ++n_fde; }
      if (ade_min == ade_min) { s_made += ade_min; ++n_made; }
      if (fde_min == fde_min) { s_mfde += fde_min; ++n_mfde; }
    }
    // ---- chained rollout of this replan (target zero-filled where missing or unpaired, prediction zero where unpaired)
    const float ct = cosf(wrap_angle(th_t)), st_ = sinf(wrap_angle(th_t));
    const float cp = cosf(wrap_angle(th_p)), sp_ = sinf(wrap_angle(th_p));
    const float* pp = pred + (((size_t)r * A + a) * K + kidx) * S * sd;
    float ltx = 0.f, lty = 0.f, lpx = 0.f, lpy = 0.f;            // previous step inside the replan (diff)
    for (int s = 0; s < rs; ++s) {
      float gx = tg[s * 5], gy = tg[s * 5 + 1];
      const bool vx = pm && gx == gx, vy = pm && gy == gy;
      gx = vx ? gx : 0.f;
      gy = vy ? gy : 0.f;
      const float px = pm ? pp[s * sd] : 0.f, py = pm ? pp[s * sd + 1] : 0.f;
      const float dtx = gx - ltx, dty = gy - lty, dpx = px - lpx, dpy = py - lpy;
      ltx = gx; lty = gy; lpx = px; lpy = py;
      ptx += dtx * ct - dty * st_;
      pty += dty * ct + dtx * st_;
      ppx += dpx * cp - dpy * sp_;
      ppy += dpy * cp + dpx * sp_;
      if (vx && vy) {
        const float ex = ptx - ppx, ey = pty - ppy;
        r_sum += sqrtf(ex * ex + ey * ey);
        ++r_cnt;
      }
    }
    {
      const float gh = tg[(rs - 1) * 5 + 2];
      th_t += (pm && gh == gh) ? gh : 0.f;
      th_p += pm ? pp[(rs - 1) * sd + 2] : 0.f;
    }
  }
  o[0] = s_ade; o[1] = s_fde; o[2] = s_made; o[3] = s_mfde;
  o[4] = (float)n_ade; o[5] = (float)n_fde; o[6] = (float)n_made; o[7] = (float)n_mfde;
  o[8] = r_sum / (float)(r_cnt > 0 ? r_cnt : 1);
  o[9] = r_cnt > 0 ? 1.f : 0.f;
}

// policy_emd += x_p after the condition layers (condition_attns.py:226)
__global__ void k_add_rows(float* __restrict__ dst, const float* __restrict__ src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

__global__ void k_fourier_test(const float* __restrict__ x4, int n, const float* __restrict__ div32, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 128) return;
  const int e = i >> 7, c = i & 127;
  out[i] = fourier_feat(x4[e * 4 + (c >> 5)], c & 31, div32);
}
__global__ void k_wrap_test(const float* __restrict__ x, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = wrap_angle(x[i]);
}

}  // namespace ps
