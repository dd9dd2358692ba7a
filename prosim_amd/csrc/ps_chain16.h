// Fused attention-layer chain, second generation (gfx950): node work on the matrix cores for up to 16 destination
// rows per workgroup, edge work one wave per destination with the relative-PE rows RECOMPUTED from 32 bytes of
// geometry per edge instead of streamed from two 384-byte operand images.
//
// Reference math: AttentionLayer.forward (prosim/models/layers/attention_layer.py:56-121), rel-PE rows
// act_decoder.py:203-221 (and twins) through FourierEmbeddingFix (fourier_embedding.py:63-78).
//
// Why (round-1 profile of the 1024-agent policy launch): k_attn_chain streams every layer's
// 960 KB of fp32 weights through each 2-4-row workgroup (21 TB/s of L2 -> CU traffic, half of the launch) and reads
// 1.05 GB of rel-PE operand images per launch (768 B per edge and layer).  Here
//   * a workgroup (4 waves) owns R <= 16 rows; every node Linear is a 16-row split-fp16 MFMA GEMM against pre-split
//     weight fragments (k_node's machinery: the fragment ring stays in flight across epilogues and barriers), so the
//     weights cross the CU once per 16 rows;
//   * the edge phase takes one WAVE per destination (rows are handed out longest first through an LDS counter; W = 8 / rows
//     waves per row below 8 rows per workgroup), walks the edge list in 16-edge tiles with an online softmax (no workgroup
//     barrier inside), and rebuilds each tile's normalised Fourier rows in registers from (2 pi dist, 2 pi rel_ori,
//     2 pi angle, rstd, -mean rstd): per (edge, frequency) one exact division by reciprocal + correction, one reduction to
//     revolutions, v_sin_f32 / v_cos_f32.  The rows feed the score MFMAs straight from registers and the aggregation
//     MFMAs through a wave-private row-major LDS tile read back with ds_read_b64_tr_b16 (recipe: tools/mb/mb_trread.hip);
//     a row's tiles are always summed in two parity classes merged (even, odd), whatever the tiling (c16_edge_phase);
//   * q~ and a_r pass between the phases through LDS slots of the workgroup, the other per-destination vectors (s, g,
//     a_v, l) through the workgroup's own rows of the EdgeIO scratch (L2-resident; same-CU visibility needs only the
//     workgroup barrier).
// The 12 policy layers stay one launch per replan; 8 waves per workgroup, one workgroup per CU (150 KB LDS, 256 registers).
#pragma once
#include "ps_attn.h"

namespace ps {

// geometry of one edge, made once per edge set (k_edge_geo) and read by every layer of the set
#ifndef PS_GEO_VARIANT
#define PS_GEO_VARIANT 0
#endif
struct EdgeGeo {
  float a0, a1, a2;   // 2 pi * (dist, rel_ori, angle): the three distinct FourierEmbeddingFix inputs, already scaled (:66)
  float rstd, nmr;    // affine-free LayerNorm of the 128-feature row: y = f * rstd + nmr (nmr = -mean * rstd)
  int src;            // source row (as in esrc)
  int pad0, pad1;
};
static_assert(sizeof(EdgeGeo) == 32, "EdgeGeo is two 16-byte loads");

// x / d for the 16 divisors dim_t[2k] of FourierEmbeddingFix(32), by reciprocal + one correction step: equals the
// IEEE quotient for every |x| in [2^-24, 65536) and for 0 (checked exhaustively over those inputs: tools/check_fdiv.c;
// the only mismatches over ALL finite inputs below 65536 sit where the quotient is denormal, |x| < 1e-34 -- a last-bit
// difference of an argument whose sine is the argument); at and beyond 65536 the caller takes the true division.
__device__ __forceinline__ float fdiv16(float x, float d, float rd) {
  const float q0 = x * rd;
  const float rem = fmaf(-q0, d, x);
  return fmaf(rem, rd, q0);
}
__device__ __forceinline__ bool fdiv16_ok(float x) {
  const float a = fabsf(x);
  return a < 65536.f;
}
// sin and cos of |a| < 2^16 on the transcendental unit: v_sin_f32 / v_cos_f32 take REVOLUTIONS, so the argument is turned
// into a / (2 pi) with the product's rounding error recovered by an fma (C1 + C2 = 1 / (2 pi) to fp64 precision) and the
// integer turns dropped before the two instructions.  Max abs error 2.5e-7 against the exact values over the arguments
// the rel-PE rows produce (tools/mb/mb_vsin.hip, measured on MI355X; torch's own sin / cos: 6e-8) -- 7 issue slots per
// pair where the Cody-Waite + degree-7 polynomial version (7.4e-8; tools/check_sincos.c) took 26: the edge phase of
// this kernel is bound by VALU issue, and the rows are rounded to 22 bits (split fp16) right after.
__device__ __forceinline__ void sincos_hw(float a, float& s, float& c) {
  constexpr float C1 = 0.15915494309189535f;
  constexpr float C2 = (float)(0.15915494309189535 - (double)0.15915494309189535f);
  const float u = a * C1;
  const float n = rintf(u);
  const float f = fmaf(a, C1, -n) + a * C2;
  s = __builtin_amdgcn_sinf(f);
  c = __builtin_amdgcn_cosf(f);
}
// one (sin, cos) pair of the embedding: argument (x 2 pi) / dim_t exactly as the reference rounds it, then sin / cos
__device__ __forceinline__ void fourier_pair(float xs, float d, float rd, bool fast, float& s, float& c) {
  if (fast) {
    sincos_hw(fdiv16(xs, d, rd), s, c);
  } else {
    sincosf(xs / d, &s, &c);
  }
}

// ---- per edge set and replan: the geometry records.  One thread per edge: the three scalars (act_decoder.py:203-217),
// then all 48 distinct (sin, cos) pairs once for the LayerNorm statistics of the 128-feature row (the angle block
// counts twice: features 96..127 repeat 64..95).
struct GeoSet {
  const int *esrc, *edst, *eoff;
  int nq;
  const float *src_ori, *dst_pos, *dst_ori;
  EdgeGeo* geo;
};
struct GeoSets {
  GeoSet s[2];
};
// raw != 0: the three inputs as they are (dist, rel_ori, angle), no Fourier statistics -- what the learnable embedding
// (ps_pe_learn.h) consumes.
// The 96 (sin, cos) values of an edge are made ONCE and kept in LDS between the two LayerNorm passes ([value][thread]: a
// wave's 64 lanes on 64 banks); round 2 recomputed them for the second pass (half of the kernel's instructions).
constexpr int GEO_THREADS = 128;
constexpr size_t GEO_LDS_BYTES = 16;   // (round 3 kept the 96 values of an edge in LDS between two LayerNorm passes)
// the record of ONE edge: source token s seen from a destination at (px, py) with heading od (cx = cosf(od), cy = sinf(od): the caller's, once per
// destination where it can).  Shared by k_edge_geo (one thread per edge of a finished CSR) and k_radius_geo (the search's own waves): same
// operations on the same values in the same order, so the records do not depend on which kernel made them.
__device__ __forceinline__ EdgeGeo geo_record(int s, float px, float py, float od, float cx, float cy, const float* __restrict__ src_pos,
                                              const float* __restrict__ src_ori, const float (&dv)[16], const float (&rdv)[16], float eps, int raw) {
  const float dx = src_pos[2 * s] - px, dy = src_pos[2 * s + 1] - py;
  float xin[3];
  xin[0] = sqrtf(dx * dx + dy * dy);
  xin[1] = wrap_angle(src_ori[s] - od);
  // torch's .sum(dim=-1) accumulates from +0, so a dot of (-0, -0) is +0 there: keep the explicit 0.f + ...
  const float dot = (0.f + cx * dx) + cy * dy;
  xin[2] = atan2f(cx * dy - cy * dx, dot);
  EdgeGeo g;
  if (raw) {
    g.a0 = xin[0]; g.a1 = xin[1]; g.a2 = xin[2];
    g.rstd = 0.f; g.nmr = 0.f;
    g.src = s; g.pad0 = 0; g.pad1 = 0;
    return g;
  }
  float xs[3];
  bool fast = true;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    xs[i] = xin[i] * PS_TWO_PI_F;
    fast = fast && fdiv16_ok(xs[i]);
  }
  // mean of the 128 features; their second moment needs no second pass: every (sin, cos) pair contributes sin^2 + cos^2 = 1, so the
  // sum of squares of the 64 pairs (the angle block counts twice) is 64 and sum (f - mean)^2 = 64 - 128 mean^2.  (Round 2 kept
  // torch's two passes bit for bit -- a last-bit change of rstd re-rolled the workload's near-cut edges; with the quieter GEMM
  // operands of round 4 the parity table holds either way, and the pass was a third of this kernel: 96 LDS round trips per edge.)
  float sm = 0.f;
#if PS_GEO_VARIANT == 1   // (hunt: scalar form -- fourier_pair per frequency, no packed instructions)
  if (fast) {
    for (int i = 0; i < 3; ++i) {
      float part = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        float sv, cv;
        fourier_pair(xs[i], dv[k], rdv[k], true, sv, cv);
        part += sv + cv;
      }
      sm += (i == 2) ? 2.f * part : part;
    }
  } else
#endif
  if (__builtin_expect(fast, 1)) {   // (the two forms in branches of their own: sharing one loop, libm's sincosf kept the kernel at 145 registers)
    // two frequencies per packed instruction through the exact division and the reduction to revolutions (round 5: feat8's scheme, the
    // operations of fourier_pair / sincos_hw in the same order on the same values -- the sums keep their order, the records their bits)
    constexpr float C1 = 0.15915494309189535f;
    constexpr float C2 = (float)(0.15915494309189535 - (double)0.15915494309189535f);
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    f32x2_ c1 = {C1, C1}, c2 = {C2, C2};
#if PS_GEO_VARIANT == 6 || PS_GEO_VARIANT == 7   // (hunt: the packed constants in VECTOR registers)
    asm volatile("" : "+v"(c1), "+v"(c2));
#endif
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float part = 0.f;
      const f32x2_ x2 = {xs[i], xs[i]};
#pragma unroll
      for (int k = 0; k < 16; k += 2) {
        f32x2_ d2 = {dv[k], dv[k + 1]}, rd2 = {rdv[k], rdv[k + 1]};
#if PS_GEO_VARIANT == 5 || PS_GEO_VARIANT == 6   // (hunt: the divisors in VECTOR registers -- no scalar-pair operand in a packed instruction)
        asm volatile("" : "+v"(d2), "+v"(rd2));
#endif
        const f32x2_ q0 = x2 * rd2;
        const f32x2_ rem = __builtin_elementwise_fma(-q0, d2, x2);
        const f32x2_ q = __builtin_elementwise_fma(rem, rd2, q0);
        const f32x2_ u = q * c1;
        const f32x2_ n = {rintf(u.x), rintf(u.y)};
        const f32x2_ f = __builtin_elementwise_fma(q, c1, -n) + q * c2;
#if PS_GEO_VARIANT == 2   // (hunt: the trans sources stay live until the sums that read the results are made)
        part += __builtin_amdgcn_sinf(f.x) + __builtin_amdgcn_cosf(f.x);
        part += __builtin_amdgcn_sinf(f.y) + __builtin_amdgcn_cosf(f.y);
        asm volatile("" : "+v"(part) : "v"(f.x), "v"(f.y));
#elif PS_GEO_VARIANT == 4   // (hunt: sin, cos and the add that consumes both in ONE asm block -- nothing can overwrite the source before the results are read)
        float sc0_, sc1_, t0_, t1_;
        asm volatile("v_sin_f32 %0, %2\n\tv_cos_f32 %1, %2\n\ts_nop 1\n\tv_add_f32 %0, %0, %1" : "=&v"(sc0_), "=&v"(t0_) : "v"(f.x));
        asm volatile("v_sin_f32 %0, %2\n\tv_cos_f32 %1, %2\n\ts_nop 1\n\tv_add_f32 %0, %0, %1" : "=&v"(sc1_), "=&v"(t1_) : "v"(f.y));
        part += sc0_;
        part += sc1_;
#else
        part += __builtin_amdgcn_sinf(f.x) + __builtin_amdgcn_cosf(f.x);
        part += __builtin_amdgcn_sinf(f.y) + __builtin_amdgcn_cosf(f.y);
#endif
      }
      sm += (i == 2) ? 2.f * part : part;
    }
  } else {
#pragma unroll 1
    for (int i = 0; i < 3; ++i) {
      float part = 0.f;
#pragma unroll 1
      for (int k = 0; k < 16; ++k) {
        float sv, cv;
        fourier_pair(xs[i], dv[k], rdv[k], false, sv, cv);
        part += sv + cv;
      }
      sm += (i == 2) ? 2.f * part : part;
    }
  }
  const float mean = sm * (1.f / 128.f);
  const float sq = 64.f - 128.f * mean * mean;
  const float rstd = 1.f / sqrtf(sq * (1.f / 128.f) + eps);
  g.a0 = xs[0]; g.a1 = xs[1]; g.a2 = xs[2];
  g.rstd = rstd;
  g.nmr = -mean * rstd;
  g.src = s;
  g.pad0 = g.pad1 = 0;
  return g;
}
__global__ __launch_bounds__(GEO_THREADS) void k_edge_geo(GeoSets sets, const float* __restrict__ src_pos, const float* __restrict__ div32,
                                                         float eps, int raw) {
  const GeoSet& S = sets.s[blockIdx.y];
  const int E = S.eoff[S.nq];
  float dv[16], rdv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    dv[k] = div32[2 * k];
    rdv[k] = 1.0f / dv[k];
  }
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
    const int d = S.edst[e], s = S.esrc[e];
    const float px = S.dst_pos[2 * d], py = S.dst_pos[2 * d + 1], od = S.dst_ori[d];
    const float cx = cosf(od), cy = sinf(od);
    S.geo[e] = geo_record(s, px, py, od, cx, cy, src_pos, S.src_ori, dv, rdv, eps, raw);
  }
}

// ---- Round 5: a radius search with geometry records as ONE launch (was: count pass, fill pass, k_edge_geo -- three dependent graph nodes per
// search, 8 + 3 searches per rollout).  One WORKGROUP of four waves per query:
//   1. ONE scan of the scene's candidates (k_radius scanned them twice, once per pass), 1024 candidates per trip, wave w ranking the trip's chunks
//      4 w .. 4 w + 3 behind the chunks before them (counts exchanged through LDS): the first cap (+ 1 with a self match to drop) hits in index order go to the
//      workgroup's LDS list -- the order k_radius's single wave gives;
//   2. the workgroup publishes its edge count (one 64-bit agent-scope atomic: bit 63 | tiles | edges) and its first wave sums the counts of the
//      queries before it straight from the flags, waiting for those not published yet: a workgroup publishes BEFORE it waits, only ever waits for
//      lower indices, and workgroups are dispatched in index order, so the lowest unfinished workgroup never waits for an undispatched one (the
//      decoupled look-back argument, without the chain: nothing is forwarded from workgroup to workgroup).  The argument holds per launch; with
//      SEVERAL such launches in flight on a full GPU (engines side by side) one launch's waiting workgroups can hold the slots another launch's
//      lowest workgroup needs, and in-order dispatch is an observation, not a contract -- so the wait is BOUNDED (spin_limit polls, ~0.5 ms) and a
//      count that has not appeared by then is computed by the waiting wave itself (rg_count_query): no workgroup depends on another one's progress;
//   3. CSR / tile offsets, esrc / edst and the 32-byte record of every edge (geo_record: k_edge_geo's arithmetic), 256 edges at a time.
// The last workgroup through its look-back (a counter behind the flags) zeroes flags and counter for the next launch -- the replays of a captured
// graph carry no launch-specific argument.  Results are bit-identical to the three-launch form (tests/test_round5_gpu.py; ps_set_search_impl(1)
// keeps that form).
// the edge count of query qi by ONE wave (k_radius<0> + k_radius_selfrank in one scan): what the look-back falls back to for a workgroup that has not
// published in time -- the kernel then needs no forward-progress guarantee between workgroups (see k_radius_geo)
__device__ __forceinline__ int rg_count_query(const RadSet& S, const float* __restrict__ qpos, const int* __restrict__ qscene, int qi, int lane) {
  const CandSet cs = S.cs;
  const int* __restrict__ cand_ok = S.cand_ok;
  const int cand_base = S.cand_base, self_base = S.self_base;
  const float r2 = S.r2;
  const int capx = S.cap + (self_base >= 0 ? 1 : 0);
  const float qx = qpos[2 * qi], qy = qpos[2 * qi + 1];
  const int b = qscene[qi];
  const int self = (self_base >= 0 && (!cand_ok || cand_ok[self_base + qi - cand_base])) ? self_base + qi : -1;
  int run = 0;
  bool selfhit = false;
  for (int rg = 0; rg < 2 && run < capx; ++rg) {
    const int* rr = rg == 0 ? cs.r1 : cs.r2;
    if (!rr) break;
    const int beg = rr[2 * b], end = rr[2 * b + 1];
    for (int i0 = beg; i0 < end && run < capx; i0 += 64) {
      const int i = i0 + lane;
      bool ok = false;
      if (i < end) ok = dist2(cs.pos[2 * i], cs.pos[2 * i + 1], qx, qy) < r2 && (!cand_ok || i < cand_base || cand_ok[i - cand_base]);
      const unsigned long long m = __ballot(ok);
      const int rank = run + __popcll(m & ((1ull << lane) - 1ull));
      selfhit = selfhit || __ballot(ok && rank < capx && i == self) != 0ull;
      run += __popcll(m);
    }
  }
  return (run < capx ? run : capx) - (selfhit ? 1 : 0);
}
struct RadSyncs {
  unsigned long long* flag[2];   // [nq] per query: bit 63 = published | 32-edge tiles << 32 | edges; then one int: the done counter
};
// For FEW queries (a single scene: latency; the host's SEARCH_WG_MAX_Q): with thousands of queries the three-launch form is faster -- measured on
// the 8-scene batch (1024 queries x 2 sets): this kernel costs the pipelined headline 2 % (the records of a 60-edge row keep three quarters of a
// four-wave workgroup's lanes idle: 2.3 x the wave-instructions of the balanced k_edge_geo launch), a one-wave-per-query form that leaves the
// records to k_edge_geo 3.7 % (four dependent trips of agent-scope flag loads per look-back against one cached sweep).
constexpr int RG_WAVES = 4, RG_CH = 4;
constexpr bool GEO = true;
__global__ __launch_bounds__(64 * RG_WAVES) void k_radius_geo(RadSets sets, GeoSets gsets, RadSyncs sy, const float* __restrict__ qpos,
                                                             const int* __restrict__ qscene, int nq, const float* __restrict__ src_pos,
                                                             const float* __restrict__ div32, float eps, int spin_limit) {
  extern __shared__ int rg_lst[];   // [cap + 1] the query's hits, in index order
  __shared__ int wc[2][RG_WAVES];
  __shared__ int sh_self, sh_pre[2];
  const RadSet& S = sets.s[blockIdx.y];
  const GeoSet& G = gsets.s[blockIdx.y];
  unsigned long long* __restrict__ flag = sy.flag[blockIdx.y];
  const CandSet cs = S.cs;
  const float r2 = S.r2;
  const int cap = S.cap, self_base = S.self_base;
  const int* __restrict__ cand_ok = S.cand_ok;
  const int cand_base = S.cand_base;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = blockIdx.x;
  const int capx = cap + (self_base >= 0 ? 1 : 0);
  // (the records' divisors and the destination's pose: requested before the scan, used after it)
  float dv[16], rdv[16];
  float px = 0.f, py = 0.f, od = 0.f;
  if (GEO) {
#pragma unroll
    for (int k = 0; k < 16; ++k) dv[k] = div32[2 * k];
    px = G.dst_pos[2 * q]; py = G.dst_pos[2 * q + 1]; od = G.dst_ori[q];
  }
  const float qx = qpos[2 * q], qy = qpos[2 * q + 1];
  const int b = qscene[q];
  const int self = (self_base >= 0 && (!cand_ok || cand_ok[self_base + q - cand_base])) ? self_base + q : -1;
  if (tid == 0) sh_self = 0;
  int run = 0, trip = 0;
  for (int rg = 0; rg < 2 && run < capx; ++rg) {
    const int* rr = rg == 0 ? cs.r1 : cs.r2;
    if (!rr) break;
    const int beg = rr[2 * b], end = rr[2 * b + 1];
    // a trip = RG_CH consecutive 64-candidate chunks per wave (wave w: candidates i0 + 64 RG_CH w ..), all position loads of a trip in flight together
    for (int i0 = beg; i0 < end && run < capx; i0 += 64 * RG_CH * RG_WAVES, ++trip) {
      unsigned long long m[RG_CH];
      int mycnt = 0;
      bool okv[RG_CH];
#pragma unroll
      for (int u = 0; u < RG_CH; ++u) {
        const int i = i0 + 64 * (RG_CH * wave + u) + lane;
        okv[u] = false;
        if (i < end) okv[u] = dist2(cs.pos[2 * i], cs.pos[2 * i + 1], qx, qy) < r2 && (!cand_ok || i < cand_base || cand_ok[i - cand_base]);
      }
#pragma unroll
      for (int u = 0; u < RG_CH; ++u) {
        m[u] = __ballot(okv[u]);
        mycnt += __popcll(m[u]);
      }
      int* w = wc[trip & 1];   // (two buffers: a wave may write the next trip's count while a slower one still reads this trip's)
      if (lane == 0) w[wave] = mycnt;
      __syncthreads();
      int base = run, tot = 0;
#pragma unroll
      for (int u = 0; u < RG_WAVES; ++u) {
        const int c = w[u];
        tot += c;
        if (u < wave) base += c;
      }
#pragma unroll
      for (int u = 0; u < RG_CH; ++u) {
        const int i = i0 + 64 * (RG_CH * wave + u) + lane;
        const int rank = base + __popcll(m[u] & ((1ull << lane) - 1ull));
        if (okv[u] && rank < capx) {   // (a chunk behind a full list takes nobody)
          rg_lst[rank] = i;
          if (i == self) sh_self = 1;
        }
        base += __popcll(m[u]);
      }
      run += tot;
    }
  }
  __syncthreads();
  const int total = run < capx ? run : capx;
  const bool selfhit = sh_self != 0;
  const int mine = total - (selfhit ? 1 : 0);
  const int nt = (mine + 31) >> 5;
  if (wave == 0) {
    if (lane == 0) {
      S.cnt[q] = mine;
      __hip_atomic_store(flag + q, (1ull << 63) | ((unsigned long long)(unsigned)nt << 32) | (unsigned long long)(unsigned)mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the sums of the queries before this one: their flags are requested four at a time, then waited for one by one
    int pe = 0, pt = 0;
    for (int i0 = 0; i0 < q; i0 += 256) {
      unsigned long long f[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 64 * u + lane;
        f[u] = i < q ? __hip_atomic_load(flag + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (1ull << 63);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 64 * u + lane;
        for (int spins = 0; !(f[u] >> 63) && spins < spin_limit; ++spins) {
          __builtin_amdgcn_s_sleep(1);
          f[u] = __hip_atomic_load(flag + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // not published within the budget (a workgroup that could not be dispatched yet on a full GPU, or a dispatch order other than the one
        // argued above): the wave counts that query's edges itself -- no workgroup ever depends on another one's progress
        unsigned long long pend = __ballot(!(f[u] >> 63));
        while (pend) {
          const int l = (int)__ffsll((long long)pend) - 1;
          const int c = rg_count_query(S, qpos, qscene, i0 + 64 * u + l, lane);
          if (lane == l) f[u] = (1ull << 63) | ((unsigned long long)(unsigned)((c + 31) >> 5) << 32) | (unsigned long long)(unsigned)c;
          pend &= pend - 1ull;
        }
        pe += (int)(unsigned)(f[u] & 0xffffffffull);
        pt += (int)(unsigned)((f[u] >> 32) & 0x7fffffffull);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      pe += __shfl_xor(pe, o);
      pt += __shfl_xor(pt, o);
    }
    if (lane == 0) {
      sh_pre[0] = pe;
      sh_pre[1] = pt;
      S.eoff[q] = pe;
      S.toff[q] = pt;
      if (q == nq - 1) {
        S.eoff[nq] = pe + mine;
        S.toff[nq] = pt + nt;
      }
    }
    if (lane < nt) S.tdst[pt + lane] = q;
    // every look-back of the launch done -> the last one out clears flags and counter for the next launch
    int* done = reinterpret_cast<int*>(flag + nq);
    int last = 0;
    // (round 6, ADVICE round 5: this workgroup's publish must be COMPLETE before its increment can be seen -- else the last workgroup could
    // zero the flags before a late publish lands and the next replay would read a stale count.  Both are agent-scope atomics (no dirty line
    // in this XCD's L2 to write back), so waiting for the store's acknowledgement orders them; a release fence here writes the whole L2 back
    // (+ 12 us per launch, measured in round 5).  The look-back's loads above were issued after the store: the wait is over when they are.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) last = __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nq - 1 ? 1 : 0;
    last = __builtin_amdgcn_readfirstlane(last);
    if (last) {
      for (int i = lane; i < nq; i += 64) __hip_atomic_store(flag + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (lane == 0) __hip_atomic_store(done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  float cx = 0.f, cy = 0.f;
  if (GEO) {
    cx = cosf(od);
    cy = sinf(od);
#pragma unroll
    for (int k = 0; k < 16; ++k) rdv[k] = 1.0f / dv[k];
  }
  __syncthreads();
  const int se = sh_pre[0];
  for (int j = tid; j < total; j += 64 * RG_WAVES) {
    const int i = rg_lst[j];
    if (i == self) continue;
    const int o = se + j - ((selfhit && self < i) ? 1 : 0);
    S.esrc[o] = i;
    S.edst[o] = q;
    if (GEO) G.geo[o] = geo_record(i, px, py, od, cx, cy, src_pos, G.src_ori, dv, rdv, eps, 0);
  }
}

// ------------------------------------------------------------------------------------------------------------
// LDS plan (bytes): k_node's node-phase buffers, then cq and the row counter, then whatever the wave-private areas of
// the edge phase (k staging 4 KB | probability tile 0.5 KB | feature tile 3.5 KB | source rows of two tiles 128 B) need
// beyond the operand planes and the GEMM result buffer, which they alias (dead while the edge phase runs).
constexpr int C16_FS = 112;   // feature-tile row stride in halfs: 96 features + 16 pad = 56 dwords, so the 8 rows x 32 B that a
                              // half-wave's transposed read touches (and the 8 rows of a b128 write group) fall on distinct banks
constexpr size_t C16_NODE_BYTES = ND_LDS_BYTES;
constexpr size_t C16_PLANES_BYTES = (size_t)2 * ND_ROWS * ND_AS * 2 + (size_t)2 * ND_ROWS * ND_AS5 * 2 + (size_t)ND_ROWS * ND_CS * 4;   // P0 | P1 | C
constexpr size_t C16_STASH_OFF = 4096 + 512 + (size_t)16 * C16_FS * 2 + 128;   // 32 x float4 a_v + 8 x (m, l): the even tiles' sums of a row (ONEW)
constexpr size_t C16_WAVE_BYTES = C16_STASH_OFF + 512 + 64;
// q~ (PRE -> edge phase) and the edge phase's a_r sums (-> POST) stay in LDS: 16 slots of [8 heads][96 (+4)] floats.  One
// wave per row: the row's a_r overwrites its q~ (slot = row).  W waves per row (rows <= 4): a_r slots row * W + part (8 at
// most), q~ of row r in slot 8 + r.  Slot stride 808 floats = 8 mod 64: the 16 rows of a fold A-fragment read spread over
// the banks two deep.
constexpr int C16_QH = 100, C16_QSL = 8 * C16_QH + 8;
constexpr size_t C16_QA_BYTES = (size_t)16 * C16_QSL * 4;
constexpr size_t C16_PB_BYTES = (size_t)2 * ND_ROWS * ND_AS * 2 + 256 * 4;   // k_chain16: the PB operand planes + the next layer's LN_dst vectors (behind QA)
constexpr size_t C16_EDGE_WAVES_BYTES = 8 * C16_WAVE_BYTES;   // (k_edge16's wave areas)
constexpr float C16_TAU = 6.f;   // (-DPS_C16_LAZY experiment only: a class's sums are rescaled when a score exceeds the reference by more than 2^6)
constexpr int C16_CTR_INTS = 52;   // [0] row counter, [1..16] rows in queue order, [17..32] their edge counts, [36..51] their first edges
template <int NWV>
constexpr size_t c16_lds_bytes() {
  return (NWV * C16_WAVE_BYTES > C16_PLANES_BYTES ? NWV * C16_WAVE_BYTES - C16_PLANES_BYTES : 0) + C16_NODE_BYTES + 16 * 8 * 4 + C16_CTR_INTS * 4 + C16_QA_BYTES +
         C16_PB_BYTES;
}

typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));

// ---- packed fp32 (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two results per issue slot) and the split straight from the packed
// hi halves.  Round 5: the tile loop of the edge phase is bound by VALU issue, and hipcc's own
// vectoriser packed only the normalise and the lo subtraction of a (sin, cos) pair; here two FREQUENCIES go through the exact
// division and the reduction to revolutions together (9 instead of 16 instructions per two pairs), and a lo half is ONE
// v_fma_mix{lo,hi}_f16 (f16(y - float(hi)) with the f16 operand read in place: the difference is exact in fp32, so the single
// conversion rounds like cvt(sub)) instead of v_cvt_f32_f16 + v_sub_f32 + v_cvt_f16_f32.  Same operations on the same values
// in the same order as the scalar form: the bits do not change (tools/mb/mb_feat.hip checks that on the GPU).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// (hi0 | hi1) = fp16 of two floats, round to nearest even: v_cvt_pk_f16_f32
__device__ __forceinline__ unsigned f16_hi_pk(float y0, float y1) {
  const half2v h = {(_Float16)y0, (_Float16)y1};
  return __builtin_bit_cast(unsigned, h);
}
// (lo0 | lo1) with lo = fp16(y - float(hi)), hi read from the packed dword -- in GROUPS of N dwords per asm block, closed by two wait states.
// Round 6: on gfx950 a VALU write of a register followed by a v_mfma that reads it needs TWO wait states (tools/mb/mb_mixhi_mfma.hip: no
// hardware interlock) -- hipcc inserts them for its own instructions and cannot for inline assembly.  With one asm statement per
// instruction the compiler was free to put a v_mfma right behind the last v_fma_mixhi of an operand; the shipped builds never did, an experiments build of
// this round did (its a_r sums were wrong in most rows, differently from run to run: a_r's A operand was read before its second half had landed).  Now:
// all the lo halves of a group first, then the hi halves (each register's two writes N - 1 instructions apart), then s_nop 1 -- whatever hipcc schedules
// behind the block is two wait states away.  Same instructions on the same values: the bits do not change; four compiler-inserted s_nop 0 per group are gone.
template <int N> struct MixAsm;
template <> struct MixAsm<2> {
  static __device__ __forceinline__ void run(const unsigned (&h)[2], const float (&y)[4], float m, unsigned (&r)[2]) {
    asm("v_fma_mixlo_f16 %0, %2, %8, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %3, %8, %6 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %2, %8, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %3, %8, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "s_nop 1"
        : "=&v"(r[0]), "=&v"(r[1]) : "v"(h[0]), "v"(h[1]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(m));
  }
};
template <> struct MixAsm<4> {
  static __device__ __forceinline__ void run(const unsigned (&h)[4], const float (&y)[8], float m, unsigned (&r)[4]) {
    asm("v_fma_mixlo_f16 %0, %4, %16, %8 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %5, %16, %10 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %2, %6, %16, %12 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %3, %7, %16, %14 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %4, %16, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %5, %16, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %2, %6, %16, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %3, %7, %16, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "s_nop 1"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
        : "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "v"(m));
  }
};
// per lane and lane-uniformly chosen half: m = -1 gives the lo halves fp16(y - float(hi)), m = -0 the hi halves fp16(y - 0) -- the
// (p hi | p lo) and (q hi | q lo) operands whose half depends on the lane take ONE instruction per value instead of both halves + a select
// 8 floats -> the lane's half (hi or lo by m) as an MFMA operand
__device__ __forceinline__ half8 f16_sel8(const float (&v)[8], float m) {
  unsigned h[4], r[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) h[j] = f16_hi_pk(v[2 * j], v[2 * j + 1]);
  MixAsm<4>::run(h, v, m, r);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(half8, u32x4{r[0], r[1], r[2], r[3]});
}
// 4 floats -> two dwords of the lane's half (the a_r MFMA's A operand)
__device__ __forceinline__ half4v f16_sel4(const float (&v)[4], float m) {
  unsigned h[2], r[2];
  h[0] = f16_hi_pk(v[0], v[1]);
  h[1] = f16_hi_pk(v[2], v[3]);
  MixAsm<2>::run(h, v, m, r);
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(half4v, u32x2_{r[0], r[1]});
}

// this lane's 8 consecutive features (4 pairs) of input block xs, normalised and split: A-fragment order of the
// score MFMAs (lane = edge + 16 kq holds features 8 kq .. 8 kq + 7 of the 32-feature block).  dv / rdv: the lane's four
// divisors and their reciprocals as two packed pairs.
__device__ __forceinline__ void feat8(float xs, float rstd, float nmr, const f32x2 (&dv)[2], const f32x2 (&rdv)[2], half8& hi, half8& lo) {
  constexpr float C1 = 0.15915494309189535f;
  constexpr float C2 = (float)(0.15915494309189535 - (double)0.15915494309189535f);
  const f32x2 x2 = {xs, xs}, c1 = {C1, C1}, c2 = {C2, C2}, rs = {rstd, rstd}, nm = {nmr, nmr};
  unsigned h[4], l[4];
  float yy[8];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    // fdiv16 (x / d exactly) and sincos_hw's reduction to revolutions, two frequencies at a time
    const f32x2 q0 = x2 * rdv[jj];
    const f32x2 rem = pk_fma(-q0, dv[jj], x2);
    const f32x2 q = pk_fma(rem, rdv[jj], q0);
    const f32x2 u = q * c1;
    const f32x2 n = {rintf(u.x), rintf(u.y)};
    const f32x2 f = pk_fma(q, c1, -n) + q * c2;
    const f32x2 sc0 = {__builtin_amdgcn_sinf(f.x), __builtin_amdgcn_cosf(f.x)}, sc1 = {__builtin_amdgcn_sinf(f.y), __builtin_amdgcn_cosf(f.y)};
    const f32x2 y0 = pk_fma(sc0, rs, nm), y1 = pk_fma(sc1, rs, nm);
    h[2 * jj] = f16_hi_pk(y0.x, y0.y);
    h[2 * jj + 1] = f16_hi_pk(y1.x, y1.y);
    yy[4 * jj] = y0.x; yy[4 * jj + 1] = y0.y; yy[4 * jj + 2] = y1.x; yy[4 * jj + 3] = y1.y;
  }
  MixAsm<4>::run(h, yy, -1.0f, l);   // lo = fp16(y - float(hi)) (f16_sel8's m = -1)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  hi = __builtin_bit_cast(half8, u32x4{h[0], h[1], h[2], h[3]});
  lo = __builtin_bit_cast(half8, u32x4{l[0], l[1], l[2], l[3]});
}
// the same outside fdiv16's checked range (a distance beyond 10 km): true division, libm sincos.  Rolled, and through LDS
// (this lane's 24 halfs of the feature tile row; `lo` selects which halves are written): the rare path must not cost the
// common one registers.
__device__ __forceinline__ void feat_slow_row(float a0, float a1, float a2, float rstd, float nmr, const float (&dv)[4], int kq,
                                              _Float16* __restrict__ frow, bool lo) {
#pragma unroll 1
  for (int i = 0; i < 12; ++i) {
    const int ks = i >> 2, j = i & 3;
    const float xs = ks == 0 ? a0 : (ks == 1 ? a1 : a2);
    const float d = j == 0 ? dv[0] : (j == 1 ? dv[1] : (j == 2 ? dv[2] : dv[3]));
    float sv, cv;
    sincosf(xs / d, &sv, &cv);
    const float ys = fmaf(sv, rstd, nmr), yc = fmaf(cv, rstd, nmr);
    const int c = 32 * ks + 8 * kq + 2 * j;
    frow[c] = lo ? f16_lo(ys) : f16_hi(ys);
    frow[c + 1] = lo ? f16_lo(yc) : f16_hi(yc);
  }
}

// ---- LDS-DMA of a tile's k rows (buffer_load_dwordx4 ... lds: 16 B per lane straight into LDS, no VGPRs; the LDS address is
// M0 + 16 * lane, wave-uniform base, so a bank-friendly layout is made by permuting the SOURCE pieces).  One round moves the
// hi (or lo) halves of 16 rows = 16 x 256 B: instruction i, lane l fills slot s = l & 15 of row rr = 4 i + (l >> 4) with piece
// s ^ rr of that row; the fragment reads (lane = row mi + 16 kq wants piece 4 ks + kq) then hit slot (4 ks + kq) ^ mi, and the 16
// lanes of every ds_read_b128 service group land on 16 distinct 16-byte slots mod 256 B (cdna guide, LDS section).
// (tools/mb/mb_glds.hip checks the recipe, also above 64 KB of LDS.)  Inline asm: hipcc neither counts these in vmcnt nor waits for them -- c16_wait_vm0() before the stage is read.
__device__ __forceinline__ void c16_blds16(unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff, unsigned lds_dst) {
  unsigned keep;   // MUBUF form: 32-bit per-lane byte offset; `soff` (SGPR) moves the memory address only -- an instruction offset would move the LDS side too
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
// One round = four DMAs behind ONE m0 swap (round 5; round 3 saved / set / restored m0 and spent an s_nop per DMA): the instruction
// offset 1024 i moves the LDS side to row group i -- and the memory side with it, which the caller takes back out of the per-lane
// offset (kaddr: + C16_DMA_BIAS - 1024 i against a descriptor based C16_DMA_BIAS bytes below the array).
constexpr unsigned C16_DMA_BIAS = 3072;
__device__ __forceinline__ void c16_blds16x4(const unsigned (&voff)[4], __amdgpu_buffer_rsrc_t rsrc, unsigned soff, unsigned lds_dst) {
#ifdef PS_C16_NO_DMA   // (diagnostic build: the same bytes through registers and ds_write_b128 -- a run-to-run difference under load was NOT the DMA's: ps_device.h kq_max3)
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const u32x4_ v = __builtin_bit_cast(u32x4_, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[i] + 1024u * i, soff, 0));
    *reinterpret_cast<__attribute__((address_space(3))) u32x4_*>((__attribute__((address_space(3))) unsigned char*)(size_t)(lds_dst + 1024u * i + 16u * (threadIdx.x & 63))) = v;
  }
  return;
#endif
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %1, %5, %6 offen lds\n\t"
               "buffer_load_dwordx4 %2, %5, %6 offen offset:1024 lds\n\t"
               "buffer_load_dwordx4 %3, %5, %6 offen offset:2048 lds\n\t"
               "buffer_load_dwordx4 %4, %5, %6 offen offset:3072 lds\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
// raw buffer over a device array (stride 0, no format conversion): 32-bit offsets instead of 64-bit pointer arithmetic per lane
// (the pointer goes through readfirstlane: a descriptor that hipcc cannot prove wave-uniform costs a waterfall loop per load)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t c16_rsrc(const void* p) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 c16_bld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  return make_float4(v.x, v.y, v.z, v.w);
}
#ifdef PS_C16_ABL_NOWAIT   // (timing experiment: how much DMA latency the two waits of a tile expose; wrong results)
__device__ __forceinline__ void c16_wait_vm0() { asm volatile("" ::: "memory"); }
#else
__device__ __forceinline__ void c16_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
__device__ __forceinline__ void c16_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Where a row's sums wait for the POST half (all in LDS; nothing of the exchange touches global memory):
//   a_r [8][96]  the row's QA slot (ONEW: slot = row, over its q~; W waves per row: slot = row * W + part, q~ in slot 8 + row)
//   l, m  [8]    columns 96 / 97 of the slot's head rows (C16_QH = 100: four spare floats per head)
//   a_v [128]    ONEW: the row's own AG row (its q, read once at the start of the row); W > 1: AG row 8 + row * W + part
__device__ __forceinline__ int c16_av_row(int lr, int part, int W) { return W == 1 ? lr : 8 + lr * W + part; }

// The edge phase of one layer for the rows of a workgroup (called once per layer).  Out of line on purpose: inlined into
// the layer loop its register pressure makes the allocator spill the weight-fragment ring of the node GEMMs; as a
// function it gets an allocation of its own.  smem = the workgroup's dynamic LDS (wave areas from its start), AG / CQ /
// ctr = the node phase's q rows, <q, kb> and the row counter.
//
// ONE summation order for every tiling from 4 rows per workgroup up: a row's 16-edge tiles are ALWAYS split by parity into
// two partial softmax sums that are merged as (even, odd).  With W = 2 (4 rows per workgroup: latency mode) two waves take
// one parity each and the POST half merges their partials.  ONEW (W = 1: 8 .. 16 rows per workgroup, throughput mode):
// the one wave of a row walks the even tiles, parks their sums (a_r in the row's LDS slot, a_v / m / l in a wave-private
// stash), walks the odd tiles in the same software pipeline (the first odd tile is prefetched under the last even one),
// and merges with the POST half's own operations -- so both modes return bit-identical results.
//
// Round 3: the k rows of a tile arrive by LDS-DMA (no registers, no ds_write pass; hi halves one tile ahead, lo halves under
// the tile's Fourier rows); the aggregation MFMAs take their probabilities from the registers the softmax left them in (the
// C layout of the score MFMA IS the A layout of the 16x16x16 aggregation MFMA: row 4 kq + j of column mi); the transposed
// feature reads of a pass are issued together; the row's sums stay in LDS (see above).  The arithmetic -- every operation
// and its order -- is the round-2 kernel's: results are bit-identical to it.
// (TAG: a second, independent copy of the function for the standalone edge kernel k_edge16 -- as a second CALLER of one copy it cost
// the policy chain 9 %: the inter-procedural register allocation is per function)
// (DIRECT: the standalone many-waves form, k_edge_rows below -- a wave takes row row0 + wave, reads q / q~ / <q, kb> straight from the
// PRE half's EdgeIO rows and writes its sums there; the even tiles' a_r park in a wave-private LDS slot behind the wave area.  Same
// operations in the same order: the results are those of the workgroup form bit for bit.)
constexpr size_t C16_PARK_BYTES = (size_t)8 * C16_QH * 4;
template <int NWV, bool ONEW, bool DIRECT>
__device__ __forceinline__ void c16_edge_body(const ChainStep* __restrict__ stp, unsigned char* c16_smem, float* AG,
                                             const float* CQ, int* ctr, float* QA, const float* __restrict__ div32, int row0, int nrows, int W,
                                             const EdgeIO& io) {
  static_assert(!DIRECT || ONEW, "the standalone form is one wave per row");
  constexpr size_t WSTRIDE = C16_WAVE_BYTES + (DIRECT ? C16_PARK_BYTES : 0);
  const ChainStep& st = *stp;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mi = lane & 15, kq = lane >> 4;
  // this lane's four frequencies (pairs 4 kq .. 4 kq + 3 of every 32-feature block) and their reciprocals
  float dv[4], rdv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    dv[j] = ldg1(div32 + 2 * (4 * kq + j));
    rdv[j] = 1.0f / dv[j];
  }
  const f32x2 dvp[2] = {{dv[0], dv[1]}, {dv[2], dv[3]}}, rdvp[2] = {{rdv[0], rdv[1]}, {rdv[2], rdv[3]}};
  unsigned char* wbase = c16_smem + (size_t)wave * WSTRIDE;
  const half8* stg = reinterpret_cast<const half8*>(wbase);                    // k staging: [16 rows][16 slots of 16 B], hi or lo halves
  float* Pt = reinterpret_cast<float*>(wbase + 4096);                         // [16 edges][8 heads] probabilities (for a_v)
  _Float16* Ft = reinterpret_cast<_Float16*>(wbase + 4096 + 512);             // [16 edges][C16_FS] feature tile (hi, then lo)
  int* Ss = reinterpret_cast<int*>(wbase + 4096 + 512 + 16 * C16_FS * 2);     // [2][16] source rows of this tile and the next
  float* stash = reinterpret_cast<float*>(wbase + C16_STASH_OFF);
  const unsigned stg_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)wbase);   // LDS byte address of the staging area
  const __amdgpu_buffer_rsrc_t rs_geo = c16_rsrc(st.geo), rs_k = c16_rsrc(reinterpret_cast<const unsigned char*>(st.khl) - C16_DMA_BIAS), rs_v = c16_rsrc(st.kv);
  // fragment reads of the staging area: row mi, piece 4 ks + kq at slot (4 ks + kq) ^ mi
  const half8* str = stg + mi * 16 + (kq ^ (mi & 3));
  const int sra = mi >> 2;
  // DMA sources: row 4 i + (lane >> 4), piece (lane & 15) ^ row
  const int dr = lane >> 4, ds_ = lane & 15;
  const bool loA = mi >= 8;
  const float selm = loA ? -1.f : -0.f;   // f16_sel_pk: lanes of the lo columns take fp16(y - float(hi)), the others fp16(y - 0)
  const int hv = (lane & 31) >> 2, eh = lane >> 5;
  for (int it = 0;; ++it) {
    int lr, part;
#ifdef PS_EDGE_ONEWAVE   // (tools only, DESIGN 7.4 item 6: wave 0 walks the whole row queue alone -- does the anomaly need a second wave?)
    if (ONEW && !DIRECT && wave != 0) break;
#endif
    if (DIRECT) {  // static: one row per wave
      if (it > 0) break;
      lr = wave;
      part = 0;
    } else if (W > 1) {   // static: wave -> (row wave / W, part wave % W)
      if (it > 0) break;
      lr = wave / W;
      part = wave - lr * W;
    } else {       // dynamic: the next row of the workgroup
      lr = 0;
      if (lane == 0) lr = atomicAdd(ctr, 1);
      lr = __builtin_amdgcn_readfirstlane(lr);
      if (lr >= nrows) break;
      lr = __builtin_amdgcn_readfirstlane(ctr[1 + lr]);   // (wave-uniform by construction; said so, the tile loop's control stays on the scalar unit)
      part = 0;
    }
    if (lr >= nrows) break;
    const int r = row0 + lr;
    float* park = DIRECT ? reinterpret_cast<float*>(wbase + C16_WAVE_BYTES) : QA + lr * C16_QSL;   // the even tiles' a_r (ONEW)
    // (wave-uniform: the tile loop's control stays on the scalar unit; the workgroup form reads what its PRE half left in LDS)
    const int e_beg = __builtin_amdgcn_readfirstlane(DIRECT ? ldgi(st.eoff + r) : ctr[36 + lr]);
    const int deg = DIRECT ? __builtin_amdgcn_readfirstlane(ldgi(st.eoff + r + 1)) - e_beg : __builtin_amdgcn_readfirstlane(ctr[17 + lr]);
    const int tstep = ONEW ? 32 : 16 * W;
    const bool two = ONEW && deg > 16;   // the row has odd tiles: two partial sums
    // Tiles of 16 edges (one score block).  A tile's geometry and source rows are requested one tile ahead (registers).
    int t0 = 16 * part;
    float4 ng;
    float nn = 0.f;
    int nsrc = 0;
    // lane mi asks for the record of edge tt + mi, or of the row's last edge past the end: the 16 source rows that lanes 0-15
    // publish are all valid (a short tile repeats its last edge; the scores of the repeats are masked)
    auto prefetch = [&](int tt) {
      const unsigned off = (unsigned)(e_beg + min(tt + mi, deg - 1)) * 32u;
      ng = c16_bld4(rs_geo, off);
      const float2 g2 = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs_geo, off + 16, 0, 0));
      nn = g2.x;
      nsrc = __float_as_int(g2.y);
    };
    if (t0 < deg) prefetch(t0);
    // B operands of the score MFMAs: lane -> column n = mi (head mi & 7, hi | lo half), k-block kq
    half8 bq[3], bk[4];
    float cqm;
    {
      const int hB = mi & 7;
      const float* qtp = DIRECT ? io.qt + (size_t)r * 1024 + hB * 128 + 8 * kq : QA + ((W == 1 ? 0 : 8) + lr) * C16_QSL + hB * C16_QH + 8 * kq;
      const float* qp = DIRECT ? io.q + (size_t)r * 128 + 8 * kq : AG + lr * ND_XS + 8 * kq;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const float4 v0 = *reinterpret_cast<const float4*>(qtp + 32 * ks), v1 = *reinterpret_cast<const float4*>(qtp + 32 * ks + 4);
        const float qv_[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        bq[ks] = f16_sel8(qv_, selm);
      }
      // q masked to head hB: the 8 columns 32 ks + 8 kq lie inside head 2 ks + (kq >> 1), so of a lane's four k-blocks only ks = hB >> 1
      // can be non-zero, and only in the lanes with (kq >> 1) == (hB & 1): one block converted, three zeros (round 4: all four were)
      {
        const int ksel = hB >> 1;
        const bool mine = (kq >> 1) == (hB & 1);
        const float4 w0 = *reinterpret_cast<const float4*>(qp + 32 * ksel), w1 = *reinterpret_cast<const float4*>(qp + 32 * ksel + 4);
        const float kv_[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        const half8 bkv = f16_sel8(kv_, selm);
        const half8 zero = {};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bk[ks] = (mine && ks == ksel) ? bkv : zero;
      }
      cqm = DIRECT ? io.cq[(size_t)r * 8 + hB] : CQ[lr * 8 + hB];
    }
    float m_run = -INFINITY, l_run = 0.f;   // of head mi & 7 (the lanes mi and mi + 8, all kq, carry copies)
    floatx4 ar[6];
#pragma unroll
    for (int cb = 0; cb < 6; ++cb) ar[cb] = floatx4{0.f, 0.f, 0.f, 0.f};
    float4 av = make_float4(0.f, 0.f, 0.f, 0.f);   // a_v partial: columns 4 (lane & 31) ..+3, edges of parity lane >> 5
    const unsigned vcol = 512u + 16u * (lane & 31);   // the v half of a k | v row, this lane's four columns
    // the DMA source offsets of a tile's k rows (hi halves; the lo halves sit 256 bytes further: the DMA's scalar offset)
    unsigned kp[4];
    int sb = 0;   // Ss buffer holding this tile's source rows
    auto kaddr = [&](int buf) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rr = 4 * i + dr;
#ifdef PS_C16_ABL_ONEROW   // (timing experiment: every k row of a tile from ONE source row -- L1 hits; wrong results)
        kp[i] = (unsigned)Ss[16 * buf] * 512u + 16u * (ds_ ^ rr) + (C16_DMA_BIAS - 1024u * i);
#else
        kp[i] = (unsigned)Ss[16 * buf + rr] * 512u + 16u * (ds_ ^ rr) + (C16_DMA_BIAS - 1024u * i);
#endif
      }
    };
    if (t0 < deg) {
      if (lane < 16) Ss[lane] = nsrc;   // (prefetch(t0)'s loads are hipcc's own: it waits for them here)
      kaddr(0);
      c16_wait_lgkm0();                 // the previous row's last reads of the staging area are done
      c16_blds16x4(kp, rs_k, 0u, stg_lds);
    }
#ifdef PS_C16_ABL_NOEDGE
    t0 = deg;
#endif
#pragma unroll 1
    for (int tn = 0; t0 < deg; t0 = tn) {
      const int n = min(16, deg - t0);
      // the tile after this one: same parity, or (ONEW) the first odd tile after the last even one
      tn = t0 + tstep;
      const bool turn = two && tn >= deg && (t0 & 16) == 0;
      if (turn) tn = 16;
      tn = __builtin_amdgcn_readfirstlane(tn);   // (uniform already; hipcc kept the tile counter in a VGPR and its compares on the VALU)
      // this tile's records (requested a tile ago)
      const float4 g0 = ng;
      const float nmr = nn;
      const int* Sc = Ss + 16 * sb;
      float sreg[4];
      half8 fl[3];
      {
        // ---- k hi halves (in flight since the previous tile): fragments, then the lo halves take the staging area
        half8 ak[4];
        c16_wait_vm0();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ak[ks] = str[4 * (ks ^ sra)];
        c16_wait_lgkm0();
#ifndef PS_C16_ABL_NOKLO
        c16_blds16x4(kp, rs_k, 256u, stg_lds);
#endif
        prefetch(tn);   // the next tile's records leave now (past the row's end: its last edge again, unused)
        floatx4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ak[ks], bk[ks], acc2, 0, 0, 0);
        // ---- this tile's Fourier rows, in registers (A fragments of the score MFMAs) and as a row-major LDS tile
        half8 fh[3];
#ifdef PS_C16_ABL_NOFEAT   // (timing experiment: what the Fourier rows cost; wrong results)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) { fh[ks] = __builtin_bit_cast(half8, ng); fl[ks] = fh[ks]; }
#else
        feat8(g0.x, g0.w, nmr, dvp, rdvp, fh[0], fl[0]);
        feat8(g0.y, g0.w, nmr, dvp, rdvp, fh[1], fl[1]);
        feat8(g0.z, g0.w, nmr, dvp, rdvp, fh[2], fl[2]);
#endif
        if (__builtin_expect(__any(!(fdiv16_ok(g0.x) && fdiv16_ok(g0.y) && fdiv16_ok(g0.z))), 0)) {
          // (a distance beyond 10 km, outside fdiv16's checked range: true division + libm, rolled, through the feature
          // tile -- twice, the hi halfs last; out of the common path's basic block)
          _Float16* frow = Ft + mi * C16_FS;
          feat_slow_row(g0.x, g0.y, g0.z, g0.w, nmr, dv, kq, frow, true);
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) fl[ks] = *reinterpret_cast<const half8*>(frow + 32 * ks + 8 * kq);
          feat_slow_row(g0.x, g0.y, g0.z, g0.w, nmr, dv, kq, frow, false);
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) fh[ks] = *reinterpret_cast<const half8*>(frow + 32 * ks + 8 * kq);
        }
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) *reinterpret_cast<half8*>(Ft + mi * C16_FS + 32 * ks + 8 * kq) = fh[ks];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[ks], bq[ks], acc, 0, 0, 0);
        // ---- k lo halves
#ifndef PS_C16_ABL_NOKLO   // (timing experiment: no wait for the lo halves, the stale hi fragments twice; wrong results)
        c16_wait_vm0();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ak[ks] = str[4 * (ks ^ sra)];
        c16_wait_lgkm0();
#endif
        if (tn < deg) {   // the next tile's source rows arrived with the prefetch: publish them, start the DMA of its k hi halves
          if (lane < 16) Ss[16 * (sb ^ 1) + lane] = nsrc;
          kaddr(sb ^ 1);
          c16_blds16x4(kp, rs_k, 0u, stg_lds);
        }
#ifndef PS_C16_ABL_NOSCORE2   // (timing experiment: the second half of the score MFMAs dropped; wrong results)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[ks], bq[ks], acc, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ak[ks], bk[ks], acc2, 0, 0, 0);
#endif
        acc += acc2;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const float v = acc[r4] + dpp_xor8(acc[r4]);   // columns h and h + 8 (q hi | q lo)
          sreg[r4] = (v + cqm) * (0.25f * 1.44269504088896341f);   // in units of log2: exp(x) = exp2(x log2 e)
        }
        if (n < 16) {   // (wave-uniform: only a parity class's last tile is short)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) sreg[r4] = (4 * kq + r4 < n) ? sreg[r4] : -INFINITY;
        }
      }
      // v rows of the tile leave now and fly under the softmax and the a_r MFMAs: gathered by source, two rows per load instruction
      float4 vv[8];
#ifdef PS_C16_ABL_NOAV
#pragma unroll
      for (int j = 0; j < 8; ++j) vv[j] = make_float4(sreg[0], sreg[1], sreg[2], sreg[3]);
#else
#pragma unroll
#ifdef PS_C16_ABL_ONEROW
      for (int j = 0; j < 8; ++j) vv[j] = c16_bld4(rs_v, (unsigned)Sc[0] * 1024u + vcol);
#else
      for (int j = 0; j < 8; ++j) vv[j] = c16_bld4(rs_v, (unsigned)Sc[2 * j + eh] * 1024u + vcol);
#endif
#endif
      // ---- online softmax over the tile (torch_geometric.utils.softmax: max-shift, exp, / (sum + 1e-16))
      float tmax = fmaxf(fmaxf(sreg[0], sreg[1]), fmaxf(sreg[2], sreg[3]));
#ifdef PS_C16_ABL_NOSOFT   // (timing experiment: no cross-lane maximum; wrong results)
      const float m_cand = fmaxf(tmax, m_run);
#else
      const float m_cand = kq_max3(tmax, m_run);   // max(m_run, the tile's maximum over the four kq lanes); finite: the tile has at least one edge
#endif
      const bool fresh = m_run == -INFINITY;   // this head's first tile of the parity class: its sums are still zero
      // Round 6, measured and NOT adopted (-DPS_C16_LAZY builds it): moving the reference point of a class's sums lazily -- while a tile's
      // maximum stays within C16_TAU (in log2 units) above the reference, keep it, let the tile's probabilities reach 2^C16_TAU and rescale
      // nothing.  Any head's maximum moving (8 heads: 99 % of a class's second tiles, 83 % of its fifth) costs the wave 8 cross-lane
      // broadcasts and 28 multiplies, 3.7 % of the policy launch by ablation; the lazy rule took the launch from 0.665 to 0.652 ms -- and, being
      // another rounding of the same sums, re-rolled configs[3] seed 0's near-cut agent #25 (1.4e-6 rad from a +-pi cut) onto the other
      // side.  Two per cent of one kernel do not buy a row of the parity table (profiles/r06_f_policy_launch_ablation.txt).
#ifdef PS_C16_LAZY   // (experiment, NOT the product's rule: see the comment above)
      const float m_new = (fresh || m_cand > m_run + C16_TAU) ? m_cand : m_run;
#else
      const float m_new = m_cand;
#endif
      // (v_exp_f32 itself: libm's exp2f wraps it in a range check + ldexp for results below 2^-126, which max-shifted
      // probabilities and scales do not need -- such a term adds nothing to sums that hold a 1)
      const float scale = fresh ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
      float psum = 0.f;
      float pr[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float p = __builtin_amdgcn_exp2f(sreg[r4] - m_new);   // 2^-inf = 0 for the slots past the edge list
        pr[r4] = p;
        psum += p;
      }
      if (mi < 8) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) Pt[(4 * kq + r4) * 8 + mi] = pr[r4];
      }
#ifndef PS_C16_ABL_NOSOFT
      psum = kq_sum(psum);
#endif
      l_run = l_run * scale + psum;
      m_run = m_new;
      // what is already accumulated shrinks by the head's scale: accumulator row 4 (lane >> 4) + r belongs to head
      // (4 (lane >> 4) + r) & 7; the a_v columns of this lane to head hv.  Skipped while no head's maximum moves
      // (most tiles after a row's first few).
      // (nor on a class's first tile: scaling zero sums by zero is the identity)
#ifdef PS_C16_ABL_NORESCALE   // (timing experiment: the accumulated sums are never rescaled; wrong results)
      if (false) {
#else
      if (__any(scale != 1.f && !fresh)) {
#endif
#ifdef PS_C16_RESCALE_R5   // (the round-5 form, for the A/B: eight ds_bpermute round trips, 7 selects, 28 multiplies -- same bits)
        float scl[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) scl[h] = __shfl(scale, h);
        const bool up = kq & 1;
        const float s0 = up ? scl[4] : scl[0], s1 = up ? scl[5] : scl[1], s2 = up ? scl[6] : scl[2], s3 = up ? scl[7] : scl[3];
#pragma unroll
        for (int cb = 0; cb < 6; ++cb) { ar[cb][0] *= s0; ar[cb][1] *= s1; ar[cb][2] *= s2; ar[cb][3] *= s3; }
        float sh = scl[0];
#pragma unroll
        for (int h = 1; h < 8; ++h) sh = (hv == h) ? scl[h] : sh;
        av.x *= sh; av.y *= sh; av.z *= sh; av.w *= sh;
#else
        // Round 6: head h's scale sits in lane h (mi = h, kq = 0) -- v_readlane_b32 into scalar registers instead of eight cross-lane
        // round trips through the LDS crossbar; the a_v lanes take theirs with ONE of those (it flies under the multiplies); the
        // accumulators two rows per v_pk_mul_f32 (natural lane order: no op_sel).  Same values, same products: the bits do not change.
        float scl[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) scl[h] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(scale), h));
        const float sh = __shfl(scale, hv);
        const bool up = kq & 1;
        const f32x2 s01 = {up ? scl[4] : scl[0], up ? scl[5] : scl[1]}, s23 = {up ? scl[6] : scl[2], up ? scl[7] : scl[3]};
#pragma unroll
        for (int cb = 0; cb < 6; ++cb) {
          const f32x2 lo = f32x2{ar[cb][0], ar[cb][1]} * s01, hi = f32x2{ar[cb][2], ar[cb][3]} * s23;
          ar[cb] = floatx4{lo.x, lo.y, hi.x, hi.y};
        }
        av.x *= sh; av.y *= sh; av.z *= sh; av.w *= sh;
#endif
      }
      // ---- a_r[h][c] += sum_e p_e,h r~_e[c] on the matrix cores (16x16x16): A = (p hi | p lo) x head, straight from this
      //      lane's probabilities (row 4 kq + j of column mi = edge 4 kq + j of head mi & 7), B = the feature tile read back
      //      transposed (4 consecutive EDGES of one feature per lane: one ds_read_b64_tr_b16), hi pass then lo pass
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      const half4v ap = f16_sel4(pr, selm);
      const _Float16* tp = Ft + (kq * 4 + (mi >> 2)) * C16_FS + (lane & 3) * 4;
#ifdef PS_C16_ABL_NOAR
      ar[0][0] += (float)ap[0] + (float)fl[0][0] + (float)fl[1][0] + (float)fl[2][0];
#else
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) *reinterpret_cast<half8*>(Ft + mi * C16_FS + 32 * ks + 8 * kq) = fl[ks];
        }
        fp16x4 tv[6];
#pragma unroll
        for (int cb = 0; cb < 6; ++cb) tv[cb] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(tp + cb * 16));
#pragma unroll
        for (int cb = 0; cb < 6; ++cb) {
          half4v bfr;
#pragma unroll
          for (int j = 0; j < 4; ++j) bfr[j] = (_Float16)tv[cb][j];
          ar[cb] = __builtin_amdgcn_mfma_f32_16x16x16f16(ap, bfr, ar[cb], 0, 0, 0);
        }
      }
#endif
      // ---- a_v[hd] += sum_e p_e,h v_src[hd] on the VALU
#ifdef PS_C16_ABL_NOAV
      av.x += vv[0].x;
#else
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float ph = Pt[(2 * j + eh) * 8 + hv];   // 0 past the edge list
        av.x = fmaf(ph, vv[j].x, av.x);
        av.y = fmaf(ph, vv[j].y, av.y);
        av.z = fmaf(ph, vv[j].z, av.z);
        av.w = fmaf(ph, vv[j].w, av.w);
      }
#endif
      sb ^= 1;
      if (ONEW && turn) {   // the even tiles are done: park their sums, start the odd tiles' from zero
        av.x = xor_add<32>(av.x); av.y = xor_add<32>(av.y); av.z = xor_add<32>(av.z); av.w = xor_add<32>(av.w);
        if (lane < 32) *reinterpret_cast<float4*>(stash + 4 * lane) = av;
        if (lane < 8) { stash[128 + 2 * lane] = m_run; stash[128 + 2 * lane + 1] = l_run; }
#pragma unroll
        for (int cb = 0; cb < 6; cb += 2) {
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const float v = swap_add32(ar[cb][r4], ar[cb + 1][r4]);
            park[(4 * ((lane >> 4) & 1) + r4) * C16_QH + (cb + (lane >> 5)) * 16 + (lane & 15)] = v;
          }
        }
        m_run = -INFINITY;
        l_run = 0.f;
#pragma unroll
        for (int cb = 0; cb < 6; ++cb) ar[cb] = floatx4{0.f, 0.f, 0.f, 0.f};
        av = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    // ---- the row's (partial) sums wait in LDS for the POST half
    const int aslot = W == 1 ? lr : lr * W + part;
    float* avrow = DIRECT ? io.av + (size_t)r * 128 : AG + c16_av_row(lr, part, W) * ND_XS;
    float* oar = DIRECT ? io.ar + (size_t)r * 1024 : QA + aslot * C16_QSL;   // [8 heads][oh]: the row's a_r
    const int oh = DIRECT ? 128 : C16_QH;
    float* ol = DIRECT ? io.l + (size_t)r * 8 : nullptr;
    av.x = xor_add<32>(av.x); av.y = xor_add<32>(av.y); av.z = xor_add<32>(av.z); av.w = xor_add<32>(av.w);
    if (ONEW && two) {   // merge (even, odd) with the operations of the POST half's merge of two waves' partials
      const float m0 = stash[128 + 2 * (lane & 7)], l0 = stash[128 + 2 * (lane & 7) + 1];   // head lane & 7
      const float mh = __shfl(m_run, lane & 7), lh = __shfl(l_run, lane & 7);              // (lane h holds head h)
      float mm = -INFINITY;
      mm = fmaxf(mm, m0);
      mm = fmaxf(mm, mh);
      const float sc0 = (m0 == -INFINITY) ? 0.f : exp2f(m0 - mm);
      const float sc1 = (mh == -INFINITY) ? 0.f : exp2f(mh - mm);
      float lm = 0.f;
      lm = fmaf(l0, sc0, lm);
      lm = fmaf(lh, sc1, lm);
      if (lane < 8) { if (DIRECT) ol[lane] = lm; else QA[aslot * C16_QSL + lane * C16_QH + 96] = lm; }
      {
        const float s0 = __shfl(sc0, hv), s1 = __shfl(sc1, hv);   // a_v columns 4 (lane & 31) ..+3 belong to head hv
        if (lane < 32) {
          const float4 a0 = *reinterpret_cast<const float4*>(stash + 4 * lane);
          float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
          o.x = fmaf(a0.x, s0, o.x); o.y = fmaf(a0.y, s0, o.y); o.z = fmaf(a0.z, s0, o.z); o.w = fmaf(a0.w, s0, o.w);
          o.x = fmaf(av.x, s1, o.x); o.y = fmaf(av.y, s1, o.y); o.z = fmaf(av.z, s1, o.z); o.w = fmaf(av.w, s1, o.w);
          *reinterpret_cast<float4*>(avrow + 4 * lane) = o;
        }
      }
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int hh = 4 * ((lane >> 4) & 1) + r4;
        const float s0 = __shfl(sc0, hh), s1 = __shfl(sc1, hh);
#pragma unroll
        for (int cb = 0; cb < 6; cb += 2) {
          const float v1 = swap_add32(ar[cb][r4], ar[cb + 1][r4]);
          const int col = (cb + (lane >> 5)) * 16 + (lane & 15);
          float o = 0.f;
          o = fmaf(park[hh * C16_QH + col], s0, o);
          o = fmaf(v1, s1, o);
          oar[hh * oh + col] = o;
        }
      }
    } else {
      if (lane < 8) {   // lane h (mi = h, kq = 0) holds head h
        if (DIRECT) ol[lane] = l_run; else QA[aslot * C16_QSL + lane * C16_QH + 96] = l_run;
        if (W > 1) QA[aslot * C16_QSL + lane * C16_QH + 97] = m_run;
      }
      if (lane < 32) *reinterpret_cast<float4*>(avrow + 4 * lane) = av;
      // rows h (p hi) and h + 8 (p lo) sit 32 lanes apart: one half-wave swap folds two column blocks at a time
#pragma unroll
      for (int cb = 0; cb < 6; cb += 2) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const float v = swap_add32(ar[cb][r4], ar[cb + 1][r4]);   // lanes < 32: block cb, lanes >= 32: block cb + 1
          oar[(4 * ((lane >> 4) & 1) + r4) * oh + (cb + (lane >> 5)) * 16 + (lane & 15)] = v;
        }
      }
    }
  }
}
template <int NWV, bool ONEW, int TAG = 0>
__device__ __noinline__ void c16_edge_phase(const ChainStep* __restrict__ stp, unsigned char* c16_smem, float* AG,
                                            const float* CQ, int* ctr, float* QA, const float* __restrict__ div32, int row0, int nrows, int W) {
  // (plain pointers into the caller's LDS, unlike the node phase's opaque LDS address: hipcc propagates the kernel's `extern __shared__` symbol into
  // this function and pays ONE look-up of the dynamic-LDS base per call for it.  Handing the wave areas over as an opaque address -- round 6 tried --
  // returned wrong sums for every row in the one-wave-per-row form, for a reason not found: profiles/r06_q_node_phase.txt)
  const EdgeIO none{};
#ifdef PS_EDGE_OPAQUE   // (tools only: the anomaly of DESIGN 7.4 item 6 -- the wave areas from an opaque LDS address, in the policy launch's copy of the function only)
  if (TAG == 2) c16_smem = lds_ptr<unsigned char>(lds_addr(c16_smem));
#endif
  c16_edge_body<NWV, ONEW, false>(stp, c16_smem, AG, CQ, ctr, QA, div32, row0, nrows, W, none);
  // nothing of this phase may still be in flight when the node phase reuses the wave areas as its operand planes (the workgroup barrier does
  // not wait for vmcnt; until round 6 the node phase happened to begin with a pointer load behind s_waitcnt vmcnt(0))
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// ---- k_attn_chain<1, 4, 3, ., ., GEO>'s edge phase (ps_attn.h; round 5): c16_edge_body's arithmetic for FOUR waves on ONE row (wave w takes
// the 16-edge tiles w, w + 4, ...; every wave its own online softmax, k_attn_chain merges the four partials like c16_node_phase's POST half
// merges W partial sums), rescheduled for LATENCY: a wave is alone on its SIMD there (one workgroup per CU, 128 workgroups), so nothing hides a
// dependent round trip but the wave's own independent work.  c16_edge_body keeps its loads short-lived (248 registers, two waves per SIMD hide
// each other): the lo halves of a tile's k rows are requested when its hi halves have landed, the v rows after its scores.  Here a tile's
// k rows (hi AND lo halves, staging areas of their own), v rows (registers) and the NEXT tile's geometry records are all requested ONE TILE AHEAD,
// the first tile's during the layer's q | s | g stages (c16_lat_pre, called at the top of the layer; the records of that tile are requested
// at the end of the PREVIOUS layer's edge phase: c16_lat_request), so that a tile waits once, for loads that left a tile ago.
static_assert(G1_QSL == C16_QSL && G1_QH == C16_QH && G1_AGS == ND_XS && G1_WAVE_FLOATS * 4 == 8192 + 512 + 16 * C16_FS * 2 + 128, "ps_attn.h's GEO layout");
struct C16LatCtx {   // a wave's addressing (cheap to rebuild: made where it is used)
  __amdgpu_buffer_rsrc_t rs_geo, rs_k, rs_v;
  unsigned stg_lds, vcol;
  int* Ss;
  int mi, kq, dr, ds_, eh;
};
__device__ __forceinline__ C16LatCtx c16_lat_ctx(const ChainStep& st, float* g1) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  unsigned char* wbase = reinterpret_cast<unsigned char*>(g1) + (size_t)wave * (G1_WAVE_FLOATS * 4);
  C16LatCtx c;
  c.rs_geo = c16_rsrc(st.geo);
  c.rs_k = c16_rsrc(reinterpret_cast<const unsigned char*>(st.khl) - C16_DMA_BIAS);
  c.rs_v = c16_rsrc(st.kv);
  c.stg_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)wbase);
  c.vcol = 512u + 16u * (lane & 31);
  c.Ss = reinterpret_cast<int*>(wbase + 8192 + 512 + 16 * C16_FS * 2);
  c.mi = lane & 15; c.kq = lane >> 4; c.dr = lane >> 4; c.ds_ = lane & 15; c.eh = lane >> 5;
  return c;
}
__device__ __forceinline__ GeoRec c16_lat_rec(const C16LatCtx& c, int e_beg, int deg, int tt) {   // lane mi: edge tt + mi (past the end: the row's last edge)
  GeoRec r;
  const unsigned off = (unsigned)(e_beg + max(min(tt + c.mi, deg - 1), 0)) * 32u;
  r.g = c16_bld4(c.rs_geo, off);
  const float2 g2 = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(c.rs_geo, off + 16, 0, 0));
  r.nn = g2.x;
  r.src = __float_as_int(g2.y);
  return r;
}
// a tile's source rows -> LDS, its k rows (hi | lo halves) by LDS-DMA, its v rows into registers
__device__ __forceinline__ void c16_lat_issue(const C16LatCtx& c, int buf, int nsrc, float4 (&vv)[8]) {
  if ((threadIdx.x & 63) < 16) c.Ss[16 * buf + (threadIdx.x & 63)] = nsrc;
  unsigned kp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rr = 4 * i + c.dr;
    kp[i] = (unsigned)c.Ss[16 * buf + rr] * 512u + 16u * (c.ds_ ^ rr) + (C16_DMA_BIAS - 1024u * i);
  }
  c16_wait_lgkm0();   // (the staging areas' last fragment reads are done)
  c16_blds16x4(kp, c.rs_k, 0u, c.stg_lds);
  c16_blds16x4(kp, c.rs_k, 256u, c.stg_lds + 4096u);
#pragma unroll
  for (int j = 0; j < 8; ++j) vv[j] = c16_bld4(c.rs_v, (unsigned)c.Ss[16 * buf + 2 * j + c.eh] * 1024u + c.vcol);
}
template <int TAG>
__device__ __forceinline__ void c16_lat_request(const ChainStep* __restrict__ stp, float* g1, int e_beg, int deg, C16LatState& S) {
  const C16LatCtx c = c16_lat_ctx(*stp, g1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (16 * wave < deg) S.r0 = c16_lat_rec(c, e_beg, deg, 16 * wave);
}
template <int TAG>
__device__ __forceinline__ void c16_lat_pre(const ChainStep* __restrict__ stp, float* g1, int e_beg, int deg, C16LatState& S) {
  const C16LatCtx c = c16_lat_ctx(*stp, g1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (16 * wave < deg) {
    c16_lat_issue(c, 0, S.r0.src, S.vv);
    S.r1 = c16_lat_rec(c, e_beg, deg, 16 * wave + 64);
  }
}
template <int TAG>
__device__ __forceinline__ void c16_lat_main(const ChainStep* __restrict__ stp, float* g1, const float* cq, const float* __restrict__ div32, int e_beg, int deg,
                                             C16LatState& S) {
  const ChainStep& st = *stp;
  const C16LatCtx c = c16_lat_ctx(st, g1);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int mi = c.mi, kq = c.kq;
  // (the divisors: loaded and inverted once per launch by the caller -- per layer their load was an exposed round trip in front of the first tile)
  const float (&dv)[4] = S.dv, (&rdv)[4] = S.rdv;
  const f32x2 dvp[2] = {{dv[0], dv[1]}, {dv[2], dv[3]}}, rdvp[2] = {{rdv[0], rdv[1]}, {rdv[2], rdv[3]}};
  unsigned char* wbase = reinterpret_cast<unsigned char*>(g1) + (size_t)wave * (G1_WAVE_FLOATS * 4);
  const half8* stg = reinterpret_cast<const half8*>(wbase);                    // k staging: hi halves [16 rows][16 slots of 16 B], lo halves 4 KB further
  float* Pt = reinterpret_cast<float*>(wbase + 8192);                          // [16 edges][8 heads] probabilities (for a_v)
  _Float16* Ft = reinterpret_cast<_Float16*>(wbase + 8192 + 512);              // [16 edges][C16_FS] feature tile (hi, then lo)
  float* QA = g1 + 4 * G1_WAVE_FLOATS;   // slots 0-3: the waves' a_r (+ l, m), slot 4: q~
  float* AG = QA + 5 * G1_QSL;           // row 0: q, rows 1-4: the waves' a_v
  const half8* str = stg + mi * 16 + (kq ^ (mi & 3));
  const int sra = mi >> 2;
  const bool loA = mi >= 8;
  const float selm = loA ? -1.f : -0.f;
  const int hv = (lane & 31) >> 2, eh = lane >> 5;
  // B operands of the score MFMAs: lane -> column n = mi (head mi & 7, hi | lo half), k-block kq
  half8 bq[3], bk[4];
  float cqm;
  {
    const int hB = mi & 7;
    const float* qtp = QA + 4 * G1_QSL + hB * G1_QH + 8 * kq;
    const float* qp = AG + 8 * kq;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const float4 v0 = *reinterpret_cast<const float4*>(qtp + 32 * ks), v1 = *reinterpret_cast<const float4*>(qtp + 32 * ks + 4);
      const float qv_[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      bq[ks] = f16_sel8(qv_, selm);
    }
    const int ksel = hB >> 1;
    const bool mine = (kq >> 1) == (hB & 1);
    const float4 w0 = *reinterpret_cast<const float4*>(qp + 32 * ksel), w1 = *reinterpret_cast<const float4*>(qp + 32 * ksel + 4);
    const float kv_[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    const half8 bkv = f16_sel8(kv_, selm);
    const half8 zero = {};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bk[ks] = (mine && ks == ksel) ? bkv : zero;
    cqm = cq[hB];
  }
  float m_run = -INFINITY, l_run = 0.f;
  floatx4 ar[6];
#pragma unroll
  for (int cb = 0; cb < 6; ++cb) ar[cb] = floatx4{0.f, 0.f, 0.f, 0.f};
  float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
  GeoRec cur = S.r0, nxt = S.r1;
  float4 vv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) vv[j] = S.vv[j];
  int sb = 0;
#pragma unroll 1
  for (int t0 = 16 * wave; t0 < deg;) {
    const int n = min(16, deg - t0);
    const int tn = __builtin_amdgcn_readfirstlane(t0 + 64);
    // ---- everything this tile reads left a tile ago: one wait, the fragments, then the NEXT tile's requests
    half8 akh[4], akl[4];
    c16_wait_vm0();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { akh[ks] = str[4 * (ks ^ sra)]; akl[ks] = str[256 + 4 * (ks ^ sra)]; }
    float4 vvn[8];
    GeoRec far = nxt;
    if (tn < deg) {
      c16_lat_issue(c, sb ^ 1, nxt.src, vvn);
      far = c16_lat_rec(c, e_beg, deg, tn + 64);
    } else {
      c16_wait_lgkm0();
#pragma unroll
      for (int j = 0; j < 8; ++j) vvn[j] = vv[j];
    }
    const float4 g0 = cur.g;
    const float nmr = cur.nn;
    float sreg[4];
    half8 fl[3];
    {
      floatx4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(akh[ks], bk[ks], acc2, 0, 0, 0);
      half8 fh[3];
      feat8(g0.x, g0.w, nmr, dvp, rdvp, fh[0], fl[0]);
      feat8(g0.y, g0.w, nmr, dvp, rdvp, fh[1], fl[1]);
      feat8(g0.z, g0.w, nmr, dvp, rdvp, fh[2], fl[2]);
      if (__builtin_expect(__any(!(fdiv16_ok(g0.x) && fdiv16_ok(g0.y) && fdiv16_ok(g0.z))), 0)) {
        _Float16* frow = Ft + mi * C16_FS;
        feat_slow_row(g0.x, g0.y, g0.z, g0.w, nmr, dv, kq, frow, true);
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) fl[ks] = *reinterpret_cast<const half8*>(frow + 32 * ks + 8 * kq);
        feat_slow_row(g0.x, g0.y, g0.z, g0.w, nmr, dv, kq, frow, false);
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) fh[ks] = *reinterpret_cast<const half8*>(frow + 32 * ks + 8 * kq);
      }
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) *reinterpret_cast<half8*>(Ft + mi * C16_FS + 32 * ks + 8 * kq) = fh[ks];
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[ks], bq[ks], acc, 0, 0, 0);
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[ks], bq[ks], acc, 0, 0, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(akl[ks], bk[ks], acc2, 0, 0, 0);
      acc += acc2;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float v = acc[r4] + dpp_xor8(acc[r4]);
        sreg[r4] = (v + cqm) * (0.25f * 1.44269504088896341f);
      }
      if (n < 16) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) sreg[r4] = (4 * kq + r4 < n) ? sreg[r4] : -INFINITY;
      }
    }
    // ---- online softmax over the tile (as c16_edge_body)
    float tmax = fmaxf(fmaxf(sreg[0], sreg[1]), fmaxf(sreg[2], sreg[3]));
    const float m_cand = kq_max3(tmax, m_run);
    const bool fresh = m_run == -INFINITY;
#ifdef PS_C16_LAZY   // (experiment: c16_edge_body)
    const float m_new = (fresh || m_cand > m_run + C16_TAU) ? m_cand : m_run;
#else
    const float m_new = m_cand;
#endif
    const float scale = fresh ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.f;
    float pr[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const float p = __builtin_amdgcn_exp2f(sreg[r4] - m_new);
      pr[r4] = p;
      psum += p;
    }
    if (mi < 8) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) Pt[(4 * kq + r4) * 8 + mi] = pr[r4];
    }
    psum = kq_sum(psum);
    l_run = l_run * scale + psum;
    m_run = m_new;
    if (__any(scale != 1.f && !fresh)) {
      float scl[8];
#pragma unroll
      for (int h = 0; h < 8; ++h) scl[h] = __shfl(scale, h);
      const bool up = kq & 1;
      const float s0 = up ? scl[4] : scl[0], s1 = up ? scl[5] : scl[1], s2 = up ? scl[6] : scl[2], s3 = up ? scl[7] : scl[3];
#pragma unroll
      for (int cb = 0; cb < 6; ++cb) { ar[cb][0] *= s0; ar[cb][1] *= s1; ar[cb][2] *= s2; ar[cb][3] *= s3; }
      float sh = scl[0];
#pragma unroll
      for (int h = 1; h < 8; ++h) sh = (hv == h) ? scl[h] : sh;
      av.x *= sh; av.y *= sh; av.z *= sh; av.w *= sh;
    }
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const half4v ap = f16_sel4(pr, selm);
    const _Float16* tp = Ft + (kq * 4 + (mi >> 2)) * C16_FS + (lane & 3) * 4;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1) {
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) *reinterpret_cast<half8*>(Ft + mi * C16_FS + 32 * ks + 8 * kq) = fl[ks];
      }
      fp16x4 tv[6];
#pragma unroll
      for (int cb = 0; cb < 6; ++cb) tv[cb] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(tp + cb * 16));
#pragma unroll
      for (int cb = 0; cb < 6; ++cb) {
        half4v bfr;
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = (_Float16)tv[cb][j];
        ar[cb] = __builtin_amdgcn_mfma_f32_16x16x16f16(ap, bfr, ar[cb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float ph = Pt[(2 * j + eh) * 8 + hv];
      av.x = fmaf(ph, vv[j].x, av.x);
      av.y = fmaf(ph, vv[j].y, av.y);
      av.z = fmaf(ph, vv[j].z, av.z);
      av.w = fmaf(ph, vv[j].w, av.w);
    }
    cur = nxt;
    nxt = far;
#pragma unroll
    for (int j = 0; j < 8; ++j) vv[j] = vvn[j];
    sb ^= 1;
    t0 = tn;
  }
  // ---- the wave's partial sums -> LDS (slot = wave)
  float* oar = QA + wave * G1_QSL;
  av.x = xor_add<32>(av.x); av.y = xor_add<32>(av.y); av.z = xor_add<32>(av.z); av.w = xor_add<32>(av.w);
  if (lane < 8) {
    oar[lane * G1_QH + 96] = l_run;
    oar[lane * G1_QH + 97] = m_run;
  }
  if (lane < 32) *reinterpret_cast<float4*>(AG + (1 + wave) * G1_AGS + 4 * lane) = av;
#pragma unroll
  for (int cb = 0; cb < 6; cb += 2) {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const float v = swap_add32(ar[cb][r4], ar[cb + 1][r4]);
      oar[(4 * ((lane >> 4) & 1) + r4) * G1_QH + (cb + (lane >> 5)) * 16 + (lane & 15)] = v;
    }
  }
}

// phase clocks of the node phases (tools only: build with -DPS_C16_PROF and run with PS_CHAIN_PROF=1; compiled out of the
// product library -- the marks split basic blocks)
#ifdef PS_C16_PROF
#define C16_MARK(i)                                                   \
  do {                                                                \
    if (prof && threadIdx.x == 0) {                                   \
      const long long now_ = clock64();                               \
      atomicAdd(prof + (i), (unsigned long long)(now_ - tprev));      \
      tprev = now_;                                                   \
    }                                                                 \
  } while (0)
#else
#define C16_MARK(i) do { (void)prof; (void)tprev; } while (0)
#endif


#ifdef PS_C16_DBG   // (tools only: six [rows][128] planes of the FIRST layer of the last k_chain16 launch: agg, u, q, s, g, fold)
__device__ float* g_c16_dbg = nullptr;
__device__ int g_c16_dbg_rows = 0;
#define C16_DBG(plane, row, colx, val) do { if (g_c16_dbg && dbg_layer == 0 && (row) < nrows) g_c16_dbg[((size_t)(plane) * g_c16_dbg_rows + row0 + (row)) * 128 + (colx)] = (val); } while (0)
#define C16_DBG_PARAM , int dbg_layer
#define C16_DBG_ARG(l) , (l)
#else
#define C16_DBG(plane, row, colx, val) do { } while (0)
#define C16_DBG_PARAM
#define C16_DBG_ARG(l)
#endif
// The node phases between two edge phases: POST of layer `post` (to_v_r fold, gate, to_out, norms, FFN) and PRE of layer
// `pre` (LN_dst, q | s | g, q~, <q, kb>, the row queue); either may be null.  Out of line like the edge phase, so that the
// weight-fragment ring gets a register allocation that no other phase's pressure can push into scratch.  (The
// residual rows leave in the kernel, after the last call.)
template <int NWV>
__device__ __noinline__ void c16_node_phase(const ChainStep* __restrict__ post, const ChainStep* __restrict__ pre,
                                            unsigned smem_a, int row0, int nrows, int W,
                                            float eps, unsigned long long* __restrict__ prof C16_DBG_PARAM) {
  unsigned char* c16_smem = lds_ptr<unsigned char>(smem_a);   // (an LDS address, not a pointer: ps_device.h lds_addr)
  long long tprev = (prof && threadIdx.x == 0) ? clock64() : 0;
  constexpr int NT = 64 * NWV;
  constexpr size_t EXTRA = NWV * C16_WAVE_BYTES > C16_PLANES_BYTES ? NWV * C16_WAVE_BYTES - C16_PLANES_BYTES : 0;
  // [wave areas beyond the planes (EXTRA)] [P0 | P1 | C] [X | AG | sp | CQ | ctr | QA]: the wave areas run from the start
  unsigned char* node_base = c16_smem + EXTRA;
  _Float16* P0h = reinterpret_cast<_Float16*>(node_base);
  _Float16* P0l = P0h + ND_ROWS * ND_AS;
  _Float16* P1h = P0l + ND_ROWS * ND_AS;
  _Float16* P1l = P1h + ND_ROWS * ND_AS5;
  float* C = reinterpret_cast<float*>(P1l + ND_ROWS * ND_AS5);   // [16][132] results of the 128-column GEMMs
  float* Cw = reinterpret_cast<float*>(P1h);                     // [16][388] wide results of the PRE half (FFN planes' memory)
  float* X = C + ND_ROWS * ND_CS;        // [16][132] residual stream
  float* AG = X + ND_ROWS * ND_XS;       // [16][132] q (PRE -> edge phase)
  float* sp = AG + ND_ROWS * ND_XS;      // [SP_SIZE] the layer's small vectors
  float* CQ = sp + SP_SIZE;              // [16][8] <q_h, kb_h>
  int* ctr = reinterpret_cast<int*>(CQ + 16 * 8);   // [0] row counter, [1..16] the rows in queue order (longest edge list first), [17..32] their edge counts
  float* QA = reinterpret_cast<float*>(ctr + C16_CTR_INTS);   // [16 slots][C16_QSL] q~ / a_r
  _Float16* PBh = reinterpret_cast<_Float16*>(QA + 16 * C16_QSL);   // [16][ND_AS] x 2: a second pair of operand planes, OUTSIDE the wave areas
  _Float16* PBl = PBh + ND_ROWS * ND_AS;
  float* NLN = reinterpret_cast<float*>(PBl + ND_ROWS * ND_AS);   // [256] LN_dst weight | bias of the next layer
  constexpr int C16_DEPTH = 3;   // fragment groups in flight per wave (k_node, alone on its CU with 512 registers: 4)
  typedef FragRingT<C16_DEPTH> Ring;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mi = lane & 15, kq = lane >> 4;
  const bool epi = tid < 256;                      // the 16 x 128 epilogues take 256 threads: row er, 8 columns from ec
  const int er = (tid >> 4) & 15, ec = (tid & 15) * 8;
  const bool live = epi && er < nrows;
  auto stage_sp = [&](const float* __restrict__ src) {
    for (int i = tid; i < SP_SIZE / 4; i += NT) *reinterpret_cast<float4*>(sp + 4 * i) = ldg4(src + 4 * i);
  };
  Ring R;
  // every pointer this call takes out of the two steps, fetched HERE in one batch and made wave-uniform: `post` / `pre` arrive in vector registers
  // (a device-function argument), so each `w.F..` below used to be a flat load of the pointer behind an s_waitcnt vmcnt(0) -- which drained the
  // fragment ring in the middle of every stage and put the pointer's own round trip on the stage chain (eight times per layer)
  const _Float16 *pFqsg = nullptr, *pFvr3 = nullptr, *pFga = nullptr, *pFout = nullptr, *pF1 = nullptr, *pF2 = nullptr, *nFqsg = nullptr, *nFkr3 = nullptr;
  const float* nsp = nullptr;
  const int* neoff = nullptr;
  if (post) {
    const auto cs = uni_const(post);
    pFqsg = cs->w.Fqsg; pFvr3 = cs->w.Fvr3; pFga = cs->w.Fga; pFout = cs->w.Fout; pF1 = cs->w.F1; pF2 = cs->w.F2;
  }
  if (pre) {
    const auto cs = uni_const(pre);
    nFqsg = cs->w.Fqsg; nFkr3 = cs->w.Fkr3; nsp = cs->w.sp; neoff = cs->eoff;
  }
  // The layer's fragment stream, one item = one ring slot's worth (item i in slot i % 3; consuming item i requests item i + 3: gemm16s):
  //   0 1   to_s | to_g of LN_dst(x) (two tiles per wave)      2   the fold's to_v_r fragments (one head per wave)      3   gate      4   to_out
  //   5..8  FFN up (four tiles per wave)      9..12  FFN down (K = 512 in four groups)      13  the NEXT layer's to_q      14 15  its to_k_r (2 x 3 tiles)
  static_assert(NWV == 8 && C16_DEPTH == 3, "the item table below is written for eight waves and a ring of three");
  auto issue = [&](auto ic) {
    constexpr int item = decltype(ic)::value, slot = item % C16_DEPTH;
    if constexpr (item <= 1) frag_issue<4, Ring, NWV>(R, slot, pFqsg + (size_t)8 * 4 * 1024, item, wave, lane);
    else if constexpr (item == 2) {
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const _Float16* f = pFvr3 + ((size_t)(wave * 3 + ks) * 2) * 512 + lane * 8;
        R.h[slot][ks] = ldgh8(f);
        R.l[slot][ks] = ldgh8(f + 512);
      }
    }
    else if constexpr (item == 3) frag_issue<4, Ring, NWV>(R, slot, pFga, 0, wave, lane);
    else if constexpr (item == 4) frag_issue<4, Ring, NWV>(R, slot, pFout, 0, wave, lane);
    else if constexpr (item <= 8) frag_issue<4, Ring, NWV>(R, slot, pF1, item - 5, wave, lane);
    else if constexpr (item <= 12) frag_issue<16, Ring, NWV>(R, slot, pF2, item - 9, wave, lane);
    else if constexpr (item == 13) { if (pre) frag_issue<4, Ring, NWV>(R, slot, nFqsg, 0, wave, lane); }
    else if constexpr (item <= 15) {
      if (pre) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {   // tiles wave + 8 g, g = 3 (item - 14) + j: frag_issue<1>'s addresses
          const _Float16* f = nFkr3 + (size_t)(wave + NWV * (3 * (item - 14) + j)) * 1024 + lane * 8;
          R.h[slot][j] = ldgh8(f);
          R.l[slot][j] = ldgh8(f + 512);
        }
      }
    }
  };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  typedef std::integral_constant<int, 2> I2;
  typedef std::integral_constant<int, 5> I5;
  typedef std::integral_constant<int, 13> I13;
  typedef std::integral_constant<int, 14> I14;
  typedef std::integral_constant<int, 15> I15;
  typedef std::integral_constant<int, 16> I16;
  // the NEXT layer's small vectors leave for registers at the top of this POST half (they do not depend on activations; two float4 per
  // thread at most) and land in LDS when the PRE half starts: the 3.3 k cycles per layer the PRE half spent waiting for them are gone
  constexpr int NSPR = (SP_SIZE / 4 + NT - 1) / NT;
  float4 spr[NSPR];
  const bool sp_early = post && pre;
  if (sp_early) {
#pragma unroll
    for (int i = 0; i < NSPR; ++i)
      if (tid + i * NT < SP_SIZE / 4) spr[i] = ldg4(nsp + 4 * (tid + i * NT));
    if (tid < 64) *reinterpret_cast<float4*>(NLN + 4 * tid) = ldg4(nsp + SP_LN_DST_W + 4 * tid);   // LN_dst weight | bias of the next layer (the last epilogue below)
  }
  int eb_pre = 0, ee_pre = 0;   // (and the rows' edge ranges of the next layer's set, for the same reason)
  if (pre && tid < 16 && tid < nrows) {
    eb_pre = ldgi(neoff + row0 + tid);
    ee_pre = ldgi(neoff + row0 + tid + 1);
  }
  const int col = 16 * wave + mi;   // this lane's column of the wave's 128-column tile (accumulator element r: row 4 kq + r)
  // all 16 rows exist and each has one wave: the common case gets straight-line code (no per-row predicates between the loads, the matrix instructions
  // and the stores of a stage: a predicate is a basic-block boundary, and hipcc schedules inside basic blocks)
  const bool full = __builtin_amdgcn_readfirstlane(nrows) == ND_ROWS && __builtin_amdgcn_readfirstlane(W) == 1;
  const float bq_n = pre ? ldg1(nsp + SP_BQ + col) : 0.f;   // the next layer's to_q bias where the q GEMM's accumulators will want it
  auto planes_put = [&](_Float16* __restrict__ Ph, _Float16* __restrict__ Pl, const float (&v)[4]) {   // accumulator layout -> split-fp16 operand planes
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // the value as an fp32 NUMBER first: left to itself hipcc folds the multiply that made it into the conversion (v_fma_mixlo_f16 x, y, 0 rounds the
      // exact product to fp16 once; planes_store8's packed conversions round the fp32 product) -- another hi | lo pair for the same value, other bits
      float x = v[r];
      asm("" : "+v"(x));
      Ph[(4 * kq + r) * ND_AS + col] = f16_hi(x);
      Pl[(4 * kq + r) * ND_AS + col] = f16_los(x);
    }
  };
  // Round 6: the elementwise steps between two GEMMs (agg, gate, q) are made on the GEMM's accumulators by the lane that holds them -- wave w owns
  // columns 16 w .. 16 w + 15 (= head w) of EVERY 128-column result, so s, g, the fold and the gate's GEMM meet in the same registers -- and
  // LN_dst(x) of the next layer is made by the thread that finishes x.  9 workgroup barriers per layer where there were 15; same operations on the
  // same values in the same order.  Operand planes: PB (outside the wave areas: LN_dst(x) survives the edge phase, so the POST half starts with its
  // GEMM), P0.
  if (post) {
    // =========================================================== POST: to_v_r fold, gate, to_out, norms, FFN   (:76-77, :100-107)
    float sv[4], gv[4], agg[4];
    {
      // to_s / to_g's x_dst half of LN_dst(x) (:106-107): x has not changed since this layer's PRE half made q from the same rows, so the two
      // projections are made HERE, next to their only use, from the planes the PRE half left in PB.
      issue(I0{}); issue(I1{}); issue(I2{});
      {   // fold: f[row][16h + d] = sum_c a_r[row][h][c] * Wvr_g3[c][16h + d], head h = wave; then agg = (a_v + f + l * vb) / (l + 1e-16)   (:89, :100)
        // (the row sums first: LDS reads and conversions that need no weights, while the layer's first fragments are on their way)
        const int h = wave;
        float av_[3][8];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
#pragma unroll
          for (int j = 0; j < 8; ++j) av_[ks][j] = 0.f;
        }
        if (mi < nrows && W == 1) {   // one wave per row: the row's sums as they are (all six loads in flight at once)
          const float* ap_ = QA + mi * C16_QSL + h * C16_QH + kq * 8;
          float4 a0[3], a1[3];
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) { a0[ks] = *reinterpret_cast<const float4*>(ap_ + ks * 32); a1[ks] = *reinterpret_cast<const float4*>(ap_ + ks * 32 + 4); }
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) {
            av_[ks][0] = a0[ks].x; av_[ks][1] = a0[ks].y; av_[ks][2] = a0[ks].z; av_[ks][3] = a0[ks].w;
            av_[ks][4] = a1[ks].x; av_[ks][5] = a1[ks].y; av_[ks][6] = a1[ks].z; av_[ks][7] = a1[ks].w;
          }
        } else if (mi < nrows) {      // W partial sums: common maximum, rescale, add
          float mm = -INFINITY;
          for (int p = 0; p < W; ++p) mm = fmaxf(mm, QA[(mi * W + p) * C16_QSL + h * C16_QH + 97]);
          for (int p = 0; p < W; ++p) {
            const float mp = QA[(mi * W + p) * C16_QSL + h * C16_QH + 97];
            const float sc = (mp == -INFINITY) ? 0.f : exp2f(mp - mm);
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
              const float* ap_ = QA + (mi * W + p) * C16_QSL + h * C16_QH + ks * 32 + kq * 8;
              const float4 a0 = *reinterpret_cast<const float4*>(ap_), a1 = *reinterpret_cast<const float4*>(ap_ + 4);
              av_[ks][0] = fmaf(a0.x, sc, av_[ks][0]); av_[ks][1] = fmaf(a0.y, sc, av_[ks][1]);
              av_[ks][2] = fmaf(a0.z, sc, av_[ks][2]); av_[ks][3] = fmaf(a0.w, sc, av_[ks][3]);
              av_[ks][4] = fmaf(a1.x, sc, av_[ks][4]); av_[ks][5] = fmaf(a1.y, sc, av_[ks][5]);
              av_[ks][6] = fmaf(a1.z, sc, av_[ks][6]); av_[ks][7] = fmaf(a1.w, sc, av_[ks][7]);
            }
          }
        }
        // the row's a_v and l for this lane's four (row, column) elements (rows beyond the workgroup's: zeros, as the epilogue threads had them)
        float av4[4], l4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * kq + r;
          av4[r] = 0.f;
          l4[r] = 0.f;
          if (row < nrows) {
            if (W == 1) {
              l4[r] = QA[row * C16_QSL + h * C16_QH + 96];
              av4[r] = AG[row * ND_XS + col];
            } else {   // merge the W partial softmax sums of the row: common maximum, rescale, add
              float mm = -INFINITY;
              for (int p = 0; p < W; ++p) mm = fmaxf(mm, QA[(row * W + p) * C16_QSL + h * C16_QH + 97]);
              for (int p = 0; p < W; ++p) {
                const int slot = row * W + p;
                const float mp = QA[slot * C16_QSL + h * C16_QH + 97];
                const float sc = (mp == -INFINITY) ? 0.f : exp2f(mp - mm);
                l4[r] = fmaf(QA[slot * C16_QSL + h * C16_QH + 96], sc, l4[r]);
                av4[r] = fmaf(AG[(8 + slot) * ND_XS + col], sc, av4[r]);
              }
            }
          }
        }
        half8 ah[3], al[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { ah[ks][j] = f16_hi(av_[ks][j]); al[ks][j] = f16_los(av_[ks][j]); }
        }
        gemm16t<4, 16, NWV, 0>(R, PBh, PBl, ND_AS, wave, lane, issue, [&](auto jc, int, const floatx4& acc) {   // tile w: s, tile w + 8: g
          constexpr int j = decltype(jc)::value;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
#ifdef PS_C16_DBG_SUMS
            if constexpr (j == 0) sv[r] = acc[r] + sp[SP_BS + col];
            else gv[r] = acc[r] + sp[SP_BG + col];
#else
            if constexpr (j == 0) { sv[r] = acc[r] + sp[SP_BS + col]; C16_DBG(3, 4 * kq + r, col, sv[r]); }
            else { gv[r] = acc[r] + sp[SP_BG + col]; C16_DBG(4, 4 * kq + r, col, gv[r]); }
#endif
          }
        });
        half8 bh[3], bl[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          bh[ks] = R.h[2][ks];   // (item 2)
          bl[ks] = R.l[2][ks];
        }
        issue(I5{});
        floatx4 acc = {0.f, 0.f, 0.f, 0.f}, acx = acc;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bh[ks], acc, 0, 0, 0);
          acx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bl[ks], acx, 0, 0, 0);
          acx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ks], bh[ks], acx, 0, 0, 0);
        }
        const float vb = sp[SP_VB + col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float f = fmaf(acx[r], PS_LO_INV, acc[r]);
          const float l = l4[r];
          const float inv = 1.f / (l + 1e-16f);
          agg[r] = (av4[r] + f + l * vb) * inv;
          C16_DBG(0, 4 * kq + r, col, agg[r]); C16_DBG(5, 4 * kq + r, col, f);
#ifdef PS_C16_DBG_SUMS   // (planes 3 | 4 carry the row's l and a_v instead of s and g)
          C16_DBG(3, 4 * kq + r, col, l4[r]); C16_DBG(4, 4 * kq + r, col, av4[r]);
#endif
        }
        planes_put(P0h, P0l, agg);
      }
      __syncthreads();
      C16_MARK(16);
      // gated update (:106-107): g = sigmoid(Wg [agg | x_dst] + bg); u = agg + g * (to_s(x_dst) - agg), on the gate GEMM's accumulators
      gemm16t<4, 8, NWV, 3>(R, P0h, P0l, ND_AS, wave, lane, issue, [&](auto, int, const floatx4& acc) {
        float u[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g = 1.f / (1.f + expf(-(acc[r] + gv[r])));
          u[r] = agg[r] + g * (sv[r] - agg[r]);
          C16_DBG(1, 4 * kq + r, col, u[r]);
        }
        planes_put(PBh, PBl, u);   // (PB retired as an operand with the barrier above)
      });
      __syncthreads();
      C16_MARK(18);
      gemm16s<4, 8, NWV, 4>(R, PBh, PBl, ND_AS, C, ND_CS, wave, lane, issue);
      __syncthreads();
      C16_MARK(21);
      {   // x = x + LN_post(to_out(u))  (:76), then LN_ffpre(x)  (:77)
        float o[8], xv[8];
        if (epi) {
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = C[er * ND_CS + ec + i] + sp[SP_BOUT + ec + i];
          row16_ln(o, sp + SP_LN_POST_W, sp + SP_LN_POST_B, ec, eps);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            xv[i] = X[er * ND_XS + ec + i] + o[i];
            X[er * ND_XS + ec + i] = xv[i];
          }
          row16_ln(xv, sp + SP_LN_FFPRE_W, sp + SP_LN_FFPRE_B, ec, eps);
          planes_store8(P0h + er * ND_AS + ec, P0l + er * ND_AS + ec, xv);
        }
      }
      __syncthreads();
      C16_MARK(23);
      gemm16s<4, 32, NWV, 5>(R, P0h, P0l, ND_AS, nullptr, 0, wave, lane, issue, P1h, P1l, ND_AS5, sp + SP_B1);
      __syncthreads();
      C16_MARK(24);
      gemm16s<16, 8, NWV, 9>(R, P1h, P1l, ND_AS5, C, ND_CS, wave, lane, issue);   // (requests the next PRE's fragments)
      __syncthreads();
      C16_MARK(25);
      if (epi) {   // x = x + LN_ffpost(FFN); and, its rows in hand, LN_dst(x) of the next layer into PB (:61 of the next layer)
        float y[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] = C[er * ND_CS + ec + i] + sp[SP_B2 + ec + i];
        row16_ln(y, sp + SP_LN_FFPOST_W, sp + SP_LN_FFPOST_B, ec, eps);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          y[i] += X[er * ND_XS + ec + i];
          X[er * ND_XS + ec + i] = y[i];
        }
        if (pre) {
          row16_ln(y, NLN, NLN + 128, ec, eps);
          planes_store8(PBh + er * ND_AS + ec, PBl + er * ND_AS + ec, y);
        }
      }
      __syncthreads();   // X is final for this layer; sp may be restaged
      C16_MARK(26);
    }
  }
  C16_MARK(2);
  if (pre) {
    // =========================================================== PRE: LN_dst, q, q~, <q, kb>, the row queue   (:61-69, :114)
    if (sp_early) {
#pragma unroll
      for (int i = 0; i < NSPR; ++i)
        if (tid + i * NT < SP_SIZE / 4) *reinterpret_cast<float4*>(sp + 4 * (tid + i * NT)) = spr[i];
    } else {
      stage_sp(nsp);
    }
    if (wave == 0) {   // (for the row queue, and so that a row's wave finds its edge range in LDS instead of behind two global loads)
      // queue order of the edge phase: rows by falling edge count, long rows first -- ranked across the 16 lanes that hold the counts
      const int mine = (tid < 16 && tid < nrows) ? ee_pre - eb_pre : -1;
      int rank = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int dj = __builtin_amdgcn_readlane(mine, j);
        rank += (dj > mine || (dj == mine && j < tid)) ? 1 : 0;
      }
      if (tid < 16) {
        ctr[17 + tid] = mine;
        ctr[36 + tid] = eb_pre;
        ctr[1 + rank] = tid;
      }
      if (tid == 0) ctr[0] = 0;
    }
    if (!post) {   // the launch's first layer: LN_dst(x) of the rows the kernel loaded
      issue(I13{}); issue(I14{}); issue(I15{});   // (later layers: requested during the previous POST's FFN)
      __syncthreads();
      C16_MARK(32);
      if (epi) {
        float xn[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xn[i] = X[er * ND_XS + ec + i];
        row16_ln(xn, sp + SP_LN_DST_W, sp + SP_LN_DST_B, ec, eps);
        planes_store8(PBh + er * ND_AS + ec, PBl + er * ND_AS + ec, xn);
      }
      __syncthreads();
      C16_MARK(33);
    }
    // q on the GEMM's accumulators: the fp32 rows for the edge phase (AG) and the operand planes of q~ (P0)
    gemm16t<4, 8, NWV, 13>(R, PBh, PBl, ND_AS, wave, lane, issue, [&](auto, int, const floatx4& acc) {
      float q[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        q[r] = acc[r] + bq_n;
#ifdef PS_C16_DBG
        if (g_c16_dbg && dbg_layer == 100 && 4 * kq + r < nrows) g_c16_dbg[((size_t)2 * g_c16_dbg_rows + row0 + 4 * kq + r) * 128 + col] = q[r];
#endif
        AG[(4 * kq + r) * ND_XS + col] = q[r];
      }
      planes_put(P0h, P0l, q);
    });
    __syncthreads();   // (... and the small vectors, the row queue)
    C16_MARK(34);
    // q~[row][h][c] = sum_d q[row][16h + d] * Wkr_g3[16h + d][c]: (head, 16-column tile) pairs over the waves
    {
      constexpr int NTQ = 6;
      if (full) {
        floatx4 qt[NTQ];
        static_for<0, NTQ>([&](auto gc) {
          constexpr int g = decltype(gc)::value, slot = (14 + g / 3) % C16_DEPTH, j = g % 3;
          const half8 bh = R.h[slot][j], bl = R.l[slot][j];
          const int t = wave + NWV * g, h = t / NTQ;
          const half8 ah = *reinterpret_cast<const half8*>(P0h + mi * ND_AS + (h >> 1) * 32 + kq * 8);
          const half8 al = *reinterpret_cast<const half8*>(P0l + mi * ND_AS + (h >> 1) * 32 + kq * 8);
          floatx4 acc = {0.f, 0.f, 0.f, 0.f}, acx = acc;
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
          acx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acx, 0, 0, 0);
          acx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acx, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) qt[g][r] = fmaf(acx[r], PS_LO_INV, acc[r]);
        });
        static_for<0, NTQ>([&](auto gc) {
          constexpr int g = decltype(gc)::value;
          const int t = wave + NWV * g, h = t / NTQ, nt = t - h * NTQ;
#pragma unroll
          for (int r = 0; r < 4; ++r) QA[(4 * kq + r) * C16_QSL + h * C16_QH + nt * 16 + mi] = qt[g][r];
        });
      } else
      static_for<0, NTQ>([&](auto gc) {   // tile t = wave + 8 g: items 14 (g < 3) and 15
        constexpr int g = decltype(gc)::value, slot = (14 + g / 3) % C16_DEPTH, j = g % 3;
        const half8 bh = R.h[slot][j], bl = R.l[slot][j];
        const int t = wave + NWV * g, h = t / NTQ, nt = t - h * NTQ;
        const half8 ah = *reinterpret_cast<const half8*>(P0h + mi * ND_AS + (h >> 1) * 32 + kq * 8);
        const half8 al = *reinterpret_cast<const half8*>(P0l + mi * ND_AS + (h >> 1) * 32 + kq * 8);
        floatx4 acc = {0.f, 0.f, 0.f, 0.f}, acx = acc;
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
        acx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acx, 0, 0, 0);
        acx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acx, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * kq + r < nrows) QA[((W == 1 ? 0 : 8) + 4 * kq + r) * C16_QSL + h * C16_QH + nt * 16 + mi] = fmaf(acx[r], PS_LO_INV, acc[r]);
      });
      if (lane < 16) {   // cq[row][h] = <q_h, kb_h>: 16 (row, head) pairs per wave, each its 16 fmas in column order
        const int idx = 16 * wave + lane, r = idx >> 3, h = idx & 7;
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) a = fmaf(AG[r * ND_XS + h * DH + d], sp[SP_KB + h * DH + d], a);
        CQ[r * 8 + h] = a;
      }
    }
    __syncthreads();   // q~ rows, cq, the row queue: visible to every wave
    C16_MARK(36);
  }
}

// NWV: waves per workgroup (8: one workgroup per CU, two waves per SIMD hide each other's latency; 4: two workgroups per CU).
// POLICY: own symbol for the per-replan policy launch (kernel traces, bench.py) + the XCD-aware block -> row mapping.
// rows: destination rows per workgroup (<= 16; the MFMA tiles are 16 rows, the rest zero).  With rows < NWV a row's edge
// list is shared by W = NWV / rows waves (tiles w, w + W, ...): every wave keeps its own running maximum and sums, the
// POST half merges the W partials.  ONEW: rows >= NWV (W = 1) -- a template parameter and not a branch, so that every kernel
// has ONE edge-phase callee (two would cost the inter-procedural register allocation: 113 registers saved per call).
// disable_tail_calls: a call that hipcc may mark `tail` makes its callee unsafe for the no-callee-saved-registers treatment
// (TargetFrameLowering::isSafeForNoCSROpt), and each phase function would then save and restore ~90 registers through
// scratch on every call (the round-2 kernel escaped that only because it passed a struct by value).
template <int NWV, bool POLICY, bool ONEW>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1) __attribute__((disable_tail_calls)) void k_chain16(float* __restrict__ x, const float* __restrict__ x_in, int Nd, int rows,
                                                    const ChainStep* __restrict__ steps, int nsteps,
                                                    const float* __restrict__ div32, float eps, int xcd,
                                                    unsigned long long* __restrict__ prof) {
  // prof (PS_CHAIN_PROF=1; nullptr in every product launch): thread 0 of each workgroup charges the cycles since the
  // previous mark to slot i: 0 PRE, 1 EDGE, 2 POST; wave 0 of the edge phase: 4 records + k issue, 5 Fourier rows,
  // 6 staging + score MFMAs, 7 softmax, 8 a_r MFMAs, 9 a_v, 10 row epilogue, 11 row prologue, 12 tiles
  long long tprev = prof ? clock64() : 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char c16_smem[];
  constexpr size_t EXTRA = NWV * C16_WAVE_BYTES > C16_PLANES_BYTES ? NWV * C16_WAVE_BYTES - C16_PLANES_BYTES : 0;
  // [wave areas beyond the planes (EXTRA)] [P0 | P1 | C] [X | AG | sp | CQ | ctr | QA]: the wave areas run from the start
  unsigned char* node_base = c16_smem + EXTRA;
  _Float16* P0h = reinterpret_cast<_Float16*>(node_base);
  _Float16* P0l = P0h + ND_ROWS * ND_AS;
  _Float16* P1h = P0l + ND_ROWS * ND_AS;
  _Float16* P1l = P1h + ND_ROWS * ND_AS5;
  float* C = reinterpret_cast<float*>(P1l + ND_ROWS * ND_AS5);   // [16][132] results of the 128-column GEMMs
  float* Cw = reinterpret_cast<float*>(P1h);                     // [16][388] wide results of the PRE half (FFN planes' memory)
  float* X = C + ND_ROWS * ND_CS;        // [16][132] residual stream
  float* AG = X + ND_ROWS * ND_XS;       // [16][132] q (PRE -> edge phase)
  float* sp = AG + ND_ROWS * ND_XS;      // [SP_SIZE] the layer's small vectors
  float* CQ = sp + SP_SIZE;              // [16][8] <q_h, kb_h>
  int* ctr = reinterpret_cast<int*>(CQ + 16 * 8);   // [0] row counter, [1..16] the rows in queue order (longest edge list first), [17..32] their edge counts
  float* QA = reinterpret_cast<float*>(ctr + C16_CTR_INTS);   // [16 slots][C16_QSL] q~ / a_r

  const int row0 = xcd_block(blockIdx.x, gridDim.x, !xcd) * rows;
  const int nrows = min(rows, Nd - row0);          // rows of this workgroup that exist (> 0: the grid is ceil(Nd / rows))
  const int W = rows < NWV ? NWV / rows : 1;       // waves per row in the edge phase (rows is a power of two)
  {
    const int tid0 = threadIdx.x, er0 = (tid0 >> 4) & 15, ec0 = (tid0 & 15) * 8;
    if (tid0 < 256) {
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (er0 < nrows) { v0 = ldg4(x_in + (size_t)(row0 + er0) * 128 + ec0); v1 = ldg4(x_in + (size_t)(row0 + er0) * 128 + ec0 + 4); }
      *reinterpret_cast<float4*>(X + er0 * ND_XS + ec0) = v0;
      *reinterpret_cast<float4*>(X + er0 * ND_XS + ec0 + 4) = v1;
    }
  }
  // PRE(0); then per layer EDGE(s), POST(s) + PRE(s + 1).  Every phase is a function of its own: the phases' register
  // needs differ (the edge phase's accumulators and operands, the node phases' weight-fragment ring) and inlined into one
  // body each pushed the other's into scratch.
  const unsigned smem_a = lds_addr(c16_smem);
  c16_node_phase<NWV>(nullptr, steps, smem_a, row0, nrows, W, eps, prof C16_DBG_ARG(100));
  C16_MARK(0);
  for (int s = 0; s < nsteps; ++s) {
#ifdef PS_EDGE_OPAQUE
    if (POLICY) c16_edge_phase<NWV, ONEW, 2>(steps + s, c16_smem, AG, CQ, ctr, QA, div32, row0, nrows, W);
    else
#endif
    c16_edge_phase<NWV, ONEW>(steps + s, c16_smem, AG, CQ, ctr, QA, div32, row0, nrows, W);
    __syncthreads();   // every row's sums are in place; the wave-private areas are dead
    C16_MARK(1);
    const bool last = s + 1 == nsteps;
    c16_node_phase<NWV>(steps + s, last ? nullptr : steps + s + 1, smem_a, row0, nrows, W, eps, prof C16_DBG_ARG(s));
    C16_MARK(2);
  }
  {   // the residual rows leave (the last POST half ends behind a barrier).  Here and not in the node phase: on gfx9 a store counts on vmcnt like
      // the fragment loads, and with one possibly pending hipcc turns every wait behind it into s_waitcnt vmcnt(0)
    const int tid0 = threadIdx.x, er0 = (tid0 >> 4) & 15, ec0 = (tid0 & 15) * 8;
    if (tid0 < 256 && er0 < nrows) {
      *reinterpret_cast<float4*>(x + (size_t)(row0 + er0) * 128 + ec0) = *reinterpret_cast<const float4*>(X + er0 * ND_XS + ec0);
      *reinterpret_cast<float4*>(x + (size_t)(row0 + er0) * 128 + ec0 + 4) = *reinterpret_cast<const float4*>(X + er0 * ND_XS + ec0 + 4);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// The edge phase alone, for the SPLIT attention layers (round 4): between the row-tile node halves (ps_rowtile.h) the scene encoder's
// s2s layers ran k_edge_small over 768 bytes of rel-PE operand images per edge (k_relpe_tiles: 0.53 GB per launch out of L2 / MALL).
// k_edge16 runs c16_edge_phase -- geometry records, rows rebuilt in registers, k rows by LDS-DMA -- on 16 destination rows per 8-wave
// workgroup: q / q~ / <q, kb> come from the PRE half's EdgeIO rows into the slots the fused chain keeps them in, the sums go back the
// same way.  Same arithmetic as a one-step k_chain16 launch's edge phase.
#ifdef PS_EXPERIMENTS   // (cross-check of k_edge_rows: ps_set_row_impl(2) in experiments builds; the product library does not carry it)
constexpr size_t c16_edge_lds_bytes() { return C16_EDGE_WAVES_BYTES + (size_t)ND_ROWS * ND_XS * 4 + 16 * 8 * 4 + C16_CTR_INTS * 4 + C16_QA_BYTES; }
__global__ __launch_bounds__(512, 1) __attribute__((disable_tail_calls)) void k_edge16(int Nd, const ChainStep* __restrict__ step, EdgeIO io,
                                                                                     const float* __restrict__ div32) {
  extern __shared__ __attribute__((aligned(16))) unsigned char c16_smem[];
  float* AG = reinterpret_cast<float*>(c16_smem + C16_EDGE_WAVES_BYTES);   // [16][ND_XS] q rows, then a_v
  float* CQ = AG + ND_ROWS * ND_XS;                                        // [16][8]
  int* ctr = reinterpret_cast<int*>(CQ + 16 * 8);                          // [0] counter, [1..16] queue, [17..32] degrees
    float* QA = reinterpret_cast<float*>(ctr + C16_CTR_INTS);                          // [16 slots][C16_QSL] q~ in, a_r | l out
  const int tid = threadIdx.x;
  const int row0 = blockIdx.x * 16;
  const int nrows = min(16, Nd - row0);
  for (int i = tid; i < 16 * 32; i += 512) {   // q rows
    const int r = i >> 5, c = (i & 31) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nrows) v = ldg4(io.q + (size_t)(row0 + r) * 128 + c);
    *reinterpret_cast<float4*>(AG + r * ND_XS + c) = v;
  }
  for (int i = tid; i < 16 * 8 * 24; i += 512) {   // q~: [row][head][96 columns] into the slot layout
    const int r = i / 192, rem = i - r * 192, h = rem / 24, c = (rem - h * 24) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nrows) v = ldg4(io.qt + (size_t)(row0 + r) * 1024 + h * 128 + c);
    *reinterpret_cast<float4*>(QA + r * C16_QSL + h * C16_QH + c) = v;
  }
  if (tid < 128) CQ[tid] = (tid >> 3) < nrows ? ldg1(io.cq + (size_t)row0 * 8 + tid) : 0.f;
  if (tid < 16) {
    const int eb = tid < nrows ? ldgi(step->eoff + row0 + tid) : 0;
    ctr[17 + tid] = tid < nrows ? ldgi(step->eoff + row0 + tid + 1) - eb : -1;
    ctr[36 + tid] = eb;
  }
  __syncthreads();
  if (tid == 0) ctr[0] = 0;
  if (tid < 16) {   // queue order: rows by falling edge count (as the fused chain's PRE half)
    const int mine = ctr[17 + tid];
    int rank = 0;
    for (int j = 0; j < 16; ++j) {
      const int dj = ctr[17 + j];
      rank += (dj > mine || (dj == mine && j < tid)) ? 1 : 0;
    }
    ctr[1 + rank] = tid;
  }
  __syncthreads();
  c16_edge_phase<8, true, 1>(step, c16_smem, AG, CQ, ctr, QA, div32, row0, nrows, 1);
  __syncthreads();
  for (int i = tid; i < 16 * 8 * 24; i += 512) {   // a_r
    const int r = i / 192, rem = i - r * 192, h = rem / 24, c = (rem - h * 24) * 4;
    if (r < nrows) *reinterpret_cast<float4*>(io.ar + (size_t)(row0 + r) * 1024 + h * 128 + c) = *reinterpret_cast<const float4*>(QA + r * C16_QSL + h * C16_QH + c);
  }
  for (int i = tid; i < 16 * 32; i += 512) {   // a_v
    const int r = i >> 5, c = (i & 31) * 4;
    if (r < nrows) *reinterpret_cast<float4*>(io.av + (size_t)(row0 + r) * 128 + c) = *reinterpret_cast<const float4*>(AG + r * ND_XS + c);
  }
  if (tid < 128 && (tid >> 3) < nrows) io.l[(size_t)row0 * 8 + tid] = QA[(tid >> 3) * C16_QSL + (tid & 7) * C16_QH + 96];
}
#endif   // PS_EXPERIMENTS


// The edge phase with NO workgroup structure (round 4): one wave per destination row, four waves per workgroup, three workgroups per CU
// (168 registers, 47 KB of LDS each) -- nothing is staged by the workgroup, there is no barrier and no row queue; a wave's latency chain
// (records -> source rows -> k rows -> scores) is covered by the eleven other waves of its CU instead of by the row's next tile, which
// is what the short rows of the scene encoder (two tiles) lack in the 16-row workgroup form.
constexpr int ER_WAVES = 4;
constexpr size_t ER_LDS_BYTES = ER_WAVES * (C16_WAVE_BYTES + C16_PARK_BYTES);
__global__ __launch_bounds__(64 * ER_WAVES, 3) __attribute__((disable_tail_calls)) void k_edge_rows(int Nd, const ChainStep* __restrict__ step, EdgeIO io,
                                                                                                   const float* __restrict__ div32, int xcd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char c16_smem[];
  const int row0 = xcd_block(blockIdx.x, gridDim.x, !xcd) * ER_WAVES;   // (rows are ordered by scene: an XCD's L2 holds the k | v rows of ~ 1/8 of the scenes)
  c16_edge_body<ER_WAVES, true, true>(step, c16_smem, nullptr, nullptr, nullptr, nullptr, div32, row0, min(ER_WAVES, Nd - row0), 1, io);
}

}  // namespace ps
