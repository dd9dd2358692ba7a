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
  const float* vb;                                  // [128]: to_v_r.weight @ beta_r + to_v_r.bias
  const float *Wga_t, *Wout_t, *bout;
  const float *ln_post_w, *ln_post_b, *ln_ffpre_w, *ln_ffpre_b;
  const float *W1_t, *b1, *W2_t, *b2, *ln_ffpost_w, *ln_ffpost_b;
  const float *Wkv_t, *bkv;                         // [128][256] K-major (k | v), [256]
};

struct ChainStep {
  AttnW w;
  const float* kv;    // [Ns][256] projected sources (k | v) for this layer
  const int* eoff;    // [Nd+1] CSR offsets by destination
  const int* esrc;    // [E] source row in kv
  const float* rt;    // [E][128] normalised relative-PE (no affine)
};

// LDS plan (floats): rows 6*128*T + 512*T | big 4*8*QP (q~ image, then per-wave partial a_r)
// | un = max(1024*T partial sums, 8*maxdeg*T scores) | avp 4*128 | ml 4*16 | cq 8*T
template <int T>
__host__ __device__ constexpr size_t attn_lds_floats(int maxdeg) {
  size_t un = (size_t)1024 * T;
  size_t sc = (size_t)8 * maxdeg * T;
  return (size_t)(6 * 128 + 512) * T + 4 * 8 * QP + (un > sc ? un : sc) + 4 * 128 + 4 * 16 + 8 * T + 64;
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

template <int T>
__global__ __launch_bounds__(WG, 1) void k_attn_chain(float* __restrict__ x, int Nd, const ChainStep* __restrict__ steps,
                                                     int nsteps, int maxdeg, float eps) {
  constexpr int W = 4 / T;  // waves per destination in the edge phase
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                 // [T][128] residual stream
  float* xn = xs + 128 * T;         // [T][128] normed / scratch row
  float* qb = xn + 128 * T;         // [T][128] q
  float* sb = qb + 128 * T;         // [T][128] to_s(x_dst)
  float* gb = sb + 128 * T;         // [T][128] to_g's x_dst half (+bias)
  float* ag = gb + 128 * T;         // [T][128] aggregated message
  float* f1 = ag + 128 * T;         // [T][512]
  float* big = f1 + 512 * T;        // [4][8][QP]
  const size_t un_sz = ((size_t)1024 * T > (size_t)8 * maxdeg * T) ? (size_t)1024 * T : (size_t)8 * maxdeg * T;
  float* un = big + 4 * 8 * QP;     // partial sums | scores
  float* avp = un + un_sz;          // [4][128]
  float* ml = avp + 4 * 128;        // [4][16]: per wave (max[8] | sum[8])
  float* cq = ml + 4 * 16;          // [T][8]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int row0 = blockIdx.x * T;
  // load the T residual rows (rows past Nd are zero-filled and never stored)
  for (int i = tid; i < T * 128; i += WG) {
    const int t = i >> 7, r = row0 + t;
    xs[i] = (r < Nd) ? x[(size_t)r * 128 + (i & 127)] : 0.f;
  }
  __syncthreads();

  for (int s = 0; s < nsteps; ++s) {
    const ChainStep& st = steps[s];
    const AttnW& w = st.w;
    // ---- pre-norm of the destination rows + q / s / gate(x) projections (:61-69, :106-107, :114)
    ln_rows<T>(xs, 128, xn, 128, w.ln_dst_w, w.ln_dst_b, eps, false);
    __syncthreads();
    gemv_rows<T, false>(xn, 128, 128, w.Wq_t, 128, w.bq, un, qb, 128, false);
    gemv_rows<T, false>(xn, 128, 128, w.Ws_t, 128, w.bs, un, sb, 128, false);
    gemv_rows<T, false>(xn, 128, 128, w.Wgx_t, 128, w.bg, un, gb, 128, false);
    // ---- q~[t][h][c] = sum_d q[t][16h+d] * Wkr_g[16h+d][c];  cq[t][h] = <q_h, kb_h>
    {
      const int h = tid >> 5, c4 = tid & 31;
      float acc[T][4];
#pragma unroll
      for (int t = 0; t < T; ++t) acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        const float4 wv = *reinterpret_cast<const float4*>(w.Wkr_g + (size_t)(h * DH + d) * 128 + 4 * c4);
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const float qv = qb[t * 128 + h * DH + d];
          acc[t][0] = fmaf(qv, wv.x, acc[t][0]);
          acc[t][1] = fmaf(qv, wv.y, acc[t][1]);
          acc[t][2] = fmaf(qv, wv.z, acc[t][2]);
          acc[t][3] = fmaf(qv, wv.w, acc[t][3]);
        }
      }
#pragma unroll
      for (int t = 0; t < T; ++t)
        *reinterpret_cast<float4*>(big + (size_t)(t * 8 + h) * QP + 4 * c4) =
            make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
      if (tid < 8 * T) {
        const int t = tid >> 3, hh = tid & 7;
        float a = 0.f;
        for (int d = 0; d < DH; ++d) a = fmaf(qb[t * 128 + hh * DH + d], w.kb[hh * DH + d], a);
        cq[tid] = a;
      }
    }
    __syncthreads();

    // ---- edge phase: wave -> (destination t, sub-wave wi); lane -> (edge slot es, 16-column chunk cc)
    const int t = wave / W, wi = wave % W;
    const int r = row0 + t;
    const int e_beg = (r < Nd) ? st.eoff[r] : 0;
    const int deg = (r < Nd) ? (st.eoff[r + 1] - e_beg) : 0;
    const int es = lane >> 3, cc = lane & 7;
    float* sc = un + (size_t)t * maxdeg * 8;
    {
      // pass 1: scores s[e][h] = (<q_h, k_src,h> + <q~_h, r~_e> + cq_h) * Dh^-0.5   (:88-90)
      float qt[8][16];
#pragma unroll
      for (int h = 0; h < 8; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 v = *reinterpret_cast<const float4*>(big + (size_t)(t * 8 + h) * QP + 4 * (cc + 8 * i));
          qt[h][4 * i] = v.x; qt[h][4 * i + 1] = v.y; qt[h][4 * i + 2] = v.z; qt[h][4 * i + 3] = v.w;
        }
      float qh[16];
#pragma unroll
      for (int d = 0; d < 16; ++d) qh[d] = qb[t * 128 + cc * DH + d];
      const float cqh = cq[t * 8 + cc];
      for (int e0 = wi * 8; e0 < deg; e0 += 8 * W) {
        const int e = e0 + es;
        const bool ok = e < deg;
        const int ee = ok ? e : deg - 1;
        const int src = st.esrc[e_beg + ee];
        const float* rr = st.rt + (size_t)(e_beg + ee) * 128;
        const float* kr = st.kv + (size_t)src * 256 + cc * DH;
        float rv[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 v = *reinterpret_cast<const float4*>(rr + 4 * (cc + 8 * i));
          rv[4 * i] = v.x; rv[4 * i + 1] = v.y; rv[4 * i + 2] = v.z; rv[4 * i + 3] = v.w;
        }
        float qk = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 v = *reinterpret_cast<const float4*>(kr + 4 * i);
          qk = fmaf(qh[4 * i], v.x, qk); qk = fmaf(qh[4 * i + 1], v.y, qk);
          qk = fmaf(qh[4 * i + 2], v.z, qk); qk = fmaf(qh[4 * i + 3], v.w, qk);
        }
        float p[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          float a = 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) a = fmaf(qt[h][i], rv[i], a);
          p[h] = a;
        }
        const float tot = reduce8_to_lane(p, cc);
        if (ok) sc[(size_t)e * 8 + cc] = (tot + qk + cqh) * 0.25f;
      }
    }
    __syncthreads();
    // softmax over the destination's edges, per head (torch_geometric.utils.softmax: max-shift,
    // exp, / (sum + 1e-16)); lanes over edges.  Every wave of the destination finds the max
    // (redundantly), then exponentiates its own slice and publishes its partial sum.
    {
      float m[8];
#pragma unroll
      for (int h = 0; h < 8; ++h) m[h] = -INFINITY;
      for (int e = lane; e < deg; e += 64) {
        const float4 a = *reinterpret_cast<const float4*>(sc + (size_t)e * 8);
        const float4 b = *reinterpret_cast<const float4*>(sc + (size_t)e * 8 + 4);
        m[0] = fmaxf(m[0], a.x); m[1] = fmaxf(m[1], a.y); m[2] = fmaxf(m[2], a.z); m[3] = fmaxf(m[3], a.w);
        m[4] = fmaxf(m[4], b.x); m[5] = fmaxf(m[5], b.y); m[6] = fmaxf(m[6], b.z); m[7] = fmaxf(m[7], b.w);
      }
#pragma unroll
      for (int h = 0; h < 8; ++h) m[h] = wave_max(m[h]);
      __syncthreads();  // all waves have read the raw scores before any wave overwrites them
      float l[8];
#pragma unroll
      for (int h = 0; h < 8; ++h) l[h] = 0.f;
      for (int e = wi * 64 + lane; e < deg; e += 64 * W) {
        float4 a = *reinterpret_cast<const float4*>(sc + (size_t)e * 8);
        float4 b = *reinterpret_cast<const float4*>(sc + (size_t)e * 8 + 4);
        a.x = expf(a.x - m[0]); a.y = expf(a.y - m[1]); a.z = expf(a.z - m[2]); a.w = expf(a.w - m[3]);
        b.x = expf(b.x - m[4]); b.y = expf(b.y - m[5]); b.z = expf(b.z - m[6]); b.w = expf(b.w - m[7]);
        l[0] += a.x; l[1] += a.y; l[2] += a.z; l[3] += a.w; l[4] += b.x; l[5] += b.y; l[6] += b.z; l[7] += b.w;
        *reinterpret_cast<float4*>(sc + (size_t)e * 8) = a;
        *reinterpret_cast<float4*>(sc + (size_t)e * 8 + 4) = b;
      }
#pragma unroll
      for (int h = 0; h < 8; ++h) l[h] = wave_sum(l[h]);
      if (lane < 8) {
        float v = l[0];
#pragma unroll
        for (int h = 1; h < 8; ++h) v = (lane == h) ? l[h] : v;
        ml[wave * 16 + 8 + lane] = v;
      }
    }
    __syncthreads();
    {
      // pass 2: a_r[h][c] = sum_e p_e,h r~_e[c],  a_v[hd] = sum_e p_e,h v_src[hd]   (:100, aggr='add')
      float ar[8][16], av[16];
#pragma unroll
      for (int h = 0; h < 8; ++h)
#pragma unroll
        for (int i = 0; i < 16; ++i) ar[h][i] = 0.f;
#pragma unroll
      for (int d = 0; d < 16; ++d) av[d] = 0.f;
      for (int e0 = wi * 8; e0 < deg; e0 += 8 * W) {
        const int e = e0 + es;
        const bool ok = e < deg;
        const int ee = ok ? e : deg - 1;
        const int src = st.esrc[e_beg + ee];
        const float* rr = st.rt + (size_t)(e_beg + ee) * 128;
        const float* vr = st.kv + (size_t)src * 256 + 128 + cc * DH;
        float4 pa = *reinterpret_cast<const float4*>(sc + (size_t)ee * 8);
        float4 pb = *reinterpret_cast<const float4*>(sc + (size_t)ee * 8 + 4);
        if (!ok) { pa = make_float4(0.f, 0.f, 0.f, 0.f); pb = pa; }
        const float p[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
        float ph = p[0];
#pragma unroll
        for (int h = 1; h < 8; ++h) ph = (cc == h) ? p[h] : ph;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 v = *reinterpret_cast<const float4*>(rr + 4 * (cc + 8 * i));
#pragma unroll
          for (int h = 0; h < 8; ++h) {
            ar[h][4 * i] = fmaf(p[h], v.x, ar[h][4 * i]);
            ar[h][4 * i + 1] = fmaf(p[h], v.y, ar[h][4 * i + 1]);
            ar[h][4 * i + 2] = fmaf(p[h], v.z, ar[h][4 * i + 2]);
            ar[h][4 * i + 3] = fmaf(p[h], v.w, ar[h][4 * i + 3]);
          }
          const float4 vv = *reinterpret_cast<const float4*>(vr + 4 * i);
          av[4 * i] = fmaf(ph, vv.x, av[4 * i]); av[4 * i + 1] = fmaf(ph, vv.y, av[4 * i + 1]);
          av[4 * i + 2] = fmaf(ph, vv.z, av[4 * i + 2]); av[4 * i + 3] = fmaf(ph, vv.w, av[4 * i + 3]);
        }
      }
      // fold the 8 edge slots of the wave (lane bits 3..5), then publish the wave's partials
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
        for (int h = 0; h < 8; ++h)
#pragma unroll
          for (int i = 0; i < 16; ++i) ar[h][i] += __shfl_xor(ar[h][i], o);
#pragma unroll
        for (int d = 0; d < 16; ++d) av[d] += __shfl_xor(av[d], o);
      }
      if (es == 0) {
#pragma unroll
        for (int h = 0; h < 8; ++h)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(big + (size_t)(wave * 8 + h) * QP + 4 * (cc + 8 * i)) =
                make_float4(ar[h][4 * i], ar[h][4 * i + 1], ar[h][4 * i + 2], ar[h][4 * i + 3]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *reinterpret_cast<float4*>(avp + wave * 128 + cc * DH + 4 * i) =
              make_float4(av[4 * i], av[4 * i + 1], av[4 * i + 2], av[4 * i + 3]);
      }
    }
    __syncthreads();
    if (W > 1) {  // sum the W sub-wave partials of each destination into its first slot
      for (int i = tid; i < T * 8 * 128; i += WG) {
        const int tt = i / 1024, hc = i % 1024, h = hc >> 7, c = hc & 127;
        float a = 0.f;
        for (int j = 0; j < W; ++j) a += big[(size_t)((tt * W + j) * 8 + h) * QP + c];
        big[(size_t)((tt * W) * 8 + h) * QP + c] = a;
      }
      for (int i = tid; i < T * 128; i += WG) {
        const int tt = i >> 7, c = i & 127;
        float a = 0.f;
        for (int j = 0; j < W; ++j) a += avp[(tt * W + j) * 128 + c];
        avp[(tt * W) * 128 + c] = a;
      }
      __syncthreads();
    }
    // ---- agg = (a_v + Wvr_g^T a_r + l * vb) / (l + 1e-16)   (to_v_r fold; :89, :100)
    gemv_rows<T, true>(big, W * 8 * QP, 128, w.Wvr_gt, 128, nullptr, un, ag, 128, false);
    for (int i = tid; i < T * 128; i += WG) {
      const int tt = i >> 7, c = i & 127, h = c >> 4;
      float l = 0.f;
      for (int j = 0; j < W; ++j) l += ml[(tt * W + j) * 16 + 8 + h];
      const int rr_ = row0 + tt;
      const bool has = (rr_ < Nd) && (st.eoff[rr_ + 1] > st.eoff[rr_]);
      const float a = (avp[(tt * W) * 128 + c] + ag[i] + l * w.vb[c]) / (l + 1e-16f);
      ag[i] = has ? a : 0.f;
    }
    __syncthreads();
    // ---- gated update (:106-107): g = sigmoid(Wg [agg | x_dst] + bg); u = agg + g * (to_s(x_dst) - agg)
    gemv_rows<T, false>(ag, 128, 128, w.Wga_t, 128, nullptr, un, f1, 512, false);
    for (int i = tid; i < T * 128; i += WG) {
      const int tt = i >> 7, c = i & 127;
      const float g = 1.f / (1.f + expf(-(f1[tt * 512 + c] + gb[i])));
      const float a = ag[i];
      ag[i] = a + g * (sb[i] - a);
    }
    __syncthreads();
    // ---- x = x + LN_post(to_out(u))  (:76)
    gemv_rows<T, false>(ag, 128, 128, w.Wout_t, 128, w.bout, un, xn, 128, false);
    ln_rows<T>(xn, 128, xn, 128, w.ln_post_w, w.ln_post_b, eps, false);
    __syncthreads();
    for (int i = tid; i < T * 128; i += WG) xs[i] += xn[i];
    __syncthreads();
    // ---- x = x + LN_ffpost(W2 relu(W1 LN_ffpre(x) + b1) + b2)  (:77)
    ln_rows<T>(xs, 128, xn, 128, w.ln_ffpre_w, w.ln_ffpre_b, eps, false);
    __syncthreads();
    gemv_rows<T, false>(xn, 128, 128, w.W1_t, 512, w.b1, un, f1, 512, true);
    gemv_rows<T, false>(f1, 512, 512, w.W2_t, 128, w.b2, un, xn, 128, false);
    ln_rows<T>(xn, 128, xn, 128, w.ln_ffpost_w, w.ln_ffpost_b, eps, false);
    __syncthreads();
    for (int i = tid; i < T * 128; i += WG) xs[i] += xn[i];
    __syncthreads();
  }
  for (int i = tid; i < T * 128; i += WG) {
    const int t = i >> 7, r = row0 + t;
    if (r < Nd) x[(size_t)r * 128 + (i & 127)] = xs[i];
  }
}

// k | v projection of source tokens for L layers: kv[l][n][0:128] = Wk LN_src(x_n),
// kv[l][n][128:256] = Wv LN_src(x_n) + bv   (attention_layer.py:61,65,115-116).  grid (ceil(Ns/T), L).
template <int T>
__global__ __launch_bounds__(WG) void k_kv_proj(const float* __restrict__ x, int Ns, const AttnW* __restrict__ layers,
                                                float* __restrict__ kv, size_t layer_stride, float eps) {
  __shared__ __attribute__((aligned(16))) float xs[T * 128];
  __shared__ __attribute__((aligned(16))) float xn[T * 128];
  __shared__ __attribute__((aligned(16))) float part[4 * T * 256];
  __shared__ __attribute__((aligned(16))) float ob[T * 256];
  const AttnW& w = layers[blockIdx.y];
  const int row0 = blockIdx.x * T, tid = threadIdx.x;
  for (int i = tid; i < T * 128; i += WG) {
    const int r = row0 + (i >> 7);
    xs[i] = (r < Ns) ? x[(size_t)r * 128 + (i & 127)] : 0.f;
  }
  __syncthreads();
  ln_rows<T>(xs, 128, xn, 128, w.ln_src_w, w.ln_src_b, eps, false);
  __syncthreads();
  gemv_rows<T, false>(xn, 128, 128, w.Wkv_t, 256, w.bkv, part, ob, 256, false);
  float* out = kv + blockIdx.y * layer_stride;
  for (int i = tid; i < T * 256; i += WG) {
    const int r = row0 + (i >> 8);
    if (r < Ns) out[(size_t)r * 256 + (i & 255)] = ob[i];
  }
}

}  // namespace ps
