// prosim_amd/csrc/ps_pe_learn.h -- the learnable relative positional encoding (edge MLP on the matrix cores).
// Included after ps_kernels.h (pn_gemm, PN_* sizes) and ps_chain16.h (EdgeGeo: the per-edge geometry records).
#pragma once
#include "ps_kernels.h"
#include "ps_chain16.h"

namespace ps {

// ------------------------------------------------------------------------------------------
// K5'  LEARNABLE relative positional encoding (*.ATTN.LEARNABLE_PE; FourierEmbedding, layers/fourier_embedding.py:11-54):
// per edge and per input i of (dist, rel_ori, angle): [cos(x_i f_ik 2 pi), sin(...), x_i] (2 * 64 + 1 features) ->
// Linear, LayerNorm, ReLU, Linear; the three results are summed; then LayerNorm, ReLU, Linear -> r[128].  The row then
// takes the same route as a fixed one: the affine-free LayerNorm every layer's attn_prenorm_r shares, split fp16, both
// MFMA operand images with all 128 columns (KR = 4 consumers).
// A workgroup (4 waves) takes TWO 32-edge tiles = 64 edge rows; every Linear is a [64 x 128] x [128 x 128] split-fp16
// MFMA GEMM (pn_gemm<4>; the raw-input column of the first Linear is a rank-1 term of its epilogue).  Seven GEMMs per
// 64 edges, 64 KB of weight fragments each: 7 KB of L2 -> CU traffic per edge, which is what this kernel costs.
struct PeLearnW {
  const float* freqs;        // [3][64]
  const _Float16* F1[3];     // mlps[i][0].weight[:, 0:128] as B fragments
  const float* w1x[3];       // mlps[i][0].weight[:, 128]
  const float* b1[3];
  const float* ln1w[3];
  const float* ln1b[3];
  const _Float16* F2[3];     // mlps[i][3].weight
  const float* b2sum;        // sum over i of mlps[i][3].bias
  const float *lnow, *lnob;  // to_out[0]
  const _Float16* Fo;        // to_out[2].weight
  const float* bo;
};
constexpr size_t PL_LDS_BYTES = (size_t)2 * PN_ROWS * PN_AS * 2 + (size_t)PN_ROWS * PN_CS * 4 + (size_t)PN_ROWS * 128 * 4 + PN_ROWS * 4 * 4;

__global__ __launch_bounds__(256) void k_pe_learn(PeLearnW w, const EdgeGeo* __restrict__ geo, const int* __restrict__ eoff,
                                                 const int* __restrict__ toff, const int* __restrict__ tdst, int nq,
                                                 _Float16* __restrict__ rtA, _Float16* __restrict__ rtT, float eps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pl_smem[];
  _Float16* Ah = reinterpret_cast<_Float16*>(pl_smem);
  _Float16* Al = Ah + PN_ROWS * PN_AS;
  float* C = reinterpret_cast<float*>(Al + PN_ROWS * PN_AS);
  float* Y = C + PN_ROWS * PN_CS;          // [64][128] sum of the three per-input embeddings
  float* xs = Y + PN_ROWS * 128;           // [64][4]: the three inputs, row valid
  _Float16(*buf)[264] = reinterpret_cast<_Float16(*)[264]>(pl_smem);   // the final rows, over the (then dead) planes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = tid >> 2, c0 = (tid & 3) * 32;   // epilogues: thread -> (row, 32-column quarter); sums meet in the lane quad
  const int ntiles = toff[nq];
  const float PI_F = 3.14159265358979323846f;
  for (int pair = blockIdx.x; 2 * pair < ntiles; pair += gridDim.x) {
    __syncthreads();   // the previous pair's image writes are done with buf
    if (tid < 64) {
      const int tile = 2 * pair + (tid >> 5), rr = tid & 31;
      float x0 = 0.f, x1 = 0.f, x2 = 0.f, ok = 0.f;
      if (tile < ntiles) {
        const int d = tdst[tile];
        const int e0 = eoff[d] + (tile - toff[d]) * 32;
        if (rr < min(32, eoff[d + 1] - e0)) {
          const EdgeGeo g = geo[e0 + rr];
          x0 = g.a0; x1 = g.a1; x2 = g.a2; ok = 1.f;
        }
      }
      xs[4 * tid] = x0; xs[4 * tid + 1] = x1; xs[4 * tid + 2] = x2; xs[4 * tid + 3] = ok;
    }
    for (int i = tid; i < PN_ROWS * 128; i += 256) Y[i] = 0.f;
    __syncthreads();
    for (int in = 0; in < 3; ++in) {
      {   // features of input `in`: thread -> (row, 16 of the 64 frequencies); columns 0..63 cos, 64..127 sin
        const float x = xs[4 * r + in];
        const int k0 = (tid & 3) * 16;
        half8 ch[2], cl[2], sh[2], sl[2];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const float v = ((x * w.freqs[in * 64 + k0 + k]) * 2.f) * PI_F;   // x * freqs * 2 * pi, left to right (:45)
          float sn, cs;
          sincosf(v, &sn, &cs);
          ch[k >> 3][k & 7] = f16_hi(cs); cl[k >> 3][k & 7] = f16_los(cs);
          sh[k >> 3][k & 7] = f16_hi(sn); sl[k >> 3][k & 7] = f16_los(sn);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          *reinterpret_cast<half8*>(Ah + r * PN_AS + k0 + 8 * h) = ch[h];
          *reinterpret_cast<half8*>(Al + r * PN_AS + k0 + 8 * h) = cl[h];
          *reinterpret_cast<half8*>(Ah + r * PN_AS + 64 + k0 + 8 * h) = sh[h];
          *reinterpret_cast<half8*>(Al + r * PN_AS + 64 + k0 + 8 * h) = sl[h];
        }
      }
      __syncthreads();
      pn_gemm<4>(Ah, Al, 4, w.F1[in], C, PN_CS, PN_ROWS, wave, lane);
      __syncthreads();
      {   // + bias + x * W[:, 128]; LayerNorm; ReLU -> the second Linear's planes
        const float x = xs[4 * r + in];
        float a[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) a[i] = C[r * PN_CS + c0 + i] + w.b1[in][c0 + i] + x * w.w1x[in][c0 + i];
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
          half8 h, l;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int c = c0 + 8 * i + j;
            const float v = fmaxf(fmaf(a[8 * i + j] * rstd, w.ln1w[in][c], w.ln1b[in][c]), 0.f);
            h[j] = f16_hi(v);
            l[j] = f16_los(v);
          }
          *reinterpret_cast<half8*>(Ah + r * PN_AS + c0 + 8 * i) = h;
          *reinterpret_cast<half8*>(Al + r * PN_AS + c0 + 8 * i) = l;
        }
      }
      __syncthreads();
      pn_gemm<4>(Ah, Al, 4, w.F2[in], C, PN_CS, PN_ROWS, wave, lane);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 32; ++i) Y[r * 128 + c0 + i] += C[r * PN_CS + c0 + i];
      // (the next writes to the planes / C come after this thread's own reads; other threads' rows are disjoint,
      //  and the GEMM that reads the planes again sits behind the barrier after the feature stage)
    }
    {   // to_out: LayerNorm, ReLU on the summed embeddings -> planes
      float a[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) a[i] = Y[r * 128 + c0 + i] + w.b2sum[c0 + i];
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
        half8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = c0 + 8 * i + j;
          const float v = fmaxf(fmaf(a[8 * i + j] * rstd, w.lnow[c], w.lnob[c]), 0.f);
          h[j] = f16_hi(v);
          l[j] = f16_los(v);
        }
        *reinterpret_cast<half8*>(Ah + r * PN_AS + c0 + 8 * i) = h;
        *reinterpret_cast<half8*>(Al + r * PN_AS + c0 + 8 * i) = l;
      }
    }
    __syncthreads();
    pn_gemm<4>(Ah, Al, 4, w.Fo, C, PN_CS, PN_ROWS, wave, lane);
    __syncthreads();   // (every wave is done reading the planes: buf may overwrite them)
    {   // + bias, the affine-free LayerNorm of attn_prenorm_r, split fp16 rows
      float a[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) a[i] = C[r * PN_CS + c0 + i] + w.bo[c0 + i];
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
      const float rstd = xs[4 * r + 3] != 0.f ? 1.f / sqrtf(sq * (1.f / 128.f) + eps) : 0.f;   // padding rows of a tile are zero
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        half8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = a[8 * i + j] * rstd;
          h[j] = f16_hi(v);
          l[j] = f16_lo(v);
        }
        *reinterpret_cast<half8*>(&buf[r][c0 + 8 * i]) = h;
        *reinterpret_cast<half8*>(&buf[r][128 + c0 + 8 * i]) = l;
      }
    }
    __syncthreads();
    for (int t = 0; t < 2; ++t) {   // both operand images of the pair's tiles (layouts: k_relpe_tiles, all four column blocks)
      const int tile = 2 * pair + t;
      if (tile >= ntiles) break;
      for (int P = tid; P < 1024; P += 256) {
        const int m = P & 15, kq = (P >> 4) & 3, ks = (P >> 6) & 3, part_ = (P >> 8) & 1, sub = P >> 9;
        *reinterpret_cast<half8*>(rtA + (size_t)tile * 8192 + (size_t)P * 8) =
            *reinterpret_cast<const half8*>(&buf[t * 32 + sub * 16 + m][part_ * 128 + ks * 32 + kq * 8]);
      }
      const int part = tid >> 7, c = tid & 127;
      _Float16* o = rtT + (size_t)tile * 8192 + part * 4096 + c * 32;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = buf[t * 32 + 8 * g + j][part * 128 + c];
        *reinterpret_cast<half8*>(o + 8 * g) = v;
      }
    }
  }
}

}  // namespace ps
