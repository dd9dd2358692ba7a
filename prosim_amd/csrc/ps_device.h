// Device-side building blocks shared by the kernels of libprosim_hip (gfx950 only, wave = 64).
// Compiled with -ffp-contract=off: every FMA is an explicit fmaf, so the neighbour-search
// distances (which must reproduce the oracle's fp32 rounding exactly) stay unfused.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace ps {

constexpr int D = 128;   // MODEL.HIDDEN_DIM
constexpr int H = 8;     // *.ATTN.NUM_HEAD
constexpr int DH = 16;   // *.ATTN.FF_DIM (= head_dim, attention_layer.py:26)
constexpr int FF = 512;  // hard-wired 4*D (attention_layer.py:39)
constexpr int QP = 132;  // padded row stride of the per-head [8][128] LDS images
constexpr int WG = 256;  // threads per workgroup of the row kernels (4 waves, one per SIMD)

#define PS_PI_F 3.14159274101257324f      // float(math.pi)
#define PS_TWO_PI_F 6.28318548202514648f  // float(2*math.pi)

// Pointers that reach a kernel inside a struct (weight tables, chain steps) are GENERIC to the
// compiler, which then emits flat_load: flat loads count on lgkmcnt as well as vmcnt, so every LDS
// wait would also drain the weight prefetch.  These helpers pin the global address space.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const v4f g_cv4f;
typedef __attribute__((address_space(1))) const v2f g_cv2f;
typedef __attribute__((address_space(1))) const float g_cf1;
typedef __attribute__((address_space(1))) const int g_ci1;
__device__ __forceinline__ float4 ldg4(const float* p) {
  const v4f v = *(g_cv4f*)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float2 ldg2(const float* p) {
  const v2f v = *(g_cv2f*)p;
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ float ldg1(const float* p) { return *(g_cf1*)p; }
__device__ __forceinline__ int ldgi(const int* p) { return *(g_ci1*)p; }

// models/utils/geometry.py:13-17 -- torch '%' is floor-mod: fmod, then shift into [0, 2pi).
__device__ __forceinline__ float wrap_angle(float a) {
  float t = a + PS_PI_F;
  float m = fmodf(t, PS_TWO_PI_F);
  if (m != 0.f && m < 0.f) m += PS_TWO_PI_F;
  return -PS_PI_F + m;
}

// ---- split-fp16 operands for the matrix cores.  x ~= hi + lo with hi = fp16(x), lo = fp16(x - hi):
// 22 mantissa bits survive, so hi*hi + hi*lo + lo*hi + lo*lo accumulated in fp32 by
// v_mfma_f32_16x16x32_f16 is fp32-class (error ~2^-21 per product) -- unlike a bf16 product (2^-9),
// which the 1e-4 closed-loop bar cannot absorb.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const half8 g_chalf8;
__device__ __forceinline__ half8 ldgh8(const _Float16* p) { return *(g_chalf8*)p; }
__device__ __forceinline__ _Float16 f16_hi(float x) { return (_Float16)x; }
__device__ __forceinline__ _Float16 f16_lo(float x) { return (_Float16)(x - (float)(_Float16)x); }
// GEMM operands (round 4): the lo half SCALED by 2^11.  A weight of magnitude 0.05 has hi = fp16(w) with an ulp of 3e-5, so its lo part
// is below 1.5e-5 -- a SUBNORMAL fp16 (step 6e-8): unscaled, the pair carries 20 - 21 bits, not 22 - 23, and that was most of the
// engine's noise against the fp64 oracle (map tokens 4.4e-6 rms; torch fp32: 1.6e-6; with scaled lo halves: 1.3e-6,
// profiles/r04_token_error.txt).  Scaled, lo is a normal fp16 for every |x| > 6e-8.  Every weight fragment the host packs
// (Builder::fragments) and every activation plane a GEMM reads carries the scaled lo; the two cross products accumulate in a tile
// of their own that joins the hi.hi tile as acc + 2^-11 x.  (Edge-phase operands -- k rows, rel-PE rows, probabilities -- keep f16_lo:
// they are O(1) activations whose four products share one accumulator.)
constexpr float PS_LO_SCALE = 2048.f, PS_LO_INV = 1.f / 2048.f;
__device__ __forceinline__ _Float16 f16_los(float x) { return (_Float16)((x - (float)(_Float16)x) * PS_LO_SCALE); }

// ---- cross-lane exchange without the LDS pipe (gfx950): v_permlane32_swap exchanges half-waves between TWO
// registers, DPP covers xor 8 / 2 / 1 inside a row of 16.
// swap_add32(lo, hi): lanes 0-31 get lo[l] + lo[l+32], lanes 32-63 get hi[l-32] + hi[l] -- one
// step of a transposing reduction (the lane's bit 5 selects which of the two values it keeps).
__device__ __forceinline__ float swap_add32(float lo, float hi) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// all-reduce over the four lanes l, l ^ 16, l ^ 32, l ^ 48 on the VALU (v_permlane16_swap / v_permlane32_swap of gfx950)
// instead of two ds_bpermute round trips through the LDS crossbar.  permlane16_swap(a, b): rows 1 / 3 of a trade places
// with rows 0 / 2 of b, so with a = b = v the two results hold (row 0, row 0, row 2, row 2) and (row 1, row 1, row 3, row 3).
__device__ __forceinline__ float kq_max(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
// max(m, v over the four kq lanes).  Round 6: plain fmaxf on the permlane results -- every instruction visible to hipcc's hazard recogniser.
// Rounds 4 - 5 wrote the two maxima as inline asm (v_max_f32 / v_max3_f32: hipcc canonicalises both halves of a permlane result before an
// fmaxf, two v_max_f32 x, x per step); an inline-asm VALU write feeding v_permlane32_swap is exactly the producer the recogniser cannot see
// (gfx950 wants two wait states there), and the builtin form returns the same bits at the same launch time (measured in round 5 and again in
// round 6: profiles/r06_n_digest.txt) -- so the form that cannot be wrong stays.
__device__ __forceinline__ float kq_max3(float v, float m) { return fmaxf(m, kq_max(v)); }
__device__ __forceinline__ float kq_sum(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float m = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float dpp_xor8(float v) {   // row_ror:8
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {   // quad_perm [2,3,0,1]
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_xor1(float v) {   // quad_perm [1,0,3,2]
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
}
// ---- xor butterflies on the VALU (round 5).  __shfl_xor compiles to ds_bpermute_b32: an LDS-pipe round trip of ~100 cycles per step,
// six dependent steps per wave reduction -- the LayerNorms, GEMV folds and softmax reductions of a k_attn_chain layer are ~250 of them
// (13 % of a single-scene policy launch was waiting for them).  On gfx950 every step of an xor butterfly exists as a VALU instruction:
// lane ^ 32 / ^ 16 through v_permlane32_swap / v_permlane16_swap, ^ 8 = row_ror:8, ^ 2 / ^ 1 quad permutes; ^ 4 is row_ror:4 once the
// data has period 8 inside a row (after the ^ 8 step of a descending butterfly), two rotations and a select otherwise.  a + partner is
// commutative, so every form below returns the bits of `v + __shfl_xor(v, O)` / fmaxf(v, __shfl_xor(v, O)).
__device__ __forceinline__ float dpp_ror4(float v) {   // row_ror:4
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_ror12(float v) {   // row_ror:12
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x12C, 0xf, 0xf, true));
}
template <int O>
__device__ __forceinline__ float xor_add(float v) {   // v + (the value of lane ^ O)
  static_assert(O == 32 || O == 16 || O == 8 || O == 4 || O == 2 || O == 1, "xor_add: one bit");
  if (O == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else if (O == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else if (O == 8) {
    return v + dpp_xor8(v);
  } else if (O == 4) {   // (general form: lanes with bit 2 clear take lane + 4, the others lane - 4)
    // row_ror:n hands lane i the value of lane (i - n) mod 16 (probed on the GPU): lanes 4-7 / 12-15 (banks 1 and 3 of the row) take
    // ror 4 = lane - 4, the others keep ror 12 = lane + 4 -- the second mov writes only those banks over the first one's result.  (As
    // a select of two full DPP moves hipcc folded the pair into one masked mov whose other lanes read 0.)
    const int up = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x12C, 0xf, 0xf, true);
    return v + __int_as_float(__builtin_amdgcn_update_dpp(up, __float_as_int(v), 0x124, 0xf, 0xa, false));
  } else if (O == 2) {
    return v + dpp_xor2(v);
  } else {
    return v + dpp_xor1(v);
  }
}
template <int O>
__device__ __forceinline__ float xor_max(float v) {   // fmaxf(v, the value of lane ^ O), O = 16 | 32
  static_assert(O == 32 || O == 16, "xor_max: 16 | 32");
  const auto r = O == 32 ? __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false)
                         : __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// the same step once the data already has period 8 inside each row of 16 (lanes i and i ^ 8 hold the same value): one rotation
__device__ __forceinline__ float xor4_add_periodic8(float v) { return v + dpp_ror4(v); }
__device__ __forceinline__ float wave_sum(float v) {   // butterfly 32, 16, 8, 4, 2, 1: the order (and the bits) of the __shfl_xor loop it replaces
  v = xor_add<32>(v);
  v = xor_add<16>(v);
  v = xor_add<8>(v);
  v = xor4_add_periodic8(v);
  v = xor_add<2>(v);
  v = xor_add<1>(v);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  v = fmaxf(v, dpp_xor8(v));
  v = fmaxf(v, dpp_ror4(v));
  v = fmaxf(v, dpp_xor2(v));
  v = fmaxf(v, dpp_xor1(v));
  return v;
}

// LayerNorm of one 128-float row by one wave: lane holds elements lane and lane+64.
// y = (x - mean) * rstd * w + b, biased variance, eps inside the sqrt (torch.nn.LayerNorm).
__device__ __forceinline__ void ln_row_wave(const float* __restrict__ x, float* __restrict__ y,
                                            const float* __restrict__ w, const float* __restrict__ b,
                                            float eps, int lane, bool relu) {
  float a0 = x[lane], a1 = x[lane + 64];
  float mean = wave_sum(a0 + a1) * (1.f / 128.f);
  float d0 = a0 - mean, d1 = a1 - mean;
  float var = wave_sum(d0 * d0 + d1 * d1) * (1.f / 128.f);
  float rstd = 1.f / sqrtf(var + eps);
  float y0 = d0 * rstd, y1 = d1 * rstd;
  if (w) {
    y0 = fmaf(y0, w[lane], b[lane]);
    y1 = fmaf(y1, w[lane + 64], b[lane + 64]);
  }
  if (relu) {
    y0 = fmaxf(y0, 0.f);
    y1 = fmaxf(y1, 0.f);
  }
  y[lane] = y0;
  y[lane + 64] = y1;
}

// Workgroup b runs on XCD b % 8 (observed placement, used for speed only: MI355X_MICROARCH.md "Workgroup dispatch").
// Destination rows are ordered by scene, and the rows of one scene gather the k|v rows of the SAME source tokens, so
// consecutive LOGICAL blocks should share an XCD (one scene's k|v = 1.5-3 MB per layer sits in that XCD's 4 MB L2
// instead of every L2 seeing every scene).  Physical block b -> logical block: XCD x takes the x-th contiguous
// eighth of the grid; a tail of G % 8 blocks keeps its index.
__device__ __forceinline__ int xcd_block(int b, int G, int off) {
  const int per = G >> 3;
  if (off || b >= (per << 3)) return b;
  return (b & 7) * per + (b >> 3);
}

// Small-N GEMV (N not a multiple of 128): one thread per output, torch layout W[N][K].
template <int T>
__device__ __forceinline__ void gemv_small(const float* x, int xs, int K, const float* __restrict__ W, int N,
                                           const float* __restrict__ bias, float* out, int os, bool relu) {
  for (int idx = threadIdx.x; idx < T * N; idx += blockDim.x) {
    const int t = idx / N, n = idx - t * N;
    float s = bias ? bias[n] : 0.f;
    const float* wr = W + (size_t)n * K;
    for (int k = 0; k < K; ++k) s = fmaf(x[t * xs + k], wr[k], s);
    if (relu) s = fmaxf(s, 0.f);
    out[t * os + n] = s;
  }
  __syncthreads();
}

// ---- 64-row GEMM on the matrix cores, shared by the PointNet encoder and the k|v projection.
// Rows live in LDS as split-fp16 planes Ah | Al (row stride PN_AS halfs: 16 B-aligned, staggered over the
// banks); weights stream from L2 as pre-split B fragments [n-tile][k-block K/32][hi|lo][lane 64][8] (lane =
// column n + 16*kq holds k = 32*ks + 8*kq ..+8), so a wave load is 1 KB contiguous; hi*hi + hi*lo + lo*hi
// accumulate in fp32 (the dropped lo*lo term is ~2^-22).  The fp32 result goes to LDS (row stride cs floats).
constexpr int PN_ROWS = 64, PN_AS = 136, PN_CS = 132;
// C[16*MT x 128] = A[16*MT x 32*k32] * W: wave w makes the 16-column tiles w and w + 4 (all 16 B-fragment loads
// of both tiles are in flight before the first MFMA).  Rows >= row_lim are not stored; ntiles < 8 for N < 128.
// The weight fragments of one GEMM for this wave (tiles w and w + 4): requested as a block so that a caller can
// start the NEXT GEMM's loads before the barrier / epilogue that separates it from the current one.
struct PnFrags {
  half8 bh[2][4], bl[2][4];
};
__device__ __forceinline__ void pn_load(PnFrags& f, const _Float16* __restrict__ F, int k32, int wave, int lane, int ntiles = 8) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const _Float16* p = F + (size_t)(wave + 4 * t) * k32 * 1024 + lane * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < k32 && wave + 4 * t < ntiles) {
        f.bh[t][ks] = ldgh8(p + ks * 1024);
        f.bl[t][ks] = ldgh8(p + ks * 1024 + 512);
      }
    }
  }
}
// (Both column tiles of the wave share one pass over the A fragments: an A fragment is read from LDS once and meets the
// B fragments of tile w and of tile w + 4 -- half the LDS reads of a tile-by-tile order.  Every accumulator still sees its
// products in the order ks ascending, hi*hi, hi*lo, lo*hi: the same bits.)
template <int MT>
__device__ __forceinline__ void pn_mma(const PnFrags& f, const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al, int k32,
                                       float* __restrict__ C, int cs, int row_lim, int wave, int lane, int ntiles = 8) {
  const int mi = lane & 15, kq = lane >> 4;
  const bool two = wave + 4 < ntiles;
  if (wave >= ntiles) return;
  floatx4 acc[2][MT], acx[2][MT];   // hi.hi | the two cross products (scaled lo halves)
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { acc[t][mt] = floatx4{0.f, 0.f, 0.f, 0.f}; acx[t][mt] = acc[t][mt]; }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    if (ks < k32) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const half8 ah = *reinterpret_cast<const half8*>(Ah + (mt * 16 + mi) * PN_AS + ks * 32 + kq * 8);
        const half8 al = *reinterpret_cast<const half8*>(Al + (mt * 16 + mi) * PN_AS + ks * 32 + kq * 8);
        acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, f.bh[0][ks], acc[0][mt], 0, 0, 0);
        acx[0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, f.bl[0][ks], acx[0][mt], 0, 0, 0);
        acx[0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, f.bh[0][ks], acx[0][mt], 0, 0, 0);
        if (two) {
          acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, f.bh[1][ks], acc[1][mt], 0, 0, 0);
          acx[1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, f.bl[1][ks], acx[1][mt], 0, 0, 0);
          acx[1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, f.bh[1][ks], acx[1][mt], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t == 1 && !two) break;
    const int nt = wave + 4 * t;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = mt * 16 + 4 * kq + r;
        if (row < row_lim) C[row * cs + nt * 16 + mi] = fmaf(acx[t][mt][r], PS_LO_INV, acc[t][mt][r]);
      }
  }
}
template <int MT>
__device__ __forceinline__ void pn_gemm(const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al, int k32,
                                        const _Float16* __restrict__ F, float* __restrict__ C, int cs, int row_lim, int wave,
                                        int lane, int ntiles = 8) {
  PnFrags f;
  pn_load(f, F, k32, wave, lane, ntiles);
  pn_mma<MT>(f, Ah, Al, k32, C, cs, row_lim, wave, lane, ntiles);
}

// LDS addresses as function arguments.  A __noinline__ phase function that is handed `extern __shared__` memory as a pointer gets the SYMBOL
// propagated into it by hipcc (every caller passes the same one), and the module-LDS lowering then turns each use into a look-up of the kernel's
// dynamic-LDS base in a table in MEMORY -- k_chain16's node phase did 14 of them per call, five as vector loads behind s_waitcnt vmcnt(0) in the
// middle of its GEMM stages.  So the kernels pass the 32-bit LDS address through an opaque register, and the callee rebuilds its pointers from it.
typedef __attribute__((address_space(3))) unsigned char lds_u8;
__device__ __forceinline__ unsigned lds_addr(const void* p) {   // p: a pointer into LDS
  unsigned o = (unsigned)reinterpret_cast<uintptr_t>((const lds_u8*)p);
  asm volatile("" : "+s"(o));
  return o;
}
template <class T>
__device__ __forceinline__ T* lds_ptr(unsigned o) {   // (the address-space inference follows the cast: accesses stay ds_ instructions)
  return (T*)reinterpret_cast<lds_u8*>((uintptr_t)o);
}
// a pointer known to be the same in every lane, as a scalar-register value (what comes out of a device-function argument or a struct in memory is a
// vector-register pair to the compiler: loads through it are flat loads with per-lane addresses)
// ... and a struct in memory that no launch writes, read through the scalar cache: a wave-uniform address in the constant address space makes every
// field access an s_load (lgkmcnt, ~200 cycles) instead of a flat load (vmcnt AND lgkmcnt: it waits behind every fragment load in flight)
template <class T>
__device__ __forceinline__ const __attribute__((address_space(4))) T* uni_const(const T* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const __attribute__((address_space(4))) T*)(((unsigned long long)hi << 32) | lo);
}
template <class T>
__device__ __forceinline__ T* uni_ptr(T* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
}  // namespace ps
