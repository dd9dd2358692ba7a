// Micro-benchmark of the edge-phase load patterns of k_attn_chain on gfx950 (not part of the library).
//   mb_gather <pattern> <table MB> <workgroups> <loads per batch> <iters> [mfma]
// pattern 0: each batch = NL contiguous 1 KB wave-loads (16 B per lane) at a random 16 KB-aligned offset
// pattern 1: row gather as pass 1 reads k rows: lane (m = l & 15, kq = l >> 4) -> 16 random 512 B rows,
//            per instruction 16 x 64 B segments
// pattern 2: like 1 but rows are 16 CONSECUTIVE rows at a random base (rel-PE rows of one destination)
// pattern 3: like 1 with the quad-contiguous lane mapping (lane = 4*row + piece): 16 random rows x 64 B per instruction
// Reports cycles per batch (wave 0 of each WG, averaged) and aggregate GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const half8 g_chalf8;
__device__ __forceinline__ half8 ldgh8(const _Float16* p) { return *(g_chalf8*)p; }

template <int NL, int PAT, bool MFMA>
__global__ __launch_bounds__(256, 2) void k_mb(const _Float16* __restrict__ tab, unsigned nrows, int iters,
                                               unsigned long long* __restrict__ cyc, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, kq = lane >> 4;
  unsigned rng = (blockIdx.x * 4 + wave) * 2654435761u + 12345u;
  floatx4 acc = {0.f, 0.f, 0.f, 0.f};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    half8 v[NL];
    if (PAT == 0) {
      rng = rng * 1664525u + 1013904223u;
      const size_t base = (size_t)((rng >> 4) % (nrows / 32)) * 32 * 256;   // 32 rows = 16 KB
#pragma unroll
      for (int j = 0; j < NL; ++j) v[j] = ldgh8(tab + base + (size_t)j * 512 + lane * 8);
    } else {
      const _Float16* p[NL / 8];
#pragma unroll
      for (int q = 0; q < NL / 8; ++q) {
        unsigned row;
        if (PAT == 1 || PAT == 3) {
          unsigned r = rng + (PAT == 3 ? (lane >> 2) : m) * 40503u + (it * 4 + q) * 2246822519u;
          r ^= r >> 15; r *= 2654435761u; r ^= r >> 13;
          row = r % nrows;
        } else {
          rng = rng * 1664525u + 1013904223u;
          row = ((rng >> 4) % (nrows / 16)) * 16 + m;
        }
        p[q] = tab + (size_t)row * 256 + 8 * (PAT == 3 ? (lane & 3) : kq);
      }
#pragma unroll
      for (int j = 0; j < NL; ++j) v[j] = ldgh8(p[j >> 3] + 32 * (j & 7));   // 8 x 64 B segments per row
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MFMA) {
#pragma unroll
      for (int j = 0; j < NL; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(v[j], v[j], acc, 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < NL; ++j) acc[0] += (float)v[j][0];
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) atomicAdd(cyc, (unsigned long long)(t1 - t0));
  if (acc[0] + acc[1] == 12345.678f) out[0] = acc[2];
}

template <int NL>
void run(int pat, const _Float16* tab, unsigned nrows, int wgs, int iters, bool mfma, unsigned long long* cyc, float* out, hipStream_t s) {
#define L(P, M) hipLaunchKernelGGL((k_mb<NL, P, M>), dim3(wgs), dim3(256), 0, s, tab, nrows, iters, cyc, out)
  if (pat == 0) { if (mfma) L(0, true); else L(0, false); }
  else if (pat == 1) { if (mfma) L(1, true); else L(1, false); }
  else if (pat == 3) { if (mfma) L(3, true); else L(3, false); }
  else { if (mfma) L(2, true); else L(2, false); }
#undef L
}

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: mb_gather pattern tableMB workgroups loads iters [mfma]\n"); return 2; }
  const int pat = atoi(argv[1]), mb = atoi(argv[2]), wgs = atoi(argv[3]), nl = atoi(argv[4]), iters = atoi(argv[5]);
  const bool mfma = argc > 6 && atoi(argv[6]);
  const size_t bytes = (size_t)mb << 20;
  const unsigned nrows = (unsigned)(bytes / 512);
  _Float16* tab; unsigned long long* cyc; float* out;
  hipMalloc(&tab, bytes); hipMalloc(&cyc, 8); hipMalloc(&out, 64);
  hipMemset(tab, 0, bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e30f; unsigned long long hc = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipMemset(cyc, 0, 8);
    hipEventRecord(a, 0);
    if (nl == 8) run<8>(pat, tab, nrows, wgs, iters, mfma, cyc, out, 0);
    else if (nl == 16) run<16>(pat, tab, nrows, wgs, iters, mfma, cyc, out, 0);
    else run<32>(pat, tab, nrows, wgs, iters, mfma, cyc, out, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) { best = ms; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost); }
  }
  const double tot = (double)wgs * 4 * iters * nl * 1024.0;
  printf("pat %d table %4d MB wgs %4d NL %2d mfma %d: %8.1f us, %7.2f TB/s total, %6.1f GB/s per CU-slot(256), %8.0f cycles per batch\n", pat, mb, wgs, nl,
         (int)mfma, best * 1e3, tot / best / 1e9, tot / best / 1e6 / 256.0, (double)hc / wgs / iters);
  return 0;
}
