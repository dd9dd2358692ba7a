// Round 6: does a transcendental VALU op (v_sin_f32 / v_cos_f32 / v_exp_f32: quarter rate, its own pipeline on gfx940+) still read its SOURCE
// register after the next instructions of the wave have issued?  k_edge_geo's LayerNorm statistics were found to differ run to run under load
// from other kernels, only ever in lanes 48-63 (the last quarter of a wave), with bit-identical inputs; the compiler's code overwrites the
// source registers of in-flight trans ops right behind them (v_pk_mul_f32 v[2:3] after v_cos_f32 v7, v3).  Probe waves run short hand-written
// sequences and compare with a safe form (result consumed before the source is touched); a second stream runs a load kernel of one kind.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/mb_transhaz.hip -o tools/mb/mb_transhaz && tools/mb/mb_transhaz
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NV = 8;   // probe variants

// safe forms: the result is read (v_mov: the hardware's own RAW interlock) before anything may touch the source
__device__ __forceinline__ float safe_sin(float x) {
  float r;
  asm volatile("v_sin_f32 %0, %1\n\ts_nop 7\n\tv_mov_b32 %0, %0\n\ts_nop 7" : "=&v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ float safe_cos(float x) {
  float r;
  asm volatile("v_cos_f32 %0, %1\n\ts_nop 7\n\tv_mov_b32 %0, %0\n\ts_nop 7" : "=&v"(r) : "v"(x));
  return r;
}

template <int V>
__global__ __launch_bounds__(256) void k_probe(int iters, unsigned long long* cnt) {
  const int lane = threadIdx.x & 63;
  unsigned bad[4] = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const float x0 = 0.001f * (float)(lane + 1) + 0.0371f * (float)(it & 15), x1 = 0.0017f * (float)(lane + 3) + 0.0213f * (float)(it & 31);
    const float junk = 0.25f + 0.125f * (float)(it & 3);
    const float e0 = safe_sin(x0), e1 = safe_cos(x0), e2 = safe_sin(x1), e3 = safe_cos(x1);
    float r0, r1, r2, r3;
    if (V == 0) {   // one trans op, its source overwritten by the NEXT instruction
      asm volatile("v_mov_b32 v20, %4\n\tv_mov_b32 v21, %5\n\ts_nop 4\n\t"
                   "v_sin_f32 v22, v20\n\tv_mov_b32 v20, %6\n\t"
                   "v_cos_f32 v23, v21\n\tv_mov_b32 v21, %6\n\t"
                   "s_nop 7\n\ts_nop 7\n\t"
                   "v_mov_b32 %0, v22\n\tv_mov_b32 %1, v23\n\tv_mov_b32 %2, v22\n\tv_mov_b32 %3, v23"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(x0), "v"(x1), "v"(junk) : "v20", "v21", "v22", "v23");
      bad[0] += r0 != e0; bad[1] += r1 != e3; bad[2] += r2 != e0; bad[3] += r3 != e3;
    } else if (V == 1) {   // four trans ops in a row, then a packed multiply over both sources (k_edge_geo's second group)
      asm volatile("v_mov_b32 v20, %4\n\tv_mov_b32 v21, %5\n\tv_mov_b32 v26, %6\n\tv_mov_b32 v27, %6\n\ts_nop 4\n\t"
                   "v_sin_f32 v22, v20\n\tv_cos_f32 v23, v20\n\tv_sin_f32 v24, v21\n\tv_cos_f32 v25, v21\n\t"
                   "v_pk_mul_f32 v[20:21], v[26:27], v[26:27]\n\t"
                   "s_nop 7\n\ts_nop 7\n\t"
                   "v_mov_b32 %0, v22\n\tv_mov_b32 %1, v23\n\tv_mov_b32 %2, v24\n\tv_mov_b32 %3, v25"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(x0), "v"(x1), "v"(junk) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
      bad[0] += r0 != e0; bad[1] += r1 != e1; bad[2] += r2 != e2; bad[3] += r3 != e3;
    } else if (V == 2) {   // k_edge_geo's first group: the add that consumes trans 1 and 2 overwrites the source of trans 4
      asm volatile("v_mov_b32 v20, %4\n\tv_mov_b32 v21, %5\n\tv_mov_b32 v26, %6\n\tv_mov_b32 v27, %6\n\ts_nop 4\n\t"
                   "v_sin_f32 v22, v20\n\tv_cos_f32 v23, v20\n\tv_sin_f32 v20, v21\n\tv_cos_f32 v24, v21\n\t"
                   "v_pk_fma_f32 v[28:29], v[26:27], v[26:27], v[26:27]\n\t"
                   "v_add_f32 v21, v22, v23\n\t"
                   "s_nop 7\n\ts_nop 7\n\t"
                   "v_mov_b32 %0, v22\n\tv_mov_b32 %1, v23\n\tv_mov_b32 %2, v20\n\tv_mov_b32 %3, v24"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(x0), "v"(x1), "v"(junk) : "v20", "v21", "v22", "v23", "v24", "v26", "v27", "v28", "v29");
      bad[0] += r0 != e0; bad[1] += r1 != e1; bad[2] += r2 != e2; bad[3] += r3 != e3;
    } else if (V == 3 || V == 4) {   // RAW: a VALU op reads a trans result 0 (V = 3) or 1 (V = 4) wait states behind it
      if (V == 3)
        asm volatile("v_mov_b32 v20, %2\n\tv_mov_b32 v21, %3\n\ts_nop 4\n\t"
                     "v_sin_f32 v22, v20\n\tv_add_f32 %0, v22, v22\n\t"
                     "v_cos_f32 v23, v21\n\tv_add_f32 %1, v23, v23\n\ts_nop 7"
                     : "=&v"(r0), "=&v"(r1) : "v"(x0), "v"(x1) : "v20", "v21", "v22", "v23");
      else
        asm volatile("v_mov_b32 v20, %2\n\tv_mov_b32 v21, %3\n\ts_nop 4\n\t"
                     "v_sin_f32 v22, v20\n\ts_nop 0\n\tv_add_f32 %0, v22, v22\n\t"
                     "v_cos_f32 v23, v21\n\ts_nop 0\n\tv_add_f32 %1, v23, v23\n\ts_nop 7"
                     : "=&v"(r0), "=&v"(r1) : "v"(x0), "v"(x1) : "v20", "v21", "v22", "v23");
      bad[0] += r0 != e0 + e0; bad[1] += r1 != e3 + e3;
    } else if (V == 5) {   // WAW: a VALU op overwrites a trans op's DESTINATION right behind it; the trans result must lose
      asm volatile("v_mov_b32 v20, %2\n\tv_mov_b32 v21, %3\n\ts_nop 4\n\t"
                   "v_sin_f32 v22, v20\n\tv_mov_b32 v22, %4\n\t"
                   "v_cos_f32 v23, v21\n\tv_mov_b32 v23, %4\n\t"
                   "s_nop 7\n\ts_nop 7\n\tv_mov_b32 %0, v22\n\tv_mov_b32 %1, v23"
                   : "=&v"(r0), "=&v"(r1) : "v"(x0), "v"(x1), "v"(junk) : "v20", "v21", "v22", "v23");
      bad[0] += r0 != junk; bad[1] += r1 != junk;
    } else if (V == 6) {   // as V = 1 with the overwrite 4 independent VALU instructions behind the last trans op
      asm volatile("v_mov_b32 v20, %4\n\tv_mov_b32 v21, %5\n\tv_mov_b32 v26, %6\n\tv_mov_b32 v27, %6\n\ts_nop 4\n\t"
                   "v_sin_f32 v22, v20\n\tv_cos_f32 v23, v20\n\tv_sin_f32 v24, v21\n\tv_cos_f32 v25, v21\n\t"
                   "v_mul_f32 v28, v26, v26\n\tv_mul_f32 v29, v26, v27\n\tv_mul_f32 v28, v27, v26\n\tv_mul_f32 v29, v27, v27\n\t"
                   "v_pk_mul_f32 v[20:21], v[26:27], v[26:27]\n\t"
                   "s_nop 7\n\ts_nop 7\n\t"
                   "v_mov_b32 %0, v22\n\tv_mov_b32 %1, v23\n\tv_mov_b32 %2, v24\n\tv_mov_b32 %3, v25"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(x0), "v"(x1), "v"(junk) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29");
      bad[0] += r0 != e0; bad[1] += r1 != e1; bad[2] += r2 != e2; bad[3] += r3 != e3;
    } else {   // V == 7: as V = 1 with the RESULTS read (hardware interlock) before the overwrite: the cure under test
      asm volatile("v_mov_b32 v20, %4\n\tv_mov_b32 v21, %5\n\tv_mov_b32 v26, %6\n\tv_mov_b32 v27, %6\n\ts_nop 4\n\t"
                   "v_sin_f32 v22, v20\n\tv_cos_f32 v23, v20\n\tv_sin_f32 v24, v21\n\tv_cos_f32 v25, v21\n\t"
                   "s_nop 0\n\tv_add_f32 v28, v24, v25\n\t"
                   "v_pk_mul_f32 v[20:21], v[26:27], v[26:27]\n\t"
                   "s_nop 7\n\ts_nop 7\n\t"
                   "v_mov_b32 %0, v22\n\tv_mov_b32 %1, v23\n\tv_mov_b32 %2, v24\n\tv_mov_b32 %3, v25"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(x0), "v"(x1), "v"(junk) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28");
      bad[0] += r0 != e0; bad[1] += r1 != e1; bad[2] += r2 != e2; bad[3] += r3 != e3;
    }
  }
  // [variant][result 0-3][lane quarter]
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (bad[r]) atomicAdd(cnt + (V * 4 + r) * 4 + (lane >> 4), (unsigned long long)bad[r]);
}

// load kernels: kind 0 trans ops, 1 plain fp32 fma, 2 global-memory streaming, 3 MFMA
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_load(int kind, int iters, const float* buf, size_t n, float* sink) {
  float a = 0.001f * threadIdx.x, b = 0.5f, c = 0.25f, d = 0.125f;
  if (kind == 0) {
    for (int i = 0; i < iters; ++i) {
      a = __builtin_amdgcn_sinf(a) + 0.1f; b = __builtin_amdgcn_exp2f(-b) + 0.2f; c = __builtin_amdgcn_cosf(c) + 0.3f; d = __builtin_amdgcn_rcpf(d + 1.f);
    }
  } else if (kind == 1) {
    for (int i = 0; i < iters; ++i) {
      a = fmaf(a, 0.999f, 0.01f); b = fmaf(b, 0.998f, 0.02f); c = fmaf(c, 0.997f, 0.03f); d = fmaf(d, 0.996f, 0.04f);
    }
  } else if (kind == 2) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (int k = 0; k < iters; ++k) {
      a += buf[i % n];
      i += (size_t)gridDim.x * 256 * 17;
    }
  } else {
    half8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(0.01f * (threadIdx.x + j)); y[j] = (_Float16)(0.02f * j); }
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc, 0, 0, 0);
    a = acc[0] + acc[1] + acc[2] + acc[3];
  }
  if (a + b + c + d == 12345.678f) sink[0] = a;
}

template <int V>
void run_probe(hipStream_t st, int iters, unsigned long long* cnt) { hipLaunchKernelGGL(k_probe<V>, dim3(512), dim3(256), 0, st, iters, cnt); }

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  hipStream_t sp, sl;
  CHK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
  CHK(hipStreamCreateWithFlags(&sl, hipStreamNonBlocking));
  unsigned long long* cnt;
  CHK(hipMalloc(&cnt, NV * 16 * sizeof(unsigned long long)));
  float *buf, *sink;
  const size_t n = (size_t)256 << 20;   // 1 GB of floats: beyond every cache
  CHK(hipMalloc(&buf, n * 4));
  CHK(hipMemset(buf, 0, n * 4));
  CHK(hipMalloc(&sink, 256));
  const char* lname[5] = {"none", "trans", "fma", "memory", "mfma"};
  const char* vname[NV] = {"WAR 1 trans, next instr", "WAR 4 trans + v_pk_mul", "WAR k_edge_geo group 1", "RAW 0 wait states", "RAW 1 wait state",
                           "WAW next instr", "WAR 4 trans, 4 instrs later", "WAR 4 trans, results read first"};
  for (int L = 0; L < 5; ++L) {
    CHK(hipMemset(cnt, 0, NV * 16 * sizeof(unsigned long long)));
    CHK(hipDeviceSynchronize());
    const int liters[5] = {0, 3000000, 12000000, 60000, 3000000};
    for (int rep = 0; rep < 2; ++rep) {
      if (L) hipLaunchKernelGGL(k_load, dim3(1024), dim3(256), 0, sl, L - 1, liters[L], (const float*)buf, n, sink);
      run_probe<0>(sp, iters, cnt); run_probe<1>(sp, iters, cnt); run_probe<2>(sp, iters, cnt); run_probe<3>(sp, iters, cnt);
      run_probe<4>(sp, iters, cnt); run_probe<5>(sp, iters, cnt); run_probe<6>(sp, iters, cnt); run_probe<7>(sp, iters, cnt);
      hipEvent_t ev;
      CHK(hipEventCreate(&ev));
      CHK(hipEventRecord(ev, sp));
      CHK(hipEventSynchronize(ev));
      const bool load_still_running = L && hipStreamQuery(sl) == hipErrorNotReady;
      CHK(hipDeviceSynchronize());
      if (rep == 1) printf("load %-7s (still running when the probes ended: %s)\n", lname[L], L ? (load_still_running ? "yes" : "NO") : "-");
      CHK(hipEventDestroy(ev));
    }
    std::vector<unsigned long long> h(NV * 16);
    CHK(hipMemcpy(h.data(), cnt, NV * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int v = 0; v < NV; ++v) {
      printf("  V%d %-34s mismatches by result x lane quarter:", v, vname[v]);
      for (int r = 0; r < 4; ++r) printf("  [%llu %llu %llu %llu]", h[(v * 4 + r) * 4], h[(v * 4 + r) * 4 + 1], h[(v * 4 + r) * 4 + 2], h[(v * 4 + r) * 4 + 3]);
      printf("\n");
    }
    fflush(stdout);
  }
  return 0;
}
