// Round 6, third pass: every packed-fp32 form against a reference made of SCALAR instructions (v_fma_f32 / v_mul_f32 / v_add_f32 on the selected
// elements), with the packed instruction isolated by s_nop padding (no producer or consumer next to it) unless the form says otherwise.  Loads: none,
// another kernel's v_mfma_f32_16x16x32_f16.  Counts: low | high half result x lane quarter.  (generated: the python block in the round-6 session log)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/mb_pksgpr3.hip -o tools/mb/mb_pksgpr3 && tools/mb/mb_pksgpr3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int NV = 16;
template <int V>
__global__ __launch_bounds__(256) void k_probe(int iters, unsigned long long* cnt, const float* __restrict__ table) {
  const int lane = threadIdx.x & 63;
  const f32x2 sp = {table[2 * (blockIdx.x & 63)], table[2 * (blockIdx.x & 63) + 1]};
  unsigned bad_lo = 0, bad_hi = 0;
  for (int it = 0; it < iters; ++it) {
    f32x2 a = {0.001f * (float)(lane + 1) + 0.0371f * (float)(it & 15), 0.0017f * (float)(lane + 3) + 0.0213f * (float)(it & 31)};
    f32x2 c = {0.25f + 0.125f * (float)(it & 3), 0.5f - 0.01f * (float)(lane & 7)};
    asm volatile("" : "+v"(a), "+v"(c));
    f32x2 vp = sp;
    asm volatile("" : "+v"(vp));
    f32x2 t, e;
    if (V == 0) {
      asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\ts_nop 7" : "=&v"(t) : "v"(a), "v"(vp), "v"(c));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.x) : "v"(a.x), "v"(vp.y), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.y) : "v"(a.y), "v"(vp.x), "v"(c.y));
    }
    else if (V == 1) {
      f32x2 q;
      asm volatile("s_nop 7\n\ts_nop 7\n\tv_mov_b64 %1, %3\n\tv_pk_fma_f32 %0, %2, %1, %4 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\ts_nop 7" : "=&v"(t), "=&v"(q) : "v"(a), "v"(vp), "v"(c));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.x) : "v"(a.x), "v"(vp.y), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.y) : "v"(a.y), "v"(vp.x), "v"(c.y));
    }
    else if (V == 2) {
      asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]\n\ts_nop 7" : "=&v"(t) : "v"(a), "v"(vp), "v"(c));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.x) : "v"(a.x), "v"(vp.y), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.y) : "v"(a.y), "v"(vp.y), "v"(c.y));
    }
    else if (V == 3) {
      asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]\n\ts_nop 7" : "=&v"(t) : "v"(a), "v"(vp), "v"(c));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.x) : "v"(a.y), "v"(vp.x), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.y) : "v"(a.y), "v"(vp.y), "v"(c.y));
    }
    else if (V == 4) {
      asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]\n\ts_nop 7" : "=&v"(t) : "v"(a), "v"(vp), "v"(c));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.x) : "v"(a.x), "v"(vp.x), "v"(c.y));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.y) : "v"(a.y), "v"(vp.y), "v"(c.y));
    }
    else if (V == 5) {
      asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]\n\ts_nop 7" : "=&v"(t) : "v"(a), "s"(sp), "v"(c));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.x) : "v"(a.x), "v"(vp.x), "v"(c.y));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.y) : "v"(a.y), "v"(vp.y), "v"(c.y));
    }
    else if (V == 6) {
      asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]\n\ts_nop 7" : "=&v"(t) : "v"(a), "s"(sp), "v"(c));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.x) : "v"(a.y), "v"(vp.x), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.y) : "v"(a.y), "v"(vp.y), "v"(c.y));
    }
    else if (V == 7) {
      asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]\n\ts_nop 7" : "=&v"(t) : "v"(a), "s"(sp), "v"(c));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.x) : "v"(a.x), "v"(vp.y), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.y) : "v"(a.y), "v"(vp.y), "v"(c.y));
    }
    else if (V == 8) {
      asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_fma_f32 %0, %1, %2, %3\n\ts_nop 7" : "=&v"(t) : "v"(a), "s"(sp), "v"(c));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.x) : "v"(a.x), "v"(vp.x), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.y) : "v"(a.y), "v"(vp.y), "v"(c.y));
    }
    else if (V == 9) {
      asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_fma_f32 %0, %1, %2, %3\n\ts_nop 7" : "=&v"(t) : "v"(a), "v"(vp), "v"(c));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.x) : "v"(a.x), "v"(vp.x), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.y) : "v"(a.y), "v"(vp.y), "v"(c.y));
    }
    else if (V == 10) {
      asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]\n\ts_nop 7" : "=&v"(t) : "v"(a), "v"(vp), "v"(c));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.x) : "v"(a.x), "v"(vp.x), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.y) : "v"(a.y), "v"(vp.x), "v"(c.y));
    }
    else if (V == 11) {
      asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]\n\ts_nop 7" : "=&v"(t) : "v"(a), "s"(sp), "v"(c));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.x) : "v"(a.x), "v"(vp.x), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.y) : "v"(a.y), "v"(vp.y), "v"(c.x));
    }
    else if (V == 12) {
      asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1]\n\ts_nop 7" : "=&v"(t) : "v"(a), "v"(vp));
      asm volatile("v_mul_f32 %0, %1, %2" : "=&v"(e.x) : "v"(a.x), "v"(vp.y));
      asm volatile("v_mul_f32 %0, %1, %2" : "=&v"(e.y) : "v"(a.y), "v"(vp.y));
    }
    else if (V == 13) {
      asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_add_f32 %0, %1, %2 op_sel:[0,1]\n\ts_nop 7" : "=&v"(t) : "v"(a), "v"(vp));
      asm volatile("v_add_f32 %0, %1, %2" : "=&v"(e.x) : "v"(a.x), "v"(vp.y));
      asm volatile("v_add_f32 %0, %1, %2" : "=&v"(e.y) : "v"(a.y), "v"(vp.y));
    }
    else if (V == 14) {
      asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[1,0]\n\ts_nop 7" : "=&v"(t) : "v"(a), "s"(sp));
      asm volatile("v_mul_f32 %0, %1, %2" : "=&v"(e.x) : "v"(a.y), "v"(vp.x));
      asm volatile("v_mul_f32 %0, %1, %2" : "=&v"(e.y) : "v"(a.y), "v"(vp.y));
    }
    else if (V == 15) {
      f32x2 q = vp;
      asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_fma_f32 %0, %3, %5, %4 op_sel:[0,1,0]\n\tv_pk_mul_f32 %2, %2, %2\n\ts_nop 7" : "=&v"(t), "+v"(q) : "v"(a), "v"(c), "v"(vp));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.x) : "v"(a.x), "v"(vp.y), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.y) : "v"(a.y), "v"(vp.y), "v"(c.y));
    }
    bad_lo += __float_as_uint(t.x) != __float_as_uint(e.x);
    bad_hi += __float_as_uint(t.y) != __float_as_uint(e.y);
  }
  if (bad_lo) atomicAdd(cnt + V * 8 + (lane >> 4), (unsigned long long)bad_lo);
  if (bad_hi) atomicAdd(cnt + V * 8 + 4 + (lane >> 4), (unsigned long long)bad_hi);
}
// load kernels: 1 VALU with scalar sources (other values), 2 VALU on vector registers only, 3 LDS traffic, 4 global loads, 5 scalar ALU + scalar loads,
// 6 MFMA, 7 transcendental ops, 8 packed instructions with scalar pairs (other values)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_load(int kind, int iters, const float* buf, size_t n, float* sink, const float* __restrict__ table) {
  __shared__ float lds[4096];
  float a = 0.001f * threadIdx.x, b = 0.5f, c = 0.25f, d = 0.125f;
  const float s0 = table[128 + (blockIdx.x & 31)], s1 = table[160 + (blockIdx.x & 31)];   // wave-uniform, NOT the probes' values
  if (kind == 1) {
    for (int i = 0; i < iters; ++i) {
      asm volatile("v_fma_f32 %0, %0, %4, %1\n\tv_fma_f32 %1, %1, %5, %2\n\tv_fma_f32 %2, %2, %4, %4\n\tv_fma_f32 %3, %3, %5, %5"
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s0), "s"(s1));
    }
  } else if (kind == 2) {
    for (int i = 0; i < iters; ++i) {
      a = fmaf(a, 0.999f, 0.01f); b = fmaf(b, 0.998f, 0.02f); c = fmaf(c, 0.997f, 0.03f); d = fmaf(d, 0.996f, 0.04f);
    }
  } else if (kind == 3) {
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = a + i;
    __syncthreads();
    int j = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
      a += lds[j & 4095];
      lds[(j + 1024) & 4095] = a;
      j += 257;
    }
  } else if (kind == 4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (int k = 0; k < iters; ++k) {
      a += buf[i % n];
      i += (size_t)gridDim.x * 256 * 17;
    }
  } else if (kind == 5) {
    unsigned x = blockIdx.x + 1, y = 12345u;
    for (int i = 0; i < iters; ++i) {
      asm volatile("s_mul_i32 %0, %0, 1664525\n\ts_add_u32 %0, %0, 1013904223\n\ts_xor_b32 %1, %1, %0\n\ts_lshr_b32 %1, %1, 1\n\ts_add_u32 %1, %1, %0" : "+s"(x), "+s"(y));
      if ((i & 63) == 0) a += table[(x >> 8) & 127];   // (a scalar load now and then)
    }
    a += (float)(x ^ y);
  } else if (kind == 6) {
    half8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(0.01f * (threadIdx.x + j)); y[j] = (_Float16)(0.02f * j); }
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc, 0, 0, 0);
    a = acc[0] + acc[1] + acc[2] + acc[3];
  } else if (kind == 7) {
    for (int i = 0; i < iters; ++i) {
      a = __builtin_amdgcn_sinf(a) + 0.1f; b = __builtin_amdgcn_exp2f(-b) + 0.2f; c = __builtin_amdgcn_cosf(c) + 0.3f; d = __builtin_amdgcn_rcpf(d + 1.f);
    }
  } else {
    const f32x2 sp = {s0, s1};
    f32x2 p = {a, b}, q = {c, d};
    for (int i = 0; i < iters; ++i) {
      asm volatile("v_pk_fma_f32 %0, %0, %2, %1\n\tv_pk_fma_f32 %1, %1, %2, %0" : "+v"(p), "+v"(q) : "s"(sp));
    }
    a = p.x; b = p.y; c = q.x; d = q.y;
  }
  if (a + b + c + d == 12345.678f) sink[0] = a;
}


template <int V>
void run_probe(hipStream_t st, int iters, unsigned long long* cnt, const float* table) { hipLaunchKernelGGL(k_probe<V>, dim3(512), dim3(256), 0, st, iters, cnt, table); }
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  hipStream_t sp, sl;
  CHK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
  CHK(hipStreamCreateWithFlags(&sl, hipStreamNonBlocking));
  unsigned long long* cnt;
  CHK(hipMalloc(&cnt, NV * 8 * sizeof(unsigned long long)));
  float *buf, *sink, *table;
  const size_t n = (size_t)16 << 20;
  CHK(hipMalloc(&buf, n * 4));
  CHK(hipMemset(buf, 0, n * 4));
  CHK(hipMalloc(&sink, 256));
  std::vector<float> ht(256);
  for (int i = 0; i < 256; ++i) ht[i] = 0.37f + 0.0131f * (float)i * (i & 1 ? 1.f : -0.5f);
  CHK(hipMalloc(&table, 1024));
  CHK(hipMemcpy(table, ht.data(), 1024, hipMemcpyHostToDevice));
  const char* lname[3] = {"none", "MFMA 16x16x32 f16", "VALU, vector registers only"};
  const char* vname[NV] = {"fma vector only, op_sel src1 + op_sel_hi src1 = 0 (halves swapped)",
                           "the same, the src1 pair written by v_mov_b64 right before",
                           "fma vector only, op_sel src1",
                           "fma vector only, op_sel src0",
                           "fma vector only, op_sel src2",
                           "fma src1 = s[n:n+1], op_sel src2 (k_edge_geo)",
                           "fma src1 = s[n:n+1], op_sel src0",
                           "fma src1 = s[n:n+1], op_sel src1",
                           "fma src1 = s[n:n+1], no modifier",
                           "fma vector only, no modifier",
                           "fma vector only, op_sel_hi src1 = 0",
                           "fma src1 = s[n:n+1], op_sel_hi src2 = 0 (k_edge_geo)",
                           "mul vector only, op_sel src1",
                           "add vector only, op_sel src1",
                           "mul src1 = s[n:n+1], op_sel src0",
                           "fma vector only, op_sel src1, operands consumed again right behind"};
  for (int L = 0; L < 3; ++L) {
    CHK(hipMemset(cnt, 0, NV * 8 * sizeof(unsigned long long)));
    CHK(hipDeviceSynchronize());
    bool still = false;
    for (int rep = 0; rep < 2; ++rep) {
      if (L == 1) hipLaunchKernelGGL(k_load, dim3(1024), dim3(256), 0, sl, 6, 8000000, (const float*)buf, n, sink, (const float*)table);
      if (L == 2) hipLaunchKernelGGL(k_load, dim3(1024), dim3(256), 0, sl, 2, 30000000, (const float*)buf, n, sink, (const float*)table);
      run_probe<0>(sp, iters, cnt, table); run_probe<1>(sp, iters, cnt, table); run_probe<2>(sp, iters, cnt, table); run_probe<3>(sp, iters, cnt, table); run_probe<4>(sp, iters, cnt, table); run_probe<5>(sp, iters, cnt, table); run_probe<6>(sp, iters, cnt, table); run_probe<7>(sp, iters, cnt, table); run_probe<8>(sp, iters, cnt, table); run_probe<9>(sp, iters, cnt, table); run_probe<10>(sp, iters, cnt, table); run_probe<11>(sp, iters, cnt, table); run_probe<12>(sp, iters, cnt, table); run_probe<13>(sp, iters, cnt, table); run_probe<14>(sp, iters, cnt, table); run_probe<15>(sp, iters, cnt, table);
      hipEvent_t ev;
      CHK(hipEventCreate(&ev));
      CHK(hipEventRecord(ev, sp));
      CHK(hipEventSynchronize(ev));
      still = L && hipStreamQuery(sl) == hipErrorNotReady;
      CHK(hipDeviceSynchronize());
      CHK(hipEventDestroy(ev));
    }
    printf("load: %s (still running when the probes ended: %s)\n", lname[L], L ? (still ? "yes" : "NO") : "-");
    std::vector<unsigned long long> h(NV * 8);
    CHK(hipMemcpy(h.data(), cnt, NV * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int v = 0; v < NV; ++v)
      printf("  X%-2d %-70s low [%llu %llu %llu %llu]  high [%llu %llu %llu %llu]\n", v, vname[v], h[v * 8], h[v * 8 + 1], h[v * 8 + 2], h[v * 8 + 3], h[v * 8 + 4],
             h[v * 8 + 5], h[v * 8 + 6], h[v * 8 + 7]);
    fflush(stdout);
  }
  return 0;
}
