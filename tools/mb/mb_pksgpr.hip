// Round 6: a packed-fp32 VALU instruction that takes a SCALAR REGISTER PAIR as a source (v_pk_fma_f32 v[0:1], v[2:3], s[12:13], v[4:5]: the
// high half multiplies by s13) -- does the read of the pair's second register hold when waves of ANOTHER kernel share the SIMD?
// Found through k_edge_geo: its LayerNorm statistics differed run to run under load from other engines, only in
// lanes 48-63, only in the packed form, only with the divisor pairs in scalar registers (hipcc puts wave-uniform values there on its own).
// Probe waves compare the instruction with a scalar-pair source against the same instruction on a vector-register copy of the pair; a second
// stream runs one kind of load.  Counts: [low half | high half] x lane quarter.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/mb_pksgpr.hip -o tools/mb/mb_pksgpr && tools/mb/mb_pksgpr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int NV = 10;

template <int V>
__global__ __launch_bounds__(256) void k_probe(int iters, unsigned long long* cnt, const float* __restrict__ table) {
  const int lane = threadIdx.x & 63;
  // a wave-uniform pair of two DIFFERENT values (scalar loads: the compiler keeps them in s[n:n+1])
  const f32x2 sp = {table[2 * (blockIdx.x & 63)], table[2 * (blockIdx.x & 63) + 1]};
  const double spd = (double)table[blockIdx.x & 63] * 1.000001;
  unsigned bad_lo = 0, bad_hi = 0;
  for (int it = 0; it < iters; ++it) {
    f32x2 a = {0.001f * (float)(lane + 1) + 0.0371f * (float)(it & 15), 0.0017f * (float)(lane + 3) + 0.0213f * (float)(it & 31)};
    f32x2 c = {0.25f + 0.125f * (float)(it & 3), 0.5f - 0.01f * (float)(lane & 7)};
    asm volatile("" : "+v"(a), "+v"(c));
    f32x2 vs = sp;
    asm volatile("" : "+v"(vs));   // the pair in VECTOR registers: the reference operand
    f32x2 t, e;
    if (V == 0) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=&v"(t) : "v"(a), "s"(sp), "v"(c));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=&v"(e) : "v"(a), "v"(vs), "v"(c));
    } else if (V == 1) {
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(t) : "v"(a), "s"(sp));
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(e) : "v"(a), "v"(vs));
    } else if (V == 2) {
      asm volatile("v_pk_add_f32 %0, %1, %2" : "=&v"(t) : "v"(a), "s"(sp));
      asm volatile("v_pk_add_f32 %0, %1, %2" : "=&v"(e) : "v"(a), "v"(vs));
    } else if (V == 3) {
      asm volatile("v_pk_fma_f32 %0, %2, %1, %3" : "=&v"(t) : "v"(a), "s"(sp), "v"(c));
      asm volatile("v_pk_fma_f32 %0, %2, %1, %3" : "=&v"(e) : "v"(a), "v"(vs), "v"(c));
    } else if (V == 4) {
      asm volatile("v_pk_fma_f32 %0, %1, %3, %2" : "=&v"(t) : "v"(a), "s"(sp), "v"(c));
      asm volatile("v_pk_fma_f32 %0, %1, %3, %2" : "=&v"(e) : "v"(a), "v"(vs), "v"(c));
    } else if (V == 5) {   // both halves from the pair's FIRST register
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=&v"(t) : "v"(a), "s"(sp), "v"(c));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=&v"(e) : "v"(a), "v"(vs), "v"(c));
    } else if (V == 6) {   // halves swapped: the LOW result reads the pair's second register
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=&v"(t) : "v"(a), "s"(sp), "v"(c));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=&v"(e) : "v"(a), "v"(vs), "v"(c));
    } else if (V == 7) {   // control: the plain 32-bit instruction with one scalar source, twice
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(t.x) : "v"(a.x), "s"(sp.x), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(t.y) : "v"(a.y), "s"(sp.y), "v"(c.y));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.x) : "v"(a.x), "v"(vs.x), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e.y) : "v"(a.y), "v"(vs.y), "v"(c.y));
    } else if (V == 8) {   // control: a 64-bit (double) instruction with a scalar-pair source
      double ad = (double)a.x, cd = (double)c.x, vd = spd, td, ed;
      asm volatile("" : "+v"(ad), "+v"(cd), "+v"(vd));
      asm volatile("v_fma_f64 %0, %1, %2, %3" : "=&v"(td) : "v"(ad), "s"(spd), "v"(cd));
      asm volatile("v_fma_f64 %0, %1, %2, %3" : "=&v"(ed) : "v"(ad), "v"(vd), "v"(cd));
      const unsigned long long tb = __builtin_bit_cast(unsigned long long, td), eb = __builtin_bit_cast(unsigned long long, ed);
      t = f32x2{__uint_as_float((unsigned)tb), __uint_as_float((unsigned)(tb >> 32))};
      e = f32x2{__uint_as_float((unsigned)eb), __uint_as_float((unsigned)(eb >> 32))};
    } else {   // V == 9: v_pk_mov_b32 from a scalar pair
      asm volatile("v_pk_mov_b32 %0, %1, %1" : "=&v"(t) : "s"(sp));
      asm volatile("v_pk_mov_b32 %0, %1, %1" : "=&v"(e) : "v"(vs));
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
  const size_t n = (size_t)256 << 20;
  CHK(hipMalloc(&buf, n * 4));
  CHK(hipMemset(buf, 0, n * 4));
  CHK(hipMalloc(&sink, 256));
  std::vector<float> ht(256);
  for (int i = 0; i < 256; ++i) ht[i] = 0.37f + 0.0131f * (float)i * (i & 1 ? 1.f : -0.5f);
  CHK(hipMalloc(&table, 1024));
  CHK(hipMemcpy(table, ht.data(), 1024, hipMemcpyHostToDevice));
  const char* lname[9] = {"none", "VALU + scalar sources", "VALU, vector registers only", "LDS", "global loads", "scalar ALU", "MFMA", "trans", "packed + scalar pairs"};
  const int liters[9] = {0, 3000000, 12000000, 1500000, 60000, 6000000, 3000000, 3000000, 6000000};
  const char* vname[NV] = {"v_pk_fma_f32 src1 = s[n:n+1]", "v_pk_mul_f32 src1 = s[n:n+1]", "v_pk_add_f32 src1 = s[n:n+1]", "v_pk_fma_f32 src0 = s[n:n+1]",
                           "v_pk_fma_f32 src2 = s[n:n+1]", "v_pk_fma_f32 src1 = s[n], both halves", "v_pk_fma_f32 src1 halves swapped", "v_fma_f32 x 2, one scalar each (control)",
                           "v_fma_f64 src1 = s[n:n+1] (control)", "v_pk_mov_b32 from s[n:n+1]"};
  for (int L = 0; L < 9; ++L) {
    CHK(hipMemset(cnt, 0, NV * 8 * sizeof(unsigned long long)));
    CHK(hipDeviceSynchronize());
    bool still = false;
    for (int rep = 0; rep < 2; ++rep) {
      if (L) hipLaunchKernelGGL(k_load, dim3(1024), dim3(256), 0, sl, L, liters[L], (const float*)buf, n, sink, (const float*)table);
      run_probe<0>(sp, iters, cnt, table); run_probe<1>(sp, iters, cnt, table); run_probe<2>(sp, iters, cnt, table); run_probe<3>(sp, iters, cnt, table);
      run_probe<4>(sp, iters, cnt, table); run_probe<5>(sp, iters, cnt, table); run_probe<6>(sp, iters, cnt, table); run_probe<7>(sp, iters, cnt, table);
      run_probe<8>(sp, iters, cnt, table); run_probe<9>(sp, iters, cnt, table);
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
      printf("  V%d %-42s low half by lane quarter [%llu %llu %llu %llu]   high half [%llu %llu %llu %llu]\n", v, vname[v], h[v * 8], h[v * 8 + 1], h[v * 8 + 2], h[v * 8 + 3],
             h[v * 8 + 4], h[v * 8 + 5], h[v * 8 + 6], h[v * 8 + 7]);
    fflush(stdout);
  }
  return 0;
}
