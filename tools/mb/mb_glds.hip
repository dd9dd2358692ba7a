// LDS-DMA recipe check (gfx950): global_load_lds_dwordx4 with a per-lane permuted source, wave-private 4 KB stage;
// dumps what lands where.  hipcc --offload-arch=gfx950 -O3 tools/mb/mb_glds.hip -o /tmp/mb_glds && /tmp/mb_glds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// the MUBUF form: 32-bit per-lane byte offset + SGPR offset (memory address only) into a raw buffer
__device__ __forceinline__ void blds16(unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
__global__ void kb(const unsigned* __restrict__ src, unsigned* __restrict__ out, int base_off, int soff) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  unsigned char* stage = smem + base_off;
  const unsigned lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)stage);
  for (int i = lane; i < 1024; i += 64) reinterpret_cast<unsigned*>(stage)[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
  const int dr = lane >> 4, ds_ = lane & 15;
  for (int i = 0; i < 4; ++i) {
    const int rr = 4 * i + dr;
    blds16((unsigned)(rr * 512 + 16 * (ds_ ^ rr)), rsrc, (unsigned)soff, lds + 1024 * i);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int i = lane; i < 1024; i += 64) out[i] = reinterpret_cast<unsigned*>(stage)[i];
  if (lane == 0) out[1024] = lds;
}
__global__ void k(const unsigned* __restrict__ src, unsigned* __restrict__ out, int base_off) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  unsigned char* stage = smem + base_off;
  const unsigned lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)stage);
  for (int i = lane; i < 1024; i += 64) reinterpret_cast<unsigned*>(stage)[i] = 0xdeadbeefu;
  __syncthreads();
  const int dr = lane >> 4, ds_ = lane & 15;
  for (int i = 0; i < 4; ++i) {
    const int rr = 4 * i + dr;
    const unsigned* p = src + rr * 128 + 4 * (ds_ ^ rr);   // row rr: 128 dwords (hi 64 | lo 64); piece = 4 dwords
    glds16(p, lds + 1024 * i);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int i = lane; i < 1024; i += 64) out[i] = reinterpret_cast<unsigned*>(stage)[i];
  if (lane == 0) out[1024] = lds;
}
int main() {
  std::vector<unsigned> h(16 * 128);
  for (int r = 0; r < 16; ++r) for (int c = 0; c < 128; ++c) h[r * 128 + c] = r * 1000 + c;
  unsigned *d, *o;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, 1025 * 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int base : {0, 4096, 61440, 65536, 98304}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 140 * 1024, 0, d, o, base);
    std::vector<unsigned> r(1025);
    if (hipMemcpy(r.data(), o, 1025 * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("memcpy failed\n"); return 1; }
    int bad = 0, dead = 0;
    for (int rr = 0; rr < 16; ++rr) for (int s = 0; s < 16; ++s) for (int j = 0; j < 4; ++j) {
      const unsigned got = r[rr * 64 + s * 4 + j], want = rr * 1000 + 4 * (s ^ rr) + j;
      if (got != want) { ++bad; if (got == 0xdeadbeefu) ++dead; }
    }
    printf("base %6d (lds addr %u): mismatches %d of 1024 (untouched %d); first words: %u %u %u %u | %u %u\n", base, r[1024], bad, dead, r[0], r[1], r[2], r[3], r[4], r[64]);
  }
  for (int soff : {0, 256}) {
    hipLaunchKernelGGL(kb, dim3(1), dim3(64), 140 * 1024, 0, d, o, 8192, soff);
    std::vector<unsigned> r(1025);
    hipMemcpy(r.data(), o, 1025 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int rr = 0; rr < 16; ++rr) for (int s = 0; s < 16; ++s) for (int j = 0; j < 4; ++j)
      if (r[rr * 64 + s * 4 + j] != (unsigned)(rr * 1000 + soff / 4 + 4 * (s ^ rr) + j)) ++bad;
    printf("buffer form, soffset %d: mismatches %d of 1024; first words %u %u | %u\n", soff, bad, r[0], r[1], r[64]);
  }
  return 0;
}
