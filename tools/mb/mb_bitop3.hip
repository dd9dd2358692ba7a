// Round 6: does gfx950 execute the v_bitop3_b32 forms hipcc emits for the staging-area swizzles of k_chain16's edge phase -- among them two
// inline constants in src1 / src2 -- as the truth table says?   hipcc --offload-arch=gfx950 -O3 tools/mb/mb_bitop3.hip -o /tmp/mb_bitop3 && /tmp/mb_bitop3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k(unsigned* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned a = (unsigned)i * 2654435761u, b = (unsigned)i * 40503u + 977u, r0, r1, r2, r3, r4, r5;
  asm volatile("v_bitop3_b32 %0, %1, 4, 12 bitop3:0x6c" : "=v"(r0) : "v"(a));
  asm volatile("v_bitop3_b32 %0, %1, 8, 12 bitop3:0x6c" : "=v"(r1) : "v"(a));
  asm volatile("v_bitop3_b32 %0, %1, 12, %1 bitop3:0xc" : "=v"(r2) : "v"(a));
  asm volatile("v_bitop3_b32 %0, %1, %2, 4 bitop3:0x36" : "=v"(r3) : "v"(a), "v"(b));
  asm volatile("v_bitop3_b32 %0, %1, %2, 3 bitop3:0x78" : "=v"(r4) : "v"(a), "v"(b));
  asm volatile("v_bitop3_b32 %0, %1, %2, 15 bitop3:0x78" : "=v"(r5) : "v"(a), "v"(b));
  out[6 * i + 0] = r0; out[6 * i + 1] = r1; out[6 * i + 2] = r2; out[6 * i + 3] = r3; out[6 * i + 4] = r4; out[6 * i + 5] = r5;
}
static unsigned tt(unsigned t, unsigned a, unsigned b, unsigned c) {
  unsigned r = 0;
  for (int i = 0; i < 32; ++i) {
    const unsigned idx = (((a >> i) & 1) << 2) | (((b >> i) & 1) << 1) | ((c >> i) & 1);
    r |= ((t >> idx) & 1u) << i;
  }
  return r;
}
int main() {
  const int n = 1 << 16;
  unsigned* d; CHK(hipMalloc(&d, 6 * n * 4));
  k<<<n / 256, 256>>>(d, n); CHK(hipDeviceSynchronize());
  unsigned* h = (unsigned*)malloc(6 * n * 4); CHK(hipMemcpy(h, d, 6 * n * 4, hipMemcpyDeviceToHost));
  long bad[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const unsigned a = (unsigned)i * 2654435761u, b = (unsigned)i * 40503u + 977u;
    const unsigned e[6] = {tt(0x6c, a, 4, 12), tt(0x6c, a, 8, 12), tt(0x0c, a, 12, a), tt(0x36, a, b, 4), tt(0x78, a, b, 3), tt(0x78, a, b, 15)};
    for (int j = 0; j < 6; ++j) if (h[6 * i + j] != e[j]) { if (!bad[j]) printf("form %d first mismatch: a %08x b %08x got %08x want %08x\n", j, a, b, h[6 * i + j], e[j]); ++bad[j]; }
  }
  printf("mismatches by form (a,4,12:0x6c | a,8,12:0x6c | a,12,a:0x0c | a,b,4:0x36 | a,b,3:0x78 | a,b,15:0x78): %ld %ld %ld %ld %ld %ld of %d\n", bad[0], bad[1], bad[2], bad[3], bad[4], bad[5], n);
  return 0;
}
