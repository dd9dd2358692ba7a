// Accuracy of the hardware v_sin_f32 / v_cos_f32 (input in revolutions) against double precision, alone and inside
// the Fourier-argument pipeline (exact fp32 argument -> revolutions by a two-term product -> v_sin / v_cos).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstdlib>
__global__ void k(const float* a, float* s, float* c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = a[i];
  // revolutions: x / (2 pi) with the rounding error of the product recovered by fma
  const float C1 = 0.15915494309189535f, C2 = (float)(0.15915494309189535 - (double)0.15915494309189535f);
  const float u = x * C1;
  const float nn = rintf(u);
  const float f = (fmaf(x, C1, -nn)) + x * C2;   // fma: exact (x*C1 - nn) before the rounding
  s[i] = __builtin_amdgcn_sinf(f);
  c[i] = __builtin_amdgcn_cosf(f);
}
int main() {
  const int n = 1 << 22;
  std::vector<float> a(n);
  srand(3);
  for (int i = 0; i < n; ++i) {
    double r = rand() / (double)RAND_MAX;
    a[i] = (i & 3) == 0 ? (float)(r * 1885.0) : ((i & 3) == 1 ? (float)((2 * r - 1) * 19.74) : ((i & 3) == 2 ? (float)(r * 6.3) : (float)((2 * r - 1) * 0.2)));
  }
  float *da, *ds, *dc;
  hipMalloc(&da, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
  hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, ds, dc, n);
  std::vector<float> s(n), c(n);
  hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
  double es[4] = {0, 0, 0, 0}, ec[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    es[i & 3] = fmax(es[i & 3], fabs((double)s[i] - sin((double)a[i])));
    ec[i & 3] = fmax(ec[i & 3], fabs((double)c[i] - cos((double)a[i])));
  }
  const char* nm[4] = {"|x| < 1885", "|x| < 19.74", "x < 6.3", "|x| < 0.2"};
  for (int j = 0; j < 4; ++j) printf("%-12s max abs err: sin %.3e cos %.3e\n", nm[j], es[j], ec[j]);
  return 0;
}
