// accuracy of v_sin_f32 / v_cos_f32 (input in revolutions) against fp64, with an fma-based reduction of z = x f 2 pi (fp32, the
// reference's argument) to r = z / (2 pi) - n in [-0.5, 0.5]:  hipcc -O3 --offload-arch=gfx950 tools/hw_sincos_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const float* z, float* s, float* c, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = z[i];
  const float c_hi = 0.15915494f, c_lo = 6.4206382e-09f;          // 1 / (2 pi) = c_hi + c_lo (c_hi = 0x3e22f983)
  const float t = x * c_hi;
  const float k_ = rintf(t);
  float r = __builtin_fmaf(x, c_hi, -k_);
  r = __builtin_fmaf(x, c_lo, r);
  s[i] = __builtin_amdgcn_sinf(r);
  c[i] = __builtin_amdgcn_cosf(r);
}
int main() {
  const int n = 1 << 22;
  std::vector<float> h(n), hs(n), hc(n);
  unsigned st = 1;
  for (int i = 0; i < n; ++i) { st = st * 1664525u + 1013904223u; const float u = (st >> 8) * (1.0f / (1 << 24)); h[i] = (u - 0.5f) * ((i & 3) == 0 ? 2000.f : (i & 3) == 1 ? 100.f : 16.f); }
  float *dz, *ds, *dc;
  hipMalloc(&dz, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
  hipMemcpy(dz, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dz, ds, dc, n);
  hipMemcpy(hs.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), dc, n * 4, hipMemcpyDeviceToHost);
  double es[3] = {0, 0, 0}, ec[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const int b = i & 3 ? ((i & 3) == 1 ? 1 : 2) : 0;
    es[b] = fmax(es[b], fabs((double)hs[i] - sin((double)h[i]))); ec[b] = fmax(ec[b], fabs((double)hc[i] - cos((double)h[i])));
  }
  const char* nm[3] = {"|z| < 1000", "|z| < 50", "|z| < 8"};
  for (int b = 0; b < 3; ++b) printf("%-10s max abs error sin %.3e cos %.3e\n", nm[b], es[b], ec[b]);
  return 0;
}
