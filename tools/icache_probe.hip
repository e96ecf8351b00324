// Does straight-line code that runs ONCE per wave cost more than the same instructions in a loop?  (k_attn_hs: 30 KB of fully
// unrolled code per launch, ~11 cycles per instruction.)  One workgroup of 8 waves per CU on 32 CUs; every lane runs N dependent-free
// v_fma instructions (4 accumulators) either as N straight-line instructions (footprint 8 N bytes) or as a loop around 64 of them.
// hipcc -O3 --offload-arch=gfx950 tools/icache_probe.hip -o /tmp/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define F4 "v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t"
#define F16 F4 F4 F4 F4
#define F64 F16 F16 F16 F16

template <int REPS64, bool LOOP>
__global__ __launch_bounds__(512) void k_code(float* out, unsigned long long* cyc, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (LOOP) {
#pragma unroll 1
    for (int i = 0; i < REPS64; ++i) asm volatile(F64 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
  } else {
#pragma unroll
    for (int i = 0; i < REPS64; ++i) asm volatile(F64 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int REPS64, bool LOOP>
int run(const char* name, float* out, unsigned long long* cyc, int grid) {
  unsigned long long h[256];
  double s = 0;
  for (int rep = 0; rep < 6; ++rep) {
    hipLaunchKernelGGL((k_code<REPS64, LOOP>), dim3(grid), dim3(512), 0, 0, out, cyc, 1.0001f, 0.5f);
    CK(hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost));
    if (rep) s += h[0];
  }
  printf("%-40s %5d instructions (%3d KB of code): %8.0f cycles = %5.2f cycles per instruction\n", name, REPS64 * 64, LOOP ? 0 : REPS64 * 64 * 8 / 1024, s / 5, s / 5 / (REPS64 * 64));
  return 0;
}

int main() {
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 256 * 8));
  run<16, true>("loop, 8 waves x 32 workgroups", out, cyc, 32);
  run<16, false>("straight line, 8 waves x 32 workgroups", out, cyc, 32);
  run<64, true>("loop", out, cyc, 32);
  run<64, false>("straight line", out, cyc, 32);
  run<128, true>("loop", out, cyc, 32);
  run<128, false>("straight line", out, cyc, 32);
  run<64, false>("straight line, 256 workgroups", out, cyc, 256);
  run<64, true>("loop, 256 workgroups", out, cyc, 256);
  return 0;
}
