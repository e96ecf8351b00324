// What does ONE compute unit get when it streams a 1 MB weight block the way k_attn_hs does (8 waves, each wave 1 KB per
// instruction, eight instructions in flight per wave) - (a) first pass in a kernel (the block was written / read by an earlier
// kernel), (b) second pass in the SAME kernel (whatever the first pass left in the L2), (c) first pass while other workgroups of the
// same launch read the same block ("warm" workgroups on idle CUs), (d) a block that this launch's other workgroups read a long
// time ago.  s_memtime cycles of wave 0 of workgroup 0.      hipcc -O3 --offload-arch=gfx950 tools/l2_probe.hip -o /tmp/l2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }

// one pass over `bytes` (multiple of 64 KB) by the 512 threads of a workgroup: 8 x (1 KB per wave) in flight per wave
__device__ __forceinline__ uint4 pass(const char* base, int bytes, int tid, size_t stride = 65536) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  const int w = tid >> 6, lane = tid & 63;
  for (int i = 0; i < bytes / 65536; ++i) {
    const size_t off = (size_t)i * stride;
    uint4 v[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) v[s] = *reinterpret_cast<const uint4*>(base + off + s * 8192 + w * 1024 + lane * 16);
#pragma unroll
    for (int s = 0; s < 8; ++s) { acc.x ^= v[s].x; acc.y ^= v[s].y; acc.z ^= v[s].z; acc.w ^= v[s].w; }
  }
  return acc;
}

__global__ __launch_bounds__(512) void k_probe(const char* blk, int bytes, int n_work, int warm, int delay, unsigned long long* out, uint4* sink, size_t stride = 65536) {
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= n_work) {                 // warm workgroups: each reads a slice once (all of them together: the whole block)
    if (!warm) return;
    const int k = ((int)blockIdx.x - n_work) >> 3, K = ((int)gridDim.x - n_work) >> 3;
    const int per = ((bytes / K) + 8191) & ~8191;
    uint4 a = make_uint4(0, 0, 0, 0);
    for (int off = k * per + tid * 16; off < min(bytes, (k + 1) * per); off += 8192) {
      const uint4 v = *reinterpret_cast<const uint4*>(blk + off);
      a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w;
    }
    if (a.x == 0x12345678u) sink[blockIdx.x] = a;
    return;
  }
  if (delay) { const unsigned long long t = now(); while (now() - t < (unsigned long long)delay) __builtin_amdgcn_s_sleep(8); }
  const unsigned long long t0 = now();
  uint4 a = pass(blk, bytes, tid, stride);
  asm volatile("s_waitcnt vmcnt(0)" :: "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w) : "memory");
  const unsigned long long t1 = now();
  uint4 b = pass(blk, bytes, tid, stride);
  asm volatile("s_waitcnt vmcnt(0)" :: "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w) : "memory");
  const unsigned long long t2 = now();
  if (blockIdx.x == 0 && tid == 0) { out[0] = t1 - t0; out[1] = t2 - t1; }
  if ((a.x ^ b.x) == 0x12345678u) sink[blockIdx.x] = a;
}

__global__ void k_touch(char* p, size_t n) {      // another kernel in between (writes something else, reads nothing of the block)
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (char)i;
}

int main() {
  const int bytes = 1 << 20;
  char *blk, *other; unsigned long long* out; uint4* sink;
  CK(hipMalloc(&blk, bytes)); CK(hipMalloc(&other, 64 << 20)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&sink, 4096 * sizeof(uint4)));
  CK(hipMemset(blk, 1, bytes));
  unsigned long long h[2];
  auto run = [&](const char* name, int n_work, int n_warm_per_xcd, int warm, int delay, bool touch) -> int {
    double s0 = 0, s1 = 0;
    for (int rep = 0; rep < 6; ++rep) {
      if (touch) hipLaunchKernelGGL(k_touch, dim3((64 << 20) / 256), dim3(256), 0, 0, other, (size_t)(64 << 20));
      const int grid = ((n_work + 7) & ~7) + 8 * n_warm_per_xcd;
      hipLaunchKernelGGL(k_probe, dim3(grid), dim3(512), 0, 0, blk, bytes, (n_work + 7) & ~7, warm, delay, out, sink);
      CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
      if (rep) { s0 += h[0]; s1 += h[1]; }
    }
    printf("%-78s first pass %7.0f cycles (%5.1f B/clk)   second pass %7.0f cycles (%5.1f B/clk)\n", name, s0 / 5, bytes / (s0 / 5), s1 / 5, bytes / (s1 / 5));
    return 0;
  };
  run("1 workgroup, back-to-back launches of the same kernel", 1, 0, 0, 0, false);
  run("1 workgroup, a 64 MB write kernel between the launches", 1, 0, 0, 0, true);
  run("32 workgroups (one per 8 CUs ...), back-to-back", 32, 0, 0, 0, false);
  run("32 workgroups, 64 MB write kernel in between", 32, 0, 0, 0, true);
  run("32 working + 24 x 8 warm workgroups, 64 MB write kernel in between", 32, 24, 1, 0, true);
  run("32 working (start delayed 8000 cycles) + 24 x 8 warm workgroups, write kernel in between", 32, 24, 1, 8000, true);
  run("32 working (start delayed 20000 cycles) + 24 x 8 warm workgroups, write kernel in between", 32, 24, 1, 20000, true);
  run("256 workgroups, write kernel in between", 256, 0, 0, 0, true);
  // address translation: the same 1 MB as sixteen 64 KB pieces, each in another 2 MB / 64 KB / 4 MB stretch of a 64 MB buffer
  for (size_t stride : {(size_t)65536, (size_t)(128 << 10), (size_t)(1 << 20), (size_t)(2 << 20), (size_t)(4 << 20)}) {
    double s0 = 0, s1 = 0;
    for (int rep = 0; rep < 6; ++rep) {
      hipLaunchKernelGGL(k_probe, dim3(8), dim3(512), 0, 0, other, bytes, 8, 0, 0, out, sink, stride);
      CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
      if (rep) { s0 += h[0]; s1 += h[1]; }
    }
    printf("16 pieces of 64 KB, %5zu KB apart, back-to-back launches:   first pass %7.0f cycles   second pass %7.0f cycles\n", stride >> 10, s0 / 5, s1 / 5);
  }
  return 0;
}
