// How fast can ONE CU pull L2-resident bytes into registers / LDS?  (k_attn_hs streams 1.05 MB of split weights per 16-row
// workgroup and sublayer at ~47 GB/s - is that the CU's limit or the kernel's dependency chain?)
//   grid = nblk workgroups of 512 threads (8 waves), every workgroup reads the SAME `bytes` region `reps` times;
//   mode 0: global_load_dwordx4 into registers, U loads in flight per wave; mode 1: global_load_lds_dwordx4 (LDS-DMA), U in flight
//   hipcc -O3 --offload-arch=gfx950 tools/cu_stream_probe.hip -o /tmp/cu_stream_probe && /tmp/cu_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int U, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_regs(const u4* __restrict__ src, int n16_per_wave, int reps, unsigned* out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  u4 acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
    const u4* p = src + (size_t)w * n16_per_wave + lane;          // each wave its own slice, 1 KB per wave instruction
    for (int i = 0; i < n16_per_wave; i += 64 * U) {
      u4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = p[i + 64 * u];
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u];
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[threadIdx.x] = acc[0];
}

template <int U, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_lds(const u4* __restrict__ src, int n16_per_wave, int reps, unsigned* out) {
  __shared__ u4 buf[WAVES][U][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned acc = 0;
  for (int r = 0; r < reps; ++r) {
    const u4* p = src + (size_t)w * n16_per_wave + lane;
    for (int i = 0; i < n16_per_wave; i += 64 * U) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        __builtin_amdgcn_global_load_lds(p + i + 64 * u, (__attribute__((address_space(3))) void*)&buf[w][u][0], 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc ^= buf[w][i & (U - 1)][lane][0];
    }
  }
  if (acc == 0x12345u) out[threadIdx.x] = acc;
}

template <typename K>
static void run(const char* name, K kern, int waves, const u4* d, size_t bytes, int reps, unsigned* out, int nblk) {
  const int n16_per_wave = (int)(bytes / 16 / waves);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(64 * waves), 0, 0, d, n16_per_wave, 2, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(64 * waves), 0, 0, d, n16_per_wave, reps, out);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double per_cu = (double)bytes * reps / (ms * 1e-3) / 1e9;
  printf("%-34s blocks %3d: %8.1f us, %7.1f GB/s per workgroup, %8.1f GB/s total\n", name, nblk, ms * 1e3, per_cu, per_cu * nblk);
}

int main() {
  const size_t bytes = 1 << 20;        // 1 MiB region, L2-resident after the first pass
  u4* d; unsigned* out;
  CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 1, bytes)); CK(hipMalloc(&out, 4096));
  const int reps = 40;
  for (int nblk : {1, 32, 256}) {
    run("regs, 8 waves, 4 x 1KB in flight", k_regs<4, 8>, 8, d, bytes, reps, out, nblk);
    run("regs, 8 waves, 8 x 1KB in flight", k_regs<8, 8>, 8, d, bytes, reps, out, nblk);
    run("regs, 8 waves, 16 x 1KB in flight", k_regs<16, 8>, 8, d, bytes, reps, out, nblk);
    run("regs, 16 waves, 8 x 1KB in flight", k_regs<8, 16>, 16, d, bytes, reps, out, nblk);
    run("regs, 4 waves, 16 x 1KB in flight", k_regs<16, 4>, 4, d, bytes, reps, out, nblk);
    run("lds-dma, 8 waves, 8 x 1KB in flight", k_lds<8, 8>, 8, d, bytes, reps, out, nblk);
    run("lds-dma, 8 waves, 16 x 1KB in flight", k_lds<16, 8>, 8, d, bytes, reps, out, nblk);
    run("lds-dma, 2 waves, 16 x 1KB in flight", k_lds<16, 2>, 2, d, bytes, reps, out, nblk);
  }
  return 0;
}
