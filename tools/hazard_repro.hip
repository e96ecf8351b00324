// Stand-alone reproducer attempt for the "wrong rows" deviation of DESIGN.md section 5.1 (gfx950, ROCm 7.2).
// Hypothesis under test: packed fp32 VALU instructions whose operands carry operand-select (broadcast) modifiers / SGPR
// sources (v_pk_fma_f32 / v_pk_mul_f32 as hipcc emits them for `pk2{s, s} * r2`) give a wrong result now and then while OTHER
// waves of the same CU run MFMA chains and the issuing wave has VMEM returns in flight.
//   waves 0-3 of a workgroup: dependent v_mfma_f32_16x16x32_f16 chains (no memory traffic)          - "matrix group"
//   waves 4-7: the accumulation of edge_attn.cuh's step(): per edge one 8-byte row load (non-temporal, so the returns keep
//              coming from HBM), pe from the data, eight v_readlane broadcasts, zz[h] = pk_fma({ph, ph}, r2, zz[h]) - "vector group"
// The two groups never synchronise.  The kernel is run with the matrix group idle (mode 0) and busy (mode 1), many times; every
// output word of every run is compared with the first mode-0 run.  PACKED = 0 builds the same loop from scalar v_fma_f32.
//   hipcc -O3 --offload-arch=gfx950 tools/hazard_repro.hip -o /tmp/hazard_repro && /tmp/hazard_repro [launches] [edges]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float pk2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int PACKED>
__global__ __launch_bounds__(512) void k_repro(const float* __restrict__ rows, int E, int mode, int mfma_iters, float* out, float* sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < 4) {                                       // matrix group
    if (!(mode & 1)) return;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    for (int it = 0; it < mfma_iters; ++it) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
    }
    if (c0[0] + c1[1] == 12345.f) sink[threadIdx.x] = c0[0];      // keep the chain alive
    return;
  }
  // vector group: row = one "destination", lane l owns columns 2l, 2l + 1
  const int row = blockIdx.x * 4 + (wave - 4);
  const float* base = rows + (size_t)row * E * 128;
  pk2 zz[8], ag = {0.f, 0.f};
  for (int h = 0; h < 8; ++h) zz[h] = pk2{0.f, 0.f};
  for (int e0 = 0; e0 < E; e0 += 6) {
    pk2 rb[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const int e = e0 + s < E ? e0 + s : E - 1;
      rb[s] = __builtin_nontemporal_load(reinterpret_cast<const pk2*>(base + (size_t)e * 128 + 2 * lane));
    }
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      if (e0 + s >= E) break;
      const pk2 r2 = rb[s];
      const float pe = __builtin_amdgcn_exp2f(-fabsf(r2[0] + r2[1]));       // a data-dependent weight in (0, 1]
      if (PACKED) {
        ag = __builtin_elementwise_fma(pk2{pe, pe}, r2, ag);
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          const float ph = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pe), 8 * h));     // SGPR
          zz[h] = __builtin_elementwise_fma(pk2{ph, ph}, r2, zz[h]);                               // v_pk_fma_f32, op_sel broadcast
        }
      } else {
        ag[0] = fmaf(pe, r2[0], ag[0]); ag[1] = fmaf(pe, r2[1], ag[1]);
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          const float ph = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pe), 8 * h));
          asm volatile("v_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %2, %4, %1" : "+v"(zz[h][0]), "+v"(zz[h][1]) : "s"(ph), "v"(r2[0]), "v"(r2[1]));
        }
      }
    }
  }
  float* o = out + (size_t)row * 18 * 64;
  for (int h = 0; h < 8; ++h) { o[(2 * h) * 64 + lane] = zz[h][0]; o[(2 * h + 1) * 64 + lane] = zz[h][1]; }
  o[16 * 64 + lane] = ag[0]; o[17 * 64 + lane] = ag[1];
}

template <int PACKED>
static long campaign(const char* name, const float* d_rows, int E, int launches, int nblk, float* d_out, float* d_sink) {
  const size_t n = (size_t)nblk * 4 * 18 * 64;
  std::vector<float> ref(n), cur(n);
  hipLaunchKernelGGL(k_repro<PACKED>, dim3(nblk), dim3(512), 0, 0, d_rows, E, 0, 0, d_out, d_sink);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(ref.data(), d_out, n * 4, hipMemcpyDeviceToHost));
  long bad_words = 0, bad_rows = 0, bad_launches = 0;
  double worst = 0.0;
  for (int mode = 0; mode < 2; ++mode) {
    long bw = 0, br = 0, bl = 0;
    for (int it = 0; it < launches; ++it) {
      CK(hipMemset(d_out, 0, n * 4));
      hipLaunchKernelGGL(k_repro<PACKED>, dim3(nblk), dim3(512), 0, 0, d_rows, E, mode, 4000 + 37 * it, d_out, d_sink);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(cur.data(), d_out, n * 4, hipMemcpyDeviceToHost));
      long w = 0;
      for (size_t r = 0; r < (size_t)nblk * 4; ++r) {
        long wr = 0;
        for (size_t k = 0; k < 18 * 64; ++k) {
          const size_t i = r * 18 * 64 + k;
          if (memcmp(&cur[i], &ref[i], 4)) { ++wr; const double d = fabs((double)cur[i] - ref[i]) / (fabs((double)ref[i]) + 1e-30); if (d > worst) worst = d; }
        }
        w += wr; br += wr != 0;
      }
      bw += w; bl += w != 0;
    }
    printf("%-7s matrix group %-4s: %d launches x %d rows: %ld launches / %ld rows / %ld words differ from the reference run\n",
           name, mode ? "BUSY" : "idle", launches, nblk * 4, bl, br, bw);
    bad_words += bw; bad_rows += br; bad_launches += bl;
  }
  if (bad_words) printf("%-7s worst relative deviation %.3e\n", name, worst);
  return bad_words;
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 200, E = argc > 2 ? atoi(argv[2]) : 600, nblk = 256;
  const size_t n_rows = (size_t)nblk * 4 * E * 128;
  std::vector<float> h(n_rows);
  unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (3.0f / (1 << 23)); }
  float *d_rows, *d_out, *d_sink;
  CK(hipMalloc(&d_rows, n_rows * 4)); CK(hipMalloc(&d_out, (size_t)nblk * 4 * 18 * 64 * 4)); CK(hipMalloc(&d_sink, 4096));
  CK(hipMemcpy(d_rows, h.data(), n_rows * 4, hipMemcpyHostToDevice));
  const long p = campaign<1>("packed", d_rows, E, launches, nblk, d_out, d_sink);
  const long q = campaign<0>("scalar", d_rows, E, launches, nblk, d_out, d_sink);
  printf("RESULT packed_bad_words=%ld scalar_bad_words=%ld\n", p, q);
  return 0;
}
