// Reproducer attempt no. 2 for DESIGN.md section 5.1, closer to the kernel that shows the deviation (k_edge_fused_p): ONE 16-wave
// workgroup per CU whose two halves never synchronise.
//   waves 8-15 ("loop half"):   the library's own edge loop - EdgeAcc<true>::step of infgen_amd/csrc/edge_attn.cuh (v_pk_mul / v_pk_fma
//                               with op_sel broadcasts and SGPR sources, permlane swaps, DPP, v_exp) over E edges per row, six K / V /
//                               rhat row loads in flight per trip, u and q from LDS
//   waves 0-7  ("matrix half"): what k_edge_fused's phases 1 and 3 execute - fragment loads from L2, split_pair (packed fp32 subtract,
//                               cvt_pkrtz), v_mfma_f32_16x16x16_f16 / 16x16x32_f16 chains, ds_write_b128 / ds_read_b128 of its own LDS tile
// mode 0: matrix half idle, mode 1: busy.  Every launch's loop results are compared bitwise with the first mode-0 launch.
// Arms (profiles/r03_hazard_bisect*.log): -DIG_EDGE_OPSEL_BROADCAST = the loop as hipcc compiles pk2{v, v} (per-lane broadcasts as op_sel
// modifiers on a 32-bit VGPR): ~40 % of the launches differ; without it (the shipped form: per-lane broadcasts as real register
// pairs): none.  With the failing arm: -DMAT_NO_MFMA none; -DMAT_NO_SPLIT / -DMAT_NO_LDS still differ; -Xclang -target-feature -Xclang
// -packed-fp32-ops (no packed fp32 instruction in the kernel) none; -DIG_EDGE_SPAIR / -DIG_EDGE_SVPAIR (SGPR broadcasts changed) still differ.
//   hipcc -O3 --offload-arch=gfx950 -I infgen_amd/csrc -DIG_EDGE_OPSEL_BROADCAST tools/hazard_repro2.hip -o /tmp/hazard_repro2 && /tmp/hazard_repro2 [launches] [edges]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "edge_attn.cuh"
#include "split.cuh"
using namespace ig;
// bisect variants (-D...): MAT_NO_MFMA replaces every MFMA of the matrix half by a VALU add of its C operand, MAT_NO_SPLIT drops
// split_pair (packed fp32 subtract + cvt_pkrtz), MAT_NO_LDS keeps the u tile out of LDS, MAT_NO_GLOBAL loads q / fragments once
#ifdef MAT_NO_MFMA
#define MFMA16(a, b, c) ((c) + f32x4{(float)(a)[0], (float)(b)[0], 0.f, 0.f})
#define MFMA32(a, b, c) ((c) + f32x4{(float)(a)[0], (float)(b)[0], 0.f, 0.f})
#else
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0)
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
constexpr int LDU = H * D + 4;

__global__ __launch_bounds__(1024, 4) void k_repro2(const float* Q, const float* Kr, const float* Vr, const float* R, const unsigned short* W,
                                                    int E, int nsrc, int mode, int mat_iters, float* out, float* sink) {
  __shared__ __attribute__((aligned(16))) float UZ[32 * LDU];
  __shared__ __attribute__((aligned(16))) float QT[32 * (D + 4)];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // the loop half's waves fill their OWN two rows of the u / q tiles: no hand-off and no synchronisation between any two waves
  if (w >= 8) {
    for (int rr = 0; rr < 2; ++rr) {
      const int rl = 16 + 2 * (w - 8) + rr;
      for (int i = lane; i < H * D; i += 64) UZ[rl * LDU + i] = Q[(size_t)((blockIdx.x * 16 + rl) * 7 % nsrc) * D + i % D] * (0.25f + 0.01f * (i / D));
      for (int i = lane; i < D; i += 64) QT[rl * (D + 4) + i] = Q[(size_t)((blockIdx.x * 16 + rl) % nsrc) * D + i];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  if (w < 8) {
    if (!(mode & 1)) return;
    const int j = lane & 15, g = lane >> 4, h = w;
    float s = 0.f;
    for (int it = 0; it < mat_iters; ++it) {
      // phase 1 of k_edge_fused: u_h = q_h W'_kr,h
      const float4 qv = *reinterpret_cast<const float4*>(Q + (size_t)((blockIdx.x * 16 + j + it) % nsrc) * D + DH * h + 4 * g);
      const unsigned short* Wk = W + (size_t)(h >> 1) * QUARTER + (size_t)((h & 1) * 8) * 2 * 256 + lane * 4;
      v4h ah[8], al[8];
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) { ah[ct] = *reinterpret_cast<const v4h*>(Wk + (ct * 2) * 256); al[ct] = *reinterpret_cast<const v4h*>(Wk + (ct * 2 + 1) * 256); }
      u32x2 qh, ql; unsigned hi, lo;
#ifndef MAT_NO_SPLIT
      split_pair(qv.x * 64.f, qv.y * 64.f, hi, lo); qh[0] = hi; ql[0] = lo;
      split_pair(qv.z * 64.f, qv.w * 64.f, hi, lo); qh[1] = hi; ql[1] = lo;
#else
      qh[0] = __float_as_uint(qv.x); ql[0] = __float_as_uint(qv.y); qh[1] = __float_as_uint(qv.z); ql[1] = __float_as_uint(qv.w);
#endif
      const v4h vqh = __builtin_bit_cast(v4h, qh), vql = __builtin_bit_cast(v4h, ql);
      f32x4 acc[8];
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) acc[ct] = MFMA16(ah[ct], vqh, (f32x4{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) acc[ct] = MFMA16(ah[ct], vql, acc[ct]);
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) acc[ct] = MFMA16(al[ct], vqh, acc[ct]);
#ifndef MAT_NO_LDS
      float* urow = UZ + j * LDU + h * D + 4 * g;
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) *reinterpret_cast<float4*>(urow + 16 * ct) = make_float4(acc[ct][0], acc[ct][1], acc[ct][2], acc[ct][3]);
#endif
      // phase 3: agg' = W'_vr,h z_h from the tile just written
      const float* zrow = UZ + j * LDU + h * D + 8 * g;
      const unsigned short* Wv = W + 8 * QUARTER + (size_t)(h >> 1) * QUARTER + (size_t)((h & 1) * 4) * 2 * 512 + lane * 8;
      f32x4 a3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const v8h a_h = *reinterpret_cast<const v8h*>(Wv + (s4 * 2) * 512), a_l = *reinterpret_cast<const v8h*>(Wv + (s4 * 2 + 1) * 512);
#ifndef MAT_NO_LDS
        const float4 z0 = *reinterpret_cast<const float4*>(zrow + 32 * s4), z1 = *reinterpret_cast<const float4*>(zrow + 32 * s4 + 4);
#else
        const float4 z0 = make_float4(acc[s4][0], acc[s4][1], acc[s4][2], acc[s4][3]), z1 = make_float4(acc[s4 + 4][0], acc[s4 + 4][1], acc[s4 + 4][2], acc[s4 + 4][3]);
#endif
        u32x4 bh, bl;
        split_pair(z0.x, z0.y, hi, lo); bh[0] = hi; bl[0] = lo; split_pair(z0.z, z0.w, hi, lo); bh[1] = hi; bl[1] = lo;
        split_pair(z1.x, z1.y, hi, lo); bh[2] = hi; bl[2] = lo; split_pair(z1.z, z1.w, hi, lo); bh[3] = hi; bl[3] = lo;
        const v8h vbh = __builtin_bit_cast(v8h, bh), vbl = __builtin_bit_cast(v8h, bl);
        a3 = MFMA32(a_h, vbh, a3);
        a3 = MFMA32(a_h, vbl, a3);
        a3 = MFMA32(a_l, vbh, a3);
      }
      s += a3[0] + a3[3];
    }
    if (s == 12345.f) sink[threadIdx.x] = s;
    return;
  }
  // loop half: wave w takes rows 2 (w - 8), 2 (w - 8) + 1 of the workgroup's 16
  const bool b3 = lane & 8;
  for (int rr = 0; rr < 2; ++rr) {
    const int rl = 16 + 2 * (w - 8) + rr, drow = blockIdx.x * 16 + (rl - 16);
    EdgeAcc<true> acc;
    acc.q = *reinterpret_cast<const float2*>(QT + rl * (D + 4) + 2 * lane);
    acc.load_u(UZ + rl * LDU, lane);
    acc.reset();
    const float* rbase = R + (size_t)drow * E * D;
    for (int i0 = 0; i0 < E; i0 += 6) {
      pk2 kb[6], vb[6], rb[6];
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        const int e = min(i0 + s, E - 1);
        const int sj = (drow * 131 + e * 17) % nsrc;
        kb[s] = ea_ld(Kr + (size_t)sj * D + 2 * lane, false);
        vb[s] = ea_ld(Vr + (size_t)sj * D + 2 * lane, false);
        rb[s] = ea_ld(rbase + (size_t)e * D + 2 * lane, true);
      }
#pragma unroll
      for (int s = 0; s < 6; ++s) acc.step(kb[s], vb[s], rb[s], i0 + s < E, b3);
    }
    float* o = out + (size_t)drow * 19 * 64;
    for (int hd = 0; hd < H; ++hd) { o[(2 * hd) * 64 + lane] = acc.zz[hd][0]; o[(2 * hd + 1) * 64 + lane] = acc.zz[hd][1]; }
    o[16 * 64 + lane] = acc.ag[0]; o[17 * 64 + lane] = acc.ag[1]; o[18 * 64 + lane] = acc.lsum;
  }
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 200, E = argc > 2 ? atoi(argv[2]) : 96, nblk = 256, nsrc = 4096;
  const size_t nrow = (size_t)nblk * 16, nR = nrow * E * D, nout = nrow * 19 * 64;
  std::vector<float> h(nR > (size_t)nsrc * D ? nR : (size_t)nsrc * D);
  unsigned s = 777u;
  auto fill = [&](size_t n, float amp) { for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) - (1 << 23)) * (amp / (1 << 23)); } };
  float *dQ, *dK, *dV, *dR, *dout, *dsink; unsigned short* dW;
  CK(hipMalloc(&dQ, (size_t)nsrc * D * 4)); CK(hipMalloc(&dK, (size_t)nsrc * D * 4)); CK(hipMalloc(&dV, (size_t)nsrc * D * 4));
  CK(hipMalloc(&dR, nR * 4)); CK(hipMalloc(&dout, nout * 4)); CK(hipMalloc(&dsink, 8192)); CK(hipMalloc(&dW, 16 * QUARTER * 2));
  fill((size_t)nsrc * D, 1.0f); CK(hipMemcpy(dQ, h.data(), (size_t)nsrc * D * 4, hipMemcpyHostToDevice));
  fill((size_t)nsrc * D, 1.0f); CK(hipMemcpy(dK, h.data(), (size_t)nsrc * D * 4, hipMemcpyHostToDevice));
  fill((size_t)nsrc * D, 1.0f); CK(hipMemcpy(dV, h.data(), (size_t)nsrc * D * 4, hipMemcpyHostToDevice));
  fill(nR, 2.5f); CK(hipMemcpy(dR, h.data(), nR * 4, hipMemcpyHostToDevice));
  { std::vector<unsigned short> wv(16 * QUARTER); for (auto& x : wv) { s = s * 1664525u + 1013904223u; x = 0x2c00 + ((s >> 12) & 0x3ff); } CK(hipMemcpy(dW, wv.data(), wv.size() * 2, hipMemcpyHostToDevice)); }
  std::vector<float> ref(nout), cur(nout);
  hipLaunchKernelGGL(k_repro2, dim3(nblk), dim3(1024), 0, 0, dQ, dK, dV, dR, dW, E, nsrc, 0, 0, dout, dsink);
  CK(hipDeviceSynchronize()); CK(hipMemcpy(ref.data(), dout, nout * 4, hipMemcpyDeviceToHost));
  double chk = 0; for (size_t i = 0; i < nout; ++i) chk += ref[i];
  printf("reference checksum %.6e (finite: %d)\n", chk, (int)std::isfinite(chk));
  for (int mode = 0; mode < 2; ++mode) {
    long bl = 0, br = 0, bw = 0; double worst = 0;
    for (int it = 0; it < launches; ++it) {
      CK(hipMemset(dout, 0, nout * 4));
      hipLaunchKernelGGL(k_repro2, dim3(nblk), dim3(1024), 0, 0, dQ, dK, dV, dR, dW, E, nsrc, mode, 6 + it % 5, dout, dsink);
      CK(hipDeviceSynchronize()); CK(hipMemcpy(cur.data(), dout, nout * 4, hipMemcpyDeviceToHost));
      long w_ = 0;
      for (size_t r = 0; r < nrow; ++r) { long wr = 0; for (size_t k = 0; k < 19 * 64; ++k) { const size_t i = r * 19 * 64 + k; if (memcmp(&cur[i], &ref[i], 4)) { ++wr; const double d = fabs((double)cur[i] - ref[i]) / (fabs((double)ref[i]) + 1e-30); if (d > worst) worst = d; } } w_ += wr; br += wr != 0; }
      bw += w_; bl += w_ != 0;
    }
    printf("matrix half %-4s: %d launches x %zu rows: %ld launches / %ld rows / %ld words differ (worst rel. %.2e)\n", mode ? "BUSY" : "idle", launches, nrow, bl, br, bw, worst);
  }
  return 0;
}
