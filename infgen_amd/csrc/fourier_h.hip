// k_fourier_h: FourierEmbedding.forward (reference infgen/modules/layers.py:142-160) on the fp16 matrix
// pipe with a three-term split, register-resident from the sin/cos features to the output row.
//
// Why: the fp32-input MFMA (32x32x2) runs at the fp32 vector rate (157 TFLOP/s); the f16 MFMA
// (32x32x16) is 16x faster.  Every operand is split into two fp16 numbers, x*2^s = hi + lo with hi the top
// 11 significand bits and lo the next 11 (power-of-two prescales chosen at pack time keep both in the
// fp16 normal range), and  x*w ~= hi_x*hi_w + hi_x*lo_w + lo_x*hi_w  is accumulated in fp32 by three
// MFMAs: operands represented to 2^-23 (round-to-nearest hi + lo, split.cuh), the dropped lo x lo term <= 2^-22 of a product;
// 16/3 = 5.3x the fp32 matrix rate.
// Logit error of a full rollout with every GEMM split like this stays at the fp32 noise floor
// (DESIGN.md section 5).
//
// Orientation: C[f][e] = sum_k W[f][k] X[k][e] - the WEIGHTS are the MFMA A operand (rows = output
// features), the 16 edges of a wave are the B operand / the C columns (16x16x32 MFMA).  A lane then
// holds 32 of the 128 features of ONE edge (lanes ^ 16, ^ 32, ^ 48 hold the rest), so
//   * LayerNorm over the features is an in-lane sum plus two cross-lane exchanges, and
//   * the C registers of one GEMM, split to fp16, ARE the B fragments of the next GEMM (the k order
//     of the weights is permuted to the C-register order at pack time) - no LDS round trip.
// LDS only holds the weights: quarter-matrices (one k-step of 32: 8 feature tiles x hi/lo x 1 KB
// fragments = 16 KB) are streamed global -> LDS with global_load_lds_dwordx4 through a ring of five
// buffers (four quarters in flight), shared by the 8 waves of the workgroup; the sequence of quarters is
// the same for every tile, so the pipeline runs across tile boundaries.
// Workgroup = 8 waves x 16 edges, one workgroup per CU (two waves per SIMD, whose VALU phases - sin/cos,
// LayerNorm, splitting - overlap the partner's MFMA phases).
// Bring-up note: early 4-wave versions with two or three workgroups per CU produced wrong rows at random (16
// consecutive rows off by ~1e-2, ~1 wave-tile in 3000, different rows every run; independent of LDS-DMA, of where the
// A fragments came from, of spills, of the fp64 slow path; never with one workgroup per CU).  The cause was never
// pinned down, and the final code no longer shows it: a 4-wave / three-buffer / two-workgroups-per-CU build of this
// file was bitwise reproducible over 160 re-runs (100 k .. 1 M rows, n = 3 and 4) and only 3 % faster, so the
// one-workgroup configuration stays; tests/test_ops_gpu.py::test_fourier_split_is_deterministic re-runs a 350 k-row
// launch bitwise, test_attn_split_is_deterministic does the same for k_attn_h (which does run two per CU).
#include "kernels.h"
#include "layout.h"
#include "tile.cuh"
#include "split.cuh"

namespace ig {


// sin/cos: three-constant Cody-Waite reduction by pi/2 with fused multiply-adds (the products n * c are
// exact inside the fma), then the classic minimax polynomials on [-pi/4, pi/4] (Cephes sinf/cosf
// coefficients); about 1 ulp, like the sleef kernels torch uses on the CPU.  Arguments of 1e5 rad and more
// (never produced by in-radius geometry) are reduced in fp64 with a three-term pi/2, good to 2^40 rad.
// (the fast path is branch-free so that the evaluations of a feature group interleave; arguments beyond the fast reduction are
// redone by sincos_big behind ONE wave-uniform branch per group - sincos_group)
__device__ __forceinline__ void sincos_fast(float z, float& s, float& c) {
  const float n = rintf(z * 0.636619772f);
  float r = __builtin_fmaf(n, -1.57079601e+00f, z);
  r = __builtin_fmaf(n, -3.13916473e-07f, r);
  r = __builtin_fmaf(n, -5.39030253e-15f, r);
  const int qi = (int)n;
  const float r2 = r * r;
  float ps = __builtin_fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f);
  ps = __builtin_fmaf(ps, r2, -1.6666654611e-1f);
  ps = __builtin_fmaf(ps * r2, r, r);
  float pc = __builtin_fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f);
  pc = __builtin_fmaf(pc, r2, 4.166664568298827e-2f);
  pc = __builtin_fmaf(pc * r2, r2, __builtin_fmaf(-0.5f, r2, 1.0f));
  // quadrant: swap by bit 0, negate by bit 1 - as bit operations on the VGPRs (selects would tie up SGPR masks)
  const unsigned m = 0u - ((unsigned)qi & 1u);
  const unsigned ups = __float_as_uint(ps), upc = __float_as_uint(pc);
  const unsigned sv = (upc & m) | (ups & ~m);
  const unsigned cv = (ups & m) | (upc & ~m);
  s = __uint_as_float(sv ^ (((unsigned)qi & 2u) << 30));
  c = __uint_as_float(cv ^ ((((unsigned)qi + 1u) & 2u) << 30));
}
// Round 4: the same pair through the hardware's v_sin_f32 / v_cos_f32 (argument in revolutions, valid on [-256, 256]).  The reference
// takes sin / cos of the ROUNDED fp32 product z = x f 2 pi, so z itself is reduced: r = z / (2 pi) - rint(z / (2 pi)) with 1 / (2 pi) as
// two floats and the products inside fused multiply-adds (z * c_hi is exact there: 24 x 24 bits; the result is rounded once at
// |r| <= 0.5, 3e-8 of a revolution = 1.9e-7 rad, for every |z| < 1e8).  Four vector instructions and two transcendental ones per
// pair instead of ~30: the sine / cosine features were a quarter of the kernel's vector work, and the kernel is bound by that work
// (tools/bench_fourier.py with the matrix phases removed: 292 of 409 us; with the vector phases removed: 195).  Measured against fp64
// (tools/hw_sincos_probe.hip): max |error| 2.6e-7 (sincos_fast: 6e-8) - the order of the three-term split's own error per product (2^-22) that
// consumes the features, and 30 x under the 2^-17 of the packed 24-bit output rows.  IG_FH_HWSIN=0 restores sincos_fast.
#ifndef IG_FH_HWSIN
#define IG_FH_HWSIN 1
#endif
__device__ __forceinline__ void sincos_hw(float z, float& s, float& c) {
  const float k = rintf(z * 0.15915494f);
  float r = __builtin_fmaf(z, 0.15915494f, -k);              // 0x3e22f983
  r = __builtin_fmaf(z, 6.4206382e-09f, r);                  // 1 / (2 pi) - 0x3e22f983
  s = __builtin_amdgcn_sinf(r);
  c = __builtin_amdgcn_cosf(r);
}
// (results by value: reference arguments of a non-inlined function would go through scratch memory)
__device__ __noinline__ f32x2 sincos_big(float z) {
  const double zd = (double)z;
  const double nd = rint(zd * 0.63661977236758134308);
  double rd = __builtin_fma(-nd, 1.57079632679489655800e+00, zd);
  rd = __builtin_fma(-nd, 6.12323399573676603587e-17, rd);
  rd = __builtin_fma(-nd, -1.49738490485916983e-33, rd);
  const float r = (float)rd;
  const int qi = (int)(nd - 4.0 * floor(nd * 0.25));
  const float r2 = r * r;
  float ps = __builtin_fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f);
  ps = __builtin_fmaf(ps, r2, -1.6666654611e-1f);
  ps = __builtin_fmaf(ps * r2, r, r);
  float pc = __builtin_fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f);
  pc = __builtin_fmaf(pc, r2, 4.166664568298827e-2f);
  pc = __builtin_fmaf(pc * r2, r2, __builtin_fmaf(-0.5f, r2, 1.0f));
  const unsigned m = 0u - ((unsigned)qi & 1u);
  const unsigned ups = __float_as_uint(ps), upc = __float_as_uint(pc);
  const unsigned sv = (upc & m) | (ups & ~m);
  const unsigned cv = (ups & m) | (upc & ~m);
  return f32x2{__uint_as_float(sv ^ (((unsigned)qi & 2u) << 30)), __uint_as_float(cv ^ ((((unsigned)qi + 1u) & 2u) << 30))};
}
// eight features of one input: z[p] = x * f[p] * 2 pi as the reference forms it (left to right, fp32), sine and cosine of each
__device__ __forceinline__ void sincos_group(float x, const float (&fr)[8], float (&sn)[8], float (&cs)[8]) {
  float z[8], zmax = 0.f;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    z[p] = x * fr[p] * 2.0f * PI_F;
    zmax = fmaxf(zmax, fabsf(z[p]));          // (a NaN argument gives NaN on the fast path as well)
#if IG_FH_HWSIN
    sincos_hw(z[p], sn[p], cs[p]);
#else
    sincos_fast(z[p], sn[p], cs[p]);
#endif
  }
  if (__builtin_expect(__any(!(zmax < 1.0e5f)), 0)) {
#pragma unroll
    for (int p = 0; p < 8; ++p)          // (unrolled: a runtime index would move z / sn / cs to scratch memory)
      if (!(fabsf(z[p]) < 1.0e5f)) { const f32x2 r = sincos_big(z[p]); sn[p] = r[0]; cs[p] = r[1]; }
  }
}

// Packed 24-bit rows (kernels.h: R24_ROW_BYTES) of a wave's 16 edges, written as whole 128-byte lines.  A lane holds four
// columns of every feature tile of ONE row, so stored from the registers a row leaves as 16 pieces of 8 and 4 bytes per lane -
// sixteen store instructions of scattered fragments per wave, ~4,400 cycles of store issue at every tile end (s_memtime).  Here the
// two planes pass through a wave-private LDS image (padded rows: conflict-free both ways) and leave as six 1 KB instructions whose
// 16- / 8-lane runs are the contiguous 256 / 128 bytes of a row's plane.  No barrier: LDS operations of one wave retire in order.
constexpr int R24_IMG_BYTES = 16 * 272;
__device__ __forceinline__ void store_rows_r24(const f32x4 (&v)[8], char* img, char* out, int row0, int E, int lane) {
  const int j = lane & 15, rg = lane >> 4;
  unsigned lo[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    unsigned u[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned b = __float_as_uint(v[t][k]);
      u[k] = b + 0x7fu + ((b >> 8) & 1u);                        // round to nearest even at bit 8
    }
    *reinterpret_cast<uint2*>(img + j * 272 + 32 * t + 8 * rg) = make_uint2((u[0] >> 16) | (u[1] & 0xffff0000u), (u[2] >> 16) | (u[3] & 0xffff0000u));
    lo[t] = ((u[0] >> 8) & 0xffu) | (u[1] & 0xff00u) | ((u[2] << 8) & 0xff0000u) | ((u[3] << 16) & 0xff000000u);
  }
  uint4 h[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) h[k] = *reinterpret_cast<const uint4*>(img + (4 * k + (lane >> 4)) * 272 + (lane & 15) * 16);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = row0 + 4 * k + (lane >> 4);
    if (r < E) *reinterpret_cast<uint4*>(out + (size_t)r * R24_ROW_BYTES + (lane & 15) * 16) = h[k];
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) *reinterpret_cast<unsigned*>(img + j * 144 + 16 * t + 4 * rg) = lo[t];
  uint4 l[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) l[k] = *reinterpret_cast<const uint4*>(img + (8 * k + (lane >> 3)) * 144 + (lane & 7) * 16);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int r = row0 + 8 * k + (lane >> 3);
    if (r < E) *reinterpret_cast<uint4*>(out + (size_t)r * R24_ROW_BYTES + R24_LO_PLANE + (lane & 7) * 16) = l[k];
  }
}

#ifndef IG_FH_TRACE
#define IG_FH_TRACE 0
#endif
#if IG_FH_TRACE
// timing experiment (variant builds only): s_memtime of waves 0 and 4 of workgroup 0 at both sides of every slot barrier
__device__ unsigned long long g_fh_trace[512];
#endif

template <int TERMS>
__global__ __launch_bounds__(FH_NT, 2) void k_fourier_h(FourierArgs a) {
#if IG_FH_AP
#include "fourier_ap_body.inc"
#else
#include "fourier_h_body.inc"
#endif
}

// up to three independent edge sets in one launch (gridDim.y = sets): the three Fourier embeddings of a decode step side by
// side when the sets are too small to fill the chip one after the other (a single tile costs 28 - 36 quarters = 35 - 45 us)
template <int TERMS>
__global__ __launch_bounds__(FH_NT, 2) void k_fourier_h_multi(FourierMultiArgs m) {
  const FourierArgs& a = m.set[blockIdx.y];
#if IG_FH_AP
#include "fourier_ap_body.inc"
#else
#include "fourier_h_body.inc"
#endif
}

#if !IG_BF16_OPERANDS
template __global__ void k_fourier_h_multi<3>(FourierMultiArgs);
#endif
template __global__ void k_fourier_h_multi<1>(FourierMultiArgs);
#if !IG_BF16_OPERANDS
template __global__ void k_fourier_h<3>(FourierArgs);
#endif
template __global__ void k_fourier_h<1>(FourierArgs);

}  // namespace ig

#if IG_FH_TRACE && !defined(k_fourier_h)          // (not again in the translation unit of the 12-wave variant, fourier_h12.hip)
extern "C" int infgen_debug_fh_trace(unsigned long long* host512) {
  return (int)hipMemcpyFromSymbol(host512, HIP_SYMBOL(ig::g_fh_trace), 512 * sizeof(unsigned long long));
}
#endif
