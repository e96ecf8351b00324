// k_fourier_h: FourierEmbedding.forward (reference infgen/modules/layers.py:142-160) on the fp16 matrix
// pipe with a three-term split, register-resident from the sin/cos features to the output row.
//
// Why: the fp32-input MFMA (32x32x2) runs at the fp32 vector rate (157 TFLOP/s); the f16 MFMA
// (32x32x16) is 16x faster.  Every operand is split into two fp16 numbers, x*2^s = hi + lo with hi the top
// 11 significand bits and lo the next 11 (power-of-two prescales chosen at pack time keep both in the
// fp16 normal range), and  x*w ~= hi_x*hi_w + hi_x*lo_w + lo_x*hi_w  is accumulated in fp32 by three
// MFMAs: 2^-21 relative error per product (fp32: 2^-24), 16/3 = 5.3x the fp32 matrix rate.
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
__device__ __forceinline__ void sincos_fast(float z, float& s, float& c) {
  const float n = rintf(z * 0.636619772f);
  float r = __builtin_fmaf(n, -1.57079601e+00f, z);
  r = __builtin_fmaf(n, -3.13916473e-07f, r);
  r = __builtin_fmaf(n, -5.39030253e-15f, r);
  int qi = (int)n;
  if (__builtin_expect(!(fabsf(z) < 1.0e5f), 0)) {
    const double zd = (double)z;
    const double nd = rint(zd * 0.63661977236758134308);
    double rd = __builtin_fma(-nd, 1.57079632679489655800e+00, zd);
    rd = __builtin_fma(-nd, 6.12323399573676603587e-17, rd);
    rd = __builtin_fma(-nd, -1.49738490485916983e-33, rd);
    r = (float)rd;
    qi = (int)(nd - 4.0 * floor(nd * 0.25));
  }
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

template <int TERMS>
__global__ __launch_bounds__(FH_NT, 2) void k_fourier_h(FourierArgs a) {
#include "fourier_h_body.inc"
}

// up to three independent edge sets in one launch (gridDim.y = sets): the three Fourier embeddings of a decode step side by
// side when the sets are too small to fill the chip one after the other (a single tile costs 28 - 36 quarters = 35 - 45 us)
template <int TERMS>
__global__ __launch_bounds__(FH_NT, 2) void k_fourier_h_multi(FourierMultiArgs m) {
  const FourierArgs& a = m.set[blockIdx.y];
#include "fourier_h_body.inc"
}

template __global__ void k_fourier_h_multi<3>(FourierMultiArgs);
template __global__ void k_fourier_h_multi<1>(FourierMultiArgs);
template __global__ void k_fourier_h<3>(FourierArgs);
template __global__ void k_fourier_h<1>(FourierArgs);

}  // namespace ig
