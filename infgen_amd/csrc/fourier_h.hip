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

constexpr int FH_WAVES = 8;
constexpr int FH_NT = 64 * FH_WAVES;     // threads per workgroup
constexpr int FH_TILE = 16 * FH_WAVES;   // edges per workgroup tile (16 per wave)

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
__device__ __forceinline__ void fourier_h_body(const FourierArgs& a) {
  __shared__ __attribute__((aligned(16))) unsigned short Wb[RING][QUARTER];    // 80 KB: also keeps the CU to ONE workgroup
  __shared__ __attribute__((aligned(16))) float Vt[FH_VEC_SIZE];
  const int E = a.count_dev ? min(*a.count_dev, a.e_cap) : a.e_cap;
  const int ntiles = (E + FH_TILE - 1) / FH_TILE;
  if ((int)blockIdx.x >= ntiles) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int j = lane & 15, rg = lane >> 4;
  if (a.prof_rows && blockIdx.x == 0 && tid == 0) atomicAdd(a.prof_rows + a.n, (unsigned long long)E);
  const float* vec = a.pack + fourier_pack_size_f32(a.n);
  const unsigned short* wg = reinterpret_cast<const unsigned short*>(vec + FH_VEC_SIZE);
  __shared__ const unsigned short* seg_ptr[2];
  __shared__ int seg_n[2];
  // quarter-matrices per tile, consumed in storage order (W1_i W2_i per dim, then W3); the table modes skip the last dim's
  // eight quarters (1) or stage nothing else (2)
  const int mode = a.dt_mode;
  const int i_lo = mode == 2 ? a.n - 1 : 0, i_hi = mode == 1 ? a.n - 1 : a.n;
  if (tid == 0) {
    seg_ptr[0] = wg + (size_t)8 * i_lo * QUARTER; seg_n[0] = mode == 0 ? 4 * (2 * a.n + 1) : 8 * (i_hi - i_lo);
    seg_ptr[1] = wg + (size_t)8 * a.n * QUARTER; seg_n[1] = 4;
  }
  for (int i = tid; i < FH_VEC_SIZE; i += FH_NT) Vt[i] = vec[i];
  __syncthreads();
  QuarterStream<FH_NT, RING> qs;
  qs.init(seg_ptr, seg_n, mode == 1 ? 2 : 1, (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x, Wb, tid);
  auto take = [&]() { return qs.take(); };
  const float inv2 = Vt[FH_HDR + 4], inv3 = Vt[FH_HDR + 5], fscale = Vt[FH_HDR + 6];
  const float* tail = Vt + FH_TAIL;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int e = tile * FH_TILE + w * 16 + j;
    const bool valid = e < E;
    float4 rawv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid && mode != 2) rawv = *reinterpret_cast<const float4*>(a.raw + 4 * (size_t)e);
    if (mode == 2) rawv = make_float4(-(float)e, -(float)e, -(float)e, -(float)e);
    u32x4 Bh[4], Bl[4];
    f32x4 acc2[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int i = i_lo; i < i_hi; ++i) {
      const float x = (i == 0) ? rawv.x : (i == 1) ? rawv.y : (i == 2) ? rawv.z : rawv.w;
      const float* fq = Vt + FH_FREQ + i * 64;
      const float* dv = Vt + FH_DIM0 + i * FHD_SIZE;
      // features: k-step s (0, 1) slot p is cos of frequency 32 s + 8 rg + p, k-step s + 2 its sin
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float4 f0 = *reinterpret_cast<const float4*>(fq + 32 * s + 8 * rg);
        const float4 f1 = *reinterpret_cast<const float4*>(fq + 32 * s + 8 * rg + 4);
        const float fr[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
        float cs[8], sn[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          // reference: x.unsqueeze(-1) * freqs.weight * 2 * math.pi  (left to right, fp32)
          const float z = x * fr[p] * 2.0f * PI_F;
          sincos_fast(z, sn[p], cs[p]);
          cs[p] *= fscale; sn[p] *= fscale;
        }
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) {
          unsigned hi, lo;
          split_pair(cs[2 * wd], cs[2 * wd + 1], hi, lo);
          Bh[s][wd] = hi; Bl[s][wd] = lo;
          split_pair(sn[2 * wd], sn[2 * wd + 1], hi, lo);
          Bh[s + 2][wd] = hi; Bl[s + 2][wd] = lo;
        }
      }
      f32x4 acc1[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) gemm_quarter<TERMS>(acc1, take(), Bh[s], Bl[s], lane);
      {
        const float inv1 = Vt[FH_HDR + i];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          acc1[t] = fma4(acc1[t], splat4(inv1), fma4(splat4(x), lds4(dv + FHD_WX + 16 * t + 4 * rg), lds4(dv + FHD_B1 + 16 * t + 4 * rg)));
        }
      }
      ln_regs<true, true>(acc1, dv + FHD_G1, dv + FHD_BE1, rg);     // gamma/beta carry the activation prescale
      regs_to_frags(acc1, Bh, Bl);
#pragma unroll
      for (int s = 0; s < 4; ++s) gemm_quarter<TERMS>(acc2, take(), Bh[s], Bl[s], lane);
    }
    if (mode == 2) {
      if (valid) {
        float* o = a.out + (size_t)e * a.ldo;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const f32x4 v = acc2[t] * splat4(inv2);
          *reinterpret_cast<float4*>(o + 16 * t + 4 * rg) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
      continue;
    }
    const float* dtrow = nullptr;
    if (mode == 1 && valid) {
      const float g = a.n == 4 ? rawv.w : a.n == 3 ? rawv.z : a.n == 2 ? rawv.y : rawv.x;
      dtrow = a.dt_tab + 128 * min(max((int)(-g), 0), DT_TAB_ROWS - 1);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int f = 16 * t + 4 * rg;
      f32x4 c = lds4(tail + FHT_B2SUM + f);
      if (a.cat && valid) c += lds4(a.cat + (size_t)e * a.ldcat + f);
      if (dtrow) c += lds4(dtrow + f);
      acc2[t] = fma4(acc2[t], splat4(inv2), c);
    }
    ln_regs<true, true>(acc2, tail + FHT_G2, tail + FHT_BE2, rg);
    regs_to_frags(acc2, Bh, Bl);
    f32x4 acc3[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc3[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) gemm_quarter<TERMS>(acc3, take(), Bh[s], Bl[s], lane);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      acc3[t] = fma4(acc3[t], splat4(inv3), lds4(tail + FHT_B3 + 16 * t + 4 * rg));
    }
    if (a.normalize) ln_regs<false, false>(acc3, nullptr, nullptr, rg);
    if (valid && a.out_r24) {
      // packed 24-bit rows (kernels.h: R24_ROW_BYTES): this lane's four columns of every feature tile
      char* o = reinterpret_cast<char*>(a.out) + (size_t)e * R24_ROW_BYTES;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        unsigned u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned b = __float_as_uint(acc3[t][k]);
          u[k] = b + 0x7fu + ((b >> 8) & 1u);                        // round to nearest even at bit 8
        }
        *reinterpret_cast<uint2*>(o + 2 * (16 * t + 4 * rg)) = make_uint2((u[0] >> 16) | (u[1] & 0xffff0000u), (u[2] >> 16) | (u[3] & 0xffff0000u));
        *reinterpret_cast<unsigned*>(o + R24_LO_PLANE + 16 * t + 4 * rg) =
            ((u[0] >> 8) & 0xffu) | (u[1] & 0xff00u) | ((u[2] << 8) & 0xff0000u) | ((u[3] << 16) & 0xff000000u);
      }
    } else if (valid) {
      float* o = a.out + (size_t)e * a.ldo;
#pragma unroll
      for (int t = 0; t < 8; ++t)
        *reinterpret_cast<float4*>(o + 16 * t + 4 * rg) = make_float4(acc3[t][0], acc3[t][1], acc3[t][2], acc3[t][3]);
    }
  }
}

template <int TERMS>
__global__ __launch_bounds__(FH_NT, 1) void k_fourier_h(FourierArgs a) { fourier_h_body<TERMS>(a); }

// up to three independent edge sets in one launch (gridDim.y = sets): the three Fourier embeddings of a decode step side by
// side when the sets are too small to fill the chip one after the other (a single tile costs 28 - 36 quarters = 35 - 45 us)
template <int TERMS>
__global__ __launch_bounds__(FH_NT, 1) void k_fourier_h_multi(FourierMultiArgs m) { fourier_h_body<TERMS>(m.set[blockIdx.y]); }

template __global__ void k_fourier_h_multi<3>(FourierMultiArgs);
template __global__ void k_fourier_h_multi<1>(FourierMultiArgs);
template __global__ void k_fourier_h<3>(FourierArgs);
template __global__ void k_fourier_h<1>(FourierArgs);

}  // namespace ig
