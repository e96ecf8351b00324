// Tile-level device helpers for gfx950 (CDNA4): fp32 MFMA row-tile GEMMs, LayerNorm on LDS
// tiles, wave reductions and the scalar math helpers that mirror the reference's
// elementwise formulas.  Everything here assumes 64-wide wavefronts and 256-thread
// workgroups (4 waves) unless stated otherwise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace ig {

constexpr int D = 128;        // hidden size (reference configs/ours_standard.yaml:60)
constexpr int H = 8;          // heads
constexpr int DH = 16;        // head dim
constexpr int TR = 32;        // rows per GEMM tile (one 32x32 MFMA M-tile)
constexpr int LDT = 132;      // LDS row stride in floats: 16-B aligned rows, conflict-free b128 reads
constexpr int NT = 256;       // threads per GEMM workgroup
constexpr float LN_EPS = 1e-5f;
constexpr float PI_F = 3.14159274101257324f;      // float(math.pi)
constexpr float TWO_PI_F = 6.28318548202514648f;  // float(2 * math.pi)
constexpr float HALF_PI_F = 1.57079637050628662f; // float(math.pi / 2)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- DPP lane exchanges (VALU data path, no LDS round trip like ds_bpermute) -----------------------
// value of lane (l ^ 1), (l ^ 2) via quad_perm; lane (7 - l%8) via row_half_mirror; lane (l ^ 8) via row_ror:8
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_xor1(float v) { return dpp_f<0xB1>(v); }        // quad_perm [1,0,3,2]
__device__ __forceinline__ float dpp_xor2(float v) { return dpp_f<0x4E>(v); }        // quad_perm [2,3,0,1]
__device__ __forceinline__ float dpp_half_mirror(float v) { return dpp_f<0x141>(v); }  // l -> 7 - l (within 8)
__device__ __forceinline__ float dpp_xor8(float v) { return dpp_f<0x128>(v); }       // row_ror:8
// sum over the 8 lanes {l & ~7 ... l | 7}; every lane gets the total
__device__ __forceinline__ float sum8(float v) {
  v += dpp_xor1(v);
  v += dpp_xor2(v);
  v += dpp_half_mirror(v);
  return v;
}

// ---- elementwise formulas of the reference ------------------------------------------------
// infgen/utils/func.py:58-62   wrap_angle(a) = -pi + (a + pi) % (2 pi)   (python-style remainder)
__device__ __forceinline__ float wrap_angle(float a) {
  float x = a + PI_F;
  float r = fmodf(x, TWO_PI_F);
  if (r != 0.0f && r < 0.0f) r += TWO_PI_F;
  return -PI_F + r;
}
// infgen/utils/func.py:30-34   atan2(cx*ny - cy*nx, (ctr * nbr).sum(-1))
// torch's sum() starts from +0, so for a zero neighbour vector (a stationary agent, or column 0 of
// a row without bos) the second argument is +0 even when both products are -0: atan2 then gives
// +-0 where the naive expression gives +-pi.  Keep the explicit "+0.0f +" (IEEE: +0 + -0 = +0).
__device__ __forceinline__ float angle_between(float cx, float cy, float nx, float ny) {
  volatile float zero = 0.0f;
  return atan2f(cx * ny - cy * nx, (zero + cx * nx) + cy * ny);
}
__device__ __forceinline__ float norm2(float x, float y) { return sqrtf(x * x + y * y); }

// ---- packed weight layout -----------------------------------------------------------------
// A logical GEMM operand B[k][n] (= torch Linear.weight[n][k]) with K padded to a multiple of
// 8 and N to a multiple of 32 is stored as Wp[k/8][n][k%8]: one lane reads the 4 k-values it
// feeds to 4 consecutive MFMAs with ONE 16-byte load, and a wave reads 1 KiB contiguous.
__host__ __device__ __forceinline__ size_t packed_index(int k, int n, int N) {
  return ((size_t)(k >> 3) * N + n) * 8 + (k & 7);
}

// acc(32x32 tile at columns [n0, n0+32)) += A[32][K] (LDS, row stride lda) * B (packed, N cols)
// v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31].
// Lane-half kh reads k = 8g + 4kh .. 8g + 4kh + 3 with one ds_read_b128 / global_load_dwordx4
// and uses them in 4 MFMAs, so MFMA m of group g contracts k in {8g + m, 8g + 4 + m}.
template <int K>
__device__ __forceinline__ void mfma_32x32(f32x16& acc, const float* __restrict__ A, int lda,
                                           const float* __restrict__ Wp, int N, int n0) {
  const int lane = lane_id();
  const int r = lane & 31, kh = lane >> 5;
  const float* ap = A + r * lda + 4 * kh;
  const float* bp = Wp + ((size_t)(n0 + r)) * 8 + 4 * kh;
#pragma unroll 8
  for (int g = 0; g < K / 8; ++g) {
    const float4 a = *reinterpret_cast<const float4*>(ap + g * 8);
    const float4 b = *reinterpret_cast<const float4*>(bp + (size_t)g * N * 8);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
}
// ---- B-fragment prefetch across a barrier -------------------------------------------------------
// The first two k-groups of a GEMM's B operand can be requested BEFORE the __syncthreads() that
// publishes its A tile (plain global loads stay in flight across s_barrier), which hides the L2
// latency that otherwise opens every short GEMM phase.
struct BPre { float4 b0, b1; };
__device__ __forceinline__ BPre b_prefetch(const float* __restrict__ Wp, int N, int n0) {
  const int lane = lane_id();
  const float* bp = Wp + ((size_t)(n0 + (lane & 31))) * 8 + 4 * (lane >> 5);
  BPre p;
  p.b0 = *reinterpret_cast<const float4*>(bp);
  p.b1 = *reinterpret_cast<const float4*>(bp + (size_t)N * 8);
  return p;
}
// same contraction as mfma_32x32<K> with the first two B groups already in registers and a
// two-deep software pipeline on the rest (K/8 >= 2)
template <int K>
__device__ __forceinline__ void mfma_32x32_pf(f32x16& acc, const float* __restrict__ A, int lda,
                                              const float* __restrict__ Wp, int N, int n0, const BPre& pre) {
  const int lane = lane_id();
  const int r = lane & 31, kh = lane >> 5;
  const float* ap = A + r * lda + 4 * kh;
  const float* bp = Wp + ((size_t)(n0 + r)) * 8 + 4 * kh;
  constexpr int G = K / 8;
  float4 bq[2] = {pre.b0, pre.b1};
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float4 a = *reinterpret_cast<const float4*>(ap + g * 8);
    const float4 b = bq[g & 1];
    if (g + 2 < G) bq[g & 1] = *reinterpret_cast<const float4*>(bp + (size_t)(g + 2) * N * 8);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
}

// ---- explicit half-GEMM pipelining ---------------------------------------------------------------
// hipcc keeps only ~2 B-operand loads in flight inside the unrolled K loop (each s_waitcnt then eats an
// L2 round trip).  BHalf holds the B fragments of 8 k-groups (64 k-values x 32 columns, 32 VGPRs); the
// kernels load the NEXT half (or the next GEMM's first half) before issuing the 32 MFMAs of the current
// one and pin that order with sched_barrier, so ~2048 MFMA cycles cover every load.
struct BHalf { float4 v[8]; };
__device__ __forceinline__ BHalf b_load_half(const float* __restrict__ Wp, int N, int n0, int half) {
  const int lane = lane_id();
  const float* bp = Wp + ((size_t)(n0 + (lane & 31))) * 8 + 4 * (lane >> 5) + (size_t)(8 * half) * N * 8;
  BHalf h;
#pragma unroll
  for (int g = 0; g < 8; ++g) h.v[g] = *reinterpret_cast<const float4*>(bp + (size_t)g * N * 8);
  return h;
}
// acc += A[32][64 k of `half`] * B-half ; A tile in LDS (row stride lda)
__device__ __forceinline__ void mfma_half(f32x16& acc, const float* __restrict__ A, int lda, int half, const BHalf& b) {
  const int lane = lane_id();
  const float* ap = A + (lane & 31) * lda + 4 * (lane >> 5) + 64 * half;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const float4 a = *reinterpret_cast<const float4*>(ap + g * 8);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.v[g].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.v[g].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.v[g].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.v[g].w, acc, 0, 0, 0);
  }
}
#define IG_PIN() __builtin_amdgcn_sched_barrier(0)

// One K = 128 GEMM (32 rows x 32 columns per wave) in the pipelined form: `cur` holds the first B half on
// entry (requested before the barrier that published A) and the NEXT GEMM's first half on exit.
template <typename NextLoad>
__device__ __forceinline__ void gemm128(f32x16& acc, const float* __restrict__ A, int lda,
                                        const float* __restrict__ Wp, int N, int n0, BHalf& cur, NextLoad next) {
  const BHalf h1 = b_load_half(Wp, N, n0, 1);
  IG_PIN();
  mfma_half(acc, A, lda, 0, cur);
  IG_PIN();
  cur = next();
  IG_PIN();
  mfma_half(acc, A, lda, 1, h1);
  IG_PIN();
}

// runtime-K variant (K multiple of 8)
__device__ __forceinline__ void mfma_32x32_rt(f32x16& acc, const float* __restrict__ A, int lda, int K,
                                              const float* __restrict__ Wp, int N, int n0) {
  const int lane = lane_id();
  const int r = lane & 31, kh = lane >> 5;
  const float* ap = A + r * lda + 4 * kh;
  const float* bp = Wp + ((size_t)(n0 + r)) * 8 + 4 * kh;
  for (int g = 0; g < K / 8; ++g) {
    const float4 a = *reinterpret_cast<const float4*>(ap + g * 8);
    const float4 b = *reinterpret_cast<const float4*>(bp + (size_t)g * N * 8);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
}

// C/D layout of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ int acc_row(int reg) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane_id() >> 5); }
__device__ __forceinline__ int acc_col() { return lane_id() & 31; }

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.0f;
  return z;
}

// acc (+ bias[col]) -> LDS tile O[32][ldo] at columns n0..n0+31
__device__ __forceinline__ void acc_to_lds(const f32x16& acc, float* O, int ldo, int n0, const float* __restrict__ bias) {
  const int col = n0 + acc_col();
  const float b = bias ? bias[col] : 0.0f;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) O[acc_row(reg) * ldo + col] = acc[reg] + b;
}

// Stage a [32][128] fp32 tile from global rows (row r -> src + rowoff(r)) into LDS with stride LDT.
// Rows >= nvalid are zero-filled.  256 threads, 16 B per thread per pass.
template <typename RowPtr>
__device__ __forceinline__ void stage_rows_128(float* dst, RowPtr rowptr, int nvalid) {
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int idx = p * NT + t;       // 0..1023 float4 slots
    const int r = idx >> 5, c4 = idx & 31;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nvalid) {
      const float* src = rowptr(r);
      if (src) v = *reinterpret_cast<const float4*>(src + c4 * 4);
    }
    *reinterpret_cast<float4*>(dst + r * LDT + c4 * 4) = v;
  }
}

// Write a [32][128] LDS tile to global rows (coalesced float4), rows < nvalid
template <typename RowPtr>
__device__ __forceinline__ void unstage_rows_128(const float* src, RowPtr rowptr, int nvalid) {
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int idx = p * NT + t;
    const int r = idx >> 5, c4 = idx & 31;
    if (r < nvalid) {
      float* d = rowptr(r);
      if (d) *reinterpret_cast<float4*>(d + c4 * 4) = *reinterpret_cast<const float4*>(src + r * LDT + c4 * 4);
    }
  }
}

// ---- row segments: thread t of a 256-thread workgroup owns row t >> 3 of a 32-row tile and the four
// 4-column groups {4*(t&7) + 32*j}; a LayerNorm reduction is 3 DPP steps over the 8 lanes of the row.
struct RowSeg { float4 v[4]; };

__device__ __forceinline__ int seg_row() { return threadIdx.x >> 3; }
__device__ __forceinline__ int seg_col(int j) { return 4 * (threadIdx.x & 7) + 32 * j; }

__device__ __forceinline__ RowSeg seg_zero() {
  RowSeg r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  return r;
}
__device__ __forceinline__ RowSeg seg_load(const float* row /* this thread's row, may be null */) {
  RowSeg r = seg_zero();
  if (row) {
#pragma unroll
    for (int j = 0; j < 4; ++j) r.v[j] = *reinterpret_cast<const float4*>(row + seg_col(j));
  }
  return r;
}
__device__ __forceinline__ void seg_store(float* row, const RowSeg& r) {
  if (row) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(row + seg_col(j)) = r.v[j];
  }
}
__device__ __forceinline__ RowSeg seg_add(const RowSeg& a, const RowSeg& b) {
  RowSeg r;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    r.v[j] = make_float4(a.v[j].x + b.v[j].x, a.v[j].y + b.v[j].y, a.v[j].z + b.v[j].z, a.v[j].w + b.v[j].w);
  return r;
}
// LayerNorm of the row this thread's segment belongs to (biased variance, eps 1e-5, torch.nn.LayerNorm);
// gamma == nullptr -> affine-free
__device__ __forceinline__ RowSeg seg_layernorm(RowSeg x, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, bool relu) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) s += (x.v[j].x + x.v[j].y) + (x.v[j].z + x.v[j].w);
  const float mean = sum8(s) * (1.0f / 128.0f);
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    x.v[j].x -= mean; x.v[j].y -= mean; x.v[j].z -= mean; x.v[j].w -= mean;
    q += (x.v[j].x * x.v[j].x + x.v[j].y * x.v[j].y) + (x.v[j].z * x.v[j].z + x.v[j].w * x.v[j].w);
  }
  const float var = sum8(q) * (1.0f / 128.0f);
  const float rstd = 1.0f / sqrtf(var + LN_EPS);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float4 g = make_float4(1.f, 1.f, 1.f, 1.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gamma) {
      g = *reinterpret_cast<const float4*>(gamma + seg_col(j));
      b = *reinterpret_cast<const float4*>(beta + seg_col(j));
    }
    float4 y = make_float4(x.v[j].x * rstd * g.x + b.x, x.v[j].y * rstd * g.y + b.y,
                           x.v[j].z * rstd * g.z + b.z, x.v[j].w * rstd * g.w + b.w);
    if (relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
    x.v[j] = y;
  }
  return x;
}

// LayerNorm over the 128 columns of each row of a [32][128] LDS tile; all 256 threads take part.
// res != nullptr -> dst = res + LN(src).  src == dst (in place) is allowed.
__device__ __forceinline__ void ln_tile(const float* src, int lds_, float* dst, int ldd,
                                        const float* __restrict__ gamma, const float* __restrict__ beta, bool relu,
                                        const float* res = nullptr, int ldr = 0) {
  const int row = seg_row();
  RowSeg y = seg_layernorm(seg_load(src + row * lds_), gamma, beta, relu);
  if (res) y = seg_add(y, seg_load(res + row * ldr));
  seg_store(dst + row * ldd, y);
}


// Warm workgroups of a small launch (k_attn_hs, k_edge_fused's one-group variant).  A decode step streams 24 MB of
// layer weights through eight 4 MB L2s, so every kernel of the chain finds its weights in the Infinity Cache, and a 16-row group's
// chain waits ~3 us for a 64 KB matrix it has to fetch itself.  A launch of a few dozen groups leaves most CUs idle: workgroups
// wg0 .. wg0 + 8 per_xcd - 1 (consecutive workgroup ids go round the XCDs) read the two regions once - the per_xcd workgroups of
// an XCD one slice each - and exit.  The regions are what the NEXT kernels of the chain will read (the library's sublayer loop
// sets them: api.hip), so the lines have a whole kernel of head start.  Placement is only assumed for speed.
struct WarmArgs { const char* p[2]; int len[2]; int wg0, per_xcd; };

// workgroup b of a launch (b >= w.wg0) reads its slice of the regions once and returns true
__device__ __forceinline__ bool warm_l2(const WarmArgs& w, int b, int tid, int nth) {
  if (w.per_xcd <= 0 || b < w.wg0) return false;
  const int k = (b - w.wg0) >> 3, K = w.per_xcd, step = 16 * nth;
  unsigned sink = 0;
  if (k < K) {
    for (int r = 0; r < 2; ++r) {
      const char* base = r ? w.p[1] : w.p[0];
      const int len = r ? w.len[1] : w.len[0];
      if (!base || len <= 0) continue;
      const int per = ((len + K - 1) / K + step - 1) / step * step;       // slice per workgroup, whole passes of the workgroup
      const int hi = min(len, (k + 1) * per);
      for (int off = k * per + tid * 16; off < hi; off += step) {
        const uint4 v = *reinterpret_cast<const uint4*>(base + off);
        sink ^= v.x ^ v.y ^ v.z ^ v.w;
      }
    }
  }
  asm volatile("" :: "v"(sink));                  // (the loads must not be dropped; nothing is stored)
  return true;
}

}  // namespace ig
