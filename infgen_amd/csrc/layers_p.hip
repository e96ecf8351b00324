// k_layers_p: ALL sublayers of a decode step (6 x [temporal, map -> agent, agent <-> agent] AttentionLayers, reference
// infgen/modules/agent_decoder.py:2123-2158 over infgen/modules/layers.py:61-113) in ONE launch, for few rows (up to 256
// 16-row groups = the BASELINE-literal batches: 64 scenes on one GPU, 8 scenes per GPU of an 8-way shard).
//
// The per-sublayer pair k_edge_fused<.., 1, ..> + k_attn_hs is a chain of ~36 dependent launches per step whose node half
// repeats the whole rows' LayerNorm / hi-lo split in all eight waves (two waves per SIMD: ~3,000 cycles per GEMM stage of
// mostly redundant vector work).  Here a workgroup of eight waves OWNS one 16-row group for the whole step:
//   * wave w owns feature tile w (= head w) of every vector of the rows: the residual stream x, agg, the gate, the FFN hidden
//     layer ... live as ONE f32x4 per lane (row j = lane & 15, features 16 w + 4 (lane >> 4) ..); nothing is replicated;
//   * a GEMM's B operand is published by its owners: each wave splits ITS tile (per (row, tile) power-of-two scale, fp16
//     hi / lo) into LDS - 16 B per lane - and every wave reads the eight tiles back as ready-made fragments;
//     the products run as v_mfma_f32_16x16x16_f16 per source tile (the halves of the 16x16x32 A fragments of the packed
//     layout, layout.h AH_*: slots 0..3 of k-step s are tile 2 s, slots 4..7 tile 2 s + 1), each tile's accumulator
//     multiplied by its own inverse scale - same three-term split (2^-21 per product), no row-wide maximum to exchange;
//   * LayerNorm statistics are merged from per-tile (mean, M2) pairs (Chan's update: as accurate as two passes);
//   * the edge loop (edge_attn.cuh: EdgeAcc, k_edge_fused's phase 2) takes q and the absorbed query u from LDS where the
//     previous sublayer's node part left them, two rows per wave; phase 3 (W'_vr z) leaves agg in the owner's registers;
//   * temporal and map sublayers need nothing from other rows of the step; the agent sublayer reads the K / V rows of its
//     whole scene, so the A_cap / 16 workgroups of a scene meet once per layer at a counter in global memory (release /
//     acquire at agent scope; all workgroups are co-resident: one per CU, grid <= 256).  K / V of the agent set are double
//     buffered by layer parity, so no workgroup overwrites rows a slower one still reads.
// Workgroup barriers are wg_barrier() (split.cuh: LDS operations retired + s_barrier): __syncthreads() would also wait for every
// weight fragment in flight (vmcnt(0)) - ten times per sublayer - and the three fragment sets requested ahead would never overlap
// anything.  Only LDS carries data between the waves of a workgroup here.
// Arithmetic differs from k_attn_hs / k_attn_h in rounding only (scale granularity, LayerNorm merge order); every result is
// a fixed-order reduction - bitwise reproducible (tests/test_ops_gpu.py::test_layers_p_*).
#include "kernels.h"
#include "layout.h"
#include "tile.cuh"
#include "split.cuh"
#include "attn_h.cuh"
#include "edge_attn.cuh"

#ifndef IG_LP_NOLOAD
#define IG_LP_NOLOAD 0           // timing experiments (wrong results): 1 no weight-fragment loads, IG_LP_NOMFMA no products in the
#endif                          // GEMM stages, IG_LP_NOEDGE empty edge lists - what each costs in wall time
#ifndef IG_LP_NOMFMA
#define IG_LP_NOMFMA 0
#endif
#ifndef IG_LP_X32
#define IG_LP_X32 7              // bit 0: W1 on 16x16x32 products (one scale per row of LN_ffpre(x)); bit 1: q / k / v, gate-x, self too
#endif                          // (LN_dst(x)); bit 2: W2 (the hidden layer's row maxima exchanged first)
#ifndef IG_LP_SYNC1
#define IG_LP_SYNC1 1              // one wave carries the agent-scope fences of the scene counter (0: every wave, the first version: +5 % of the launch)
#endif
#ifndef IG_LP_NOAUX
#define IG_LP_NOAUX 0            // (timing) no W'kr / W'vr loads, IG_LP_NOSYNC no scene counter, IG_LP_NOKV no K / V stores
#endif
#ifndef IG_LP_NOSYNC
#define IG_LP_NOSYNC 0
#endif
#ifndef IG_LP_NOKV
#define IG_LP_NOKV 0
#endif
#ifndef IG_LP_NOEDGE
#define IG_LP_NOEDGE 0
#endif
#ifndef IG_LP_TRACE
#define IG_LP_TRACE 0          // 1: s_memtime stamps at the phase boundaries (tools/lp_trace.py; build with -DIG_LP_TRACE=1)
#endif

namespace ig {

namespace {

constexpr int LP_LDU = H * D + 4;          // row stride of the U / Z tile in floats (k_edge_fused's)
constexpr int LP_LDA = D + 4;
#ifndef IG_LP_G
#define IG_LP_G 6
#endif
constexpr int LP_G = IG_LP_G;              // edges per trip of the edge loop

struct AFragP {                // the four k-steps of ONE feature tile of a 128 x 128 matrix (attn_hs.hip: AFrag)
  v8h h[4], l[4];
  __device__ __forceinline__ void load(const unsigned short* W, int w, int lane) {
#if IG_LP_NOLOAD          // timing experiment (wrong results): the node part without its weight stream
    asm volatile("" : "+v"(h[0]), "+v"(l[0]));
    return;
#endif
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      h[s] = *reinterpret_cast<const v8h*>(W + (size_t)s * QUARTER + w * 1024 + lane * 8);
      l[s] = *reinterpret_cast<const v8h*>(W + (size_t)s * QUARTER + w * 1024 + 512 + lane * 8);
    }
  }
};

__device__ __forceinline__ v4h half_lo(v8h a) { return __builtin_shufflevector(a, a, 0, 1, 2, 3); }
__device__ __forceinline__ v4h half_hi(v8h a) { return __builtin_shufflevector(a, a, 4, 5, 6, 7); }

// own tile -> LDS as B fragment pieces of source tile w: [w][lane] = {hi(v0, v1), hi(v2, v3), lo(v0, v1), lo(v2, v3)}, and the
// tile's inverse scale of row j at SC[j][w]
__device__ __forceinline__ void publish_frag(f32x4 v, uint4* FR, float* SC, int w, int lane) {
  float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
  m = xor_lanes_max(m);
  unsigned eb = __float_as_uint(m) >> 23;
  eb = min(max(eb, 15u), 253u);
  const float sc = __uint_as_float((268u - eb) << 23), inv = __uint_as_float((eb - 14u) << 23);
  unsigned h0, l0, h1, l1;
  split_pair(v[0] * sc, v[1] * sc, h0, l0);
  split_pair(v[2] * sc, v[3] * sc, h1, l1);
  FR[w * 64 + lane] = make_uint4(h0, h1, l0, l1);
  if ((lane >> 4) == 0) SC[(lane & 15) * 8 + w] = inv;
}

// per-tile LayerNorm statistics of row j: (mean over the tile's 16 features, sum of squared deviations from it)
__device__ __forceinline__ void publish_stats(f32x4 v, float2* ST, int w, int lane) {
  const float mt = xor_lanes((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 16.0f);
  const float d0 = v[0] - mt, d1 = v[1] - mt, d2 = v[2] - mt, d3 = v[3] - mt;
  const float m2 = xor_lanes(fmaf(d0, d0, d1 * d1) + fmaf(d2, d2, d3 * d3));
  if ((lane >> 4) == 0) ST[(lane & 15) * 8 + w] = make_float2(mt, m2);
}
// (x32 path) the same plus the tile's largest |value - tile mean| + |tile mean| bound pieces: (min, max) of the tile, in a second
// array - every wave derives the SAME bound of the row's LayerNorm output from the eight pairs
__device__ __forceinline__ void publish_minmax(f32x4 v, float2* MM, int w, int lane) {
  const float mx = xor_lanes_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
  const float mn = -xor_lanes_max(-fminf(fminf(v[0], v[1]), fminf(v[2], v[3])));
  if ((lane >> 4) == 0) MM[(lane & 15) * 8 + w] = make_float2(mn, mx);
}
__device__ __forceinline__ float row_dev(const float2* MM, int j, float mean) {
  const float4* p = reinterpret_cast<const float4*>(MM + j * 8);
  const float4 a = p[0], b = p[1], c = p[2], d = p[3];
  const float lo = fminf(fminf(fminf(a.x, a.z), fminf(b.x, b.z)), fminf(fminf(c.x, c.z), fminf(d.x, d.z)));
  const float hi = fmaxf(fmaxf(fmaxf(a.y, a.w), fmaxf(b.y, b.w)), fmaxf(fmaxf(c.y, c.w), fmaxf(d.y, d.w)));
  return fmaxf(hi - mean, mean - lo);
}
// A GEMM operand whose rows have ONE scale for all eight tiles multiplies on v_mfma_f32_16x16x32_f16 (tiles 2 s, 2 s + 1 are the
// halves of k-step s): 12 instead of 24 matrix instructions per GEMM and wave.  LayerNorm outputs get that scale without another
// exchange: |LN(x)_i| <= dev * rstd * max|gamma| + max|beta|, dev = max_i |x_i - mean| from the per-tile (min, max) that travel
// with the statistics, max|gamma| / max|beta| from the pack header (packing.py) - every wave derives the same power of two.  (A
// value far below the bound keeps its 22 bits unless its low half goes subnormal: absolute error <= 2^-39 of the bound.)
__device__ __forceinline__ unsigned scale_bits(float bound) {
  unsigned eb = __float_as_uint(bound) >> 23;
  return min(max(eb, 15u), 253u);
}
__device__ __forceinline__ float scale_inv(unsigned eb) { return __uint_as_float((eb - 14u) << 23); }
__device__ __forceinline__ void publish_frag_common(f32x4 v, uint4* FR, unsigned eb, int w, int lane) {
  const float sc = __uint_as_float((268u - eb) << 23);
  unsigned h0, l0, h1, l1;
  split_pair(v[0] * sc, v[1] * sc, h0, l0);
  split_pair(v[2] * sc, v[3] * sc, h1, l1);
  FR[w * 64 + lane] = make_uint4(h0, h1, l0, l1);
}
// merged over the eight tiles: mean and 1 / sqrt(var + eps) of row j (biased variance, eps 1e-5: split.cuh ln_stats)
__device__ __forceinline__ void row_stats(const float2* ST, int j, float& mean, float& rstd) {
  const float4* p = reinterpret_cast<const float4*>(ST + j * 8);
  const float4 a = p[0], b = p[1], c = p[2], d = p[3];
  mean = (((a.x + a.z) + (b.x + b.z)) + ((c.x + c.z) + (d.x + d.z))) * 0.125f;
  const float e0 = a.x - mean, e1 = a.z - mean, e2 = b.x - mean, e3 = b.z - mean;
  const float e4 = c.x - mean, e5 = c.z - mean, e6 = d.x - mean, e7 = d.z - mean;
  const float within = ((a.y + a.w) + (b.y + b.w)) + ((c.y + c.w) + (d.y + d.w));
  const float between = (fmaf(e0, e0, e1 * e1) + fmaf(e2, e2, e3 * e3)) + (fmaf(e4, e4, e5 * e5) + fmaf(e6, e6, e7 * e7));
  const float var_eps = fmaf(16.0f, between, within) * (1.0f / 128.0f) + LN_EPS;
  rstd = ln_rstd(var_eps);
}
__device__ __forceinline__ f32x4 ln_own(f32x4 v, float mean, float rstd, const float* g, const float* b) {
  const f32x4 y = (v - splat4(mean)) * splat4(rstd);
  return fma4(y, lds4(g), lds4(b));
}

// out tile w = sum over source tiles t of inv_t[row] * (W[16 w .., tile t] x B_t): 8 x 3 MFMAs (K = 16), four source tiles at a
// time (four independent accumulator chains keep the matrix pipe busy; eight at once cost 64 registers of fragments + accumulators)
__device__ __forceinline__ f32x4 gemm_tiles(const AFragP& f, const uint4* FR, const float* SC, int lane) {
  __builtin_amdgcn_sched_barrier(0);           // (independent GEMMs side by side would multiply the live fragments / accumulators)
  const int j = lane & 15;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
#if IG_LP_NOMFMA
  { const uint4 b0 = FR[lane]; o[0] = __uint_as_float(b0.x) + SC[j * 8] + (float)f.h[0][0]; return o; }
#endif
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const float4 sv = *reinterpret_cast<const float4*>(SC + j * 8 + 4 * g);
    const float inv[4] = {sv.x, sv.y, sv.z, sv.w};
    f32x4 acc[4];
    uint4 b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) b[t] = FR[(4 * g + t) * 64 + lane];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int tt = 4 * g + t;
      const v4h bh = __builtin_bit_cast(v4h, u32x2{b[t].x, b[t].y});
      const v4h ah = (tt & 1) ? half_hi(f.h[tt >> 1]) : half_lo(f.h[tt >> 1]);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int tt = 4 * g + t;
      const v4h bl = __builtin_bit_cast(v4h, u32x2{b[t].z, b[t].w});
      const v4h ah = (tt & 1) ? half_hi(f.h[tt >> 1]) : half_lo(f.h[tt >> 1]);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl, acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int tt = 4 * g + t;
      const v4h bh = __builtin_bit_cast(v4h, u32x2{b[t].x, b[t].y});
      const v4h al = (tt & 1) ? half_hi(f.l[tt >> 1]) : half_lo(f.l[tt >> 1]);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh, acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) o = fma4(acc[t], splat4(inv[t]), o);
  }
  __builtin_amdgcn_sched_barrier(0);
  return o;
}

// x32 path: out tile w (unscaled) = W[16 w .., :] x B, B read from the published fragments (tiles 2 s | 2 s + 1 = k-step s)
__device__ __forceinline__ f32x4 gemm32(const AFragP& f, const uint4* FR, int lane) {
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc[4];
  u32x4 bh[4], bl[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint4 t0 = FR[(2 * s) * 64 + lane], t1 = FR[(2 * s + 1) * 64 + lane];
    bh[s] = u32x4{t0.x, t0.y, t1.x, t1.y};
    bl[s] = u32x4{t0.z, t0.w, t1.z, t1.w};
  }
#pragma unroll
  for (int s = 0; s < 4; ++s)
    acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.h[s], __builtin_bit_cast(v8h, bh[s]), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
  for (int s = 0; s < 4; ++s)
    acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.h[s], __builtin_bit_cast(v8h, bl[s]), acc[s], 0, 0, 0);
#pragma unroll
  for (int s = 0; s < 4; ++s)
    acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.l[s], __builtin_bit_cast(v8h, bh[s]), acc[s], 0, 0, 0);
  const f32x4 o = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __builtin_amdgcn_sched_barrier(0);
  return o;
}

}  // namespace

template <bool R24, int ROWS>
__global__ __launch_bounds__(512, 1) void k_layers_p(LayersPArgs a) {
  __shared__ __attribute__((aligned(16))) float UZ[16 * LP_LDU];        // u, then z; the FFN's hidden-layer fragments in between
  __shared__ __attribute__((aligned(16))) float AG[16 * LP_LDA];        // q tile, then agg
  __shared__ __attribute__((aligned(16))) float SG[16 * H];
  __shared__ __attribute__((aligned(16))) float Vt[VT_SIZE + 4];
  __shared__ __attribute__((aligned(16))) uint4 FRx[512];               // LN_dst(x): the layer's query / gate / self input
  __shared__ __attribute__((aligned(16))) uint4 FRp[2][512];
  __shared__ __attribute__((aligned(16))) float SCx[128], SCp[2][128], SCH[4][128];
  __shared__ __attribute__((aligned(16))) float2 STp[2][128];
  __shared__ __attribute__((aligned(16))) float2 MMp[2][128];           // (x32 path) per-tile (min, max), with the statistics
  __shared__ __attribute__((aligned(16))) float HM[128];                 // (x32 path) the hidden layer's per-(row, wave) maxima
  uint4* FRH = reinterpret_cast<uint4*>(UZ);                            // [4][512] (32 KB of the 66 KB tile)

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 15, rg = lane >> 4;
  // R rows per workgroup: 16, or 8 for the smallest batches (twice the workgroups: ONE row per wave in the edge loop, the phase the
  // agent sublayers spend most of their time in; lanes j >= 8 then shadow rows j - 8 - same loads, no stores, never read)
  constexpr int R = ROWS;                                               // (compile time: a run-time row count costs ~3 %)
  const int gps = a.A_cap / R;                                          // workgroups per scene
  const int L = a.num_layers;
  // a scene's workgroups on ONE XCD (consecutive workgroups go to consecutive XCDs): its K / V rows and its counter share an L2
  int grp = blockIdx.x;
  if (a.xcd_order) {
    const int per = 8 * gps, bq = grp / per, br = grp % per;
    grp = bq * per + (br % 8) * gps + br / 8;
  }
  const int r0 = a.row0 + R * grp, row = r0 + (j & (R - 1));        // (row0: first row of this launch's chunk of scenes)
  const bool own_row = j < R;                                           // (this lane's row is not a shadow: it may store)
  const int scene = r0 / a.A_cap;
  const int own = 16 * w + 4 * rg;                                      // this lane's four features
  int* ctr = a.sync + scene;
  int fp = 0, sp = 0;                                                   // ping-pong indices of FRp / SCp and STp
  float inv_x = 1.0f;                                                   // (x32 path) inverse scale of the rows' LN_dst(x) fragments
  // diagnostics (INFGEN_LP_TRACE): s_memtime stamps of workgroup 0's wave 0 at the phase boundaries -> a.trace[]
  int n_trace = 0;
  auto STAMP = [&](int id) __attribute__((always_inline)) {
#if IG_LP_TRACE
    if (a.trace && blockIdx.x == 0 && tid == 0 && n_trace < 1020) {
      a.trace[2 * n_trace] = (unsigned long long)id; a.trace[2 * n_trace + 1] = __builtin_amdgcn_s_memtime(); ++n_trace;
    }
#endif
  };

  f32x4 x;
  { const float4 t = *reinterpret_cast<const float4*>(a.X + (size_t)row * D + own); x = f32x4{t.x, t.y, t.z, t.w}; }
  // Three fragment sets in rotation over the fifteen 128 x 128 matrices a sublayer consumes (consumption order: gate-agg fa,
  // gate-x fb, self fc, Wo fa, W1 chunks 0..3 fb fc fa fb, W2 chunks 0..3 fc fa fb fc, the NEXT layer's Wq fa, Wk fb, Wv fc): a set
  // is requested again right after its use, two matrices ahead of its next one; the first three of a sublayer are requested
  // BEFORE its edge loop and arrive under it.  (Measured, 8 scenes: a fourth set - three matrices ahead, but gate-x / self then
  // requested after the edge loop because four live sets do not fit next to its accumulators - was 8 % slower; the CU's
  // vector-memory path takes the loads in order at ~64 B per cycle, so a longer queue only delays the set needed next.  Helper
  // workgroups on idle CUs that touched the next sublayer's pack to keep it in the XCD's L2, and all sublayers reading ONE pack,
  // changed nothing: the stream is bound inside the CU, not by where the weights come from.)
  AFragP fa, fb, fc;

  // the vector table of a sublayer (post part of P, pre part of NP): two 16-byte slots per thread (attn_hs.hip)
  auto table_fetch = [&](const float* P, const float* NP, float4& tv0, float4& tv1, int& o0, int& o1) __attribute__((always_inline)) {
    const int d0 = 4 * tid, d1 = 4 * (tid + 512);
    const float* anyp = P ? P : NP;
    const float* ts0 = attn_table_src(d0, P, NP, 0);
    const float* ts1 = d1 < VT_SIZE ? attn_table_src(d1, P, NP, 0) : nullptr;
    tv0 = *reinterpret_cast<const float4*>(ts0 ? ts0 : anyp);
    tv1 = *reinterpret_cast<const float4*>(ts1 ? ts1 : anyp);
    o0 = ts0 ? d0 : VT_SIZE;
    o1 = ts1 ? d1 : VT_SIZE;
  };

  // The edge lists of the wave's two rows, requested a sublayer ahead (three dependent round trips - counts / offsets, source
  // indices, then the rows themselves - would otherwise open every edge loop).  The 16 rows are ranked by their edge counts in
  // every wave (16 compares); wave w takes the rows of rank w and 15 - w: longest with shortest.
  int eE[2], eB[2], eR[2], eS[2];
  int el_cnt, el_off;
  // (two halves: the loads' return is in order, so whoever consumes a load waits for every load issued before it - the counts /
  // offsets are requested BEFORE a batch of weight fragments and consumed after the GEMMs that wait for those fragments anyway)
  auto edge_lists_request = [&](const EdgeSet& es) __attribute__((always_inline)) {
    el_cnt = es.cnt[r0 + (lane & (R - 1))];
    el_off = es.off[r0 + (lane & (R - 1))];
  };
  auto edge_lists_resolve = [&](const EdgeSet& es) __attribute__((always_inline)) {
    const int rl = lane & 15;
    const int off = el_off;
    const int cnt = rl < R ? el_cnt : -1;                        // (shadow rows rank last: ranks 0 .. R - 1 are the real rows)
    int rank = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int ck = __shfl(cnt, k, 64);
      rank += (ck > cnt || (ck == cnt && k < rl)) ? 1 : 0;
    }
#pragma unroll
    for (int i = 0; i < (R == 16 ? 2 : 1); ++i) {
      // R = 16: rows of rank w and 15 - w; R = 8: rank w only; R = 4: rank w >> 1, and the row's list is halved between waves 2 q
      // and 2 q + 1 (first half | second half), whose partial softmax states are merged afterwards
      const int want = R == 4 ? (w >> 1) : (i == 0 ? w : 15 - w);
      const unsigned long long m = __ballot(rank == want && lane < 16);
      const int r = __builtin_ctzll(m);
      eR[i] = r;
      int E = __builtin_amdgcn_readlane(cnt, r), B = __builtin_amdgcn_readlane(off, r);
      if constexpr (R == 4) {
        const int first = (E + 1) >> 1;
        B += (w & 1) ? first : 0;
        E = (w & 1) ? E - first : first;
      }
      eE[i] = E;
      eB[i] = B;
      eS[i] = es.src[B + max(min(lane, E - 1), 0)];              // (an empty list reads the entry at its offset: in bounds, unused)
    }
  };

  // ---- pre part of layer NP on the rows' current x: LN_dst -> q (-> u), k, v.  q / u stay in LDS for the edge loop.
  // In: fa / fb / fc = Wq / Wk / Wv of NP.
  auto node_pre = [&](const float* NP, float* nK, float* nV) __attribute__((always_inline)) {
    const unsigned short* pre = reinterpret_cast<const unsigned short*>(NP + AH_PRE);
    const float* hdr = Vt + VT_N_HDR;
    STAMP(16);
    publish_stats(x, STp[sp], w, lane);
    if (IG_LP_X32 & 2) publish_minmax(x, MMp[sp], w, lane);
    wg_barrier();
    STAMP(17);
    float mean, rstd;
    row_stats(STp[sp], j, mean, rstd);
    const f32x4 xn = ln_own(x, mean, rstd, Vt + VT_N_LN_G + own, Vt + VT_N_LN_B + own);
    if (IG_LP_X32 & 2) {
      const unsigned ebx = scale_bits(fmaf(row_dev(MMp[sp], j, mean) * rstd, hdr[12], hdr[13]));
      inv_x = scale_inv(ebx);
      publish_frag_common(xn, FRx, ebx, w, lane);
    } else {
      publish_frag(xn, FRx, SCx, w, lane);
    }
    sp ^= 1;
    wg_barrier();
    STAMP(18);
    f32x4 q = (IG_LP_X32 & 2) ? gemm32(fa, FRx, lane) * splat4(inv_x) : gemm_tiles(fa, FRx, SCx, lane);
    q = fma4(q, splat4(hdr[0]), lds4(Vt + VT_N_BQ + own));
    if (nK) {
      f32x4 kk = (IG_LP_X32 & 2) ? gemm32(fb, FRx, lane) * splat4(inv_x) : gemm_tiles(fb, FRx, SCx, lane);
      f32x4 vv = (IG_LP_X32 & 2) ? gemm32(fc, FRx, lane) * splat4(inv_x) : gemm_tiles(fc, FRx, SCx, lane);
      kk = kk * splat4(hdr[2]);
      vv = fma4(vv, splat4(hdr[3]), lds4(Vt + VT_N_BV + own));
      if (own_row && !IG_LP_NOKV) {
        *reinterpret_cast<float4*>(nK + (size_t)row * D + own) = make_float4(kk[0], kk[1], kk[2], kk[3]);
        *reinterpret_cast<float4*>(nV + (size_t)row * D + own) = make_float4(vv[0], vv[1], vv[2], vv[3]);
      }
    }
    STAMP(19);
    // u_w = q_w W'_kr,w (attn_hs.hip / edge_fused.hip phase 1): K = 16, the head's query is this wave's own tile
    {
      const unsigned short* Wk = pre + (size_t)(4 + (w >> 1)) * QUARTER + (size_t)((w & 1) * 8) * 2 * 256 + lane * 4;
      v4h ah[8], al[8];
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) {
#if IG_LP_NOAUX
        asm volatile("" : "=v"(ah[ct]), "=v"(al[ct]));
#else
        ah[ct] = *reinterpret_cast<const v4h*>(Wk + (ct * 2) * 256);
        al[ct] = *reinterpret_cast<const v4h*>(Wk + (ct * 2 + 1) * 256);
#endif
      }
      *reinterpret_cast<float4*>(AG + j * LP_LDA + own) = make_float4(q[0], q[1], q[2], q[3]);
      float m = fmaxf(fmaxf(fabsf(q[0]), fabsf(q[1])), fmaxf(fabsf(q[2]), fabsf(q[3])));
      m = xor_lanes_max(m);
      unsigned ebits = __float_as_uint(m) >> 23;
      ebits = min(max(ebits, 15u), 253u);
      const float sc = __uint_as_float((268u - ebits) << 23), inv = __uint_as_float((ebits - 14u) << 23);
      u32x2 qh, ql;
      {
        unsigned hi, lo;
        split_pair(q[0] * sc, q[1] * sc, hi, lo); qh[0] = hi; ql[0] = lo;
        split_pair(q[2] * sc, q[3] * sc, hi, lo); qh[1] = hi; ql[1] = lo;
      }
      const v4h vqh = __builtin_bit_cast(v4h, qh), vql = __builtin_bit_cast(v4h, ql);
      const float cq = inv * hdr[1];
      f32x4 acc[8];
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[ct], vqh, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[ct], vql, acc[ct], 0, 0, 0);
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(al[ct], vqh, acc[ct], 0, 0, 0);
      float* urow = UZ + j * LP_LDU + w * D + 4 * rg;
#pragma unroll
      for (int ct = 0; ct < 8; ++ct)
        *reinterpret_cast<float4*>(urow + 16 * ct) = make_float4(acc[ct][0] * cq, acc[ct][1] * cq, acc[ct][2] * cq, acc[ct][3] * cq);
    }
  };

  // ---- one sublayer: edge loop over `es` -> phase 3 -> post part of P -> pre part of NP (q, u; k / v into nK / nV when given).
  // es_next: the edge set of the sublayer after this one (its lists are requested here)
  auto sublayer = [&](const EdgeSet& es, const EdgeSet& es_next, const float* Ksrc, const float* Vsrc, bool kv_once, const float* P,
                      const float* NP, float* nK, float* nV) __attribute__((always_inline)) {
    const unsigned short* post = reinterpret_cast<const unsigned short*>(P + AH_POST);
    const unsigned short* pre_any = reinterpret_cast<const unsigned short*>((NP ? NP : P) + AH_PRE);
    float4 tv0, tv1; int to0, to1;
    table_fetch(P, NP, tv0, tv1, to0, to1);         // consumed after the edge loop, like the first three matrices of the node part
    fa.load(post + 4 * QUARTER, w, lane);
    fb.load(post + 8 * QUARTER, w, lane);
    fc.load(post + 12 * QUARTER, w, lane);
    wg_barrier();                                   // q / u of the previous node part
    STAMP(1);

    // ---- edge loop (k_edge_fused's phase 2): this wave's two rows, lists requested a sublayer ago
    {
      const bool b3 = lane & 8;
      const unsigned lo8 = 8u * (unsigned)lane;
      auto ld8 = [&](const float* base, bool nt) {
        return ea_ld(reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + lo8), nt);
      };
      auto ld_r24 = [&](size_t e) {
        const char* rowp = reinterpret_cast<const char*>(es.rhat) + e * R24_ROW_BYTES;
        const unsigned hi = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(rowp + 4 * lane));
        const unsigned lo = __builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(rowp + R24_LO_PLANE + 2 * lane));
        return pk2{__uint_as_float((hi << 16) | ((lo & 0xffu) << 8)), __uint_as_float((hi & 0xffff0000u) | (lo & 0xff00u))};
      };
      for (int ri = 0; ri < (R == 16 ? 2 : 1); ++ri) {
        const int rl = ri == 0 ? eR[0] : eR[1];
        const int E = IG_LP_NOEDGE ? 0 : (ri == 0 ? eE[0] : eE[1]);
        const int e_base = ri == 0 ? eB[0] : eB[1];
        int sv = ri == 0 ? eS[0] : eS[1];
        float* uz = UZ + rl * LP_LDU;
        EdgeAcc<true> acc;
        acc.q = *reinterpret_cast<const float2*>(AG + rl * LP_LDA + 2 * lane);
        acc.load_u(uz, lane);
        acc.reset();
        for (int c0 = 0; c0 < E; c0 += 64) {
          const int mc = min(64, E - c0);
          if (c0 > 0) sv = es.src[e_base + c0 + min(lane, mc - 1)];
          for (int i0 = 0; i0 < mc; i0 += LP_G) {
            pk2 kb[LP_G], vb[LP_G], rb[LP_G];
#pragma unroll
            for (int s = 0; s < LP_G; ++s) {
              const int ic = min(i0 + s, mc - 1);
              const int sj = __builtin_amdgcn_readlane(sv, ic);
              kb[s] = ld8(Ksrc + (size_t)sj * D, kv_once);
              vb[s] = ld8(Vsrc + (size_t)sj * D, kv_once);
              if constexpr (R24) rb[s] = ld_r24((size_t)(e_base + c0 + ic));
              else rb[s] = ld8(es.rhat + (size_t)(e_base + c0 + ic) * D, true);
            }
#pragma unroll
            for (int s = 0; s < LP_G; ++s) acc.step(kb[s], vb[s], rb[s], i0 + s < mc, b3);
          }
        }
        if constexpr (R == 4) {
          // two waves share the row: the odd one parks its un-normalised state in spare rows of the U / Z tile (rows 4 .. 11 are
          // unused with four rows per workgroup), the even one merges it into its own (running maxima may differ) and finalises
          float* ps = UZ + (4 + 2 * (w >> 1)) * LP_LDU;
          if (w & 1) {
#pragma unroll
            for (int hd = 0; hd < H; ++hd) *reinterpret_cast<float2*>(ps + hd * D + 2 * lane) = make_float2(acc.zz[hd][0], acc.zz[hd][1]);
            *reinterpret_cast<float2*>(ps + LP_LDU + 2 * lane) = make_float2(acc.ag[0], acc.ag[1]);
            ps[LP_LDU + 128 + lane] = acc.m;
            ps[LP_LDU + 192 + lane] = acc.lsum;
          }
          wg_barrier();
          if (w & 1) continue;
          const float m1 = ps[LP_LDU + 128 + lane], l1 = ps[LP_LDU + 192 + lane];
          const float mm = fmaxf(acc.m, m1);
          const float s0 = acc.m == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(acc.m - mm);
          const float s1 = m1 == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m1 - mm);
          acc.lsum = fmaf(l1, s1, acc.lsum * s0);
          const float2 a1 = *reinterpret_cast<const float2*>(ps + LP_LDU + 2 * lane);
          acc.ag = pk2{fmaf(a1.x, s1, acc.ag[0] * s0), fmaf(a1.y, s1, acc.ag[1] * s0)};
#pragma unroll
          for (int hd = 0; hd < H; ++hd) {
            const float h0 = readlane_f(s0, 8 * hd), h1 = readlane_f(s1, 8 * hd);
            const float2 z1 = *reinterpret_cast<const float2*>(ps + hd * D + 2 * lane);
            acc.zz[hd] = pk2{fmaf(z1.x, h1, acc.zz[hd][0] * h0), fmaf(z1.y, h1, acc.zz[hd][1] * h0)};
          }
        }
        const float inv = 1.0f / (acc.lsum + 1e-16f);
        *reinterpret_cast<float2*>(AG + rl * LP_LDA + 2 * lane) = make_float2(acc.ag[0] * inv, acc.ag[1] * inv);
#pragma unroll
        for (int hd = 0; hd < H; ++hd) {
          const float ih = readlane_f(inv, 8 * hd);
          *reinterpret_cast<float2*>(uz + hd * D + 2 * lane) = make_float2(acc.zz[hd][0] * ih, acc.zz[hd][1] * ih);
        }
        if ((lane & 7) == 0) SG[rl * H + (lane >> 3)] = acc.lsum * inv;
      }
    }
    STAMP(2);
    *reinterpret_cast<float4*>(Vt + to0) = tv0;
    *reinterpret_cast<float4*>(Vt + to1) = tv1;
    edge_lists_request(es_next);                    // (the next sublayer's lists: three round trips under this node part)
    v8h p3h[4], p3l[4];                             // phase 3's W'vr fragments land under the barrier wait
    {
      const unsigned short* Wv = post + (size_t)(w >> 1) * QUARTER + (size_t)((w & 1) * 4) * 2 * 512 + lane * 8;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#if IG_LP_NOAUX
        asm volatile("" : "=v"(p3h[s]), "=v"(p3l[s]));
#else
        p3h[s] = *reinterpret_cast<const v8h*>(Wv + (s * 2) * 512);
        p3l[s] = *reinterpret_cast<const v8h*>(Wv + (s * 2 + 1) * 512);
#endif
      }
    }
    wg_barrier();
    STAMP(3);

    // ---- phase 3: agg' = agg + W'_vr,w z_w + b'_w sigma_w  (wave w = head w = feature tile w; result stays in registers)
    f32x4 ago;
    {
      const float* hdr = Vt + VT_HDR;
      const float* zrow = UZ + j * LP_LDU + w * D + 8 * rg;
      const float zs = 1024.0f, zinv = hdr[4] * (1.0f / 1024.0f);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4 z0 = *reinterpret_cast<const float4*>(zrow + 32 * s);
        const float4 z1 = *reinterpret_cast<const float4*>(zrow + 32 * s + 4);
        u32x4 bh, bl;
        unsigned hi, lo;
        split_pair(z0.x * zs, z0.y * zs, hi, lo); bh[0] = hi; bl[0] = lo;
        split_pair(z0.z * zs, z0.w * zs, hi, lo); bh[1] = hi; bl[1] = lo;
        split_pair(z1.x * zs, z1.y * zs, hi, lo); bh[2] = hi; bl[2] = lo;
        split_pair(z1.z * zs, z1.w * zs, hi, lo); bh[3] = hi; bl[3] = lo;
        const v8h vbh = __builtin_bit_cast(v8h, bh), vbl = __builtin_bit_cast(v8h, bl);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(p3h[s], vbh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(p3h[s], vbl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(p3l[s], vbh, acc, 0, 0, 0);
      }
      const float sg = SG[j * H + w];
      const f32x4 bvr = lds4(Vt + VT_BVR + own);
      const f32x4 ag = lds4(AG + j * LP_LDA + own);
#pragma unroll
      for (int r = 0; r < 4; ++r) ago[r] = ag[r] + (acc[r] * zinv + bvr[r] * sg);
    }

    // ---- post part (layers.py:94-99, 74-75, 110-112): gate / self / update, out projection + post-norm, FFN + post-norm
    const float* hdr = Vt + VT_HDR;
    STAMP(4);
    publish_frag(ago, FRp[fp], SCp[fp], w, lane);
    wg_barrier();                                   // (also: every wave has finished reading z, the q tile and SG)
    STAMP(5);
    f32x4 upd;
    {
      const f32x4 ga = gemm_tiles(fa, FRp[fp], SCp[fp], lane);
      fa.load(post + 16 * QUARTER, w, lane);                          // Wo
      const f32x4 gx = (IG_LP_X32 & 2) ? gemm32(fb, FRx, lane) * splat4(inv_x) : gemm_tiles(fb, FRx, SCx, lane);
      fb.load(post + 20 * QUARTER, w, lane);                          // W1, chunk 0
      const f32x4 sf = (IG_LP_X32 & 2) ? gemm32(fc, FRx, lane) * splat4(inv_x) : gemm_tiles(fc, FRx, SCx, lane);
      fc.load(post + (size_t)28 * QUARTER, w, lane);                  // W1, chunk 1
      fp ^= 1;
      const f32x4 bg = lds4(Vt + VT_BG + own), bs = lds4(Vt + VT_BS + own);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gate = 1.0f / (1.0f + expf(-((ga[r] * hdr[5] + gx[r] * hdr[5]) + bg[r])));
        upd[r] = ago[r] + gate * ((sf[r] * hdr[6] + bs[r]) - ago[r]);
      }
    }
    edge_lists_resolve(es_next);                    // (counts / offsets arrived with gate-x / self above; the indices travel under the FFN)
    STAMP(6);
    publish_frag(upd, FRp[fp], SCp[fp], w, lane);
    wg_barrier();
    STAMP(7);
    {
      f32x4 o = gemm_tiles(fa, FRp[fp], SCp[fp], lane);
      fp ^= 1;
      fa.load(post + (size_t)36 * QUARTER, w, lane);                  // W1, chunk 2
      o = fma4(o, splat4(hdr[7]), lds4(Vt + VT_BO + own));
      STAMP(8);
      publish_stats(o, STp[sp], w, lane);
      wg_barrier();
      STAMP(9);
      float mean, rstd;
      row_stats(STp[sp], j, mean, rstd); sp ^= 1;
      x += ln_own(o, mean, rstd, Vt + VT_LNP_G + own, Vt + VT_LNP_B + own);      // x1 = x + LN_post(out)
    }
    {
      publish_stats(x, STp[sp], w, lane);
      if (IG_LP_X32 & 1) publish_minmax(x, MMp[sp], w, lane);
      wg_barrier();
      STAMP(10);
      float mean, rstd;
      row_stats(STp[sp], j, mean, rstd);
      const f32x4 fin = ln_own(x, mean, rstd, Vt + VT_LNF_G + own, Vt + VT_LNF_B + own);
      float inv_f = 1.0f;
      if (IG_LP_X32 & 1) {
        const unsigned ebf = scale_bits(fmaf(row_dev(MMp[sp], j, mean) * rstd, hdr[10], hdr[11]));
        inv_f = scale_inv(ebf);
        publish_frag_common(fin, FRp[fp], ebf, w, lane);
      } else {
        publish_frag(fin, FRp[fp], SCp[fp], w, lane);
      }
      sp ^= 1;
      wg_barrier();
      STAMP(11);
      f32x4 hd4[4];
      float inv_h = 1.0f;
      // FFN: the four 128-wide chunks of the hidden layer (tile w of each), published as fragments behind ONE barrier
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        f32x4 hd;
        if (IG_LP_X32 & 1) hd = (cc == 1 ? gemm32(fc, FRp[fp], lane) : cc == 2 ? gemm32(fa, FRp[fp], lane) : gemm32(fb, FRp[fp], lane)) * splat4(inv_f);
        else hd = cc == 1 ? gemm_tiles(fc, FRp[fp], SCp[fp], lane) : cc == 2 ? gemm_tiles(fa, FRp[fp], SCp[fp], lane)
                                                                          : gemm_tiles(fb, FRp[fp], SCp[fp], lane);
        if (cc == 0) fb.load(post + (size_t)44 * QUARTER, w, lane);   // W1, chunk 3
        if (cc == 1) fc.load(post + (size_t)24 * QUARTER, w, lane);   // W2, chunk 0
        if (cc == 2) fa.load(post + (size_t)32 * QUARTER, w, lane);   // W2, chunk 1
        if (cc == 3) fb.load(post + (size_t)40 * QUARTER, w, lane);   // W2, chunk 2
        hd = fma4(hd, splat4(hdr[8]), lds4(Vt + VT_B1 + 128 * cc + own));
        hd = __builtin_elementwise_max(hd, splat4(0.f));
        if (IG_LP_X32 & 4) hd4[cc] = hd;
        else publish_frag(hd, FRH + cc * 512, SCH[cc], w, lane);
      }
      fp ^= 1;
      if (IG_LP_X32 & 4) {
        // one scale per row for the whole hidden layer: the waves exchange their per-row maxima first (one more barrier)
        const f32x4 m4 = __builtin_elementwise_max(__builtin_elementwise_max(hd4[0], hd4[1]), __builtin_elementwise_max(hd4[2], hd4[3]));
        const float mo = xor_lanes_max(fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3])));      // (ReLU outputs: no fabs)
        if (rg == 0) HM[j * 8 + w] = mo;
        wg_barrier();
        const float4 hm0 = *reinterpret_cast<const float4*>(HM + j * 8), hm1 = *reinterpret_cast<const float4*>(HM + j * 8 + 4);
        const unsigned ebh = scale_bits(fmaxf(fmaxf(fmaxf(hm0.x, hm0.y), fmaxf(hm0.z, hm0.w)), fmaxf(fmaxf(hm1.x, hm1.y), fmaxf(hm1.z, hm1.w))));
        inv_h = scale_inv(ebh);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) publish_frag_common(hd4[cc], FRH + cc * 512, ebh, w, lane);
      }
      STAMP(12);
      wg_barrier();
      STAMP(13);
      f32x4 f = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        f32x4 part;
        if (IG_LP_X32 & 4) part = (cc == 1 ? gemm32(fa, FRH + cc * 512, lane) : cc == 2 ? gemm32(fb, FRH + cc * 512, lane) : gemm32(fc, FRH + cc * 512, lane)) * splat4(inv_h);
        else part = cc == 1 ? gemm_tiles(fa, FRH + cc * 512, SCH[cc], lane) : cc == 2 ? gemm_tiles(fb, FRH + cc * 512, SCH[cc], lane)
                                                                                  : gemm_tiles(fc, FRH + cc * 512, SCH[cc], lane);
        // W2 chunk 3, then the next layer's Wq, Wk, Wv - unconditional (a sublayer without a pre part / without k, v re-reads
        // matrices it ignores): a conditional load is a basic block of its own whose results hipcc merges with copies and waits
        if (cc == 0) fc.load(post + (size_t)48 * QUARTER, w, lane);
        if (cc == 1) fa.load(pre_any, w, lane);
        if (cc == 2) fb.load(pre_any + 8 * QUARTER, w, lane);
        if (cc == 3) fc.load(pre_any + 12 * QUARTER, w, lane);
        f = fma4(part, splat4(hdr[9]), f);
      }
      f = f + lds4(Vt + VT_B2 + own);
      STAMP(14);
      publish_stats(f, STp[sp], w, lane);
      wg_barrier();                                 // (also: the hidden-layer fragments in the U / Z tile are dead)
      STAMP(15);
      float m2, r2;
      row_stats(STp[sp], j, m2, r2); sp ^= 1;
      x += ln_own(f, m2, r2, Vt + VT_LNO_G + own, Vt + VT_LNO_B + own);          // x2 = x1 + LN_ffpost(ffn)
    }
    if (NP) node_pre(NP, nK, nV);
    STAMP(20);
  };

  // ---- the step: pre part of the first temporal layer, then 6 x (temporal, map, agent)
  {
    float4 tv0, tv1; int to0, to1;
    table_fetch(nullptr, a.attn_t[0], tv0, tv1, to0, to1);
    const unsigned short* pre = reinterpret_cast<const unsigned short*>(a.attn_t[0] + AH_PRE);
    edge_lists_request(a.et);
    fa.load(pre, w, lane);
    fb.load(pre + 8 * QUARTER, w, lane);
    fc.load(pre + 12 * QUARTER, w, lane);
    edge_lists_resolve(a.et);
    *reinterpret_cast<float4*>(Vt + to0) = tv0;
    *reinterpret_cast<float4*>(Vt + to1) = tv1;
    wg_barrier();
    node_pre(a.attn_t[0], a.ringK[0] + a.slot_off, a.ringV[0] + a.slot_off);
  }
  // ONE copy of the sublayer body: sublayer k = 3 i + kind (0 temporal, 1 map -> agent, 2 agent <-> agent of layer i)
  for (int k = 0; k < 3 * L; ++k) {
    const int i = k / 3, kind = k - 3 * i;
    const bool last = k + 1 == 3 * L;
    float* Ka = a.Ka[i & 1];
    float* Va = a.Va[i & 1];
    const EdgeSet es = kind == 0 ? a.et : kind == 1 ? a.em : a.ea;
    const EdgeSet es_next = kind == 0 ? a.em : kind == 1 ? a.ea : a.et;
    const float* Ksrc = kind == 0 ? a.ringK[i] : kind == 1 ? a.mapK[i] : Ka;
    const float* Vsrc = kind == 0 ? a.ringV[i] : kind == 1 ? a.mapV[i] : Va;
    const float* P = kind == 0 ? a.attn_t[i] : kind == 1 ? a.attn_m[i] : a.attn_a[i];
    const float* NP = kind == 0 ? a.attn_m[i] : kind == 1 ? a.attn_a[i] : last ? nullptr : a.attn_t[i + 1];
    float* nK = kind == 0 ? nullptr : kind == 1 ? Ka : last ? nullptr : a.ringK[i + 1] + a.slot_off;
    float* nV = kind == 0 ? nullptr : kind == 1 ? Va : last ? nullptr : a.ringV[i + 1] + a.slot_off;
    STAMP(30 + kind);
    if (kind == 2 && !IG_LP_NOSYNC) {
      // the scene's K / V rows of this layer: every workgroup of the scene has written its 16 before any reads them
#if IG_LP_SYNC1
      // ONE wave carries the agent-scope fences: the others' K / V stores are complete when they reach the barrier (vmcnt(0)), the
      // L2 write-back / the L1 + L2 invalidate act on caches the whole workgroup shares, and the barriers order the rest
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (w == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (tid == 0) {
          __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int target = (i + 1) * gps;
          unsigned spins = 0;
          while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (a.spin_limit && ++spins > a.spin_limit) __builtin_trap();
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
#else
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int target = (i + 1) * gps;
        // (api.hip: layers_p_launch never launches more workgroups than the device keeps resident for this kernel and orders the
        // k_layers_p launches of the process's streams behind each other, so every workgroup is resident or will be as soon as kernels
        // of other streams leave the CUs: the wait is bounded by other work's duration.  The default launch (layers_p = 1) is a PLAIN
        // one with a.spin_limit = 2^24 polls (tens of seconds: only another PROCESS running such a kernel on the same GPU can hold
        // it that long - it traps instead of hanging the device); layers_p = 2 launches cooperatively (device-wide queue, safe next
        // to other processes) and waits without a limit; launches captured into a HIP graph trap after 2^22 polls)
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(2);
          if (a.spin_limit && ++spins > a.spin_limit) __builtin_trap();
        }
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    }
    STAMP(0);
    sublayer(es, es_next, Ksrc, Vsrc, kind == 0, P, NP, nK, nV);
  }
  if (own_row) *reinterpret_cast<float4*>(a.X + (size_t)row * D + own) = make_float4(x[0], x[1], x[2], x[3]);
}

template __global__ void k_layers_p<true, 16>(LayersPArgs);
template __global__ void k_layers_p<false, 16>(LayersPArgs);
template __global__ void k_layers_p<true, 8>(LayersPArgs);
template __global__ void k_layers_p<false, 8>(LayersPArgs);
template __global__ void k_layers_p<true, 4>(LayersPArgs);
template __global__ void k_layers_p<false, 4>(LayersPArgs);

}  // namespace ig
