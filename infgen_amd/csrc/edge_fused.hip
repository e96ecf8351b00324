// k_edge_fused: the edge side of one AttentionLayer (reference infgen/modules/layers.py:78-99,109) for a tile of 16
// destination rows with the absorbed relative-position query U and the aggregate Z kept on chip:
//
//   phase 1 (matrix pipe)  u_h = q_h W'_kr,h for the 16 rows, wave w = head w; three-term fp16 split like k_attn_h
//                          (split.cuh), weights straight from L2 as MFMA A fragments; result -> LDS tile [16][8][128];
//                          the q tile is parked in LDS as well (in the buffer that later takes agg)
//   phase 2 (vector pipe)  the edge loop (edge_attn.cuh: EdgeAcc - online softmax with PyG's + 1e-16, one wave per row,
//                          rows dealt to the 8 waves through an LDS counter), u and q read from LDS, the normalised
//                          z_h = sum_e a_e,h rhat_e written back over the row's own u
//   phase 3 (matrix pipe)  agg' = agg + W'_vr,h z_h + b'_h sigma_h, wave w = head w, -> global AGG
//
// so that the node kernel (k_attn_h / k_attn_post with has_pos = 0) never sees U, Z or SIG: per row and sublayer 512 B of
// q in and 512 B of agg' out instead of 9.2 KB (4 KB of U in, 4 KB of Z out, q, agg, sigma) here and 8 KB in the node
// kernel.  The GEMM arithmetic (operand scaling, split, order of the products) is that of k_attn_h's z-GEMM and u-GEMM.
// 1024 threads = 16 waves = two 16-row halves (wave w: head w & 7 of half w >> 3 in the matrix phases), 150 KB of LDS: ONE
// workgroup per CU.  Two co-resident 8-wave workgroups (75 KB each) were ~5 % faster and WRONG: the workgroup that is not the
// first on its CU got single rows off by ~1e-2, different rows every run (tools/determinism_probe2.py; never with one
// workgroup per CU - the "bring-up scare" of fourier_h.hip again, now reproducible: DESIGN.md section 5.1).
#include "kernels.h"
#include "layout.h"
#include "tile.cuh"
#include "split.cuh"
#include "edge_attn.cuh"

namespace ig {

constexpr int EF_HALVES = 2;                // 16-row halves per workgroup (the N dimension of the 16x16 MFMAs is 16 rows)
constexpr int EF_WAVES = 8 * EF_HALVES;     // one per (head, half) in the matrix phases
constexpr int EF_NT = 64 * EF_WAVES;
constexpr int EF_ROWS = 16 * EF_HALVES;
constexpr int EF_LDU = H * D + 4;           // row stride of the U / Z tile in floats (+4: conflict-free b128 column writes)
constexpr int EF_LDA = D + 4;

// G = edges per trip of the edge loop (their K / V / rhat rows are requested together); R24: rhat rows in the packed 24-bit
// format (kernels.h); HALVES = 16-row groups per workgroup: 2 when the launch fills the chip, 1 for small launches (up to 256
// groups = 4 k rows: twice the workgroups, and the 16 waves take ONE row each in the edge loop instead of two - the loop of a
// small launch is the latency of its longest rows; the 8 waves without a (head, half) idle in the matrix phases)
// WAVES = 16: one 1024-thread workgroup per CU; WAVES = 8 (HALVES = 1 only, 75 KB of LDS): two co-resident workgroups per CU, whose
// matrix phases run under each other's edge loops - the configuration that gave wrong rows in round 2 and is exact since the
// loop's per-lane broadcasts are real register pairs (edge_attn.cuh: bc_v)
// timing experiment (build with -DIG_EF_TRACE=1, run with INFGEN_EDGE_DBG bit 7): s_memtime of wave 0 of workgroup 2600 -> a.dbgbuf
#ifndef IG_EF_TRACE
#define IG_EF_TRACE 0
#endif
#if IG_EF_TRACE
#define EF_STAMP(i) do { if ((a.dbg & 128) && blockIdx.x == 2600 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(a.dbgbuf)[i] = __builtin_readcyclecounter(); } while (0)
#else
#define EF_STAMP(i) do {} while (0)
#endif
template <int G, bool R24, int HALVES, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 4) void k_edge_fused(EdgeFusedArgs a) {
  static_assert(WAVES == 16 || (WAVES == 8 && HALVES == 1), "8-wave workgroups take one 16-row group");
  constexpr int ROWS = 16 * HALVES;
  __shared__ __attribute__((aligned(16))) float UZ[ROWS * EF_LDU];
  __shared__ __attribute__((aligned(16))) float AG[ROWS * EF_LDA];     // q tile (phase 1 -> 2), then agg (phase 2 -> 3)
  __shared__ float SG[ROWS * H];
  __shared__ int next_row;
  const int ngroups = a.groups ? *a.n_groups : (a.rows + 15) / 16;        // 16-row groups; a tile takes HALVES of them
  const int ntiles = (ngroups + HALVES - 1) / HALVES;
  // XCD-aware tile order: consecutive workgroups go to consecutive XCDs (b % 8), each with its own L2.  With tps tiles per
  // scene, workgroups b, b + 8, ..., b + 8 (tps - 1) - one XCD - take the tiles of ONE scene, so that the scene's K / V rows
  // (agent set: read by every row of the scene) are fetched into one L2 instead of tps of them.
  if (HALVES == 1 && WAVES == 16 && warm_l2(a.warm, blockIdx.x, threadIdx.x, 64 * WAVES)) return;       // kernels.h: WarmArgs
  int tile = blockIdx.x;
  if (a.tiles_per_scene > 1) {
    const int tps = a.tiles_per_scene, grp = 8 * tps;
    const int bq = tile / grp, br = tile % grp;
    tile = bq * grp + (br % 8) * tps + br / 8;
  }
  if (tile >= ntiles) return;
  EF_STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int h = w & 7, half = w >> 3, hp = h >> 1, hh = h & 1;
  // first row of each half (-1: no such group)
  int r0h[HALVES];
#pragma unroll
  for (int b = 0; b < HALVES; ++b) {
    const int gi = HALVES * tile + b;
    r0h[b] = gi < ngroups ? 16 * (a.groups ? a.groups[gi] : gi) : -1;
  }
  const bool mat = half < HALVES;                  // this wave has a (head, half) of the matrix phases
  const int r0 = (HALVES > 1 && half) ? r0h[HALVES - 1] : r0h[0];
  const int jl = 16 * half + j;                    // this lane's row of the LDS tiles in the matrix phases
  const int row = r0 + j;
  const bool valid = mat && r0 >= 0 && row < a.rows;
  const float* hdr = a.pack + AH_HDR;
  if (threadIdx.x == 0) next_row = 0;
  // Longest rows first: the tile's rows are dealt to the waves in descending order of their edge counts (the agent set's lists
  // run from a few to 60+ edges; dealt in index order a long row that comes last is the tile's tail while the other waves
  // wait at the barrier).  Only the ORDER in which rows are picked changes - every row is still summed edge by edge by one wave,
  // results are bitwise the same.  The last wave ranks the rows (a load of ROWS counts, ROWS compares) while phase 1 runs.
  constexpr bool SORT_ROWS = !(HALVES == 1 && WAVES == 16);            // (one row per wave: nothing to order)
  __shared__ unsigned char row_order[ROWS];
  if (SORT_ROWS && w == WAVES - 1) {
    const int rl = lane & (ROWS - 1);
    const int rb = (HALVES > 1 && (rl >> 4)) ? r0h[HALVES - 1] : r0h[0];
    const int dr = rb + (rl & 15);
    const int cnt = (rb >= 0 && dr < a.rows) ? a.es.cnt[dr] : 0;
    int rank = 0;
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
      const int ck = __shfl(cnt, k, 64);
      rank += (ck > cnt || (ck == cnt && k < rl)) ? 1 : 0;
    }
    if (lane < ROWS) row_order[rank] = (unsigned char)rl;
  }

  EF_STAMP(1);
  // ---- phase 1: u_h = q_h W'_kr,h (K = 16: v_mfma_f32_16x16x16_f16; B fragment = the head's 16 query values of row j)
  if (mat) {
    float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) qv = *reinterpret_cast<const float4*>(a.Q + (size_t)row * D + DH * h + 4 * g);
    const unsigned short* Wk = reinterpret_cast<const unsigned short*>(a.pack + AH_PRE) + (size_t)(4 + hp) * QUARTER +
                               (size_t)(hh * 8) * 2 * 256 + lane * 4;
    v4h ah[8], al[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      ah[ct] = *reinterpret_cast<const v4h*>(Wk + (ct * 2) * 256);
      al[ct] = *reinterpret_cast<const v4h*>(Wk + (ct * 2 + 1) * 256);
    }
    *reinterpret_cast<float4*>(AG + jl * EF_LDA + DH * h + 4 * g) = qv;
    // per (row, head) power-of-two scale into the fp16 range, as frags_scaled does per row
    float m = fmaxf(fmaxf(fabsf(qv.x), fabsf(qv.y)), fmaxf(fabsf(qv.z), fabsf(qv.w)));
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    unsigned ebits = __float_as_uint(m) >> 23;
    ebits = min(max(ebits, 15u), 253u);
    const float sc = __uint_as_float((268u - ebits) << 23), inv = __uint_as_float((ebits - 14u) << 23);
    u32x2 qh, ql;
    {
      unsigned hi, lo;
      split_pair(qv.x * sc, qv.y * sc, hi, lo); qh[0] = hi; ql[0] = lo;
      split_pair(qv.z * sc, qv.w * sc, hi, lo); qh[1] = hi; ql[1] = lo;
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
    float* urow = UZ + jl * EF_LDU + h * D + 4 * g;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
      *reinterpret_cast<float4*>(urow + 16 * ct) = make_float4(acc[ct][0] * cq, acc[ct][1] * cq, acc[ct][2] * cq, acc[ct][3] * cq);
  }
  EF_STAMP(2);
  __syncthreads();
  EF_STAMP(3);

  // ---- phase 2: edge loop, one wave per destination row
  {
    const bool b3 = lane & 8;
    const bool kv_once = a.kv_once != 0;
    const unsigned lo8 = 8u * (unsigned)lane;
    auto ld8 = [&](const float* base, bool nt) {
      return ea_ld(reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + lo8), nt);
    };
    // packed 24-bit rhat row e (kernels.h): this lane's two columns = one dword of the 16-bit plane + one short of the 8-bit plane
    auto ld_r24 = [&](size_t e) {
      const char* rowp = reinterpret_cast<const char*>(a.es.rhat) + e * R24_ROW_BYTES;
      const unsigned hi = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(rowp + 4 * lane));
      const unsigned lo = __builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(rowp + R24_LO_PLANE + 2 * lane));
      return pk2{__uint_as_float((hi << 16) | ((lo & 0xffu) << 8)), __uint_as_float((hi & 0xffff0000u) | (lo & 0xff00u))};
    };
    // rows are dealt to the waves through an LDS counter (the agent set's lists vary in length)
    auto take_row = [&]() {
      int r = 0;
      if (lane == 0) { r = atomicAdd(&next_row, 1); if (SORT_ROWS && r < ROWS) r = row_order[r]; }
      return __builtin_amdgcn_readfirstlane(r);
    };
    for (int rl = take_row(); rl < ROWS; rl = take_row()) {
      const int rbase = (HALVES > 1 && (rl >> 4)) ? r0h[HALVES - 1] : r0h[0];
      const int drow = rbase + (rl & 15);
      const bool live = rbase >= 0 && drow < a.rows && !(a.dbg & 1);
      const int E = live ? __builtin_amdgcn_readfirstlane(a.es.cnt[drow]) : 0;
      const int e_base = live ? __builtin_amdgcn_readfirstlane(a.es.off[drow]) : 0;
      int sv = E > 0 ? a.es.src[e_base + min(lane, E - 1)] : 0;            // source indices of up to 64 edges in one register
      float* uz = UZ + rl * EF_LDU;
      EdgeAcc<true> acc;
      acc.q = *reinterpret_cast<const float2*>(AG + rl * EF_LDA + 2 * lane);
      acc.load_u(uz, lane);
      acc.reset();
      for (int c0 = 0; c0 < E; c0 += 64) {
        const int mc = min(64, E - c0);
        if (c0 > 0) sv = a.es.src[e_base + c0 + min(lane, mc - 1)];        // lists beyond 64 edges: next chunk of indices
        // G edges per trip: all their K / V / rhat rows are requested at the top of the trip (index clamped at the end of the
        // list: no branch around the loads, the waits are counted ones) and consumed in turn; nothing is carried in registers
        // from trip to trip (edge_attn.cuh explains why), the trip's fill latency is hidden by the SIMD's other waves.  (A
        // tail trip of exactly the remaining length was tried: same time on lists of any raggedness - the loop is bound by
        // its gathers, DESIGN.md section 9 - and ten spilled registers.)
        for (int i0 = 0; i0 < mc; i0 += G) {
          pk2 kb[G], vb[G], rb[G];
#pragma unroll
          for (int s = 0; s < G; ++s) {
            const int ic = min(i0 + s, mc - 1);
            const int sj = __builtin_amdgcn_readlane(sv, ic);
            kb[s] = ld8(a.Ksrc + (size_t)sj * D, kv_once);
            vb[s] = ld8(a.Vsrc + (size_t)sj * D, kv_once);
            if constexpr (R24) rb[s] = ld_r24((size_t)(e_base + c0 + ic));
            else rb[s] = ld8(a.es.rhat + (size_t)(e_base + c0 + ic) * D, true);
          }
#pragma unroll
          for (int s = 0; s < G; ++s) acc.step(kb[s], vb[s], rb[s], i0 + s < mc, b3);
        }
      }
      const float inv = 1.0f / (acc.lsum + 1e-16f);
      *reinterpret_cast<float2*>(AG + rl * EF_LDA + 2 * lane) = make_float2(acc.ag[0] * inv, acc.ag[1] * inv);
#pragma unroll
      for (int hd = 0; hd < H; ++hd) {
        const float ih = readlane_f(inv, 8 * hd);
        *reinterpret_cast<float2*>(uz + hd * D + 2 * lane) = make_float2(acc.zz[hd][0] * ih, acc.zz[hd][1] * ih);
      }
      if ((lane & 7) == 0) SG[rl * H + (lane >> 3)] = acc.lsum * inv;
    }
  }
  EF_STAMP(4);
  // (phase 3's weight fragments are requested BEFORE the barrier: a wave that has finished its rows waits there anyway, and the
  // fragments' L2 latency runs under that wait)
  v8h p3h[4], p3l[4];
  if (mat) {
    const unsigned short* Wv = reinterpret_cast<const unsigned short*>(a.pack + AH_POST) + (size_t)hp * QUARTER +
                               (size_t)(hh * 4) * 2 * 512 + lane * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      p3h[s] = *reinterpret_cast<const v8h*>(Wv + (s * 2) * 512);
      p3l[s] = *reinterpret_cast<const v8h*>(Wv + (s * 2 + 1) * 512);
    }
  }
  EF_STAMP(5);
  __syncthreads();
  EF_STAMP(6);

  // ---- phase 3: agg' = agg + W'_vr,h z_h + b'_h sigma_h  (k_attn_h's z-GEMM: |z| <= sqrt(127), static prescale 1024)
  if (mat) {
    const float* zrow = UZ + jl * EF_LDU + h * D + 8 * g;
    v8h (&ah)[4] = p3h;
    v8h (&al)[4] = p3l;
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
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s], vbh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s], vbl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[s], vbh, acc, 0, 0, 0);
    }
    if (valid) {
      const float sg = SG[jl * H + h];
      const float4 bvr = *reinterpret_cast<const float4*>(a.pack + AL_BVR + DH * h + 4 * g);
      const float4 ag = *reinterpret_cast<const float4*>(AG + jl * EF_LDA + DH * h + 4 * g);
      float4 o;
      o.x = ag.x + (acc[0] * zinv + bvr.x * sg);
      o.y = ag.y + (acc[1] * zinv + bvr.y * sg);
      o.z = ag.z + (acc[2] * zinv + bvr.z * sg);
      o.w = ag.w + (acc[3] * zinv + bvr.w * sg);
      *reinterpret_cast<float4*>(a.AGG + (size_t)row * D + DH * h + 4 * g) = o;
    }
  }
  EF_STAMP(7);
}

template __global__ void k_edge_fused<4, false, 2, 16>(EdgeFusedArgs);
template __global__ void k_edge_fused<6, false, 2, 16>(EdgeFusedArgs);
template __global__ void k_edge_fused<8, false, 2, 16>(EdgeFusedArgs);
template __global__ void k_edge_fused<4, true, 2, 16>(EdgeFusedArgs);
template __global__ void k_edge_fused<6, true, 2, 16>(EdgeFusedArgs);
template __global__ void k_edge_fused<8, true, 2, 16>(EdgeFusedArgs);
template __global__ void k_edge_fused<6, false, 1, 16>(EdgeFusedArgs);
template __global__ void k_edge_fused<6, true, 1, 16>(EdgeFusedArgs);
template __global__ void k_edge_fused<6, false, 1, 8>(EdgeFusedArgs);
template __global__ void k_edge_fused<6, true, 1, 8>(EdgeFusedArgs);

}  // namespace ig
