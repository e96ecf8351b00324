// The per-destination edge-attention loop shared by k_edge_attn / k_edge_attn_wide (edge_kernels.hip) and
// the fused tile kernel k_edge_fused (edge_fused.hip): one wavefront per destination row, single pass over the row's
// incoming edges with an online (running-max) softmax; every global access is a coalesced 512-byte row.
#pragma once
#include "kernels.h"

namespace ig {

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// running (un-normalised) attention state of one wavefront: lane l owns columns 2l, 2l+1 (head l >> 3)
struct AttnState {
  float2 z[H];      // sum_e p_e,h * rhat_e   (own columns, every head)
  float2 ag;        // sum_e p_e,head(l) * v_src (own columns)
  float m, lsum;    // running max / sum of exp for head (l >> 3)
};

// ---- the edge loop (k_edge_fused, k_edge_attn, k_edge_attn_wide) -----------------------------------------------------------
// consumes edges e = e_first, e_first + e_step, ... < E of destination `row`.  The straightforward form (running maximum,
// expf, one head's product at a time, a dependent scalar load of the source index per edge) costs ~140 vector instructions
// per edge; this one ~90:
//   * scores live in the log2 domain (one multiply by log2 e per edge), exp is one v_exp_f32;
//   * the softmax reference m is only moved when a score exceeds it by more than 8 (a factor 256): the accumulators are
//     rescaled a handful of times per row instead of at most edges (with eight heads some head sets a new maximum on most
//     edges); terms stay <= 2^8, the reference edge itself contributes exactly 1, so lsum >= 1 and PyG's
//     exp(s - max) / (sum + 1e-16) is reproduced to rounding (the epsilon only matters for rows without edges: exact 0);
//   * the eight u_h . rhat partial products are formed for two heads at a time (v_pk_mul / v_pk_fma);
//   * source indices of up to 64 edges sit in one register (v_readlane per edge instead of a dependent scalar load), and
//     the K / V / rhat rows of PF edges are requested together.
typedef float pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk2 pk_fma(pk2 a, pk2 b, pk2 c) { return __builtin_elementwise_fma(a, b, c); }
// Broadcast operands of the packed fp32 instructions.  hipcc folds pk2{v, v} of a per-lane value into an operand-select modifier on
// a 32-bit VGPR (`v_pk_fma_f32 v[90:91], v[86:87], v[84:85], v[56:57] op_sel_hi:[0,1,1]`: the pair's other register is whatever
// lives there).  On this gfx950 / ROCm 7.2 stack such instructions now and then compute with the OTHER register of the pair while
// other waves of the CU execute MFMAs: tools/hazard_repro2.hip (this loop next to k_edge_fused's matrix phases, no data shared)
// shows it in ~40 % of its launches, never without the MFMAs, never without packed fp32 instructions, never once the per-lane
// broadcasts are real register pairs - while wave-uniform broadcasts from SGPRs (`s[36:37] op_sel_hi:[0,..]`) are innocent
// (profiles/r03_hazard_bisect*.log).  So a per-lane broadcast is materialised as a register pair (one v_mov;
// three per edge).  IG_EDGE_OPSEL_BROADCAST restores hipcc's own form (the reproducer's failing arm), IG_EDGE_SPAIR / IG_EDGE_SVPAIR
// are the SGPR arms of the bisect.
__device__ __forceinline__ pk2 bc_s(float s) {          // s is wave-uniform (a v_readlane result)
#if defined(IG_EDGE_SPAIR)
  const unsigned u = __builtin_amdgcn_readfirstlane(__float_as_uint(s));
  unsigned long long p = ((unsigned long long)u << 32) | u;
  asm volatile("" : "+s"(p));
  return __builtin_bit_cast(pk2, p);
#elif defined(IG_EDGE_SVPAIR)
  pk2 p = {s, s};
  asm volatile("" : "+v"(p));
  return p;
#else
  return pk2{s, s};
#endif
}
__device__ __forceinline__ pk2 bc_v(float v) {
#if defined(IG_EDGE_OPSEL_BROADCAST)
  return pk2{v, v};
#else
  pk2 p = {v, v};
  asm("" : "+v"(p));        // (not volatile: the scheduler may move it; it only hides the value's origin from the op_sel fold)
  return p;
#endif
}
constexpr float EA_LOG2E = 1.44269504088896340736f;
constexpr float EA_TAU = 8.0f;

// per-wave running state of one destination row and the work of one edge; lane l owns columns 2l, 2l + 1 (head l >> 3)
template <bool HASR>
struct EdgeAcc {
  pk2 ux[H / 2], uy[H / 2];     // absorbed query, heads paired: ux[i] = (u[2i].x, u[2i+1].x)
  pk2 zz[H];                    // sum_e p_e,h rhat_e (own columns, every head)
  pk2 ag;                       // sum_e p_e,head(l) v_src (own columns)
  float2 q;
  float m, lsum;                // softmax reference (log2 domain) / sum of the terms for head (l >> 3)
  __device__ __forceinline__ void reset() {
#pragma unroll
    for (int h = 0; h < H; ++h) zz[h] = pk2{0.f, 0.f};
    ag = pk2{0.f, 0.f};
    m = -INFINITY;
    lsum = 0.f;
  }
  // u: [8][128] fp32 of this row (LDS or global)
  __device__ __forceinline__ void load_u(const float* u, int lane) {
#pragma unroll
    for (int i = 0; i < H / 2; ++i) {
      const float2 u0 = *reinterpret_cast<const float2*>(u + (2 * i) * D + 2 * lane);
      const float2 u1 = *reinterpret_cast<const float2*>(u + (2 * i + 1) * D + 2 * lane);
      ux[i] = pk2{u0.x, u1.x};
      uy[i] = pk2{u0.y, u1.y};
    }
  }
  // one edge; live = false (a slot beyond the end of the list in an unrolled tail): the score is -inf and the edge contributes
  // exactly nothing - cheaper than a branch around the accumulator updates, whose join makes hipcc copy all of them
  __device__ __forceinline__ void step(pk2 k2, pk2 v2, pk2 r2, bool live, bool b3) {
    float val = fmaf(q.y, k2[1], q.x * k2[0]);
    if constexpr (HASR) {
      pk2 pp[H / 2];
#ifdef IG_EDGE_SDROP          // timing experiment (wrong results): the u . rhat products of two of the eight heads only
      pp[0] = pk_fma(uy[0], bc_v(r2[1]), ux[0] * bc_v(r2[0]));
      pp[1] = pp[2] = pp[3] = pp[0];
#else
#pragma unroll
      for (int j = 0; j < H / 2; ++j) pp[j] = pk_fma(uy[j], bc_v(r2[1]), ux[j] * bc_v(r2[0]));
#endif
      // halving exchange over lane bits 5 and 4 (gfx950 half / row swaps): after swapping the upper half of X with the
      // lower half of Y, X + Y holds the X sum in the lower lanes and the Y sum in the upper ones
      float k4[4];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const u32x2_t s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(pp[j][0]), __float_as_uint(pp[j + 2][0]), false, false);
        const u32x2_t s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(pp[j][1]), __float_as_uint(pp[j + 2][1]), false, false);
        k4[2 * j] = __uint_as_float(s0[0]) + __uint_as_float(s0[1]);          // heads 2 j | 2 j + 4
        k4[2 * j + 1] = __uint_as_float(s1[0]) + __uint_as_float(s1[1]);      // heads 2 j + 1 | 2 j + 5
      }
      float k2v[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const u32x2_t sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(k4[j]), __float_as_uint(k4[2 + j]), false, false);
        k2v[j] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);             // heads j | j + 2 | j + 4 | j + 6
      }
      const float send = b3 ? k2v[0] : k2v[1];
      const float keep = b3 ? k2v[1] : k2v[0];
      val += keep + dpp_xor8(send);
    }
    // log2-domain score of head (lane >> 3), uniform over its 8 lanes (the dead-slot select is on the constant so that the
    // score - and with it the K row's load - cannot be sunk into a branch on `live`)
    val = fminf(sum8(val) * EA_LOG2E, live ? INFINITY : -INFINITY);
    const bool grow = val > m + EA_TAU;                // first edge: m = -inf
    if (__any(grow)) {
      const float mn = grow ? val : m;
      const float sc = __builtin_amdgcn_exp2f(m - mn);      // 0 on the first edge, 1 for heads that keep their reference
      lsum *= sc;
      ag *= bc_v(sc);
      if constexpr (HASR) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
          const float sh = readlane_f(sc, 8 * h);
          zz[h] *= bc_s(sh);
        }
      }
      m = mn;
    }
    const float pe = __builtin_amdgcn_exp2f(val - m);
    lsum += pe;
#if defined(IG_EDGE_OPSEL_BROADCAST)
    ag = pk_fma(bc_v(pe), v2, ag);
#else
    ag = pk2{fmaf(pe, v2[0], ag[0]), fmaf(pe, v2[1], ag[1])};      // two scalar FMAs: a register pair for pe would cost the same
#endif
    if constexpr (HASR) {
#ifdef IG_EDGE_ZDROP          // timing experiment (wrong results): one of the eight z accumulations only
      constexpr int HZ = 1;
#else
      constexpr int HZ = H;
#endif
      float ph[HZ];
#pragma unroll
      for (int h = 0; h < HZ; ++h) ph[h] = readlane_f(pe, 8 * h);
#pragma unroll
      for (int h = 0; h < HZ; ++h) zz[h] = pk_fma(bc_s(ph[h]), r2, zz[h]);
    }
  }
};

// rhat rows are read exactly once per launch: non-temporal loads, so that they do not evict the K / V rows a tile (and its
// scene's other tiles on the same XCD) re-reads from L2 - the agent set reads every K / V row ~20 times.  kv_once: the K / V
// rows are read once as well (temporal set: every edge has its own ring row).
__device__ __forceinline__ pk2 ea_ld(const float* p, bool nt) {
  return nt ? __builtin_nontemporal_load(reinterpret_cast<const pk2*>(p)) : *reinterpret_cast<const pk2*>(p);
}

template <int PF, bool ULDS, bool HASR>
__device__ __forceinline__ void edge_attn_wave2(const EdgeAttnArgs& a, int row, int E, int e_base, int e_first,
                                                int e_step, AttnState& st, const float* u_lds = nullptr, bool kv_once = false) {
  const int lane = lane_id();
  const bool b3 = lane & 8;
  EdgeAcc<HASR> acc;
  acc.q = *reinterpret_cast<const float2*>(a.Q + (size_t)row * D + 2 * lane);
  if constexpr (ULDS) acc.load_u(u_lds, lane);     // (compile-time: a dead a.U path with a null U crashes this hipcc build's simplifycfg pass)
  else if constexpr (HASR) acc.load_u(a.U + (size_t)row * (H * D), lane);
  acc.reset();
  const int n = E > e_first ? (E - e_first + e_step - 1) / e_step : 0;      // edges of this wave
  for (int c0 = 0; c0 < n; c0 += 64) {
    const int mc = min(64, n - c0);
    const int srcv = a.es.src[e_base + e_first + (c0 + min(lane, mc - 1)) * e_step];
    // PF edges per trip: all their K / V / rhat rows are requested at the top of the trip (index clamped at the end of the
    // list: no branch around the loads, the waits are counted ones) and consumed in turn.  Nothing is carried in registers
    // from trip to trip on purpose: hipcc pipelines loop-carried load destinations through staging registers and rotates
    // them with copies at the back edge, and a copy of a pending load's destination is a full wait (measured: 2, 3 or 5
    // edges carried in flight, same time) - the trip's fill latency is hidden by the other waves of the SIMD instead.
    for (int i0 = 0; i0 < mc; i0 += PF) {
      pk2 kb[PF], vb[PF], rb[PF];
#pragma unroll
      for (int s = 0; s < PF; ++s) {
        const int ic = min(i0 + s, mc - 1);
        const int sj = __builtin_amdgcn_readlane(srcv, ic);
        const size_t e = (size_t)(e_base + e_first + (c0 + ic) * e_step);
        kb[s] = ea_ld(a.Ksrc + (size_t)sj * D + 2 * lane, kv_once);
        vb[s] = ea_ld(a.Vsrc + (size_t)sj * D + 2 * lane, kv_once);
        if constexpr (HASR) rb[s] = __builtin_nontemporal_load(reinterpret_cast<const pk2*>(a.es.rhat + e * D + 2 * lane));
      }
#pragma unroll
      for (int s = 0; s < PF; ++s) acc.step(kb[s], vb[s], HASR ? rb[s] : pk2{0.f, 0.f}, i0 + s < mc, b3);
    }
  }
#pragma unroll
  for (int h = 0; h < H; ++h) st.z[h] = make_float2(acc.zz[h][0], acc.zz[h][1]);
  st.ag = make_float2(acc.ag[0], acc.ag[1]);
  st.m = acc.m;
  st.lsum = acc.lsum;
}

__device__ __forceinline__ void edge_attn_write(const EdgeAttnArgs& a, int row, const AttnState& st) {
  const int lane = lane_id();
  const float inv = 1.0f / (st.lsum + 1e-16f);
  *reinterpret_cast<float2*>(a.AGG + (size_t)row * D + 2 * lane) = make_float2(st.ag.x * inv, st.ag.y * inv);
  if (a.Z) {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float ih = readlane_f(inv, 8 * h);
      *reinterpret_cast<float2*>(a.Z + (size_t)row * (H * D) + h * D + 2 * lane) =
          make_float2(st.z[h].x * ih, st.z[h].y * ih);
    }
  }
  if ((lane & 7) == 0) a.SIG[(size_t)row * H + (lane >> 3)] = st.lsum * inv;
}

}  // namespace ig
