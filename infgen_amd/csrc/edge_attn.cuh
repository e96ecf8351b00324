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

// consume edges e = e_first, e_first + e_step, ... < E of destination `row`
// u_lds != null: the absorbed query u_h = q_h W'_kr,h of this row is read from LDS ([8][128] fp32, k_edge_fused) instead of a.U
__device__ __forceinline__ void edge_attn_wave(const EdgeAttnArgs& a, int row, int E, int e_base, int e_first,
                                               int e_step, bool has_r, AttnState& st,
                                               const float* u_lds = nullptr) {
  const int lane = lane_id();
  const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
  const float2 q = *reinterpret_cast<const float2*>(a.Q + (size_t)row * D + 2 * lane);
  float2 u[H];
  if (u_lds) {
    // k_edge_fused: the absorbed query of this row was left in LDS by the tile's u-GEMM ([8][128] fp32)
#pragma unroll
    for (int h = 0; h < H; ++h) {
      u[h] = *reinterpret_cast<const float2*>(u_lds + h * D + 2 * lane);
      st.z[h] = make_float2(0.f, 0.f);
    }
  } else {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      u[h] = has_r ? *reinterpret_cast<const float2*>(a.U + (size_t)row * (H * D) + h * D + 2 * lane)
                   : make_float2(0.f, 0.f);
      st.z[h] = make_float2(0.f, 0.f);
    }
  }
  st.ag = make_float2(0.f, 0.f);
  st.m = -INFINITY;
  st.lsum = 0.f;
  // software pipeline: operands of the next edge are requested before the current one is consumed
  float2 kn = make_float2(0.f, 0.f), vn = kn, rn = kn;
  if (e_first < E) {
    const int s0 = __builtin_amdgcn_readfirstlane(a.es.src[e_base + e_first]);
    kn = *reinterpret_cast<const float2*>(a.Ksrc + (size_t)s0 * D + 2 * lane);
    vn = *reinterpret_cast<const float2*>(a.Vsrc + (size_t)s0 * D + 2 * lane);
    if (has_r) rn = *reinterpret_cast<const float2*>(a.es.rhat + (size_t)(e_base + e_first) * D + 2 * lane);
  }
  for (int e = e_first; e < E; e += e_step) {
    const float2 k2 = kn, v2 = vn, r2 = rn;
    if (e + e_step < E) {
      const int s1 = __builtin_amdgcn_readfirstlane(a.es.src[e_base + e + e_step]);
      kn = *reinterpret_cast<const float2*>(a.Ksrc + (size_t)s1 * D + 2 * lane);
      vn = *reinterpret_cast<const float2*>(a.Vsrc + (size_t)s1 * D + 2 * lane);
      if (has_r) rn = *reinterpret_cast<const float2*>(a.es.rhat + (size_t)(e_base + e + e_step) * D + 2 * lane);
    }
    float val = fmaf(q.y, k2.y, q.x * k2.x);
    if (has_r) {
      float p[H];
#pragma unroll
      for (int h = 0; h < H; ++h) p[h] = fmaf(u[h].y, r2.y, u[h].x * r2.x);
      // halving exchange over lane bits 5 and 4 with the gfx950 half / row swaps: after swapping the upper half of X
      // with the lower half of Y, X + Y holds the X sum in the lower lanes and the Y sum in the upper ones
      float k4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x2_t sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(p[i]), __float_as_uint(p[4 + i]), false, false);
        k4[i] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      }
      float k2v[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x2_t sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(k4[i]), __float_as_uint(k4[2 + i]), false, false);
        k2v[i] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      }
      {
        const float send = b3 ? k2v[0] : k2v[1];
        const float keep = b3 ? k2v[1] : k2v[0];
        val += keep + dpp_xor8(send);
      }
    }
    val = sum8(val);                          // score of head (lane >> 3), uniform over its 8 lanes
    const float mn = fmaxf(st.m, val);
    const float pe = expf(val - mn);
    if (__any(mn > st.m)) {                   // some head's running max grew: rescale the accumulators
      const float sc = expf(st.m - mn);       // exp(-inf) = 0 on the first edge (accumulators are 0)
      st.lsum *= sc;
      st.ag.x *= sc; st.ag.y *= sc;
      if (has_r) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
          const float sh = readlane_f(sc, 8 * h);
          st.z[h].x *= sh; st.z[h].y *= sh;
        }
      }
      st.m = mn;
    }
    st.lsum += pe;
    st.ag.x = fmaf(pe, v2.x, st.ag.x);
    st.ag.y = fmaf(pe, v2.y, st.ag.y);
    if (has_r) {
      float ph[H];                     // all eight broadcasts first: their SGPR results are not needed back to back
#pragma unroll
      for (int h = 0; h < H; ++h) ph[h] = readlane_f(pe, 8 * h);
#pragma unroll
      for (int h = 0; h < H; ++h) {
        st.z[h].x = fmaf(ph[h], r2.x, st.z[h].x);
        st.z[h].y = fmaf(ph[h], r2.y, st.z[h].y);
      }
    }
  }
}

// ---- second form of the same loop (k_edge_fused, k_edge_attn, k_edge_attn_wide) ------------------------------------------
// The loop above is bound by vector-instruction issue (~140 instructions per edge, 4 cycles each).  This form keeps the
// arithmetic and drops a third of the instructions:
//   * scores live in the log2 domain (one multiply by log2 e per edge), exp is one v_exp_f32;
//   * the softmax reference m is only moved when a score exceeds it by more than 8 (a factor 256): the accumulators are
//     rescaled a handful of times per row instead of at most edges (with eight heads some head sets a new maximum on most
//     edges); terms stay <= 2^8, the reference edge itself contributes exactly 1, so lsum >= 1 and PyG's
//     exp(s - max) / (sum + 1e-16) is reproduced to rounding (the epsilon only matters for rows without edges: exact 0);
//   * the eight u_h . rhat partial products are formed for two heads at a time (v_pk_mul / v_pk_fma);
//   * source indices of up to 64 edges sit in one register (v_readlane per edge instead of a dependent scalar load), and
//     the K / V / rhat rows of PF edges are in flight (index clamped at the end of the list: no branch around the loads,
//     so the waits are counted ones).
typedef float pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk2 pk_fma(pk2 a, pk2 b, pk2 c) { return __builtin_elementwise_fma(a, b, c); }
constexpr float EA_LOG2E = 1.44269504088896340736f;
constexpr float EA_TAU = 8.0f;

template <int PF, bool ULDS, bool HASR>
__device__ __forceinline__ void edge_attn_wave2(const EdgeAttnArgs& a, int row, int E, int e_base, int e_first,
                                                int e_step, AttnState& st, const float* u_lds = nullptr) {
  constexpr bool has_r = HASR;
  const int lane = lane_id();
  const bool b3 = lane & 8;
  const float2 q = *reinterpret_cast<const float2*>(a.Q + (size_t)row * D + 2 * lane);
  pk2 ux[H / 2], uy[H / 2], zz[H];
#pragma unroll
  for (int i = 0; i < H / 2; ++i) {
    float2 u0 = make_float2(0.f, 0.f), u1 = u0;
    if constexpr (ULDS) {      // (compile-time: a dead a.U path with a null U crashes this hipcc build's simplifycfg pass)
      u0 = *reinterpret_cast<const float2*>(u_lds + (2 * i) * D + 2 * lane);
      u1 = *reinterpret_cast<const float2*>(u_lds + (2 * i + 1) * D + 2 * lane);
    } else if constexpr (HASR) {
      u0 = *reinterpret_cast<const float2*>(a.U + (size_t)row * (H * D) + (2 * i) * D + 2 * lane);
      u1 = *reinterpret_cast<const float2*>(a.U + (size_t)row * (H * D) + (2 * i + 1) * D + 2 * lane);
    }
    ux[i] = pk2{u0.x, u1.x};
    uy[i] = pk2{u0.y, u1.y};
  }
#pragma unroll
  for (int h = 0; h < H; ++h) zz[h] = pk2{0.f, 0.f};
  pk2 ag = {0.f, 0.f};
  float m = -INFINITY, lsum = 0.f;
  const int n = E > e_first ? (E - e_first + e_step - 1) / e_step : 0;      // edges of this wave
  for (int c0 = 0; c0 < n; c0 += 64) {
    const int mc = min(64, n - c0);
    const int srcv = a.es.src[e_base + e_first + (c0 + min(lane, mc - 1)) * e_step];
    pk2 kb[PF], vb[PF], rb[PF];
    auto issue = [&](int slot, int i) {
      const int ic = min(i, mc - 1);
      const int sj = __builtin_amdgcn_readlane(srcv, ic);
      const size_t e = (size_t)(e_base + e_first + (c0 + ic) * e_step);
      kb[slot] = *reinterpret_cast<const pk2*>(a.Ksrc + (size_t)sj * D + 2 * lane);
      vb[slot] = *reinterpret_cast<const pk2*>(a.Vsrc + (size_t)sj * D + 2 * lane);
      if constexpr (HASR) rb[slot] = *reinterpret_cast<const pk2*>(a.es.rhat + e * D + 2 * lane);
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) issue(s, s);
    for (int i0 = 0; i0 < mc; i0 += PF) {
#pragma unroll
      for (int s = 0; s < PF; ++s) {
        const int i = i0 + s;
        const pk2 k2 = kb[s], v2 = vb[s], r2 = HASR ? rb[s] : pk2{0.f, 0.f};
        issue(s, i + PF);
        float val = fmaf(q.y, k2[1], q.x * k2[0]);
        if constexpr (HASR) {
          pk2 pp[H / 2];
#pragma unroll
          for (int j = 0; j < H / 2; ++j) pp[j] = pk_fma(uy[j], pk2{r2[1], r2[1]}, ux[j] * pk2{r2[0], r2[0]});
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
        // log2-domain score of head (lane >> 3), uniform over its 8 lanes; a slot beyond the end of the list (the unrolled
        // tail) scores -inf and contributes exactly nothing - cheaper than a branch around the accumulator updates, whose
        // join makes hipcc copy all of them
        val = i < mc ? sum8(val) * EA_LOG2E : -INFINITY;
        const bool grow = val > m + EA_TAU;       // first edge: m = -inf
        if (__any(grow)) {
          const float mn = grow ? val : m;
          const float sc = __builtin_amdgcn_exp2f(m - mn);      // 0 on the first edge, 1 for heads that keep their reference
          lsum *= sc;
          ag *= pk2{sc, sc};
          if constexpr (HASR) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
              const float sh = readlane_f(sc, 8 * h);
              zz[h] *= pk2{sh, sh};
            }
          }
          m = mn;
        }
        const float pe = __builtin_amdgcn_exp2f(val - m);
        lsum += pe;
        ag = pk_fma(pk2{pe, pe}, v2, ag);
        if constexpr (HASR) {
          float ph[H];
#pragma unroll
          for (int h = 0; h < H; ++h) ph[h] = readlane_f(pe, 8 * h);
#pragma unroll
          for (int h = 0; h < H; ++h) zz[h] = pk_fma(pk2{ph[h], ph[h]}, r2, zz[h]);
        }
      }
    }
  }
#pragma unroll
  for (int h = 0; h < H; ++h) st.z[h] = make_float2(zz[h][0], zz[h][1]);
  st.ag = make_float2(ag[0], ag[1]);
  st.m = m;
  st.lsum = lsum;
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
