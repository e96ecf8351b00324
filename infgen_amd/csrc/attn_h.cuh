// Pieces shared by the node-side split kernels k_attn_h (attn_h.hip: 64 / 128-row tiles) and k_attn_hs (attn_hs.hip:
// one 16-row tile per workgroup, feature tiles dealt to the waves): the LDS vector table and row helpers.
#pragma once
#include "kernels.h"
#include "layout.h"
#include "tile.cuh"
#include "split.cuh"

namespace ig {

enum AttnHVec : int {       // fp32 per-feature vectors copied to LDS
  VT_LND_G = 0, VT_LND_B = 128, VT_BVR = 256, VT_BG = 384, VT_BS = 512, VT_BO = 640,
  VT_LNP_G = 768, VT_LNP_B = 896, VT_LNF_G = 1024, VT_LNF_B = 1152, VT_B1 = 1280 /* 512 */, VT_B2 = 1792,
  VT_LNO_G = 1920, VT_LNO_B = 2048, VT_HDR = 2176 /* 16 */,
  VT_N_LN_G = 2192, VT_N_LN_B = 2320, VT_N_BQ = 2448, VT_N_BV = 2576, VT_N_HDR = 2704 /* 16 */,
  VT_SIZE = 2720,
};

// Source of a 16-byte slot of the vector table Vt (first float d, a multiple of 4) in the packs of the layer whose post part runs
// (P) / whose pre part runs (NP); nullptr: the slot has no source in this launch.  Selects only (the segments ascend in Vt, the
// last one that starts at or before d is the slot's): as an if / else tree the lanes of a wave went through twenty branches.
__device__ __forceinline__ const float* attn_table_src(int d, const float* P, const float* NP, int next_src_ln) {
  constexpr int NSEG = 20;
  constexpr int DST[NSEG] = {VT_LND_G, VT_LND_B, VT_BVR, VT_BG, VT_BS, VT_BO, VT_LNP_G, VT_LNP_B, VT_LNF_G, VT_LNF_B, VT_B1, VT_B2,
                             VT_LNO_G, VT_LNO_B, VT_HDR, VT_N_LN_G, VT_N_LN_B, VT_N_BQ, VT_N_BV, VT_N_HDR};
  constexpr int SRC[NSEG] = {AL_LN_DST_G, AL_LN_DST_B, AL_BVR, AL_BG, AL_BS, AL_BO, AL_LN_POST_G, AL_LN_POST_B, AL_LN_FFPRE_G,
                             AL_LN_FFPRE_B, AL_B1, AL_B2, AL_LN_FFPOST_G, AL_LN_FFPOST_B, AH_HDR, -1, -2, AL_BQ, AL_BV, AH_HDR};
  const int nlg = next_src_ln ? AL_LN_SRC_G : AL_LN_DST_G, nlb = next_src_ln ? AL_LN_SRC_B : AL_LN_DST_B;
  int off = 0;
#pragma unroll
  for (int k = 0; k < NSEG; ++k) {
    const int sk = SRC[k] == -1 ? nlg : SRC[k] == -2 ? nlb : SRC[k];
    off = d >= DST[k] ? sk + (d - DST[k]) : off;
  }
  const float* base = d >= VT_N_LN_G ? NP : P;
  return (base && d < VT_SIZE) ? base + off : nullptr;
}
// the same row helpers without the null test (callers clamp the row instead: see k_attn_hs)
__device__ __forceinline__ void load_row_nc(f32x4 (&v)[8], const float* row, int rg) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float4 x = *reinterpret_cast<const float4*>(row + 16 * t + 4 * rg);
    v[t] = f32x4{x.x, x.y, x.z, x.w};
  }
}

__device__ __forceinline__ void load_row(f32x4 (&v)[8], const float* row, int rg) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row) x = *reinterpret_cast<const float4*>(row + 16 * t + 4 * rg);
    v[t] = f32x4{x.x, x.y, x.z, x.w};
  }
}
__device__ __forceinline__ void store_row(float* row, const f32x4 (&v)[8], int rg) {
#ifdef IG_AH_NOSTORE          // timing experiment (wrong results): what the row stores cost
  if (rg >= 0) return;
#endif
  if (!row) return;
#pragma unroll
  for (int t = 0; t < 8; ++t)
    *reinterpret_cast<float4*>(row + 16 * t + 4 * rg) = make_float4(v[t][0], v[t][1], v[t][2], v[t][3]);
}
__device__ __forceinline__ void zero_acc(f32x4 (&v)[8]) {
#pragma unroll
  for (int t = 0; t < 8; ++t) v[t] = f32x4{0.f, 0.f, 0.f, 0.f};
}
// v = v * s + bias (bias: per-feature vector in LDS, may be null)
__device__ __forceinline__ void scale_bias(f32x4 (&v)[8], float s, const float* bias, int rg) {
  const f32x4 s4 = splat4(s);
#pragma unroll
  for (int t = 0; t < 8; ++t) v[t] = bias ? fma4(v[t], s4, lds4(bias + 16 * t + 4 * rg)) : v[t] * s4;
}

}  // namespace ig
