// k_edge_fused: the edge side of one AttentionLayer (reference infgen/modules/layers.py:78-99,109) for a tile of 16
// destination rows with the absorbed relative-position query U and the aggregate Z kept on chip:
//
//   phase 1 (matrix pipe)  u_h = q_h W'_kr,h for the 16 rows, wave w = head w; three-term fp16 split like k_attn_h
//                          (split.cuh), weights straight from L2 as MFMA A fragments; result -> LDS tile [16][8][128]
//   phase 2 (vector pipe)  the per-destination edge loop of edge_attn.cuh (online softmax with PyG's + 1e-16, one wave
//                          per row, rows dealt to the 8 waves through an LDS counter), u read from LDS, the normalised
//                          z_h = sum_e a_e,h rhat_e written back over the row's own u
//   phase 3 (matrix pipe)  agg' = agg + W'_vr,h z_h + b'_h sigma_h, wave w = head w, -> global AGG
//
// so that the node kernel (k_attn_h / k_attn_post with has_pos = 0) never sees U, Z or SIG: per row and sublayer 512 B of
// q in and 512 B of agg' out instead of 9.2 KB (4 KB of U in, 4 KB of Z out, q, agg, sigma) here and 8 KB in the node
// kernel.  The GEMM arithmetic (operand scaling, split, order of the products) is that of k_attn_h's z-GEMM and u-GEMM.
// 512 threads, 75 KB of LDS: two workgroups per CU.
#include "kernels.h"
#include "layout.h"
#include "tile.cuh"
#include "split.cuh"
#include "edge_attn.cuh"

namespace ig {

constexpr int EF_ROWS = 16;                 // destination rows per workgroup (the N dimension of the 16x16 MFMAs)
constexpr int EF_WAVES = 8;                 // one per head in the matrix phases
constexpr int EF_NT = 64 * EF_WAVES;
constexpr int EF_LDU = H * D + 4;           // row stride of the U / Z tile in floats (+4: conflict-free b128 column writes)
constexpr int EF_LDA = D + 4;

template <int LOOP>
__global__ __launch_bounds__(EF_NT, 4) void k_edge_fused(EdgeFusedArgs a) {
  __shared__ __attribute__((aligned(16))) float UZ[EF_ROWS * EF_LDU];
  __shared__ __attribute__((aligned(16))) float AG[EF_ROWS * EF_LDA];
  __shared__ float SG[EF_ROWS * H];
  __shared__ int next_row;
  const int ngroups = a.groups ? *a.n_groups : (a.rows + EF_ROWS - 1) / EF_ROWS;
  if ((int)blockIdx.x >= ngroups) return;
  const int r0 = (a.groups ? a.groups[blockIdx.x] : (int)blockIdx.x) * EF_ROWS;
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int h = w, hp = h >> 1, hh = h & 1;
  const int row = r0 + j;
  const bool valid = row < a.rows;
  const float* hdr = a.pack + AH_HDR;
  if (tid == 0) next_row = 0;

  // ---- phase 1: u_h = q_h W'_kr,h (K = 16: v_mfma_f32_16x16x16_f16; B fragment = the head's 16 query values of row j)
  {
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
    // per (row, head) power-of-two scale into the fp16 range, as frags_scaled does per row
    float m = fmaxf(fmaxf(fabsf(qv.x), fabsf(qv.y)), fmaxf(fabsf(qv.z), fabsf(qv.w)));
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    unsigned eb = __float_as_uint(m) >> 23;
    eb = min(max(eb, 15u), 253u);
    const float sc = __uint_as_float((268u - eb) << 23), inv = __uint_as_float((eb - 14u) << 23);
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
    float* urow = UZ + j * EF_LDU + h * D + 4 * g;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
      *reinterpret_cast<float4*>(urow + 16 * ct) = make_float4(acc[ct][0] * cq, acc[ct][1] * cq, acc[ct][2] * cq, acc[ct][3] * cq);
  }
  __syncthreads();

  // ---- phase 2: edge loop, one wave per destination row
  {
    EdgeAttnArgs ea;
    ea.rows = a.rows; ea.Q = a.Q; ea.U = nullptr; ea.Ksrc = a.Ksrc; ea.Vsrc = a.Vsrc; ea.es = a.es;
    ea.AGG = nullptr; ea.Z = nullptr; ea.SIG = nullptr; ea.wkr = nullptr; ea.n_agents = nullptr; ea.A_cap = 0; ea.margin = 0;
    // (a compile-time `true` here sends this hipcc build's simplifycfg pass into a crash; rhat is always present)
    const bool has_r = a.es.rhat != nullptr;
    auto take_row = [&]() {
      int r = 0;
      if (lane == 0) r = atomicAdd(&next_row, 1);
      return __builtin_amdgcn_readfirstlane(r);
    };
    for (int rl = take_row(); rl < EF_ROWS; rl = take_row()) {
      const int drow = r0 + rl;
      float* uz = UZ + rl * EF_LDU;
      const bool live = drow < a.rows;
      const int E = live ? __builtin_amdgcn_readfirstlane(a.es.cnt[drow]) : 0;
      const int e_base = live ? __builtin_amdgcn_readfirstlane(a.es.off[drow]) : 0;
      AttnState st;
      if constexpr (LOOP == 2) edge_attn_wave2<3, true, true>(ea, live ? drow : 0, E, e_base, 0, 1, st, uz);
      else edge_attn_wave(ea, live ? drow : 0, E, e_base, 0, 1, has_r, st, uz);
      const float inv = 1.0f / (st.lsum + 1e-16f);
      *reinterpret_cast<float2*>(AG + rl * EF_LDA + 2 * lane) = make_float2(st.ag.x * inv, st.ag.y * inv);
#pragma unroll
      for (int hd = 0; hd < H; ++hd) {
        const float ih = readlane_f(inv, 8 * hd);
        *reinterpret_cast<float2*>(uz + hd * D + 2 * lane) = make_float2(st.z[hd].x * ih, st.z[hd].y * ih);
      }
      if ((lane & 7) == 0) SG[rl * H + (lane >> 3)] = st.lsum * inv;
    }
  }
  __syncthreads();

  // ---- phase 3: agg' = agg + W'_vr,h z_h + b'_h sigma_h  (k_attn_h's z-GEMM: |z| <= sqrt(127), static prescale 1024)
  {
    const float* zrow = UZ + j * EF_LDU + h * D + 8 * g;
    const unsigned short* Wv = reinterpret_cast<const unsigned short*>(a.pack + AH_POST) + (size_t)hp * QUARTER +
                               (size_t)(hh * 4) * 2 * 512 + lane * 8;
    v8h ah[4], al[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      ah[s] = *reinterpret_cast<const v8h*>(Wv + (s * 2) * 512);
      al[s] = *reinterpret_cast<const v8h*>(Wv + (s * 2 + 1) * 512);
    }
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
      const float sg = SG[j * H + h];
      const float4 bvr = *reinterpret_cast<const float4*>(a.pack + AL_BVR + DH * h + 4 * g);
      const float4 ag = *reinterpret_cast<const float4*>(AG + j * EF_LDA + DH * h + 4 * g);
      float4 o;
      o.x = ag.x + (acc[0] * zinv + bvr.x * sg);
      o.y = ag.y + (acc[1] * zinv + bvr.y * sg);
      o.z = ag.z + (acc[2] * zinv + bvr.z * sg);
      o.w = ag.w + (acc[3] * zinv + bvr.w * sg);
      *reinterpret_cast<float4*>(a.AGG + (size_t)row * D + DH * h + 4 * g) = o;
    }
  }
}

template __global__ void k_edge_fused<1>(EdgeFusedArgs);
template __global__ void k_edge_fused<2>(EdgeFusedArgs);

}  // namespace ig
