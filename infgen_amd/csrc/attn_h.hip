// k_attn_h: the node side of one AttentionLayer (reference infgen/modules/layers.py:61-113) on the fp16
// matrix pipe with the three-term hi/lo split of split.cuh - the same arithmetic as k_attn_post /
// k_attn_pre (layer_kernels.hip), which run on the fp32-input MFMA at 1/16 of the rate.
//
//   post (layer `pack`):   agg' = AGG + W'_vr,h z_h + b'_h sigma_h            (z-GEMM, one feature tile per head)
//                          g = sigmoid(W_g [agg' ; LN_dst(x)] + b_g);  upd = agg' + g (W_s LN_dst(x) + b_s - agg')
//                          x1 = x + LN_post(W_o upd + b_o);  x2 = x1 + LN_ffpost(W_2 relu(W_1 LN_ffpre(x1) + b_1) + b_2)
//   pre (layer `next_pack`): xn = LN(x2);  q = s (W_q xn + b_q);  u_h = q_h W'_kr,h;  k = W_k xn;  v = W_v xn + b_v
//
// Register-resident per wave: a wave owns 16 rows, lane (j = lane & 15, rg = lane >> 4) holds features
// 16 t + 4 rg + r (tile t, register r) of row j; weights are the MFMA A operand and stream through the
// five-buffer LDS ring as quarter-matrices (16 KB); activations are split to fp16 hi/lo B fragments with a
// per-row power-of-two scale (frags_scaled), so no bound on their magnitude is assumed.
// WAVES = 4: 64-row tiles, two workgroups per CU (measured 5-11 % faster than one 8-wave workgroup from 32 k rows up,
// and - unlike k_fourier_h, see fourier_h.hip - bitwise reproducible with several workgroups per CU: 40 re-runs each of
// 16 k / 32 k / 64 k / 100 k rows; tests/test_ops_gpu.py::test_attn_split_is_deterministic keeps watching it).
// WAVES = 8 (128-row tiles, one workgroup per CU) is kept for comparison.
#include <type_traits>
#include "kernels.h"
#include "layout.h"
#include "tile.cuh"
#include "split.cuh"
#include "attn_h.cuh"

namespace ig {

template <int WAVES, int TERMS>
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? 2 : 1) void k_attn_h(AttnHArgs a) {
  constexpr int NTH = 64 * WAVES, TILE = 16 * WAVES;
  // 4 waves: three quarter buffers (59 KB with the vectors) and <= 256 registers -> two workgroups per CU;
  // 8 waves: the five-buffer ring of split.cuh, one workgroup per CU
  constexpr int XRING = WAVES == 4 ? IG_AH_RING4 : IG_AH_RING8;
  __shared__ __attribute__((aligned(16))) unsigned short Wb[XRING][QUARTER];
  __shared__ __attribute__((aligned(16))) float Vt[VT_SIZE];
  __shared__ const unsigned short* seg_ptr[5];
  __shared__ int seg_n[5];
  // with a group list (rows padded to A_cap per scene, insertion on) only the 16-row groups that hold agents are visited
  const int ngroups = a.groups ? *a.n_groups : (a.rows + 15) / 16;
  const int ntiles = (ngroups + WAVES - 1) / WAVES;
  if ((int)blockIdx.x >= ntiles) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int j = lane & 15, rg = lane >> 4;
  const float* P = a.pack;
  const float* NP = a.next_pack;
  const bool need_q = NP && (a.nQ || a.nU), need_u = NP && a.nU, need_kv = NP && a.nK;

  // ---- the quarter stream of one tile: up to five runs of consecutive quarters
  if (tid == 0) {
    const unsigned short* post = P ? reinterpret_cast<const unsigned short*>(P + AH_POST) : nullptr;
    const unsigned short* pre = NP ? reinterpret_cast<const unsigned short*>(NP + AH_PRE) : nullptr;
    seg_ptr[0] = post;                                seg_n[0] = (P && a.has_pos) ? 4 : 0;     // W'vr
    seg_ptr[1] = post ? post + 4 * QUARTER : nullptr; seg_n[1] = P ? 48 : 0;                   // Wg_a Wg_x Ws Wo (W1 W2) x 4
    seg_ptr[2] = pre;                                 seg_n[2] = need_q ? 4 : 0;               // Wq
    seg_ptr[3] = pre ? pre + 4 * QUARTER : nullptr;   seg_n[3] = need_u ? 4 : 0;               // W'kr
    seg_ptr[4] = pre ? pre + 8 * QUARTER : nullptr;   seg_n[4] = need_kv ? 8 : 0;              // Wk Wv
  }
  if (P) {
    for (int i = tid; i < 128; i += NTH) {
      Vt[VT_LND_G + i] = P[AL_LN_DST_G + i]; Vt[VT_LND_B + i] = P[AL_LN_DST_B + i];
      Vt[VT_BVR + i] = P[AL_BVR + i]; Vt[VT_BG + i] = P[AL_BG + i]; Vt[VT_BS + i] = P[AL_BS + i]; Vt[VT_BO + i] = P[AL_BO + i];
      Vt[VT_LNP_G + i] = P[AL_LN_POST_G + i]; Vt[VT_LNP_B + i] = P[AL_LN_POST_B + i];
      Vt[VT_LNF_G + i] = P[AL_LN_FFPRE_G + i]; Vt[VT_LNF_B + i] = P[AL_LN_FFPRE_B + i];
      Vt[VT_B2 + i] = P[AL_B2 + i];
      Vt[VT_LNO_G + i] = P[AL_LN_FFPOST_G + i]; Vt[VT_LNO_B + i] = P[AL_LN_FFPOST_B + i];
    }
    for (int i = tid; i < 512; i += NTH) Vt[VT_B1 + i] = P[AL_B1 + i];
    if (tid < 16) Vt[VT_HDR + tid] = P[AH_HDR + tid];
  }
  if (NP) {
    for (int i = tid; i < 128; i += NTH) {
      Vt[VT_N_LN_G + i] = NP[(a.next_src_ln ? AL_LN_SRC_G : AL_LN_DST_G) + i];
      Vt[VT_N_LN_B + i] = NP[(a.next_src_ln ? AL_LN_SRC_B : AL_LN_DST_B) + i];
      Vt[VT_N_BQ + i] = NP[AL_BQ + i]; Vt[VT_N_BV + i] = NP[AL_BV + i];
    }
    if (tid < 16) Vt[VT_N_HDR + tid] = NP[AH_HDR + tid];
  }
  __syncthreads();
#if IG_QSU
  typename std::conditional<WAVES == 8, QuarterStreamU<NTH>, QuarterStream<NTH, XRING>>::type qs;
#else
  QuarterStream<NTH, XRING> qs;
#endif
  qs.dbg = a.dbg;
  qs.init(seg_ptr, seg_n, 5, (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x, Wb, tid);
  auto take = [&]() { return qs.take(); };
  auto gemm_unit = [&](f32x4 (&acc)[8], const u32x4 (&Bh)[4], const u32x4 (&Bl)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) gemm_quarter<TERMS>(acc, take(), Bh[s], Bl[s], lane);
  };

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int gi = tile * WAVES + w;
    const int grp = gi < ngroups ? (a.groups ? a.groups[gi] : gi) : -1;
    const int row = grp * 16 + j;
    const bool valid = grp >= 0 && row < a.rows;
    float* xrow = valid ? a.X + (size_t)row * D : nullptr;
    // (rows beyond the end load the last row and are never stored: a conditional load gets a basic block of its own in which
    // hipcc waits for it before it issues the next one - section 5.4 of DESIGN.md)
    const int rowc = valid ? row : a.rows - 1;
    f32x4 x[8];
    load_row_nc(x, a.X + (size_t)rowc * D, rg);
    u32x4 Bh[4], Bl[4];

    if (P) {
      const float* hdr = Vt + VT_HDR;
      f32x4 ag[8];
      load_row_nc(ag, a.AGG + (size_t)rowc * D, rg);
      if (a.has_pos) {
        // z-GEMM: head h is feature tile h; B fragments straight from Z[row][h][:] (k = 32 s + 8 rg + p), |z| <= sqrt(127)
        const float* zrow = a.Z + (size_t)rowc * (H * D);
        const float zs = 1024.0f, zinv = hdr[4] * (1.0f / 1024.0f);
        for (int hp = 0; hp < 4; ++hp) {
          const unsigned short* Wl = take();
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int h = 2 * hp + hh;
            float4 z[4][2];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              z[s][0] = *reinterpret_cast<const float4*>(zrow + h * D + 32 * s + 8 * rg);
              z[s][1] = *reinterpret_cast<const float4*>(zrow + h * D + 32 * s + 8 * rg + 4);
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              u32x4 bh, bl;
              unsigned hi, lo;
              split_pair(z[s][0].x * zs, z[s][0].y * zs, hi, lo); bh[0] = hi; bl[0] = lo;
              split_pair(z[s][0].z * zs, z[s][0].w * zs, hi, lo); bh[1] = hi; bl[1] = lo;
              split_pair(z[s][1].x * zs, z[s][1].y * zs, hi, lo); bh[2] = hi; bl[2] = lo;
              split_pair(z[s][1].z * zs, z[s][1].w * zs, hi, lo); bh[3] = hi; bl[3] = lo;
              const v8h ah = *reinterpret_cast<const v8h*>(Wl + ((hh * 4 + s) * 2) * 512 + lane * 8);
              const v8h al = *reinterpret_cast<const v8h*>(Wl + ((hh * 4 + s) * 2 + 1) * 512 + lane * 8);
              const v8h vbh = __builtin_bit_cast(v8h, bh), vbl = __builtin_bit_cast(v8h, bl);
              acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, vbh, acc, 0, 0, 0);
              if constexpr (TERMS == 3) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, vbl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, vbh, acc, 0, 0, 0);
              }
            }
            const float sg = valid ? a.SIG[(size_t)row * H + h] : 0.f;
            const float4 bvr = *reinterpret_cast<const float4*>(Vt + VT_BVR + 16 * h + 4 * rg);
            // static tile index: the head loop is unrolled by hand through the switch below
#define IG_ADD_HEAD(T) case T: ag[T][0] += acc[0] * zinv + bvr.x * sg; ag[T][1] += acc[1] * zinv + bvr.y * sg; \
                               ag[T][2] += acc[2] * zinv + bvr.z * sg; ag[T][3] += acc[3] * zinv + bvr.w * sg; break;
            switch (h) { IG_ADD_HEAD(0) IG_ADD_HEAD(1) IG_ADD_HEAD(2) IG_ADD_HEAD(3) IG_ADD_HEAD(4) IG_ADD_HEAD(5) IG_ADD_HEAD(6) IG_ADD_HEAD(7) }
#undef IG_ADD_HEAD
          }
        }
      }
      // gate / self projection / update (layers.py:94-99)
      f32x4 upd[8];
      {
        f32x4 xn[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) xn[t] = x[t];
        ln_regs<true, false>(xn, Vt + VT_LND_G, Vt + VT_LND_B, rg);
        u32x4 Xh[4], Xl[4];
        const float inv_x = frags_scaled(xn, Xh, Xl);
        const float inv_a = frags_scaled(ag, Bh, Bl);
        f32x4 ga[8], gx[8], sf[8];
        zero_acc(ga); zero_acc(gx); zero_acc(sf);
        gemm_unit(ga, Bh, Bl);
        gemm_unit(gx, Xh, Xl);
        gemm_unit(sf, Xh, Xl);
        const float ca = inv_a * hdr[5], cx = inv_x * hdr[5], cs = inv_x * hdr[6];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float4 bg = *reinterpret_cast<const float4*>(Vt + VT_BG + 16 * t + 4 * rg);
          const float4 bs = *reinterpret_cast<const float4*>(Vt + VT_BS + 16 * t + 4 * rg);
          const float bgv[4] = {bg.x, bg.y, bg.z, bg.w}, bsv[4] = {bs.x, bs.y, bs.z, bs.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float gate = 1.0f / (1.0f + expf(-((ga[t][r] * ca + gx[t][r] * cx) + bgv[r])));
            const float agv = ag[t][r];
            upd[t][r] = agv + gate * ((sf[t][r] * cs + bsv[r]) - agv);
          }
        }
      }
      {
        const float inv_u = frags_scaled(upd, Bh, Bl);
        f32x4 o[8];
        zero_acc(o);
        gemm_unit(o, Bh, Bl);
        scale_bias(o, inv_u * hdr[7], Vt + VT_BO, rg);
        ln_regs<true, false>(o, Vt + VT_LNP_G, Vt + VT_LNP_B, rg);
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] += o[t];                              // x1 = x + LN_post(out)
      }
      {
        f32x4 xf[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) xf[t] = x[t];
        ln_regs<true, false>(xf, Vt + VT_LNF_G, Vt + VT_LNF_B, rg);
        u32x4 Fh[4], Fl[4];
        const float inv_f = frags_scaled(xf, Fh, Fl);
        f32x4 f[8];
        zero_acc(f);
        for (int cc = 0; cc < 4; ++cc) {
          f32x4 hdn[8];
          zero_acc(hdn);
          gemm_unit(hdn, Fh, Fl);
          scale_bias(hdn, inv_f * hdr[8], Vt + VT_B1 + 128 * cc, rg);
#pragma unroll
          for (int t = 0; t < 8; ++t) hdn[t] = __builtin_elementwise_max(hdn[t], splat4(0.f));
          const float inv_h = frags_scaled(hdn, Bh, Bl);
          f32x4 part[8];
          zero_acc(part);
          gemm_unit(part, Bh, Bl);
          const float ch = inv_h * hdr[9];
#pragma unroll
          for (int t = 0; t < 8; ++t) f[t] = fma4(part[t], splat4(ch), f[t]);
        }
        scale_bias(f, 1.0f, Vt + VT_B2, rg);
        ln_regs<true, false>(f, Vt + VT_LNO_G, Vt + VT_LNO_B, rg);
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] += f[t];                              // x2 = x1 + LN_ffpost(ffn)
      }
      store_row(xrow, x, rg);
    }

    if (NP) {
      const float* hdr = Vt + VT_N_HDR;
      ln_regs<true, false>(x, Vt + VT_N_LN_G, Vt + VT_N_LN_B, rg);
      const float inv_n = frags_scaled(x, Bh, Bl);
      if (need_q) {
        f32x4 q[8];
        zero_acc(q);
        gemm_unit(q, Bh, Bl);
        scale_bias(q, inv_n * hdr[0], Vt + VT_N_BQ, rg);
        if (a.nQ) store_row(valid ? a.nQ + (size_t)row * D : nullptr, q, rg);
        if (need_u) {
          // u_h = q_h W'_kr,h: K = 16 per head (16x16x16 MFMA); the B fragment of head h is C tile h of q
          u32x4 Qh[4], Ql[4];
          const float cq = frags_scaled(q, Qh, Ql) * hdr[1];
          float* urow = valid ? a.nU + (size_t)row * (H * D) : nullptr;
          for (int hp = 0; hp < 4; ++hp) {
            const unsigned short* Wl = take();
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              // tiles 2 hp, 2 hp + 1 of q live in k-step fragment hp: words (0, 1) and (2, 3)
              u32x2 qh, ql;
#define IG_PICK(S) case S: qh = hh ? u32x2{Qh[S][2], Qh[S][3]} : u32x2{Qh[S][0], Qh[S][1]}; \
                           ql = hh ? u32x2{Ql[S][2], Ql[S][3]} : u32x2{Ql[S][0], Ql[S][1]}; break;
              switch (hp) { IG_PICK(0) IG_PICK(1) IG_PICK(2) default: IG_PICK(3) }
#undef IG_PICK
              const v4h vqh = __builtin_bit_cast(v4h, qh), vql = __builtin_bit_cast(v4h, ql);
#pragma unroll
              for (int ct = 0; ct < 8; ++ct) {
                const v4h ah = *reinterpret_cast<const v4h*>(Wl + ((hh * 8 + ct) * 2) * 256 + lane * 4);
                const v4h al = *reinterpret_cast<const v4h*>(Wl + ((hh * 8 + ct) * 2 + 1) * 256 + lane * 4);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, vqh, acc, 0, 0, 0);
                if constexpr (TERMS == 3) {
                  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, vql, acc, 0, 0, 0);
                  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(al, vqh, acc, 0, 0, 0);
                }
                if (urow)
                  *reinterpret_cast<float4*>(urow + (2 * hp + hh) * D + 16 * ct + 4 * rg) =
                      make_float4(acc[0] * cq, acc[1] * cq, acc[2] * cq, acc[3] * cq);
              }
            }
          }
        }
      }
      if (need_kv) {
        f32x4 kk[8];
        zero_acc(kk);
        gemm_unit(kk, Bh, Bl);
        scale_bias(kk, inv_n * hdr[2], nullptr, rg);
        store_row(valid ? a.nK + (size_t)row * D : nullptr, kk, rg);
        zero_acc(kk);
        gemm_unit(kk, Bh, Bl);
        scale_bias(kk, inv_n * hdr[3], Vt + VT_N_BV, rg);
        store_row(valid && a.nV ? a.nV + (size_t)row * D : nullptr, kk, rg);
      }
    }
  }
}

#if !IG_BF16_OPERANDS
template __global__ void k_attn_h<4, 3>(AttnHArgs);
template __global__ void k_attn_h<8, 3>(AttnHArgs);
#endif
template __global__ void k_attn_h<4, 1>(AttnHArgs);

// k_active_groups: the 16-row groups of the [S][A_cap] row layout that hold at least one row below n_agents[s] + margin
// (margin = rows a decode step may append), in ascending order.  One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void k_active_groups(ActiveGroupsArgs a) {
  __shared__ int wave_cnt[16];
  __shared__ int base;
  const int total = (a.S * a.A_cap + 15) / 16;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int g0 = 0; g0 < total; g0 += 1024) {
    const int g = g0 + threadIdx.x;
    bool on = false;
    if (g < total) {
      const int r0 = 16 * g, r1 = min(16 * g + 15, a.S * a.A_cap - 1);
      const int s0 = r0 / a.A_cap, s1 = r1 / a.A_cap;
      on = (r0 - s0 * a.A_cap) < a.n_agents[s0] + a.margin;            // first row of the group in its scene
      if (s1 != s0) on = on || a.n_agents[s1] + a.margin > 0;            // the group runs into the next scene
    }
    const unsigned long long m = __ballot(on);
    if (lane == 0) wave_cnt[w] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int k = 0; k < w; ++k) off += wave_cnt[k];
    if (on) a.groups[off + __popcll(m & ((1ull << lane) - 1))] = g;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < 16; ++k) t += wave_cnt[k]; base += t; }
    __syncthreads();
  }
  if (threadIdx.x == 0) *a.n_groups = base;
}

}  // namespace ig
