// k_attn_hs: the node side of one AttentionLayer (reference infgen/modules/layers.py:61-113) for FEW rows - the arithmetic
// of k_attn_h (attn_h.hip: fp16 matrix pipe, three-term hi/lo split) laid out for latency instead of throughput.
//
// k_attn_h gives a wave 16 rows and all eight 16-feature tiles of every GEMM: one tile's post + pre chain is ~1250 MFMAs
// behind each other in one wave (69 us for a lone tile); the fp32-MFMA kernels k_attn_post / k_attn_pre need 46-53 us per
// launch whatever the row count.  That latency is what the insertion sub-loop (512 seed rows, ~15 dependent node launches
// per iteration) and small batches (8 scenes = 512 rows: 13 of 31 ms per rollout) pay.  Here ONE 16-row group is a
// workgroup of eight waves and wave w computes feature tile w of every GEMM (12 MFMAs instead of 96), with its A fragments
// read straight from L2 in the packed quarter layout (1 KB per wave instruction, requested a GEMM ahead); the C tiles
// are exchanged through LDS (one barrier, ping-pong buffers) and every wave then holds the whole 128-feature rows in
// registers exactly like k_attn_h, so LayerNorm and the per-row scaled hi/lo split are k_attn_h's own code (redundant in
// the eight waves: vector work is cheap here, the chain is what costs).  Per accumulator the products are issued in
// k_attn_h's order (k-step major; hi x hi, hi x lo, lo x hi), so the GEMM results are bit-identical to k_attn_h's.
//   z-GEMM / u-GEMM: head h is feature tile h - wave w takes head w (k_edge_fused's per-(row, head) scaling for u).
// (the LayerNorm tables stay read-where-used here: with split.cuh's ln_fetch the kernel needs 40 registers more than it has)
#define IG_LN_PREFETCH 0
#include "kernels.h"
#include "layout.h"
#include "tile.cuh"
#include "split.cuh"
#include "attn_h.cuh"

namespace ig {

namespace {

template <int TERMS>
struct AFrag {                 // the four k-steps of ONE feature tile of a 128 x 128 matrix
  v8h h[4], l[4];
  // W: first quarter of the matrix (quarter = k-step: [tile 8][hi, lo][lane 64][8 fp16])
  __device__ __forceinline__ void load(const unsigned short* W, int w, int lane) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      h[s] = *reinterpret_cast<const v8h*>(W + (size_t)s * QUARTER + w * 1024 + lane * 8);
      if constexpr (TERMS == 3) l[s] = *reinterpret_cast<const v8h*>(W + (size_t)s * QUARTER + w * 1024 + 512 + lane * 8);
    }
  }
};

template <int TERMS>
__device__ __forceinline__ f32x4 mm_step(const AFrag<TERMS>& f, int s, u32x4 Bh, u32x4 Bl, f32x4 acc) {
  const v8h bh = __builtin_bit_cast(v8h, Bh), bl = __builtin_bit_cast(v8h, Bl);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.h[s], bh, acc, 0, 0, 0);
  if constexpr (TERMS == 3) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.h[s], bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.l[s], bh, acc, 0, 0, 0);
  }
  return acc;
}

template <int TERMS>
__device__ __forceinline__ f32x4 mm_own(const AFrag<TERMS>& f, const u32x4 (&Bh)[4], const u32x4 (&Bl)[4]) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; ++s) acc = mm_step<TERMS>(f, s, Bh[s], Bl[s], acc);
  return acc;
}

// every wave leaves its C tile in LDS and takes all eight: the whole rows, in k_attn_h's register layout
__device__ __forceinline__ void exchange(float* buf, f32x4 own, f32x4 (&v)[8], int w, int lane) {
  *reinterpret_cast<float4*>(buf + (w * 64 + lane) * 4) = make_float4(own[0], own[1], own[2], own[3]);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 8; ++t) v[t] = lds4(buf + (t * 64 + lane) * 4);
}

}  // namespace

template <int TERMS>
__global__ __launch_bounds__(512, 1) void k_attn_hs(AttnHArgs a) {
  __shared__ __attribute__((aligned(16))) float Xb[2][8 * 64 * 4];
  __shared__ __attribute__((aligned(16))) float Hb[4][8 * 64 * 4];      // the FFN's hidden layer, four 128-wide chunks
  __shared__ __attribute__((aligned(16))) float Vt[VT_SIZE + 4];        // (+ 4: the spare slot)
  const int tid = threadIdx.x;
  if (warm_l2(a.warm, blockIdx.x, tid, 512)) return;            // kernels.h: WarmArgs
  const int ngroups = a.groups ? *a.n_groups : (a.rows + 15) / 16;
  if ((int)blockIdx.x >= ngroups) return;
  const int lane = tid & 63, w = tid >> 6;
  const int j = lane & 15, rg = lane >> 4;
  const float* P = a.pack;
  const float* NP = a.next_pack;
  const bool need_q = NP && (a.nQ || a.nU), need_u = NP && a.nU, need_kv = NP && a.nK;
  const unsigned short* post = P ? reinterpret_cast<const unsigned short*>(P + AH_POST) : nullptr;
  const unsigned short* pre = NP ? reinterpret_cast<const unsigned short*>(NP + AH_PRE) : nullptr;

  const int grp = a.groups ? a.groups[blockIdx.x] : (int)blockIdx.x;
  const int row = grp * 16 + j;
  const bool valid = row < a.rows;
  float* xrow = valid ? a.X + (size_t)row * D : nullptr;
  // Request order = arrival order (one in-order counter): the small vector tables first (their LDS copy and the barrier then
  // wait for 30 KB instead of for 220 KB), the rows next (the first LayerNorm runs while the weights are still arriving), then the
  // A fragments of the first three GEMMs.  The fragment sets are three register sets used round-robin by the layer's fifteen
  // 128 x 128 matrices in consumption order (gate a / gate x / self / out / W1 chunks 0..3 / W2 chunks 0..3 / q / k / v): set i % 3
  // is requested again as soon as GEMM i has issued its products, i.e. two to three GEMMs ahead of its use instead of one.
  // Loads only moved: arithmetic and results are unchanged (bitwise).
  // (tables: two 16-byte slots per thread, loaded unconditionally - a thread without a slot re-reads the first one and stores to
  // the spare slot behind the table; rows beyond the end load the last row and are never stored.  A conditional load gets a basic
  // block of its own in which hipcc waits for it before the next one is issued: every `valid ? load : 0` and every per-table copy
  // loop of the first version was a memory round trip of ~3,000 cycles on its own.)
  const int d0 = 4 * tid, d1 = 4 * (tid + 512);
  const float* anyp = P ? P : NP;
  const float* ts0 = attn_table_src(d0, P, NP, a.next_src_ln);
  const float* ts1 = d1 < VT_SIZE ? attn_table_src(d1, P, NP, a.next_src_ln) : nullptr;
  const float4 tv0 = *reinterpret_cast<const float4*>(ts0 ? ts0 : anyp);
  const float4 tv1 = *reinterpret_cast<const float4*>(ts1 ? ts1 : anyp);
  const int rowc = valid ? row : a.rows - 1;
  f32x4 x[8];
  load_row_nc(x, a.X + (size_t)rowc * D, rg);
  const int own = 16 * w + 4 * rg;               // first of this lane's four features of the wave's tile
  // this wave's tile of agg, and (has_pos = 0) the whole rows for the gate GEMM; the first three fragment sets - all of it
  // unconditional (a launch without a post part reads its x rows / its q, k, v matrices here, which it needs anyway or ignores)
  const float* aggp = (P ? a.AGG : a.X) + (size_t)rowc * D;
  f32x4 ago;
  { const float4 t = *reinterpret_cast<const float4*>(aggp + own); ago = f32x4{t.x, t.y, t.z, t.w}; }
  f32x4 ag[8];
  load_row_nc(ag, aggp, rg);
  AFrag<TERMS> fa, fb, fc;
  fa.load(P ? post + 4 * QUARTER : pre, w, lane);
  fb.load(P ? post + 8 * QUARTER : pre + 8 * QUARTER, w, lane);
  fc.load(P ? post + 12 * QUARTER : pre + 12 * QUARTER, w, lane);
  *reinterpret_cast<float4*>(Vt + (ts0 ? d0 : VT_SIZE)) = tv0;
  *reinterpret_cast<float4*>(Vt + (ts1 ? d1 : VT_SIZE)) = tv1;
  __syncthreads();

  int pp = 0;                                    // ping-pong index of the exchange buffer
  u32x4 Bh[4], Bl[4];

  if (P) {
    const float* hdr = Vt + VT_HDR;
    if (a.has_pos) {
      // z-GEMM of head w (k_attn_h's, k_edge_fused's phase 3): B fragments straight from Z[row][w][:], |z| <= sqrt(127)
      const float* zrow = a.Z + (size_t)rowc * (H * D) + w * D;
      const unsigned short* Wl = post + (size_t)(w >> 1) * QUARTER + (size_t)((w & 1) * 4) * 2 * 512 + lane * 8;
      const float zs = 1024.0f, zinv = hdr[4] * (1.0f / 1024.0f);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4 z0 = *reinterpret_cast<const float4*>(zrow + 32 * s + 8 * rg);
        const float4 z1 = *reinterpret_cast<const float4*>(zrow + 32 * s + 8 * rg + 4);
        u32x4 bh, bl;
        unsigned hi, lo;
        split_pair(z0.x * zs, z0.y * zs, hi, lo); bh[0] = hi; bl[0] = lo;
        split_pair(z0.z * zs, z0.w * zs, hi, lo); bh[1] = hi; bl[1] = lo;
        split_pair(z1.x * zs, z1.y * zs, hi, lo); bh[2] = hi; bl[2] = lo;
        split_pair(z1.z * zs, z1.w * zs, hi, lo); bh[3] = hi; bl[3] = lo;
        const v8h ah = *reinterpret_cast<const v8h*>(Wl + (s * 2) * 512);
        const v8h al = *reinterpret_cast<const v8h*>(Wl + (s * 2 + 1) * 512);
        const v8h vbh = __builtin_bit_cast(v8h, bh), vbl = __builtin_bit_cast(v8h, bl);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, vbh, acc, 0, 0, 0);
        if constexpr (TERMS == 3) {
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, vbl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, vbh, acc, 0, 0, 0);
        }
      }
      const float sg = a.SIG[(size_t)rowc * H + w];
      const f32x4 bvr = lds4(Vt + VT_BVR + own);
#pragma unroll
      for (int r = 0; r < 4; ++r) ago[r] += acc[r] * zinv + bvr[r] * sg;
      exchange(Xb[pp], ago, ag, w, lane); pp ^= 1;
    }
    // gate / self projection / update (layers.py:94-99)
    f32x4 upd_own;
    {
      f32x4 xn[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) xn[t] = x[t];
      ln_regs<true, false>(xn, Vt + VT_LND_G, Vt + VT_LND_B, rg);
      u32x4 Xh[4], Xl[4];
      const float inv_x = frags_scaled(xn, Xh, Xl);
      const float inv_a = frags_scaled(ag, Bh, Bl);
      f32x4 ga = {0.f, 0.f, 0.f, 0.f}, gx = ga, sf = ga;
#pragma unroll
      for (int s = 0; s < 4; ++s) {             // three independent accumulators: no back-to-back dependent MFMAs
        ga = mm_step<TERMS>(fa, s, Bh[s], Bl[s], ga);
        gx = mm_step<TERMS>(fb, s, Xh[s], Xl[s], gx);
        sf = mm_step<TERMS>(fc, s, Xh[s], Xl[s], sf);
      }
      fa.load(post + 16 * QUARTER, w, lane);    // Wo
      fb.load(post + 20 * QUARTER, w, lane);    // W1, chunk 0
      fc.load(post + (size_t)28 * QUARTER, w, lane);                                      // W1, chunk 1
      const float ca = inv_a * hdr[5], cx = inv_x * hdr[5], cs = inv_x * hdr[6];
      const f32x4 bg = lds4(Vt + VT_BG + own), bs = lds4(Vt + VT_BS + own);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gate = 1.0f / (1.0f + expf(-((ga[r] * ca + gx[r] * cx) + bg[r])));
        upd_own[r] = ago[r] + gate * ((sf[r] * cs + bs[r]) - ago[r]);
      }
    }
    {
      f32x4 t8[8];
      exchange(Xb[pp], upd_own, t8, w, lane); pp ^= 1;
      const float inv_u = frags_scaled(t8, Bh, Bl);
      f32x4 o = mm_own<TERMS>(fa, Bh, Bl);
      fa.load(post + (size_t)36 * QUARTER, w, lane);                                      // W1, chunk 2
      o = fma4(o, splat4(inv_u * hdr[7]), lds4(Vt + VT_BO + own));
      exchange(Xb[pp], o, t8, w, lane); pp ^= 1;
      ln_regs<true, false>(t8, Vt + VT_LNP_G, Vt + VT_LNP_B, rg);
#pragma unroll
      for (int t = 0; t < 8; ++t) x[t] += t8[t];                              // x1 = x + LN_post(out)
    }
    {
      f32x4 t8[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) t8[t] = x[t];
      ln_regs<true, false>(t8, Vt + VT_LNF_G, Vt + VT_LNF_B, rg);
      u32x4 Fh[4], Fl[4];
      const float inv_f = frags_scaled(t8, Fh, Fl);
      f32x4 f = {0.f, 0.f, 0.f, 0.f};
      // FFN: the four 128-wide chunks of the hidden layer first (W1 chunk cc from the same B fragments; A fragments alternate
      // between fa and fc, requested a chunk ahead), parked in LDS behind ONE barrier, then the four W2 chunks - three barriers
      // fewer than chunk-by-chunk; per chunk the arithmetic (scale, products, order of the partial sums) is unchanged
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        // W1 chunks 0, 1, 2, 3 sit in sets fb, fc, fa, fb; the freed set takes W1 chunk 3 / W2 chunks 0, 1, 2
        f32x4 hd = cc == 1 ? mm_own<TERMS>(fc, Fh, Fl) : cc == 2 ? mm_own<TERMS>(fa, Fh, Fl) : mm_own<TERMS>(fb, Fh, Fl);
        if (cc == 0) fb.load(post + (size_t)44 * QUARTER, w, lane);                       // W1, chunk 3
        if (cc == 1) fc.load(post + (size_t)24 * QUARTER, w, lane);                       // W2, chunk 0
        if (cc == 2) fa.load(post + (size_t)32 * QUARTER, w, lane);                       // W2, chunk 1
        if (cc == 3) fb.load(post + (size_t)40 * QUARTER, w, lane);                       // W2, chunk 2
        hd = fma4(hd, splat4(inv_f * hdr[8]), lds4(Vt + VT_B1 + 128 * cc + own));
        hd = __builtin_elementwise_max(hd, splat4(0.f));
        *reinterpret_cast<float4*>(Hb[cc] + (w * 64 + lane) * 4) = make_float4(hd[0], hd[1], hd[2], hd[3]);
      }
      __syncthreads();
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
        for (int t = 0; t < 8; ++t) t8[t] = lds4(Hb[cc] + (t * 64 + lane) * 4);
        const float inv_h = frags_scaled(t8, Bh, Bl);
        // W2 chunks 0, 1, 2, 3 sit in sets fc, fa, fb, fc; the freed set takes W2 chunk 3 / the next layer's q, k, v
        const f32x4 part = cc == 1 ? mm_own<TERMS>(fa, Bh, Bl) : cc == 2 ? mm_own<TERMS>(fb, Bh, Bl) : mm_own<TERMS>(fc, Bh, Bl);
        if (cc == 0) fc.load(post + (size_t)48 * QUARTER, w, lane);                       // W2, chunk 3
        if (cc == 1 && need_q) fa.load(pre, w, lane);
        if (cc == 2 && need_kv) fb.load(pre + 8 * QUARTER, w, lane);
        if (cc == 3 && need_kv) fc.load(pre + 12 * QUARTER, w, lane);
        f = fma4(part, splat4(inv_h * hdr[9]), f);
      }
      f = fma4(f, splat4(1.0f), lds4(Vt + VT_B2 + own));
      exchange(Xb[pp], f, t8, w, lane); pp ^= 1;
      ln_regs<true, false>(t8, Vt + VT_LNO_G, Vt + VT_LNO_B, rg);
#pragma unroll
      for (int t = 0; t < 8; ++t) x[t] += t8[t];                              // x2 = x1 + LN_ffpost(ffn)
    }
    if (w == 0) store_row(xrow, x, rg);
  }

  if (NP) {
    const float* hdr = Vt + VT_N_HDR;
    ln_regs<true, false>(x, Vt + VT_N_LN_G, Vt + VT_N_LN_B, rg);        // (q / k / v fragments: requested under the FFN, or at the top)
    const float inv_n = frags_scaled(x, Bh, Bl);
    if (need_q) {
      f32x4 q = mm_own<TERMS>(fa, Bh, Bl);
      q = fma4(q, splat4(inv_n * hdr[0]), lds4(Vt + VT_N_BQ + own));
      if (a.nQ && valid) *reinterpret_cast<float4*>(a.nQ + (size_t)row * D + own) = make_float4(q[0], q[1], q[2], q[3]);
      if (need_u) {
        // u_w = q_w W'_kr,w (K = 16: v_mfma_f32_16x16x16_f16): the head's query is this wave's own C tile; per (row, head)
        // power-of-two scale into the fp16 range (k_edge_fused's phase 1)
        const unsigned short* Wk = pre + (size_t)(4 + (w >> 1)) * QUARTER + (size_t)((w & 1) * 8) * 2 * 256 + lane * 4;
        float m = fmaxf(fmaxf(fabsf(q[0]), fabsf(q[1])), fmaxf(fabsf(q[2]), fabsf(q[3])));
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
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
        float* urow = valid ? a.nU + (size_t)row * (H * D) + w * D + 4 * rg : nullptr;
        f32x4 acc[8];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
          const v4h ah = *reinterpret_cast<const v4h*>(Wk + (ct * 2) * 256);
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, vqh, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          if constexpr (TERMS == 3) {
            const v4h al = *reinterpret_cast<const v4h*>(Wk + (ct * 2 + 1) * 256);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, vql, acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(al, vqh, acc[ct], 0, 0, 0);
          }
        }
        if (urow) {
#pragma unroll
          for (int ct = 0; ct < 8; ++ct)
            *reinterpret_cast<float4*>(urow + 16 * ct) = make_float4(acc[ct][0] * cq, acc[ct][1] * cq, acc[ct][2] * cq, acc[ct][3] * cq);
        }
      }
    }
    if (need_kv) {
      f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = kk;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        kk = mm_step<TERMS>(fb, s, Bh[s], Bl[s], kk);
        vv = mm_step<TERMS>(fc, s, Bh[s], Bl[s], vv);
      }
      kk = kk * splat4(inv_n * hdr[2]);
      vv = fma4(vv, splat4(inv_n * hdr[3]), lds4(Vt + VT_N_BV + own));
      if (valid) {
        *reinterpret_cast<float4*>(a.nK + (size_t)row * D + own) = make_float4(kk[0], kk[1], kk[2], kk[3]);
        if (a.nV) *reinterpret_cast<float4*>(a.nV + (size_t)row * D + own) = make_float4(vv[0], vv[1], vv[2], vv[3]);
      }
    }
  }
}

#if !IG_BF16_OPERANDS
template __global__ void k_attn_hs<3>(AttnHArgs);
#endif
template __global__ void k_attn_hs<1>(AttnHArgs);

}  // namespace ig
