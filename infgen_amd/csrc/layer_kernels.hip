// Node-side kernels of one AttentionLayer (reference infgen/modules/layers.py:61-113) and the
// token/state heads.  The relative-position projections are ABSORBED into the query and the
// aggregate (never materialised per edge):
//     sim_e,h   = q_h . (k_j,h + W_kr,h LN(r_e))          = q_h . k_j,h + u_h . rhat_e + const(h)
//     sum_e a_e (v_j,h + W_vr,h LN(r_e) + b)             = sum_e a_e v_j,h + W'_vr,h z_h + b'_h sigma_h
// with rhat_e the affine-free LayerNorm of r_e, u_h = gamma (.) W_kr,h^T q_h, z_h = sum_e a_e rhat_e,
// W'_vr = W_vr diag(gamma), b' = W_vr beta + b_vr, sigma_h = sum_e a_e.  The per-destination constant
// q_h . W_kr,h beta cancels in the softmax (max-shift and ratio are shift invariant).
#include "kernels.h"

namespace ig {

// ------------------------------------------------------------------------------------------
// k_attn_pre: Xn = LN(X); Q = scale*(Xn Wq^T + bq); K = Xn Wk^T; V = Xn Wv^T + bv; U_h = Q_h W'_kr,h
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT, 2) void k_attn_pre(AttnPreArgs a);

// ------------------------------------------------------------------------------------------
// k_attn_post: agg = AGG + W'_vr z + b' sigma ; gate ; out-proj ; post-norm residual ; FFN residual
// The residual stream of the tile lives in registers (RowSeg layout), three LDS tiles hold the GEMM
// operands -> 50.7 KB of LDS, three workgroups per CU.  Optionally runs the next layer's
// prenorm + q/k/v/u projections on the freshly computed rows (saves a launch and an X round trip).
// ------------------------------------------------------------------------------------------
// q/k/v/u projections of one 32-row tile whose normalised input sits in LDS (Xn); `cur` carries the first
// B half of to_q (requested by the caller before its barrier)
__device__ __forceinline__ void pre_from_lds(const float* Xn, float* Qs, const float* P, int row0, int nvalid,
                                             float* Q, float* U, float* K, float* V, BHalf& cur) {
  const int n0 = 32 * wave_id();
  const int col = n0 + acc_col();
  const int lane = lane_id();
  {
    f32x16 acc = zero16();
    gemm128(acc, Xn, LDT, P + AL_WQ, 128, n0, cur, [&] { return b_load_half(P + AL_WK, 128, n0, 0); });
    const float bq = P[AL_BQ + col];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int r = acc_row(reg);
      const float q = acc[reg] + bq;
      Qs[r * LDT + col] = q;
      if (Q && r < nvalid) Q[(size_t)(row0 + r) * D + col] = q;
    }
  }
  if (K) {
    f32x16 acc = zero16();
    gemm128(acc, Xn, LDT, P + AL_WK, 128, n0, cur, [&] { return b_load_half(P + AL_WV, 128, n0, 0); });
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int r = acc_row(reg);
      if (r < nvalid) K[(size_t)(row0 + r) * D + col] = acc[reg];
    }
    f32x16 accv = zero16();
    gemm128(accv, Xn, LDT, P + AL_WV, 128, n0, cur, [&] { return cur; });
    const float bv = P[AL_BV + col];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int r = acc_row(reg);
      if (r < nvalid) V[(size_t)(row0 + r) * D + col] = accv[reg] + bv;
    }
  }
  if (U) {
    // u_h = q_h W'_kr,h : eight K = 16 GEMMs; the 16 B fragments of this wave's column slice are requested
    // up front, the barrier publishes Qs meanwhile
    const float* bp = P + AL_WKR + ((size_t)(n0 + (lane & 31))) * 8 + 4 * (lane >> 5);
    float4 bu[H][2];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      bu[h][0] = *reinterpret_cast<const float4*>(bp + h * (DH * D));
      bu[h][1] = *reinterpret_cast<const float4*>(bp + h * (DH * D) + 128 * 8);
    }
    __syncthreads();
    const float* ap = Qs + (lane & 31) * LDT + 4 * (lane >> 5);
#pragma unroll
    for (int h = 0; h < H; ++h) {
      f32x16 acc = zero16();
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const float4 a = *reinterpret_cast<const float4*>(ap + DH * h + 8 * g);
        const float4 b = bu[h][g];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
      }
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = acc_row(reg);
        if (r < nvalid) U[(size_t)(row0 + r) * (H * D) + h * D + col] = acc[reg];
      }
    }
  }
}

__global__ __launch_bounds__(NT, 2) void k_attn_post(AttnPostArgs a) {
  __shared__ __attribute__((aligned(16))) float B1[TR * LDT];   // Z head / LN_dst(x) -> LN_ffpre(x1) -> LN_next(x2)
  __shared__ __attribute__((aligned(16))) float B2[TR * LDT];   // agg -> to_out(...) -> ffn out
  __shared__ __attribute__((aligned(16))) float B3[TR * LDT];   // Z head / upd -> relu(hidden chunk) -> q of the next layer
  const int row0 = blockIdx.x * TR;
  const int nvalid = min(TR, a.rows - row0);
  if (nvalid <= 0) return;
  const int w = wave_id(), n0 = 32 * w, lane = lane_id();
  const float* P = a.pack;
  const int srow = seg_row();
  float* xrow = srow < nvalid ? a.X + (size_t)(row0 + srow) * D : nullptr;

  RowSeg x = seg_load(xrow);
  seg_store(B2 + srow * LDT, seg_load(srow < nvalid ? a.AGG + (size_t)(row0 + srow) * D : nullptr));

  if (a.has_pos) {
    // z-GEMM: out[row][16h + c] = sum_d Z[row][h][d] * B_h[d][c] on v_mfma_f32_16x16x4_f32
    // (lane l supplies A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; C: col = l&15, row = 4*(l>>4) + reg).
    // Z[:, h, :] is staged through LDS two heads per round (coalesced 512-B rows, next round's rows
    // prefetched into registers): wave w handles head 2p + (w >> 1), row half w & 1.
    const int i16 = lane & 15, kq = lane >> 4;
    const int mt = w & 1, hsel = w >> 1;
    const float* zbase = srow < nvalid ? a.Z + (size_t)(row0 + srow) * (H * D) : nullptr;
    RowSeg za = seg_load(zbase), zb = seg_load(zbase ? zbase + D : nullptr);
    for (int p = 0; p < 4; ++p) {
      seg_store(B1 + srow * LDT, za);
      seg_store(B3 + srow * LDT, zb);
      if (p + 1 < 4) {
        za = seg_load(zbase ? zbase + (2 * p + 2) * D : nullptr);
        zb = seg_load(zbase ? zbase + (2 * p + 3) * D : nullptr);
      }
      __syncthreads();
      const int h = 2 * p + hsel;
      const float* Zs = (hsel ? B3 : B1) + (mt * 16 + i16) * LDT + 4 * kq;
      const float* Bh = P + AL_WVR + h * (DH * D);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const float4 av = *reinterpret_cast<const float4*>(Zs + 16 * s);
        const float4 bv = *reinterpret_cast<const float4*>(Bh + ((s * 16 + i16) * 4 + kq) * 4);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
      }
      const int c = DH * h + i16;
      const float bvr = P[AL_BVR + c];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int rr = mt * 16 + 4 * kq + reg;
        const float sg = rr < nvalid ? a.SIG[(size_t)(row0 + rr) * H + h] : 0.f;
        B2[rr * LDT + c] += acc[reg] + bvr * sg;
      }
      __syncthreads();
    }
  }
  seg_store(B1 + srow * LDT, seg_layernorm(x, P + AL_LN_DST_G, P + AL_LN_DST_B, false));
  BHalf cur = b_load_half(P + AL_WG, 128, n0, 0);
  __syncthreads();

  // gate / self projection / update (layers.py:94-99)
  {
    f32x16 accg = zero16(), accs = zero16();
    gemm128(accg, B2, LDT, P + AL_WG, 128, n0, cur, [&] { return b_load_half(P + AL_WG + 16384, 128, n0, 0); });
    gemm128(accg, B1, LDT, P + AL_WG + 16384, 128, n0, cur, [&] { return b_load_half(P + AL_WS, 128, n0, 0); });
    gemm128(accs, B1, LDT, P + AL_WS, 128, n0, cur, [&] { return b_load_half(P + AL_WO, 128, n0, 0); });
    const int col = n0 + acc_col();
    const float bg = P[AL_BG + col], bs = P[AL_BS + col];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int r = acc_row(reg);
      const float gate = 1.0f / (1.0f + expf(-(accg[reg] + bg)));
      const float ag = B2[r * LDT + col];
      B3[r * LDT + col] = ag + gate * ((accs[reg] + bs) - ag);
    }
  }
  __syncthreads();
  {
    f32x16 acco = zero16();
    gemm128(acco, B3, LDT, P + AL_WO, 128, n0, cur, [&] { return b_load_half(P + AL_W1, 512, n0, 0); });
    acc_to_lds(acco, B2, LDT, n0, P + AL_BO);
  }
  __syncthreads();
  // x1 = x + LN_post(out) stays in registers; LN_ffpre(x1) feeds the FFN
  x = seg_add(x, seg_layernorm(seg_load(B2 + srow * LDT), P + AL_LN_POST_G, P + AL_LN_POST_B, false));
  seg_store(B1 + srow * LDT, seg_layernorm(x, P + AL_LN_FFPRE_G, P + AL_LN_FFPRE_B, false));
  __syncthreads();
  f32x16 accf = zero16();
  const float* NP = a.next_pack;
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) {
    const float* Wd = P + AL_W2 + (size_t)(16 * cc) * 128 * 8;
    f32x16 acch = zero16();
    gemm128(acch, B1, LDT, P + AL_W1, 512, 128 * cc + n0, cur, [&] { return b_load_half(Wd, 128, n0, 0); });
    {
      const int col = n0 + acc_col();
      const float b1 = P[AL_B1 + 128 * cc + col];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) B3[acc_row(reg) * LDT + col] = fmaxf(acch[reg] + b1, 0.f);
    }
    __syncthreads();
    gemm128(accf, B3, LDT, Wd, 128, n0, cur, [&] {
      return cc + 1 < 4 ? b_load_half(P + AL_W1, 512, 128 * (cc + 1) + n0, 0)
                        : (NP ? b_load_half(NP + AL_WQ, 128, n0, 0) : cur);
    });
    __syncthreads();
  }
  acc_to_lds(accf, B2, LDT, n0, P + AL_B2);
  __syncthreads();
  x = seg_add(x, seg_layernorm(seg_load(B2 + srow * LDT), P + AL_LN_FFPOST_G, P + AL_LN_FFPOST_B, false));
  seg_store(xrow, x);                                             // x2 = x1 + LN(ffn)
  if (NP) {
    seg_store(B1 + srow * LDT, seg_layernorm(x, NP + AL_LN_DST_G, NP + AL_LN_DST_B, false));
    __syncthreads();
    pre_from_lds(B1, B3, NP, row0, nvalid, a.nQ, a.nU, a.nK, a.nV, cur);
  }
}

__global__ __launch_bounds__(NT, 2) void k_attn_pre(AttnPreArgs a) {
  __shared__ __attribute__((aligned(16))) float Xs[TR * LDT];
  __shared__ __attribute__((aligned(16))) float Qs[TR * LDT];
  const int row0 = blockIdx.x * TR;
  const int nvalid = min(TR, a.rows - row0);
  if (nvalid <= 0) return;
  const int n0 = 32 * wave_id();
  const int srow = seg_row();
  const float* g = a.pack + (a.use_src_ln ? AL_LN_SRC_G : AL_LN_DST_G);
  const float* b = a.pack + (a.use_src_ln ? AL_LN_SRC_B : AL_LN_DST_B);
  BHalf cur = b_load_half(a.pack + ((a.Q || a.U) ? AL_WQ : AL_WK), 128, n0, 0);
  seg_store(Xs + srow * LDT, seg_layernorm(seg_load(srow < nvalid ? a.X + (size_t)(row0 + srow) * D : nullptr), g, b, false));
  __syncthreads();
  if (a.Q || a.U) {
    pre_from_lds(Xs, Qs, a.pack, row0, nvalid, a.Q, a.U, a.K, a.V, cur);
  } else {
    // K/V only (map tokens as a bipartite source)
    const int col = n0 + acc_col();
    f32x16 acc = zero16();
    gemm128(acc, Xs, LDT, a.pack + AL_WK, 128, n0, cur, [&] { return b_load_half(a.pack + AL_WV, 128, n0, 0); });
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int r = acc_row(reg);
      if (r < nvalid && a.K) a.K[(size_t)(row0 + r) * D + col] = acc[reg];
    }
    f32x16 accv = zero16();
    gemm128(accv, Xs, LDT, a.pack + AL_WV, 128, n0, cur, [&] { return cur; });
    const float bv = a.pack[AL_BV + col];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int r = acc_row(reg);
      if (r < nvalid && a.V) a.V[(size_t)(row0 + r) * D + col] = accv[reg] + bv;
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_heads: token_predict_head (128 -> 128 LN ReLU -> token_size) with fused arg-max and optional
// logits store; state_predict_head (128 -> 128 LN ReLU -> 3) arg-max.
// reference agent_decoder.py:2161-2167 (greedy: motion_beam_size = 1)
// MLPLayer pack: [0] P(128,128) W0 | [16384] b0 | [16512] ln g | [16640] ln b | [16768] W3 ... | b3
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT, 2) void k_heads(HeadsArgs a) {
  __shared__ __attribute__((aligned(16))) float Xs[TR * LDT];
  __shared__ __attribute__((aligned(16))) float Hs[TR * LDT];
  __shared__ float red_v[4][TR];
  __shared__ int red_i[4][TR];
  const int row0 = blockIdx.x * TR;
  const int nvalid = min(TR, a.rows - row0);
  if (nvalid <= 0) return;
  const int w = wave_id(), n0 = 32 * w, lane = lane_id();
  BHalf cur = b_load_half(a.tok_pack, 128, n0, 0);
  stage_rows_128(Xs, [&](int r) { return a.X + (size_t)(row0 + r) * D; }, nvalid);
  __syncthreads();
  {
    const float* P = a.tok_pack;
    f32x16 acc = zero16();
    const float* W3p = P + 16768;
    const int nchunk = a.token_size / 128, ns = a.nsplit > 1 ? a.nsplit : 1, sp = blockIdx.y;
    const int p0 = nchunk * sp / ns, p1 = nchunk * (sp + 1) / ns;       // this workgroup's logit chunks
    gemm128(acc, Xs, LDT, P, 128, n0, cur, [&] { return b_load_half(W3p, a.token_size, 128 * p0 + n0, 0); });
    acc_to_lds(acc, Hs, LDT, n0, P + 16384);
    __syncthreads();
    ln_tile(Hs, LDT, Hs, LDT, P + 16512, P + 16640, true);
    __syncthreads();
    const float* W3 = P + 16768;
    const float* b3 = W3 + (size_t)128 * a.token_size;
    float bv[16];
    int bi[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) { bv[reg] = -INFINITY; bi[reg] = 0; }
    for (int p = p0; p < p1; ++p) {
      const int c0 = 128 * p + n0;
      f32x16 acc2 = zero16();
      const bool last = p + 1 >= p1;
      gemm128(acc2, Hs, LDT, W3, a.token_size, c0, cur, [&] {
        return last ? b_load_half(a.st_pack, 128, n0, 0) : b_load_half(W3, a.token_size, c0 + 128, 0);
      });
      const int col = c0 + acc_col();
      const float bb = b3[col];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const float v = acc2[reg] + bb;
        const int r = acc_row(reg);
        if (a.logits && r < nvalid) a.logits[(size_t)(row0 + r) * a.token_size + col] = v;
        if (v > bv[reg]) { bv[reg] = v; bi[reg] = col; }
      }
    }
    // reduce over the 32 lanes (columns) that share a row; ties -> smaller index
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      float v = bv[reg];
      int i = bi[reg];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(i, o, 64);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
      }
      if ((lane & 31) == 0) { red_v[w][acc_row(reg)] = v; red_i[w][acc_row(reg)] = i; }
    }
    __syncthreads();
    if (threadIdx.x < nvalid) {
      const int r = threadIdx.x;
      float v = red_v[0][r];
      int i = red_i[0][r];
      for (int ww = 1; ww < 4; ++ww) {
        const float ov = red_v[ww][r];
        const int oi = red_i[ww][r];
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
      }
      if (ns == 1) a.next_token[row0 + r] = i;
      else {
        // ordered key: larger value first, then the smaller index (torch.argmax: first maximum); -0 counts as +0
        unsigned u = __float_as_uint(v + 0.0f);
        u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
        atomicMax(a.part + row0 + r, ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - (unsigned)i));
      }
    }
    if (sp != 0) return;                 // the state head is the first split's
  }
  __syncthreads();
  {
    const float* P = a.st_pack;
    f32x16 acc = zero16();
    gemm128(acc, Xs, LDT, P, 128, n0, cur, [&] { return cur; });
    acc_to_lds(acc, Hs, LDT, n0, P + 16384);
    __syncthreads();
    ln_tile(Hs, LDT, Hs, LDT, P + 16512, P + 16640, true);
    __syncthreads();
    if (threadIdx.x < nvalid) {
      const int r = threadIdx.x;
      const float* W3 = P + 16768;        // [3][128] row-major
      const float* b3 = W3 + 3 * 128;
      float best = -INFINITY;
      int bi = 0;
      for (int o = 0; o < 3; ++o) {
        float s = 0.f;
        for (int k = 0; k < 128; ++k) s = fmaf(Hs[r * LDT + k], W3[o * 128 + k], s);
        s += b3[o];
        if (s > best) { best = s; bi = o; }
      }
      a.next_state[row0 + r] = bi;
    }
  }
}

__global__ __launch_bounds__(NT) void k_heads_finish(const unsigned long long* part, int rows, int* next_token) {
  const int r = blockIdx.x * NT + threadIdx.x;
  if (r < rows) next_token[r] = (int)(0xffffffffu - (unsigned)(part[r] & 0xffffffffull));
}

}  // namespace ig
