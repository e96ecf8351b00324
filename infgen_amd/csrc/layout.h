// Packed-weight layouts shared by the kernels and (through infgen_layout_query) the host packer.
// All offsets are in floats.  "P(K,N)" = a GEMM operand packed as Wp[K/8][N][8] (tile.cuh).
#pragma once

namespace ig {

// ---- AttentionLayer (reference infgen/modules/layers.py:16-113) ------------------------------
enum AttnLayout : int {
  AL_LN_SRC_G = 0,                       // attn_prenorm_x_src
  AL_LN_SRC_B = AL_LN_SRC_G + 128,
  AL_LN_DST_G = AL_LN_SRC_B + 128,       // attn_prenorm_x_dst (== src when not bipartite)
  AL_LN_DST_B = AL_LN_DST_G + 128,
  AL_WQ = AL_LN_DST_B + 128,             // P(128,128) of head_dim^-0.5 * to_q.weight
  AL_BQ = AL_WQ + 16384,                 // head_dim^-0.5 * to_q.bias
  AL_WK = AL_BQ + 128,                   // P(128,128) to_k
  AL_WV = AL_WK + 16384,                 // P(128,128) to_v
  AL_BV = AL_WV + 16384,
  AL_WKR = AL_BV + 128,                  // 8 x P(16,128): B_h[c][d] = to_k_r.weight[16h+c][d] * ln_r.gamma[d]
  AL_WVR = AL_WKR + 16384,               // 8 x [8][16][4][4]: B_h[d][c] = to_v_r.weight[16h+c][d] * ln_r.gamma[d]
  AL_BVR = AL_WVR + 16384,               // to_v_r.weight @ ln_r.beta + to_v_r.bias
  AL_WS = AL_BVR + 128,                  // P(128,128) to_s
  AL_BS = AL_WS + 16384,
  AL_WG = AL_BS + 128,                   // P(256,128) to_g  (k < 128: agg part, k >= 128: x_dst part)
  AL_BG = AL_WG + 32768,
  AL_WO = AL_BG + 128,                   // P(128,128) to_out
  AL_BO = AL_WO + 16384,
  AL_LN_POST_G = AL_BO + 128,
  AL_LN_POST_B = AL_LN_POST_G + 128,
  AL_LN_FFPRE_G = AL_LN_POST_B + 128,
  AL_LN_FFPRE_B = AL_LN_FFPRE_G + 128,
  AL_W1 = AL_LN_FFPRE_B + 128,           // P(128,512) ff_mlp.0
  AL_B1 = AL_W1 + 65536,
  AL_W2 = AL_B1 + 512,                   // P(512,128) ff_mlp.3
  AL_B2 = AL_W2 + 65536,
  AL_LN_FFPOST_G = AL_B2 + 128,
  AL_LN_FFPOST_B = AL_LN_FFPOST_G + 128,
  AL_SIZE_F32 = AL_LN_FFPOST_B + 128,
  // ---- fp16-split section (k_attn_h, attn_h.hip): inverse weight scales, then quarter-matrices (8192 fp16 =
  // 4096 floats each; packing.pack_matrix_h / pack_wvr_h / pack_wkr_h) in consumption order
  AH_HDR = AL_SIZE_F32,                  // [16]: 1/scale of  0 wq  1 wkr  2 wk  3 wv  4 wvr  5 wg  6 ws  7 wo  8 w1  9 w2; 10, 11 max |gamma|, |beta| of ff_prenorm; 12, 13 of attn_prenorm_x_dst; 14: AH_HDR_VERSION
  AH_PRE = AH_HDR + 16,                  // 16 quarters: Wq (4)  W'kr (4: two heads each)  Wk (4)  Wv (4)
  AH_POST = AH_PRE + 16 * 4096,          // 52 quarters: W'vr (4: two heads each)  Wg[:, :128] (4)  Wg[:, 128:] (4)  Ws (4)
                                         //   Wo (4)  then per 128-wide FFN chunk c: W1[128c:128c+128, :] (4)  W2[:, 128c:128c+128] (4)
  AL_SIZE = AH_POST + 52 * 4096,
};

// ---- FourierEmbedding (layers.py:116-160), n input dims (n <= 4) -------------------------------
// header, then n per-dim blocks, then the tail
enum FourierLayout : int {
  FE_FREQ = 0,                 // [4][64] freqs.weight (rows >= n unused)
  FE_DIM0 = 256,               // start of per-dim blocks
  // per-dim block
  FD_W1 = 0,                   // P(128,128): k < 64 -> cos weights, k >= 64 -> sin weights (mlps.i.0.weight[:, :128])
  FD_W1X = FD_W1 + 16384,      // mlps.i.0.weight[:, 128]
  FD_B1 = FD_W1X + 128,
  FD_LN_G = FD_B1 + 128,
  FD_LN_B = FD_LN_G + 128,
  FD_W2 = FD_LN_B + 128,       // P(128,128) mlps.i.3.weight
  FD_SIZE = FD_W2 + 16384,
  // tail (offsets relative to FE_DIM0 + n * FD_SIZE)
  FT_B2SUM = 0,                // sum_i mlps.i.3.bias
  FT_LN_G = 128,
  FT_LN_B = 256,
  FT_W3 = 384,                 // P(128,128) to_out.2
  FT_B3 = FT_W3 + 16384,
  FT_SIZE = FT_B3 + 128,
};
__host__ __device__ inline int fourier_pack_size_f32(int n) { return FE_DIM0 + n * FD_SIZE + FT_SIZE; }

// ---- fp16-split section appended to the Fourier pack (k_fourier_h, fourier_h.hip) ---------------
// a table of fp32 per-feature vectors (copied to LDS), then 2 (2 n + 1) half-matrices of fp16
// MFMA A-fragments in consumption order  W1_0 W2_0 W1_1 W2_1 ... W3 :
//   half-matrix = [k-step 4][feature tile 4][hi, lo][lane 64][8 fp16]       (16384 fp16 = 8192 floats)
//   element (ks = 4 half + k-step, ft, lane = (i, g), s) = W[32 ft + i][k] * 2^sw with
//     k = 16 ks + 8 g + s                                   for W1 (input = [cos 64 | sin 64])
//     k = 32 (ks >> 1) + 16 (ks & 1) + (s & 3) + 8 (s >> 2) + 4 g    for W2 / W3 (input = C registers of the previous GEMM)
enum FourierHLayout : int {
  FH_HDR = 0,                  // [0..3] 1 / (sw1_i * sf)   [4] 1 / (sw2 * sa1)   [5] 1 / (sw3 * sa2)   [6] sf (feature prescale)
  FH_FREQ = 16,                // [4][64] freqs.weight
  FH_DIM0 = FH_FREQ + 256,     // four per-dim blocks
  FHD_WX = 0,                  // mlps.i.0.weight[:, 128]
  FHD_B1 = 128,
  FHD_G1 = 256,                // mlps.i.1.weight * sa1
  FHD_BE1 = 384,               // mlps.i.1.bias * sa1
  FHD_SIZE = 512,
  FH_TAIL = FH_DIM0 + 4 * FHD_SIZE,
  FHT_B2SUM = 0,
  FHT_G2 = 128,                // to_out.0.weight * sa2
  FHT_BE2 = 256,
  FHT_B3 = 384,
  FH_VEC_SIZE = FH_TAIL + 512,
  FH_HALF_MAT_FLOATS = 8192,
};
__host__ __device__ inline int fourier_pack_size(int n) {
  return fourier_pack_size_f32(n) + FH_VEC_SIZE + 2 * (2 * n + 1) * FH_HALF_MAT_FLOATS;
}

// value of header slot 14 of an attention pack whose slots 10..13 (LayerNorm bounds for k_layers_p) are filled (packing.py)
constexpr float AH_HDR_VERSION = 2.0f;

}  // namespace ig
