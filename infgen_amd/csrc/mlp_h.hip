// Small MLP chains on the fp16 matrix pipe (three-term split, split.cuh), register resident per 16-row wave like
// k_attn_h: 4 waves per workgroup (64-row tiles), three quarter buffers, two workgroups per CU.
//   k_mlpemb_h  MLPEmbedding (reference infgen/modules/layers.py:163-179) with a K0 = 128 j input:
//               Linear(K0,128) LN ReLU Linear(128,128) LN ReLU Linear(128,128)   - the fusion embedding of the raw
//               per-column feature (agent_decoder.py:2265-2287), three k_linear launches before
//   k_heads_h   token_predict_head / state_predict_head + greedy arg-max (agent_decoder.py:2161-2167), k_heads before
#include "kernels.h"
#include "layout.h"
#include "tile.cuh"
#include "split.cuh"

namespace ig {

__device__ __forceinline__ void mh_load_row(f32x4 (&v)[8], const float* row, int rg) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row) x = *reinterpret_cast<const float4*>(row + 16 * t + 4 * rg);
    v[t] = f32x4{x.x, x.y, x.z, x.w};
  }
}
__device__ __forceinline__ void mh_zero(f32x4 (&v)[8]) {
#pragma unroll
  for (int t = 0; t < 8; ++t) v[t] = f32x4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void mh_scale_bias(f32x4 (&v)[8], float s, const float* bias, int rg) {
  const f32x4 s4 = splat4(s);
#pragma unroll
  for (int t = 0; t < 8; ++t) v[t] = fma4(v[t], s4, lds4(bias + 16 * t + 4 * rg));
}

constexpr int MH_NT = 256, MH_TILE = 64, MH_RING = 3;

template <int TERMS>
__global__ __launch_bounds__(MH_NT, 2) void k_mlpemb_h(MlpEmbHArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned short Wb[MH_RING][QUARTER];
  __shared__ __attribute__((aligned(16))) float Vt[16 + 7 * 128];      // hdr | b0 g0 be0 | b1 g1 be1 | b2
  __shared__ const unsigned short* seg_ptr[1];
  __shared__ int seg_n[1];
  const int ntiles = (a.rows + MH_TILE - 1) / MH_TILE;
  if ((int)blockIdx.x >= ntiles) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, rg = lane >> 4;
  const int nchunk = a.K0 / 128;
  const float* P = a.pack;
  const int o2 = a.K0 * 128 + 3 * 128, o3 = o2 + 16384 + 3 * 128, oh = o3 + 16384 + 128;   // fp32 stages, then the split section
  if (tid == 0) { seg_ptr[0] = reinterpret_cast<const unsigned short*>(P + oh + 16); seg_n[0] = 4 * (nchunk + 2); }
  if (tid < 16) Vt[tid] = P[oh + tid];
  for (int i = tid; i < 384; i += MH_NT) {
    Vt[16 + i] = P[a.K0 * 128 + i];
    Vt[16 + 384 + i] = P[o2 + 16384 + i];
  }
  for (int i = tid; i < 128; i += MH_NT) Vt[16 + 768 + i] = P[o3 + 16384 + i];
  __syncthreads();
  QuarterStream<MH_NT, MH_RING> qs;
  qs.init(seg_ptr, seg_n, 1, (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x, Wb, tid);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row = tile * MH_TILE + w * 16 + j;
    const bool valid = row < a.rows;
    u32x4 Bh[4], Bl[4];
    f32x4 h[8];
    mh_zero(h);
    for (int c = 0; c < nchunk; ++c) {
      f32x4 xc[8];
      mh_load_row(xc, valid ? a.X + (size_t)row * a.ldx + 128 * c : nullptr, rg);
      const float inv = frags_scaled(xc, Bh, Bl) * Vt[0];
      f32x4 part[8];
      mh_zero(part);
#pragma unroll
      for (int s = 0; s < 4; ++s) gemm_quarter<TERMS>(part, qs.take(), Bh[s], Bl[s], lane);
#pragma unroll
      for (int t = 0; t < 8; ++t) h[t] = fma4(part[t], splat4(inv), h[t]);
    }
    mh_scale_bias(h, 1.0f, Vt + 16, rg);
    ln_regs<true, true>(h, Vt + 16 + 128, Vt + 16 + 256, rg);
    {
      const float inv = frags_scaled(h, Bh, Bl) * Vt[1];
      mh_zero(h);
#pragma unroll
      for (int s = 0; s < 4; ++s) gemm_quarter<TERMS>(h, qs.take(), Bh[s], Bl[s], lane);
      mh_scale_bias(h, inv, Vt + 16 + 384, rg);
      ln_regs<true, true>(h, Vt + 16 + 512, Vt + 16 + 640, rg);
    }
    {
      const float inv = frags_scaled(h, Bh, Bl) * Vt[2];
      mh_zero(h);
#pragma unroll
      for (int s = 0; s < 4; ++s) gemm_quarter<TERMS>(h, qs.take(), Bh[s], Bl[s], lane);
      mh_scale_bias(h, inv, Vt + 16 + 768, rg);
    }
    if (valid) {
      float* o = a.Y + (size_t)row * a.ldy;
#pragma unroll
      for (int t = 0; t < 8; ++t)
        *reinterpret_cast<float4*>(o + 16 * t + 4 * rg) = make_float4(h[t][0], h[t][1], h[t][2], h[t][3]);
    }
  }
}

template <int TERMS>
__global__ __launch_bounds__(MH_NT, 2) void k_heads_h(HeadsArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned short Wb[MH_RING][QUARTER];
  // token head: hdr | b0 g0 be0 ; state head: hdr | b0 g0 be0 | W3 [3][128] | b3 [3]
  __shared__ __attribute__((aligned(16))) float Vt[16 + 384 + 16 + 384 + 384 + 16];
  __shared__ const unsigned short* seg_ptr[3];
  __shared__ int seg_n[3];
  const int ntiles = (a.rows + MH_TILE - 1) / MH_TILE;
  if ((int)blockIdx.x >= ntiles) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, rg = lane >> 4;
  const int nchunk = a.token_size / 128;
  const float* PT = a.tok_pack;
  const float* PS = a.st_pack;
  const int W3T = 16768;                                            // P(128, token_size), then b3
  const int oht = W3T + 128 * a.token_size + a.token_size;          // split section of the token head pack
  const int ohs = 16768 + 3 * 128 + 3;                              // ... of the state head pack (W3 [3][128] row-major, b3 [3])
  const int ohs_al = (ohs + 3) & ~3;
  float* VS = Vt + 16 + 384;
  if (tid == 0) {
    const unsigned short* qt = reinterpret_cast<const unsigned short*>(PT + oht + 16);
    seg_ptr[0] = qt;                                                              seg_n[0] = 4;           // token W0
    seg_ptr[1] = reinterpret_cast<const unsigned short*>(PS + ohs_al + 16);       seg_n[1] = 4;           // state W0
    seg_ptr[2] = qt + 4 * QUARTER;                                                seg_n[2] = 4 * nchunk;  // token W3 chunks
  }
  if (tid < 16) { Vt[tid] = PT[oht + tid]; VS[tid] = PS[ohs_al + tid]; }
  for (int i = tid; i < 384; i += MH_NT) {
    Vt[16 + i] = PT[16384 + i];
    VS[16 + i] = PS[16384 + i];
    VS[16 + 384 + i] = PS[16768 + i];
  }
  if (tid < 3) VS[16 + 768 + tid] = PS[16768 + 384 + tid];
  __syncthreads();
  QuarterStream<MH_NT, MH_RING> qs;
  qs.init(seg_ptr, seg_n, 3, (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x, Wb, tid);
  const float* b3 = PT + W3T + (size_t)128 * a.token_size;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row = tile * MH_TILE + w * 16 + j;
    const bool valid = row < a.rows;
    u32x4 Bh[4], Bl[4];
    f32x4 x[8], ht[8], hs[8];
    mh_load_row(x, valid ? a.X + (size_t)row * D : nullptr, rg);
    const float inv_x = frags_scaled(x, Bh, Bl);
    mh_zero(ht); mh_zero(hs);
#pragma unroll
    for (int s = 0; s < 4; ++s) gemm_quarter<TERMS>(ht, qs.take(), Bh[s], Bl[s], lane);
#pragma unroll
    for (int s = 0; s < 4; ++s) gemm_quarter<TERMS>(hs, qs.take(), Bh[s], Bl[s], lane);
    mh_scale_bias(ht, inv_x * Vt[0], Vt + 16, rg);
    ln_regs<true, true>(ht, Vt + 16 + 128, Vt + 16 + 256, rg);
    mh_scale_bias(hs, inv_x * VS[0], VS + 16, rg);
    ln_regs<true, true>(hs, VS + 16 + 128, VS + 16 + 256, rg);
    // state head: three outputs, plain dot products over the lane's 32 features, then the four lanes of the row
    {
      float best = -INFINITY;
      int bi = 0;
#pragma unroll
      for (int o = 0; o < 3; ++o) {
        f32x4 acc = splat4(0.f);
#pragma unroll
        for (int t = 0; t < 8; ++t) acc = fma4(hs[t], lds4(VS + 16 + 384 + o * 128 + 16 * t + 4 * rg), acc);
        const float sdot = xor_lanes((acc[0] + acc[1]) + (acc[2] + acc[3])) + VS[16 + 768 + o];
        if (sdot > best) { best = sdot; bi = o; }
      }
      if (valid && rg == 0) a.next_state[row] = bi;
    }
    // token head: logits in 128-wide chunks, running arg-max (first maximum)
    const float inv_h = frags_scaled(ht, Bh, Bl) * Vt[1];
    float bv = -INFINITY;
    int bidx = 0x7fffffff;
    for (int c = 0; c < nchunk; ++c) {
      f32x4 lg[8];
      mh_zero(lg);
#pragma unroll
      for (int s = 0; s < 4; ++s) gemm_quarter<TERMS>(lg, qs.take(), Bh[s], Bl[s], lane);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int col = 128 * c + 16 * t + 4 * rg;
        const float4 bb = *reinterpret_cast<const float4*>(b3 + col);
        const float v0 = lg[t][0] * inv_h + bb.x, v1 = lg[t][1] * inv_h + bb.y;
        const float v2 = lg[t][2] * inv_h + bb.z, v3 = lg[t][3] * inv_h + bb.w;
        if (a.logits && valid)
          *reinterpret_cast<float4*>(a.logits + (size_t)row * a.token_size + col) = make_float4(v0, v1, v2, v3);
        if (v0 > bv) { bv = v0; bidx = col; }
        if (v1 > bv) { bv = v1; bidx = col + 1; }
        if (v2 > bv) { bv = v2; bidx = col + 2; }
        if (v3 > bv) { bv = v3; bidx = col + 3; }
      }
    }
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
      const float ov = __shfl_xor(bv, off, 64);
      const int oi = __shfl_xor(bidx, off, 64);
      if (ov > bv || (ov == bv && oi < bidx)) { bv = ov; bidx = oi; }
    }
    if (valid && rg == 0) a.next_token[row] = bidx;
  }
}

#if !IG_BF16_OPERANDS
template __global__ void k_mlpemb_h<3>(MlpEmbHArgs);
#endif
template __global__ void k_mlpemb_h<1>(MlpEmbHArgs);
#if !IG_BF16_OPERANDS
template __global__ void k_heads_h<3>(HeadsArgs);
#endif
template __global__ void k_heads_h<1>(HeadsArgs);

}  // namespace ig
