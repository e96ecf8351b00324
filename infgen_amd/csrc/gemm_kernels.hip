// Generic row-tile Linear (+LayerNorm/ReLU) kernel and the fused FourierEmbedding kernel.
// fp32 MFMA (v_mfma_f32_32x32x2_f32), 32-row tiles, 256 threads (4 waves x 32 output columns).
#include "kernels.h"

namespace ig {

// ------------------------------------------------------------------------------------------
// k_linear:  Y = epilogue( LNpre?(X[gather]) @ W^T + b )
// Used for everything off the per-step hot path (embedding tables, heads' first layers,
// map K/V, fusion MLP) and as the reference implementation of the tile helpers in tests.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void linear_body(const LinearArgs& a);

__global__ __launch_bounds__(NT) void k_linear(LinearArgs a) { linear_body(a); }

// several independent Linear launches in one (the heads of the insertion sub-loop all read the same few rows):
// blockIdx.z picks the descriptor
__global__ __launch_bounds__(NT) void k_linear_multi(LinearMultiArgs m) {
  const LinearArgs& a = m.d[blockIdx.z];
  if ((int)blockIdx.x * TR >= a.rows) return;
  linear_body(a);
}

__device__ __forceinline__ void linear_body(const LinearArgs& a) {
  __shared__ __attribute__((aligned(16))) float As[TR * LDT];
  __shared__ __attribute__((aligned(16))) float Os[TR * LDT];
  const int row0 = blockIdx.x * TR;
  const int nvalid = min(TR, a.rows - row0);
  if (nvalid <= 0) return;
  const int w = wave_id();
  const int kchunks = (a.K + 127) / 128;
  const int passes = (a.Np + 127) / 128;

  auto stage = [&](int kc) {
    for (int idx = threadIdx.x; idx < TR * 128; idx += NT) {
      const int r = idx >> 7, c = idx & 127;
      float v = 0.f;
      if (r < nvalid && kc + c < a.K) {
        long src = a.gather ? (long)a.gather[row0 + r] : (long)(row0 + r);
        if (src >= 0) v = a.X[src * a.ldx + kc + c];
      }
      As[r * LDT + c] = v;
    }
  };

  if (kchunks == 1) {
    stage(0);
    __syncthreads();
    if (a.pre_g) {
      ln_tile(As, LDT, As, LDT, a.pre_g, a.pre_b, false);
      __syncthreads();
    }
    // a wide output (token / grid heads) over few rows: the column passes are spread over gridDim.y
    for (int p = blockIdx.y; p < passes; p += gridDim.y) {
      const int n0 = p * 128 + 32 * w;
      f32x16 acc = zero16();
      if (n0 < a.Np) mfma_32x32_rt(acc, As, LDT, a.Kp, a.Wp, a.Np, n0);
      if (a.post_g) {   // N == 128, single pass
        acc_to_lds(acc, Os, LDT, n0, a.bias);
        __syncthreads();
        ln_tile(Os, LDT, Os, LDT, a.post_g, a.post_b, a.relu != 0);
        __syncthreads();
        unstage_rows_128(Os, [&](int r) { return a.Y + (size_t)(row0 + r) * a.ldy; }, nvalid);
      } else if (n0 < a.Np) {
        const int col = n0 + acc_col();
        if (col < a.N) {
          const float b = a.bias ? a.bias[col] : 0.f;
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int r = acc_row(reg);
            if (r < nvalid) {
              float y = acc[reg] + b;
              if (a.relu) y = fmaxf(y, 0.f);
              a.Y[(size_t)(row0 + r) * a.ldy + col] = y;
            }
          }
        }
      }
    }
    return;
  }
  // K > 128: single pass of <= 128 output columns, accumulate over K chunks
  if (blockIdx.y) return;
  const int n0 = 32 * w;
  f32x16 acc = zero16();
  for (int kc = 0; kc < a.K; kc += 128) {
    stage(kc);
    __syncthreads();
    const int kk = min(128, a.Kp - kc);
    if (n0 < a.Np) mfma_32x32_rt(acc, As, LDT, kk, a.Wp + (size_t)(kc >> 3) * a.Np * 8, a.Np, n0);
    __syncthreads();
  }
  if (a.post_g) {
    acc_to_lds(acc, Os, LDT, n0, a.bias);
    __syncthreads();
    ln_tile(Os, LDT, Os, LDT, a.post_g, a.post_b, a.relu != 0);
    __syncthreads();
    unstage_rows_128(Os, [&](int r) { return a.Y + (size_t)(row0 + r) * a.ldy; }, nvalid);
  } else if (n0 < a.Np) {
    const int col = n0 + acc_col();
    if (col < a.N) {
      const float b = a.bias ? a.bias[col] : 0.f;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = acc_row(reg);
        if (r < nvalid) {
          float y = acc[reg] + b;
          if (a.relu) y = fmaxf(y, 0.f);
          a.Y[(size_t)(row0 + r) * a.ldy + col] = y;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_fourier: FourierEmbedding.forward (reference infgen/modules/layers.py:142-160) over a flat
// list of E rows with n continuous inputs each (raw[e] is a float4, first n used).
//   per dim i:  z = x_i * freqs[i, :] * 2 * pi ; [cos z, sin z, x_i] -> Linear(129,128) -> LN -> ReLU -> Linear(128,128)
//   sum over i (+ categorical embedding row) -> LN -> ReLU -> Linear(128,128)
// `normalize` additionally applies the affine-free LayerNorm shared by the six layers'
// attn_prenorm_r (each layer's gamma/beta are folded into its to_k_r/to_v_r operands).
// The K = 129 first Linear runs as a K = 128 MFMA GEMM plus a rank-1 update for the x column.
// Persistent grid-stride over 32-row tiles; the row count may live on the device.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT, 2) void k_fourier(FourierArgs a) {
  __shared__ __attribute__((aligned(16))) float Fs[TR * LDT];
  __shared__ __attribute__((aligned(16))) float Hs[TR * LDT];
  __shared__ float xs[TR];
  const int E = a.count_dev ? min(*a.count_dev, a.e_cap) : a.e_cap;
  const int ntiles = (E + TR - 1) / TR;
  const int w = wave_id();
  const int n0 = 32 * w;
  if (a.prof_rows && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.prof_rows + a.n, (unsigned long long)E);
  const float* tail = a.pack + FE_DIM0 + a.n * FD_SIZE;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int e0 = tile * TR;
    const int nvalid = min(TR, E - e0);
    f32x16 acc2 = zero16();
    for (int i = 0; i < a.n; ++i) {
      const float* blk = a.pack + FE_DIM0 + i * FD_SIZE;
      const float* freq = a.pack + FE_FREQ + i * 64;
      __syncthreads();   // previous users of Fs/Hs/xs are done
      for (int idx = threadIdx.x; idx < TR * 64; idx += NT) {
        const int r = idx >> 6, f = idx & 63;
        float x = 0.f;
        if (r < nvalid) x = a.raw[(size_t)(e0 + r) * 4 + i];
        // reference: x.unsqueeze(-1) * freqs.weight * 2 * math.pi  (left to right, fp32)
        const float z = x * freq[f] * 2.0f * PI_F;
        float sn, cs;
        sincosf(z, &sn, &cs);
        Fs[r * LDT + f] = cs;
        Fs[r * LDT + 64 + f] = sn;
        if (f == 0) xs[r] = x;
      }
      __syncthreads();
      f32x16 acc1 = zero16();
      mfma_32x32<128>(acc1, Fs, LDT, blk + FD_W1, 128, n0);
      {
        const int col = n0 + acc_col();
        const float wx = blk[FD_W1X + col], b1 = blk[FD_B1 + col];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int r = acc_row(reg);
          Hs[r * LDT + col] = acc1[reg] + xs[r] * wx + b1;
        }
      }
      __syncthreads();
      ln_tile(Hs, LDT, Hs, LDT, blk + FD_LN_G, blk + FD_LN_B, true);
      __syncthreads();
      mfma_32x32<128>(acc2, Hs, LDT, blk + FD_W2, 128, n0);
    }
    __syncthreads();
    {
      const int col = n0 + acc_col();
      const float b2 = tail[FT_B2SUM + col];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = acc_row(reg);
        float v = acc2[reg] + b2;
        if (a.cat && r < nvalid) v += a.cat[(size_t)(e0 + r) * a.ldcat + col];
        Hs[r * LDT + col] = v;
      }
    }
    __syncthreads();
    ln_tile(Hs, LDT, Hs, LDT, tail + FT_LN_G, tail + FT_LN_B, true);
    __syncthreads();
    f32x16 acc3 = zero16();
    mfma_32x32<128>(acc3, Hs, LDT, tail + FT_W3, 128, n0);
    acc_to_lds(acc3, Fs, LDT, n0, tail + FT_B3);
    __syncthreads();
    if (a.normalize) {
      ln_tile(Fs, LDT, Fs, LDT, nullptr, nullptr, false);
      __syncthreads();
    }
    unstage_rows_128(Fs, [&](int r) { return a.out + (size_t)(e0 + r) * a.ldo; }, nvalid);
  }
}


// ------------------------------------------------------------------------------------------
// k_layernorm: Y = LN(X) over 128 columns (gamma == nullptr: affine-free).  torch.nn.LayerNorm.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_layernorm(const float* X, int rows, const float* g, const float* b, float* Y) {
  __shared__ __attribute__((aligned(16))) float Xs[TR * LDT];
  const int row0 = blockIdx.x * TR;
  const int nvalid = min(TR, rows - row0);
  if (nvalid <= 0) return;
  stage_rows_128(Xs, [&](int r) { return X + (size_t)(row0 + r) * D; }, nvalid);
  __syncthreads();
  ln_tile(Xs, LDT, Xs, LDT, g, b, false);
  __syncthreads();
  unstage_rows_128(Xs, [&](int r) { return Y + (size_t)(row0 + r) * D; }, nvalid);
}

// ------------------------------------------------------------------------------------------
// k_stream_read<W>: calibration of the rocprofv3 FETCH_SIZE counter (tools/calibrate_fetch.sh): a plain streaming read of
// `n` bytes with W bytes per lane (8: the edge kernel's row pieces, 16: the node kernels' float4 rows), one partial sum per
// workgroup written out so that the loads stay.  The guide documents the factor only for 16 B per lane.
// ------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(256) void k_stream_read(const float* __restrict__ p, size_t n_floats, float* out) {
  constexpr int F = W / 4;
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * 256 * F;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * F; i + F <= n_floats; i += stride) {
    typedef float v2 __attribute__((ext_vector_type(2)));
    if constexpr (F == 2) { const v2 v = __builtin_nontemporal_load(reinterpret_cast<const v2*>(p + i)); acc += v[0] + v[1]; }
    else { const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + i)); acc += (v[0] + v[1]) + (v[2] + v[3]); }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) atomicAdd(out + blockIdx.x, acc);
}
template __global__ void k_stream_read<8>(const float*, size_t, float*);
template __global__ void k_stream_read<16>(const float*, size_t, float*);

}  // namespace ig
