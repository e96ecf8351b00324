// Shared pieces of the fp16 three-term-split kernels (fourier_h.hip, attn_h.hip): operand splitting, the
// quarter-matrix LDS ring, the 16x16x32 MFMA k-step, LayerNorm and fragment conversion in the
// "transposed" register layout (weights = MFMA A operand, 16 rows of a wave = B operand / C columns;
// register r of C tile t in lane (j = lane & 15, rg = lane >> 4) is feature 16 t + 4 rg + r of row j).
#pragma once
#include "tile.cuh"

namespace ig {

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int QUARTER = 8192;            // fp16 elements of one quarter-matrix in LDS (16 KB)
constexpr int RING = 5;                  // quarter buffers in LDS (80 KB: at most one workgroup per CU)
constexpr int DIST = RING - 1;           // quarters in flight ahead of the one being consumed

// (Round 6 tried the remainders through v_fma_mixlo_f16 / v_fma_mixhi_f16 - convert, subtract and round in one instruction each, three
// per pair instead of five, bit-identical (tools/split_mix_probe.hip): k_fourier_h 0 - 2 % faster, i.e. its vector phases are not
// bound by instruction count - and the two half-register writes in front of an MFMA need more wait states than hipcc pads for an
// asm block, depending on their order (tools/split_mix_mfma_probe.hip: k_attn_hs came out wrong at the fp16 level).  Not adopted.)
// (a, b) -> packed fp16 pairs hi, lo with a = hi_a + lo_a (+ <= 2^-23 |a|): hi = a rounded to nearest even at 11 significand bits
// (v_cvt_pk_f16_f32), lo = the remainder a - hi (exact in fp32, |a - hi| <= 2^-11 |a|) rounded to nearest even again.  Same five
// instructions as a truncating split (round-toward-zero conversions: <= 2^-21 |a|), four times the accuracy: with both operands of a
// product split this way, hi hi + hi lo + lo hi misses the exact product by the dropped lo lo term (<= 2^-22) plus the two
// representation errors (<= 2^-23 each).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// IG_BF16_OPERANDS = 1 (the *_b16.hip translation units: the kernels behind gemm_terms = 2, BASELINE config C5's "bf16"): an operand is
// rounded to bf16 precision (8 significant bits, round to nearest even: v_cvt_pk_bf16_f32) before it enters the f16 pipe.  Every
// operand is pre-scaled by a power of two into the fp16 range, so the bf16-rounded value is exactly representable in fp16: the
// products are those of a bf16 MFMA, accumulated in fp32 - bf16 arithmetic on the f16 pipe, same rate.  hi only (TERMS = 1).
#ifndef IG_BF16_OPERANDS
#define IG_BF16_OPERANDS 0
#endif
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
#if IG_BF16_OPERANDS
  const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
  const f32x2 r = {__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
  lo = 0u;
#else
  const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
  const f32x2 r = f32x2{a, b} - __builtin_convertvector(h, f32x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
#endif
}

// sum over lanes ^ 16, ^ 32 (the four lanes that hold one row): the gfx950 row / half swaps with both operands equal leave
// (own, partner) in the two results - one VALU instruction per step instead of a ds_bpermute round trip through the LDS queue
// (which the partner wave's fragment reads keep busy); the sums are the same two-operand additions, bitwise
typedef unsigned u32x2_sw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float xor_lanes(float v) {
  u32x2_sw a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
__device__ __forceinline__ float xor_lanes_max(float v) {
  u32x2_sw a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}

// LDS-DMA by hand.  Through the builtin (__builtin_amdgcn_global_load_lds) hipcc treats every such load as a store to LDS that
// any later LDS access may alias: it puts s_waitcnt vmcnt(0) in front of the next ds_read and in front of every __syncthreads()
// (whose release fence waits for "LDS stores"), i.e. each quarter's pieces were waited for right after their issue and the ring
// of quarters in flight never was one - the 1.2 - 1.3 us per-quarter cadence of rounds 1 - 2.  As inline assembly the loads are
// invisible to that bookkeeping; the streams order them themselves (counted s_waitcnt vmcnt + wg_barrier below, as they always
// did), and loads hipcc issues on its own only make its own counted waits longer, never shorter (vmcnt retires in order).
// M0 = LDS byte address of lane 0's 16 bytes (wave-uniform); one wait state between the M0 write and the LDS-DMA instruction.
#ifndef IG_DMA_BUILTIN
#define IG_DMA_BUILTIN 0
#endif
#ifndef IG_LN_RSQ
#define IG_LN_RSQ 1
#endif
#ifndef IG_GQ_INTERLEAVE
#define IG_GQ_INTERLEAVE 0
#endif
#ifndef IG_GU_NOREAD      // timing experiment (wrong results): gemm_unit reads its A fragments from LDS once per GEMM instead of eight times
#define IG_GU_NOREAD 0
#endif
__device__ __forceinline__ void lds_dma16(const void* gptr, unsigned lds_byte_addr_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :: "s"(lds_byte_addr_uniform), "v"(gptr) : "memory");     // (M0 is reserved - not a legal clobber; nothing else in these kernels uses it: tests/test_boundary_cpu.py scans for that)
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}
// workgroup barrier without __syncthreads()' fence (which would wait for every outstanding LDS-DMA piece): LDS operations of
// this wave retired (lgkmcnt), then s_barrier; callers wait for the LDS-DMA pieces they need with a counted vmcnt before it
__device__ __forceinline__ void wg_barrier() {
#if IG_DMA_BUILTIN
  __syncthreads();
#else
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#endif
}
template <int NTH>
__device__ __forceinline__ void stage_quarter(const unsigned short* __restrict__ gsrc, unsigned short* ldst, int tid) {
  // 16 KB = 1024 sixteen-byte units; a wave instruction lands 1 KB at (uniform base + lane * 16)
#if IG_DMA_BUILTIN
#pragma unroll
  for (int c = 0; c < 1024 / NTH; ++c) {
    const int unit = c * NTH + tid;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + unit * 8),
                                     (__attribute__((address_space(3))) void*)(ldst + (c * NTH + (tid & ~63)) * 8),
                                     16, 0, 0);
  }
#else
  const unsigned lbase = __builtin_amdgcn_readfirstlane(lds_addr(ldst) + (unsigned)(tid & ~63) * 16u);
#pragma unroll
  for (int c = 0; c < 1024 / NTH; ++c) lds_dma16(gsrc + (c * NTH + tid) * 8, lbase + (unsigned)(c * NTH) * 16u);
#endif
}

// acc[t] += W[16 t .., this k-step] * B   (three MFMAs per feature tile; the two A fragments of tile t + 1 are
// read from LDS while the MFMAs of tile t run).  TERMS = 1 keeps only the hi x hi product: plain fp16 arithmetic
// (11 significant bits per operand, truncated) at a third of the matrix work and half of the LDS reads - the reduced-
// precision mode behind infgen_set_gemm_terms(1), never the default.
template <int TERMS = 3>
__device__ __forceinline__ void gemm_quarter(f32x4 (&acc)[8], const unsigned short* Wl, u32x4 Bh, u32x4 Bl, int lane) {
  const v8h bh = __builtin_bit_cast(v8h, Bh);
  const v8h bl = __builtin_bit_cast(v8h, Bl);
  const unsigned short* p = Wl + lane * 8;
  // Four feature tiles at a time: all their A fragments are requested from LDS first, then the products are issued term
  // by term ACROSS the tiles, so that consecutive MFMAs never share an accumulator (a dependent MFMA waits ~40 cycles, and
  // any instruction between two MFMAs on one accumulator costs another ~43).  The scheduling barriers keep hipcc from
  // sinking each tile's ds_reads back in front of its own MFMAs, which serialises LDS latency and MFMA latency per tile.
  // Round 3: the second group's fragments are requested in the shadow of the first group's products (one read after every
  // product) instead of in a block between the groups, where their issue left the matrix pipe idle (gemm_unit below).
#if IG_GQ_INTERLEAVE
  v8h ah[2][4], al[2][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    ah[0][t] = *reinterpret_cast<const v8h*>(p + t * 1024);
    if constexpr (TERMS == 3) al[0][t] = *reinterpret_cast<const v8h*>(p + t * 1024 + 512);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    ah[1][t] = *reinterpret_cast<const v8h*>(p + (4 + t) * 1024);
    if constexpr (TERMS == 3) al[1][t] = *reinterpret_cast<const v8h*>(p + (4 + t) * 1024 + 512);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0][t], bh, acc[t], 0, 0, 0);
  if constexpr (TERMS == 3) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0][t], bl, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[0][t], bh, acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < (TERMS == 3 ? 8 : 4); ++r) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  }
  if constexpr (TERMS == 3) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[4 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1][t], bh, acc[4 + t], 0, 0, 0);
  if constexpr (TERMS == 3) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[4 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1][t], bl, acc[4 + t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[4 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[1][t], bh, acc[4 + t], 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
#elif defined(IG_GQ_NOMFMA)      // timing experiment (wrong results): no fragment reads, no products
  asm volatile("" : "+v"(acc[0]), "+v"(acc[7]));
#else
#pragma unroll
  for (int g = 0; g < 8; g += 4) {
    v8h ah[4], al[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#ifdef IG_GQ_NOREAD            // timing experiment (wrong results): the products without their A-fragment reads from LDS
      ah[t] = bh; al[t] = bl;
#else
      ah[t] = *reinterpret_cast<const v8h*>(p + (g + t) * 1024);
      if constexpr (TERMS == 3) al[t] = *reinterpret_cast<const v8h*>(p + (g + t) * 1024 + 512);
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[g + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh, acc[g + t], 0, 0, 0);
    if constexpr (TERMS == 3) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[g + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl, acc[g + t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[g + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh, acc[g + t], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#endif
}

// One whole 128 x 128 GEMM (four quarters resident in LDS, no barrier inside): acc[t] += W[16 t .., :] * B.  Eight groups of four
// feature tiles; the A fragments of the next group are requested while the products of the current one are issued (two fragment
// sets in registers).  Issue order inside a group: one fragment read of the NEXT group in the shadow of every product - an MFMA
// occupies the matrix pipe for 16 cycles and blocks only its own issue slot; eight reads in a row ahead of the products left the
// pipe idle while they were issued (2,250 instead of ~1,600 cycles per GEMM, s_memtime).  Products term by term across the four
// tiles as in gemm_quarter: the same products in the same order per accumulator as four gemm_quarter calls, bitwise equal.
template <int TERMS>
struct GemmUnit {
  v8h ah[2][4], al[2][4];
  const unsigned short* p;
  __device__ __forceinline__ void request(int grp, int buf) {       // group grp = quarter grp >> 1, feature tiles 4 (grp & 1) ..
    const unsigned short* q = p + (grp >> 1) * QUARTER + (grp & 1) * 4096;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      ah[buf][t] = *reinterpret_cast<const v8h*>(q + t * 1024);
      if constexpr (TERMS == 3) al[buf][t] = *reinterpret_cast<const v8h*>(q + t * 1024 + 512);
    }
  }
  template <bool LAST>
  __device__ __forceinline__ void group(f32x4 (&acc)[8], int grp, u32x4 Bh, u32x4 Bl) {
    const int buf = grp & 1, o = 4 * (grp & 1);
    if (!LAST && !IG_GU_NOREAD) request(grp + 1, buf ^ 1);
    const v8h bh = __builtin_bit_cast(v8h, Bh);
    const v8h bl = __builtin_bit_cast(v8h, Bl);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[o + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[buf][t], bh, acc[o + t], 0, 0, 0);
    if constexpr (TERMS == 3) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[o + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[buf][t], bl, acc[o + t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[o + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[buf][t], bh, acc[o + t], 0, 0, 0);
    }
    if constexpr (!LAST) {
#pragma unroll
      for (int r = 0; r < (TERMS == 3 ? 8 : 4); ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      if constexpr (TERMS == 3) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
};
template <int TERMS = 3>
__device__ __forceinline__ void gemm_unit(f32x4 (&acc)[8], const unsigned short* Wu, const u32x4 (&Bh)[4], const u32x4 (&Bl)[4], int lane) {
  GemmUnit<TERMS> u;
  u.p = Wu + lane * 8;
  u.request(0, 0);
  if (IG_GU_NOREAD) u.request(1, 1);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int grp = 0; grp < 7; ++grp) u.template group<false>(acc, grp, Bh[grp >> 1], Bl[grp >> 1]);
  u.template group<true>(acc, 7, Bh[3], Bl[3]);
}

// The same GEMM with ONE set of fragment registers (32 instead of 64; three waves per SIMD have 168 registers each): a tile's hi
// fragment is dead after its second product and its lo fragment after the third, so the next group's fragments are requested into
// the registers the running group has just finished with - hi(t) behind the hi x lo product of tile t, lo(t) behind lo x hi -
// seven to eight products (~120 cycles) ahead of their first use.  Same products in the same order per accumulator.
template <int TERMS = 3>
__device__ __forceinline__ void gemm_unit_sb(f32x4 (&acc)[8], const unsigned short* Wu, const u32x4 (&Bh)[4], const u32x4 (&Bl)[4], int lane) {
  static_assert(TERMS == 3, "the single-buffered schedule is written for the three-term split");
  const unsigned short* p = Wu + lane * 8;
  v8h ah[4], al[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    ah[t] = *reinterpret_cast<const v8h*>(p + t * 1024);
    al[t] = *reinterpret_cast<const v8h*>(p + t * 1024 + 512);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int grp = 0; grp < 8; ++grp) {
    const int o = 4 * (grp & 1);
    const unsigned short* q = p + ((grp + 1) >> 1) * QUARTER + ((grp + 1) & 1) * 4096;      // the next group's fragments
    const v8h bh = __builtin_bit_cast(v8h, Bh[grp >> 1]);
    const v8h bl = __builtin_bit_cast(v8h, Bl[grp >> 1]);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[o + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh, acc[o + t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[o + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl, acc[o + t], 0, 0, 0);
      if (grp < 7) ah[t] = *reinterpret_cast<const v8h*>(q + t * 1024);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[o + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh, acc[o + t], 0, 0, 0);
      if (grp < 7) al[t] = *reinterpret_cast<const v8h*>(q + t * 1024 + 512);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    if (grp < 7) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    } else {
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// LayerNorm over the 128 features of this lane's edge (32 here, the rest in lanes ^ 16, ^ 32, ^ 48), biased
// variance, eps 1e-5; register r of tile t is feature 16 t + 4 rg + r.
// Packed-math helpers: whole-f32x4 expressions lower to v_pk_* (two floats per instruction); fused multiply-adds
// are written explicitly because the library is built with -ffp-contract=off.
__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
// (a per-lane broadcast is materialised as a real register pair: hipcc would fold f32x4{x, x, x, x} into an op_sel modifier on one
// 32-bit VGPR, the instruction form that misbehaves next to other waves' MFMAs - edge_attn.cuh: bc_v)
__device__ __forceinline__ f32x4 splat4(float x) {
  if (__builtin_constant_p(x)) return f32x4{x, x, x, x};
  typedef float pk2_t __attribute__((ext_vector_type(2)));
  pk2_t p = {x, x};
  asm("" : "+v"(p));
  return f32x4{p[0], p[1], p[0], p[1]};
}
__device__ __forceinline__ f32x4 lds4(const float* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  return f32x4{v.x, v.y, v.z, v.w};
}

// 1 / sqrt(var + eps): v_rsq_f32 (1 ulp) + one Newton step instead of the correctly rounded square root and division (~25 dependent
// instructions of a chain that a lone wave pays in full: s_memtime traces of k_fourier_h); var + eps >= 1e-5, far from the denormal
// range; the result is within an ulp of the correctly rounded one (IG_LN_RSQ=0: sqrtf and a division; round 4, with the vector
// phases of k_fourier_h shortened: +0.7 % per 1024-scene rollout, it was +0.3 % in round 3 and off)
__device__ __forceinline__ float ln_rstd(float var_eps) {
#if IG_LN_RSQ
  const float y = __builtin_amdgcn_rsqf(var_eps);
  const float e = __builtin_fmaf(-(var_eps * y), y, 1.0f);
  return __builtin_fmaf(0.5f * y, e, y);
#else
  return 1.0f / sqrtf(var_eps);
#endif
}
// (two halves so that a kernel can put a slot barrier between them: ln_stats centres v and returns 1 / sqrt(var + eps), ln_apply
// scales and applies the affine part / ReLU; ln_regs = both, the same operations in the same order)
__device__ __forceinline__ float ln_stats(f32x4 (&v)[8]) {
  f32x4 s4 = (v[0] + v[1]) + (v[2] + v[3]);
  s4 += (v[4] + v[5]) + (v[6] + v[7]);
  const float mean = xor_lanes((s4[0] + s4[1]) + (s4[2] + s4[3])) * (1.0f / 128.0f);
  const f32x4 m4 = splat4(mean);
  f32x4 q4 = splat4(0.f), q5 = splat4(0.f);
#pragma unroll
  for (int t = 0; t < 8; t += 2) {
    v[t] -= m4; v[t + 1] -= m4;
    q4 = fma4(v[t], v[t], q4);
    q5 = fma4(v[t + 1], v[t + 1], q5);
  }
  q4 += q5;
  const float var_eps = xor_lanes((q4[0] + q4[1]) + (q4[2] + q4[3])) * (1.0f / 128.0f) + LN_EPS;
  return ln_rstd(var_eps);
}
template <bool AFFINE, bool RELU>
__device__ __forceinline__ void ln_apply(f32x4 (&v)[8], float rstd, const float* gtab, const float* btab, int rg) {
  const f32x4 r4 = splat4(rstd);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    f32x4 y = v[t] * r4;
    if (AFFINE) y = fma4(y, lds4(gtab + 16 * t + 4 * rg), lds4(btab + 16 * t + 4 * rg));
    if (RELU) y = __builtin_elementwise_max(y, splat4(0.f));
    v[t] = y;
  }
}
// The affine tables of a LayerNorm in registers.  ln_apply reads gamma / beta of a feature tile right where it uses them, and hipcc
// keeps one or two of those sixteen LDS reads in flight (`s_waitcnt lgkmcnt(1)` in front of every pair of products): sixteen exposed
// LDS round trips per LayerNorm, each queued behind the fragment reads of the SIMD's other wave - the LayerNorm phases of k_fourier_h
// took ~4,000 cycles for ~300 instructions (s_memtime slot traces, round 4).  ln_fetch requests all sixteen in one block (gamma and
// beta of a tile next to each other: the first product waits for two reads, not nine); callers put it in front of the statistics'
// dependency chain, fenced with a scheduling barrier, so the reads land under it.  Same values, same operations: bitwise equal.
#ifndef IG_LN_PREFETCH
#define IG_LN_PREFETCH 1
#endif
struct LnTab { f32x4 g[8], b[8]; };
__device__ __forceinline__ void ln_fetch(LnTab& tb, const float* gtab, const float* btab, int rg) {
#pragma unroll
  for (int t = 0; t < 8; ++t) { tb.g[t] = lds4(gtab + 16 * t + 4 * rg); tb.b[t] = lds4(btab + 16 * t + 4 * rg); }
}
template <bool RELU>
__device__ __forceinline__ void ln_apply_tab(f32x4 (&v)[8], float rstd, const LnTab& tb) {
  const f32x4 r4 = splat4(rstd);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    f32x4 y = fma4(v[t] * r4, tb.g[t], tb.b[t]);
    if (RELU) y = __builtin_elementwise_max(y, splat4(0.f));
    v[t] = y;
  }
}
template <bool AFFINE, bool RELU>
__device__ __forceinline__ void ln_regs(f32x4 (&v)[8], const float* gtab, const float* btab, int rg) {
  if constexpr (AFFINE && IG_LN_PREFETCH) {
    LnTab tb;
    ln_fetch(tb, gtab, btab, rg);
    __builtin_amdgcn_sched_barrier(0);
    const float rstd = ln_stats(v);
    ln_apply_tab<RELU>(v, rstd, tb);
  } else {
    const float rstd = ln_stats(v);
    ln_apply<AFFINE, RELU>(v, rstd, gtab, btab, rg);
  }
}

// C registers -> B fragments of the next GEMM: k-step s takes tiles 2 s (slots 0..3) and 2 s + 1 (slots 4..7)
__device__ __forceinline__ void regs_to_frags(const f32x4 (&v)[8], u32x4 (&Bh)[4], u32x4 (&Bl)[4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      unsigned hi, lo;
      split_pair(v[2 * s + (w >> 1)][2 * (w & 1)], v[2 * s + (w >> 1)][2 * (w & 1) + 1], hi, lo);
      Bh[s][w] = hi;
      Bl[s][w] = lo;
    }
}

// The weight stream of a workgroup: up to four runs of consecutive quarter-matrices (tables in LDS, filled by thread
// 0 before init), cycled once per tile through a ring of NR quarter buffers.  take() returns the next quarter after
// waiting for its LDS-DMA (counted vmcnt: "all but the (NR - 2) * GLDS most recent VMEM operations" always covers it;
// hipcc does not order LDS-DMA against ds_read on its own), a barrier, and a refill of the slot just released.
template <int NTH, int NR>
struct QuarterStream {
  static constexpr int GLDS = 1024 / NTH, ND = NR - 1;
  const unsigned short* const* seg_ptr;
  const int* seg_n;
  unsigned short (*Wb)[QUARTER];
  int nseg, total, consumed, slot, slot_stage, sseg, soff, tid;
  int dbg = 0;
  __device__ __forceinline__ void stage_next() {
    while (soff >= seg_n[sseg]) { soff = 0; sseg = (sseg + 1 == nseg) ? 0 : sseg + 1; }
    stage_quarter<NTH>(seg_ptr[sseg] + (size_t)soff * QUARTER, Wb[slot_stage], tid);
    ++soff;
    slot_stage = (slot_stage + 1 == NR) ? 0 : slot_stage + 1;
  }
  __device__ __forceinline__ void init(const unsigned short* const* sp, const int* sn, int nseg_, int my_tiles,
                                       unsigned short (*wb)[QUARTER], int tid_) {
    seg_ptr = sp; seg_n = sn; nseg = nseg_; Wb = wb; tid = tid_;
    int nq = 0;
    for (int i = 0; i < nseg; ++i) nq += sn[i];
    total = my_tiles * nq;
    consumed = slot = slot_stage = sseg = soff = 0;
    for (int d = 0; d < ND && d < total; ++d) stage_next();
  }
  __device__ __forceinline__ const unsigned short* take() {
    if (dbg & 3) {                       // timing experiments (wrong results): the ring is never restaged; 1: no barrier either -
                                         // what free-running waves would cost; 2: the barrier stays - what the LDS-DMA waits cost
      if (consumed == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); wg_barrier(); }
      else if (dbg & 2) wg_barrier();
      const unsigned short* cur = Wb[slot];
      slot = (slot + 1 == NR) ? 0 : slot + 1;
      ++consumed;
      return cur;
    }
    if (consumed + ND <= total) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((ND - 1) * GLDS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
    if (consumed + ND < total) stage_next();
    const unsigned short* cur = Wb[slot];
    slot = (slot + 1 == NR) ? 0 : slot + 1;
    ++consumed;
    return cur;
  }
};

// QuarterStreamU: the same stream with ONE barrier per GEMM unit (four quarters) instead of one per quarter.  Two units live
// in LDS (8 x 16 KB): at a unit boundary every wave waits for its LDS-DMA pieces of the unit it is about to read (staged a whole
// unit earlier), the workgroup meets once, and the slots of the unit just finished are refilled with the unit after next.
// Inside a unit the waves run free, so they drift apart by up to a unit: the two waves of a SIMD are no longer in their matrix
// and vector phases at the same instants.  Every GEMM of the kernels that use it consumes exactly four quarters.
template <int NTH>
struct QuarterStreamU {
  static constexpr int GLDS = 1024 / NTH, NR = 8;
  const unsigned short* const* seg_ptr;
  const int* seg_n;
  unsigned short (*Wb)[QUARTER];
  int nseg, total, consumed, staged, sseg, soff, tid;
  int dbg = 0;
  __device__ __forceinline__ void stage_next() {
    while (soff >= seg_n[sseg]) { soff = 0; sseg = (sseg + 1 == nseg) ? 0 : sseg + 1; }
    stage_quarter<NTH>(seg_ptr[sseg] + (size_t)soff * QUARTER, Wb[staged & (NR - 1)], tid);
    ++soff;
    ++staged;
  }
  __device__ __forceinline__ void init(const unsigned short* const* sp, const int* sn, int nseg_, int my_tiles,
                                       unsigned short (*wb)[QUARTER], int tid_) {
    seg_ptr = sp; seg_n = sn; nseg = nseg_; Wb = wb; tid = tid_;
    int nq = 0;
    for (int i = 0; i < nseg; ++i) nq += sn[i];
    total = my_tiles * nq;
    consumed = staged = sseg = soff = 0;
    for (int d = 0; d < NR && d < total; ++d) stage_next();          // units 0 and 1
  }
  __device__ __forceinline__ const unsigned short* take() {
    if ((consumed & 3) == 0) {
      // unit boundary: my pieces of this unit have landed once nothing older than the NEXT unit's pieces is outstanding
      if (consumed == 0 && total > 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * GLDS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      wg_barrier();
      if (consumed >= 4) for (int d = 0; d < 4 && staged < total; ++d) stage_next();    // the unit after next, into the slots just freed
    }
    const unsigned short* cur = Wb[consumed & (NR - 1)];
    ++consumed;
    return cur;
  }
};

// Row-scaled fragments: the 128-vector of this lane's row is multiplied by the power of two that brings its
// largest magnitude into [2^14, 2^15) (exact), split into fp16 hi/lo B fragments, and the inverse factor is
// returned - scaling a column of B scales the same column of C, so the caller multiplies its GEMM result by it.
__device__ __forceinline__ float frags_scaled(const f32x4 (&v)[8], u32x4 (&Bh)[4], u32x4 (&Bl)[4]) {
  float m = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t)
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[t][0]), fabsf(v[t][1]))), fmaxf(fabsf(v[t][2]), fabsf(v[t][3])));
  m = xor_lanes_max(m);
  unsigned eb = __float_as_uint(m) >> 23;
  eb = min(max(eb, 15u), 253u);
  const float sc = __uint_as_float((268u - eb) << 23), inv = __uint_as_float((eb - 14u) << 23);
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      unsigned hi, lo;
      split_pair(v[2 * s + (w >> 1)][2 * (w & 1)] * sc, v[2 * s + (w >> 1)][2 * (w & 1) + 1] * sc, hi, lo);
      Bh[s][w] = hi;
      Bl[s][w] = lo;
    }
  return inv;
}

}  // namespace ig
