// k_edge_mfma_z: the FIRST form of the matrix-pipe edge loop (commit 0529c2c: scores AND z aggregation of 16-edge trips
// as MFMAs, the rhat tile through LDS-DMA and gfx950's transposing LDS reads), re-cut into workgroups of FOUR waves that own 8
// destination rows: 61 KB of LDS and one wave per SIMD per workgroup, so that TWO workgroups share a CU (255 registers per wave) and
// one's phases 1 / 3 / launch run under the other's edge loop - the 8-wave form serialised 7 us of fixed cost per 16 rows.
// wave w: heads 2 w, 2 w + 1 in the matrix phases (16-column MFMAs, 8 columns used), rows at positions w and 7 - w of the tile's
// length-sorted order in the loop.  Opt-in: INFGEN_EDGE_MFMA=2.  Reference: infgen/modules/layers.py:78-92,109.
#include "kernels.h"
#include "layout.h"
#include "tile.cuh"
#include "split.cuh"
#include "edge_attn.cuh"

namespace ig {

constexpr int EM_ROWS = 8;                  // destination rows per workgroup
constexpr int EM_WAVES = 4;
constexpr int EM_NT = 64 * EM_WAVES;
constexpr int EM_LDU = H * D + 4;           // floats per row image: U' fragments (4 KB), later the row's normalised z (fp32 [8][128])
constexpr int EM_LDA = D + 4;
constexpr int EM_HI = 16 * 256;             // bytes of a wave's R^ tile: fp16 plane, then fp8 plane
constexpr int EM_LO = 16 * 128;
constexpr int EM_TILE = EM_HI + EM_LO;
constexpr float EM_PSCALE = 64.0f;          // p <= 2^8 (EA_TAU) -> 64 p <= 2^14 in fp16, its remainder stays a normal number for p >= 2^-9

typedef short v4s_t __attribute__((vector_size(8)));
typedef int v2i_t __attribute__((vector_size(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));

// 16-byte chunk swizzle of the fp16 plane (16 chunks per 256-byte row): chunk c of edge row m lives at position c ^ em_key(m).
// A bijection of 0..15 whose upper three bits are distinct over rows 0..7 and over rows 8..15 (the eight rows a half wave touches
// in a transposing read) and whose values differing in bit 0 belong to rows of one ds_read_b128 lane group
__device__ __forceinline__ int em_key(int m) { return (((m & 7) ^ ((m >> 3) << 2)) << 1) | (m >> 3); }

__device__ __forceinline__ v8h cvt8(uint2 b) {
  const v2h a0 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)b.x, 1.0f, false);
  const v2h a1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)b.x, 1.0f, true);
  const v2h a2 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)b.y, 1.0f, false);
  const v2h a3 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)b.y, 1.0f, true);
  return v8h{a0[0], a0[1], a1[0], a1[1], a2[0], a2[1], a3[0], a3[1]};
}
__device__ __forceinline__ v4h cvt4(unsigned b) {
  const v2h a0 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)b, 1.0f, false);
  const v2h a1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)b, 1.0f, true);
  return v4h{a0[0], a0[1], a1[0], a1[1]};
}
__device__ __forceinline__ float swap32_sum(float x, float y) {      // lower lanes: x + x(lane + 32); upper lanes: y(lane - 32) + y
  const u32x2_sw s = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}

// timing experiment (INFGEN_EDGE_DBG bit 7): s_memtime of wave 0 of workgroup 2600 (a later round: warm instruction cache) at the marked points -> a.dbgbuf [64] x 64 bit
#define EM_STAMP(i) do { if ((a.dbg & 128) && blockIdx.x == 2600 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(a.dbgbuf)[i] = __builtin_readcyclecounter(); } while (0)
template <bool KV_ONCE>
__global__ __launch_bounds__(EM_NT, 2) void k_edge_mfma_z(EdgeFusedArgs a) {
  __shared__ __attribute__((aligned(16))) float UZ[EM_ROWS * EM_LDU];
  __shared__ __attribute__((aligned(16))) float AG[EM_ROWS * EM_LDA];     // q tile (phase 1 -> 2), then agg (phase 2 -> 3)
  __shared__ __attribute__((aligned(16))) char RB[EM_WAVES * EM_TILE];    // per wave: the R^ rows of 16 edges
  __shared__ float SG[EM_ROWS * H];
  __shared__ float SCL[EM_ROWS * H];                                      // 1 / (scale of U' x 2048) per (row, head)
  __shared__ int M_CNT[EM_ROWS], M_OFF[EM_ROWS];
  __shared__ unsigned char row_order[EM_ROWS];
  const int ngroups = (a.rows + EM_ROWS - 1) / EM_ROWS;          // (no row-group lists: the launcher keeps those on k_edge_fused)
  int tile = blockIdx.x;
  if (a.tiles_per_scene > 1) {              // XCD-aware tile order (edge_fused.hip): a scene's tiles share an L2
    const int tps = a.tiles_per_scene, grp = 8 * tps;
    const int bq = tile / grp, br = tile % grp;
    tile = bq * grp + (br % 8) * tps + br / 8;
  }
  if (tile >= ngroups) return;
  EM_STAMP(0);
  if (a.dbg & 16) return;          // timing experiments (INFGEN_EDGE_DBG): 16 empty workgroups, 32 phase 1 only, 64 no row loop
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int r0 = EM_ROWS * tile;
  const int row = r0 + j;
  const bool valid = j < EM_ROWS && row < a.rows;       // (the matrix phases' tiles have 16 columns: 8 are rows)
  const float* hdr = a.pack + AH_HDR;
  // the tile's edge lists (count, first edge) -> LDS, and the order in which the waves take the rows: sorted by length, wave w takes
  // positions w and 7 - w (longest with shortest).  Only the ORDER changes: every row is summed trip by trip by one wave.
  if (w == EM_WAVES - 1) {
    const int rl = lane & (EM_ROWS - 1);
    const int dr = r0 + rl;
    const bool lv = dr < a.rows && !(a.dbg & 1);
    const int cnt = lv ? a.es.cnt[dr] : 0;
    const int eoff = lv ? a.es.off[dr] : 0;
    int rank = 0;
#pragma unroll
    for (int k = 0; k < EM_ROWS; ++k) {
      const int ck = __shfl(cnt, k, 64);
      rank += (ck > cnt || (ck == cnt && k < rl)) ? 1 : 0;
    }
    if (lane < EM_ROWS) { row_order[rank] = (unsigned char)rl; M_CNT[rl] = cnt; M_OFF[rl] = eoff; }
  }

  EM_STAMP(1);
  // ---- phase 1: u_h = q_h W'_kr,h (wave w: heads 2 w, 2 w + 1; k_edge_fused's arithmetic), left in LDS as the B fragments of the score product
#pragma unroll 1
  for (int hh = 0; hh < 2; ++hh) {
    const int h = 2 * w + hh, hp = w;
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
    if (j < EM_ROWS) *reinterpret_cast<float4*>(AG + j * EM_LDA + DH * h + 4 * g) = qv;
    float m = fmaxf(fmaxf(fabsf(qv.x), fabsf(qv.y)), fmaxf(fabsf(qv.z), fabsf(qv.w)));
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    unsigned ebits = __float_as_uint(m) >> 23;
    ebits = min(max(ebits, 15u), 253u);
    const float sc = __uint_as_float((268u - ebits) << 23), inv = __uint_as_float((ebits - 14u) << 23);
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
    // acc[ct][i] * cq = u_h[feature 16 ct + 4 g + i] of row j.  One power-of-two scale per (row, head) into the fp16 range,
    // then hi -> column h, lo -> column h + 8 of the row's U' image: [k-step ks][lane (n + 16 kg)][8 fp16], feature 32 ks + 8 kg + idx
    float um = 0.f;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      acc[ct] *= splat4(cq);
      um = fmaxf(um, fmaxf(fmaxf(fabsf(acc[ct][0]), fabsf(acc[ct][1])), fmaxf(fabsf(acc[ct][2]), fabsf(acc[ct][3]))));
    }
    um = xor_lanes_max(um);
    unsigned eu = __float_as_uint(um) >> 23;
    eu = min(max(eu, 15u), 253u);
    const float su = __uint_as_float((268u - eu) << 23);
    char* img = reinterpret_cast<char*>(UZ + (j & (EM_ROWS - 1)) * EM_LDU);
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      if (j >= EM_ROWS) break;
      unsigned h01, l01, h23, l23;
      split_pair(acc[ct][0] * su, acc[ct][1] * su, h01, l01);
      split_pair(acc[ct][2] * su, acc[ct][3] * su, h23, l23);
      const int ks = ct >> 1, kg = 2 * (ct & 1) + (g >> 1);
      char* p = img + ks * 1024 + (h + 16 * kg) * 16 + 8 * (g & 1);
      *reinterpret_cast<uint2*>(p) = make_uint2(h01, h23);
      *reinterpret_cast<uint2*>(p + 8 * 16) = make_uint2(l01, l23);
    }
    if (g == 0 && j < EM_ROWS) SCL[j * H + h] = __uint_as_float((eu - 14u) << 23) * (1.0f / 2048.0f);
  }
  EM_STAMP(2);
  __syncthreads();
  EM_STAMP(3);
  if (a.dbg & 32) return;

  // ---- phase 2: the edge loop, one wave per destination row, 16 edges per trip
  if (!(a.dbg & 64)) {
    const int n = lane & 15;                 // column of S^T / row of P': head n & 7, fp16 hi (n < 8) or lo part
    const int hd = n & 7, half = n >> 3;
    char* rb_hi = RB + w * EM_TILE;
    char* rb_lo = rb_hi + EM_HI;
    const unsigned lds_hi = __builtin_amdgcn_readfirstlane(lds_addr(rb_hi));
    const unsigned lds_lo = lds_hi + EM_HI;
    // LDS-DMA source offsets inside an H8 row: instruction k of the fp16 plane lands rows 4 k + (lane >> 4), 16-byte position
    // lane & 15; of the fp8 plane rows 8 k + (lane >> 3), position lane & 7
    int dma_hi_off[4], dma_lo_off[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_hi_off[k] = ((lane & 15) ^ em_key(4 * k + g)) * 16;
#pragma unroll
    for (int k = 0; k < 2; ++k) dma_lo_off[k] = H8_LO_PLANE + ((lane & 7) ^ ((8 * k + (lane >> 3)) >> 1)) * 16;
    // A fragments of the score product: edge row n (= lane & 15), features 32 ks + 8 g .. + 7
    const char* a_hi[4]; const char* a_lo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      a_hi[ks] = rb_hi + n * 256 + ((4 * ks + g) ^ em_key(n)) * 16;
      a_lo[ks] = rb_lo + n * 128 + ((4 * ks + g) ^ (n & 14)) * 8;
    }
    // B fragments of the z product through the transposing reads: this lane SUPPLIES the address of
    //   fp16: edge row 4 g + (n >> 2), features 16 t + 4 (n & 3) .. + 3      (8 bytes)
    //   fp8 : edge row 4 g + ((n >> 1) & 3), features 16 t + 8 (n & 1) .. + 7  (8 bytes; rows repeat: the read delivers eight)
    // and RECEIVES feature 16 t + n of edges 4 g .. 4 g + 3
    const int trow = 4 * g + (n >> 2), tq = n & 3;
    const char* t_hi = rb_hi + trow * 256 + (tq & 1) * 8;
    const int t_hi_key = em_key(trow) ^ (tq >> 1);
    const int brow = 4 * g + ((n >> 1) & 3);
    const char* t_lo = rb_lo + brow * 128;
    const int t_lo_key = (brow & 14) ^ (n & 1);
    const char* rh = reinterpret_cast<const char*>(a.es.rhat);

    // this wave's two rows (positions w and 7 - w of the sorted order), their lists and the source indices of their first 64 edges
    int rlx[2], Ex[2], obx[2], svx[2];
#pragma unroll
    for (int ri = 0; ri < 2; ++ri) {
      rlx[ri] = __builtin_amdgcn_readfirstlane((int)row_order[ri ? EM_ROWS - 1 - w : w]);
      Ex[ri] = __builtin_amdgcn_readfirstlane(M_CNT[rlx[ri]]);
      obx[ri] = __builtin_amdgcn_readfirstlane(M_OFF[rlx[ri]]);
      svx[ri] = Ex[ri] > 0 ? a.es.src[obx[ri] + min(lane, Ex[ri] - 1)] : 0;
    }
    int si[4];
    f32x4 kf[4][2], vf[4][2];
    // requests of one 16-edge trip (E_, eb_, sv_: the list it belongs to; on = false: aimed at one line, see below)
    auto load_si = [&](int sv_, int t0, int E_) {
#pragma unroll
      for (int i = 0; i < 4; ++i) si[i] = __shfl(sv_, min(t0 + 4 * g + i, E_ - 1) & 63, 64);
    };
    auto issue_dma = [&](int t0, int E_, int eb_, bool on) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int er = min(t0 + 4 * k + g, E_ - 1);
        lds_dma16(on ? rh + (size_t)(eb_ + er) * H8_ROW_BYTES + dma_hi_off[k] : rh, lds_hi + k * 1024);
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int er = min(t0 + 8 * k + (lane >> 3), E_ - 1);
        lds_dma16(on ? rh + (size_t)(eb_ + er) * H8_ROW_BYTES + dma_lo_off[k] : rh, lds_lo + k * 1024);
      }
    };
    auto issue_kv = [&](const float* base, f32x4 (&dst)[4][2], bool on) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* kp = on ? base + (size_t)si[i] * D + DH * hd + 8 * half : base;
        if constexpr (KV_ONCE) {
          dst[i][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(kp));
          dst[i][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(kp + 4));
        } else {
          dst[i][0] = *reinterpret_cast<const f32x4*>(kp);
          dst[i][1] = *reinterpret_cast<const f32x4*>(kp + 4);
        }
      }
    };
    EM_STAMP(4);
    bool inflight = false;                   // the current row's first trip has been requested (under the previous row's last trip)
    for (int ri = 0; ri < 2; ++ri) {
      const int rl = ri ? rlx[1] : rlx[0];
      const int E = ri ? Ex[1] : Ex[0];
      const int e_base = ri ? obx[1] : obx[0];
      int sv = ri ? svx[1] : svx[0];
      // what follows this row in the wave's stream (ri == 0: the second row, if it has edges)
      const int En = ri ? 0 : Ex[1], ebn = ri ? 0 : obx[1], svn = svx[1];
      const char* img = reinterpret_cast<const char*>(UZ + rl * EM_LDU);
      v8h ub[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) ub[ks] = *reinterpret_cast<const v8h*>(img + ks * 1024 + lane * 16);
      const float4 q0 = *reinterpret_cast<const float4*>(AG + rl * EM_LDA + DH * hd + 8 * half);
      const float4 q1 = *reinterpret_cast<const float4*>(AG + rl * EM_LDA + DH * hd + 8 * half + 4);
      const float cs = SCL[rl * H + hd];
      f32x4 zacc[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) zacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      float ag8[8];
#pragma unroll
      for (int d = 0; d < 8; ++d) ag8[d] = 0.f;
      float m = -INFINITY, lsum = 0.f;

      // Software pipeline over the wave's 16-edge trips: the K rows of the NEXT trip (of this row, or the first of the wave's
      // second row) are requested as soon as this trip's scores are formed, its rhat tile as soon as the last fragment of this
      // trip has been read, its V rows after this trip's aggregation - one wait per trip.  Requests beyond the wave's last trip
      // are unconditional (a conditional load is a basic block whose results hipcc merges with copies) but aimed at one line.
      if (E > 0 && !inflight) {
        load_si(sv, 0, E);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_kv(a.Ksrc, kf, true);
        issue_dma(0, E, e_base, true);
        issue_kv(a.Vsrc, vf, true);
      }
      for (int t0 = 0; t0 < E; t0 += 16) {
        const int t1 = t0 + 16;
        const bool more = t1 < E;                              // the next trip belongs to this row
        const bool hop = !more && En > 0;                      // ... is the first of the wave's second row
        const bool nx = more || hop;
        const int nt0 = more ? t1 : 0, nE = more ? E : (hop ? En : 1), neb = more ? e_base : ebn;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // trip t0 has landed (LDS-DMA retires through vmcnt)

        // ---- scores
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const v8h*>(a_hi[ks]), ub[ks], sacc, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(cvt8(*reinterpret_cast<const uint2*>(a_lo[ks])), ub[ks], sacc, 0, 0, 0);
        float val[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float qk = q0.x * kf[i][0].x;
          qk = fmaf(q0.y, kf[i][0].y, qk); qk = fmaf(q0.z, kf[i][0].z, qk); qk = fmaf(q0.w, kf[i][0].w, qk);
          qk = fmaf(q1.x, kf[i][1].x, qk); qk = fmaf(q1.y, kf[i][1].y, qk); qk = fmaf(q1.z, kf[i][1].z, qk); qk = fmaf(q1.w, kf[i][1].w, qk);
          const float s = fmaf(sacc[i], cs, qk);               // this lane's part: (hi or lo of u) . r^  +  half of q . k
          const float tot = s + dpp_xor8(s);
          val[i] = (t0 + 4 * g + i < E) ? tot * EA_LOG2E : -INFINITY;
        }
        if (more && (t1 & 63) == 0) sv = a.es.src[e_base + min(t1 + lane, E - 1)];      // lists beyond 64 edges: the next chunk of indices
        load_si(more ? sv : svn, nt0, nE);
        asm volatile("" ::: "memory");
        issue_kv(a.Ksrc, kf, nx);
        asm volatile("" ::: "memory");
        // ---- online softmax (log2 domain; the reference only moves when a score exceeds it by more than EA_TAU)
        const float vmax = fmaxf(fmaxf(val[0], val[1]), fmaxf(val[2], val[3]));
        if (__any(vmax > m + EA_TAU)) {
          const float vm = xor_lanes_max(vmax);                // over the head's four lane groups (columns n and n ^ 8 already agree)
          const float mn = vm > m + EA_TAU ? vm : m;
          const float scl = __builtin_amdgcn_exp2f(m - mn);    // 0 at the first trip, 1 for heads that keep their reference
          lsum *= scl;
#pragma unroll
          for (int d = 0; d < 8; ++d) ag8[d] *= scl;
          float sh[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) sh[k] = readlane_f(scl, k);         // lane k: column k = head k
          f32x4 f;
#pragma unroll
          for (int i = 0; i < 4; ++i) f[i] = (g & 1) ? sh[4 + i] : sh[i];  // rows 4 g + i of Z' belong to head (4 g + i) & 7
#pragma unroll
          for (int t = 0; t < 8; ++t) zacc[t] *= f;
          m = mn;
        }
        float p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { p[i] = __builtin_amdgcn_exp2f(val[i] - m); lsum += p[i]; }
        // ---- z: row n of P' = 64 p as fp16 hi (n < 8) or its remainder (n >= 8)
        unsigned h01, l01, h23, l23;
        split_pair(p[0] * EM_PSCALE, p[1] * EM_PSCALE, h01, l01);
        split_pair(p[2] * EM_PSCALE, p[3] * EM_PSCALE, h23, l23);
        const v4h pa = __builtin_bit_cast(v4h, half ? u32x2{l01, l23} : u32x2{h01, h23});
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const v4s_t bh = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) v4s_t*)(t_hi + ((2 * t) ^ t_hi_key) * 16));
          zacc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(pa, __builtin_bit_cast(v4h, bh), zacc[t], 0, 0, 0);
          const v2i_t bl = __builtin_amdgcn_ds_read_tr8_b64_v2i32(
              (__attribute__((address_space(3))) v2i_t*)(t_lo + ((2 * t) ^ t_lo_key) * 8));
          zacc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(pa, cvt4((unsigned)bl[0]), zacc[t], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every fragment of this trip has been read: the buffer may be refilled
        issue_dma(nt0, nE, neb, nx);
        // ---- agg: this lane's 8 dims of the head's value rows
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ag8[0] = fmaf(p[i], vf[i][0].x, ag8[0]); ag8[1] = fmaf(p[i], vf[i][0].y, ag8[1]);
          ag8[2] = fmaf(p[i], vf[i][0].z, ag8[2]); ag8[3] = fmaf(p[i], vf[i][0].w, ag8[3]);
          ag8[4] = fmaf(p[i], vf[i][1].x, ag8[4]); ag8[5] = fmaf(p[i], vf[i][1].y, ag8[5]);
          ag8[6] = fmaf(p[i], vf[i][1].z, ag8[6]); ag8[7] = fmaf(p[i], vf[i][1].w, ag8[7]);
        }
        asm volatile("" ::: "memory");
        issue_kv(a.Vsrc, vf, nx);
      }

      inflight = E > 0 && En > 0;
      EM_STAMP(5 + 2 * ri);
      // ---- the row's results -> LDS (z over the row's own U' image)
      const float lt = xor_lanes(lsum);
      const float inv = 1.0f / (lt + 1e-16f);
#pragma unroll
      for (int d = 0; d < 8; ++d) ag8[d] = xor_lanes(ag8[d]) * inv;
      if (g == 0) {
        float* ap = AG + rl * EM_LDA + DH * hd + 8 * half;
        *reinterpret_cast<float4*>(ap) = make_float4(ag8[0], ag8[1], ag8[2], ag8[3]);
        *reinterpret_cast<float4*>(ap + 4) = make_float4(ag8[4], ag8[5], ag8[6], ag8[7]);
        if (half == 0) SG[rl * H + hd] = lt * inv;
      }
      float ih[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) ih[k] = readlane_f(inv, k) * (1.0f / (EM_PSCALE * 2048.0f));
      float* zr = UZ + rl * EM_LDU;
      const int hb = 4 * (g & 1), tb = 4 * (g >> 1);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // rows 4 g + i (hi part, g < 2) and 4 g + i + 8 (lo part) of Z' are 32 lanes apart: lower lanes take tile t, upper t + 4
          const float z = swap32_sum(zacc[t][i], zacc[t + 4][i]);
          const float fi = (g & 1) ? ih[4 + i] : ih[i];
          zr[(hb + i) * D + 16 * (tb + t) + n] = z * fi;
        }
      EM_STAMP(6 + 2 * ri);
    }
  }
  EM_STAMP(9);
  EM_STAMP(10);
  __syncthreads();
  EM_STAMP(11);

  // ---- phase 3: agg' = agg + W'_vr,h z_h + b'_h sigma_h (k_edge_fused's; wave w: heads 2 w, 2 w + 1)
#pragma unroll 1
  for (int hh = 0; hh < 2; ++hh) {
    const int h = 2 * w + hh, hp = w;
    v8h p3h[4], p3l[4];
    const unsigned short* Wv = reinterpret_cast<const unsigned short*>(a.pack + AH_POST) + (size_t)hp * QUARTER +
                               (size_t)(hh * 4) * 2 * 512 + lane * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      p3h[s] = *reinterpret_cast<const v8h*>(Wv + (s * 2) * 512);
      p3l[s] = *reinterpret_cast<const v8h*>(Wv + (s * 2 + 1) * 512);
    }
    const int jr = j & (EM_ROWS - 1);
    const float* zrow = UZ + jr * EM_LDU + h * D + 8 * g;
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
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(p3h[s], vbh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(p3h[s], vbl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(p3l[s], vbh, acc, 0, 0, 0);
    }
    if (valid) {
      const float sg = SG[j * H + h];
      const float4 bvr = *reinterpret_cast<const float4*>(a.pack + AL_BVR + DH * h + 4 * g);
      const float4 ag = *reinterpret_cast<const float4*>(AG + j * EM_LDA + DH * h + 4 * g);
      float4 o;
      o.x = ag.x + (acc[0] * zinv + bvr.x * sg);
      o.y = ag.y + (acc[1] * zinv + bvr.y * sg);
      o.z = ag.z + (acc[2] * zinv + bvr.z * sg);
      o.w = ag.w + (acc[3] * zinv + bvr.w * sg);
      *reinterpret_cast<float4*>(a.AGG + (size_t)row * D + DH * h + 4 * g) = o;
    }
  }
  EM_STAMP(12);
}

template __global__ void k_edge_mfma_z<false>(EdgeFusedArgs);
template __global__ void k_edge_mfma_z<true>(EdgeFusedArgs);

}  // namespace ig
