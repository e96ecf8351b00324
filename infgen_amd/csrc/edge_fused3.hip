// k_edge_fused3: k_edge_fused (edge_fused.hip: the edge side of one AttentionLayer, reference infgen/modules/layers.py:78-99,109,
// for a tile of 16 destination rows with the absorbed query U and the aggregate Z on chip) with another LANE LAYOUT in the edge loop.
//
// k_edge_fused gives lane l the columns (2 l, 2 l + 1) of every 128-wide row (K, V, rhat): per edge the eight dot products
// u_h . rhat_e over 128 columns are 8 packed products plus an all-to-all reduction over the 64 lanes (4 + 2 lane swaps, 10 adds,
// selects), and the eight aggregates z_h += p_h rhat_e need every head's p_h in every lane (8 v_readlane + 8 packed FMAs):
// ~61 vector instructions per edge, the kernel's bound (vector pipe 77 % busy).
//
// Here lane l = (head h = l >> 3, slice i = l & 7) owns, for ITS head only, 16 columns of rhat - the four 4-column chunks
// 32 k + 4 i .. + 3 (k = 0..3), so that the eight slices' 16-byte reads of one chunk tile 128 contiguous bytes (no LDS bank conflict):
//   score    u_h . rhat_e   = 8 packed FMAs in the lane + the 8-lane sum that q_h . k_j needs anyway (3 DPP adds)
//   z_h     += p_h rhat_e   = 8 packed FMAs with the lane's OWN p_h (no cross-lane broadcast)
// ~34 vector instructions per edge.  Every head needs all 128 columns of an edge's rhat row, so the row goes global -> LDS once
// (LDS-DMA, 1 KB = two consecutive edges' rows per instruction: a destination's rows are contiguous) and each lane reads its
// 64 bytes from there (4 ds_read_b128; the 8 heads read the same addresses - broadcasts).  The ring needs no LDS of its own: a
// wave's ring is the U / Z slot of the ROW IT IS PROCESSING (4112 B = 8 rows of 512 B) - the row's u tile is in registers by then
// and its z is written into the slot when the row is done.  Rows are dealt statically: the tile's rows sorted by edge count,
// wave w takes the (w + 1)-th longest and then the (w + 1)-th shortest.  K / V rows keep the (2 l, 2 l + 1) mapping (lane l's two
// columns belong to head l >> 3).  Phases 1 (u = q W'_kr) and 3 (agg' = agg + W'_vr z + b' sigma) are k_edge_fused's.
//
// 512 threads = 8 waves = one 16-row group, 75 KB of LDS: two workgroups per CU, like k_edge_fused<6, *, 1, 8>.
#include "kernels.h"
#include "layout.h"
#include "tile.cuh"
#include "split.cuh"
#include "edge_attn.cuh"

namespace ig {

constexpr int E3_LDU = H * D + 4;           // row stride of the U / Z tile in floats (+4: conflict-free b128 column writes)
constexpr int E3_LDA = D + 4;
static_assert(E3_LDU * 4 >= 8 * 512, "a row's slot holds the ring of eight 512-byte rhat rows");

// G = edges per trip (even, <= 8): their rhat rows (G / 2 LDS-DMA instructions) and K / V rows (2 G loads) are requested together
template <int G>
__global__ __launch_bounds__(512, 4) void k_edge_fused3(EdgeFusedArgs a) {
  static_assert(G % 2 == 0 && G >= 2 && G <= 8, "trips of 2, 4, 6 or 8 edges");
  __shared__ __attribute__((aligned(16))) float UZ[16 * E3_LDU];
  __shared__ __attribute__((aligned(16))) float AG[16 * E3_LDA];     // q tile (phase 1 -> 2), then agg (phase 2 -> 3)
  __shared__ float SG[16 * H];
  const int ngroups = a.groups ? *a.n_groups : (a.rows + 15) / 16;
  int tile = blockIdx.x;
  if (a.tiles_per_scene > 1) {              // XCD-aware tile order (edge_fused.hip): a scene's tiles share an L2
    const int tps = a.tiles_per_scene, grp = 8 * tps;
    const int bq = tile / grp, br = tile % grp;
    tile = bq * grp + (br % 8) * tps + br / 8;
  }
  if (tile >= ngroups) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int h = w, hp = h >> 1, hh = h & 1;
  const int r0 = 16 * (a.groups ? a.groups[tile] : tile);
  const int row = r0 + j;
  const bool valid = row < a.rows;
  const float* hdr = a.pack + AH_HDR;
  // The lists' bookkeeping first: the two small loads (edge counts, first edges of the tile's 16 rows) go out in FRONT of the q tile and
  // the 64 KB of weight fragments - vmcnt retires in order, so they return after one short round trip instead of behind that burst
  // (cycle stamps: ~6,000 cycles for the burst) - and the rows' source indices are requested while the burst is still landing.
  const int bk_rl = lane & 15;
  const int bk_dr = r0 + bk_rl;
  const int bk_cnt = bk_dr < a.rows ? a.es.cnt[bk_dr] : 0;
  const int bk_off = bk_dr < a.rows ? a.es.off[bk_dr] : 0;
  // ---- phase 1: u_h = q_h W'_kr,h (K = 16: v_mfma_f32_16x16x16_f16; B fragment = the head's 16 query values of row j), wave = head
  float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid) qv = *reinterpret_cast<const float4*>(a.Q + (size_t)row * D + DH * h + 4 * g);
  v4h ah[8], al[8];
  {
    const unsigned short* Wk = reinterpret_cast<const unsigned short*>(a.pack + AH_PRE) + (size_t)(4 + hp) * QUARTER +
                               (size_t)(hh * 8) * 2 * 256 + lane * 4;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      ah[ct] = *reinterpret_cast<const v4h*>(Wk + (ct * 2) * 256);
      al[ct] = *reinterpret_cast<const v4h*>(Wk + (ct * 2 + 1) * 256);
    }
  }
  // Every wave ranks the tile's rows by edge count itself (16 counts, 16 compares: the same two loads in all eight waves) and takes
  // the (w + 1)-th longest and the (w + 1)-th shortest row; the first 64 source indices of both rows are requested HERE, in front of
  // phase 1's arithmetic: a workgroup's life is a chain of dependent memory round trips (count -> indices -> K / V rows, for two rows),
  // and with two workgroups per CU that chain, not the arithmetic, is what a launch with short lists costs (map set: 5 edges per row,
  // ~110 us per launch).  Cycle stamps of one workgroup (map set, ~26,000 cycles): first loads back 2,700, ranked 3,300, indices
  // requested 6,800 (queued behind the weight burst), phase 1 done 9,400, rows 5,000 each, phase 3 3,000.
  int E2[2], eb2[2], sv2[2], rl2[2];
  {
    const int rl = bk_rl, cnt = bk_cnt, off = bk_off;
    int rank = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int ck = __builtin_amdgcn_readlane(cnt, k);          // (lanes 0..15 hold the 16 counts: no LDS round trip like __shfl)
      rank += (ck > cnt || (ck == cnt && k < rl)) ? 1 : 0;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int want = p ? 15 - w : w;
      const unsigned long long hit = __ballot(rank == want && lane < 16);
      const int rr = __builtin_ctzll(hit);                 // (ranks are a permutation of 0..15: exactly one lane)
      rl2[p] = rr;
      E2[p] = __builtin_amdgcn_readlane(cnt, rr);
      eb2[p] = __builtin_amdgcn_readlane(off, rr);
      sv2[p] = E2[p] > 0 ? a.es.src[eb2[p] + min(lane, min(E2[p], 64) - 1)] : 0;
    }
  }
  {
    *reinterpret_cast<float4*>(AG + j * E3_LDA + DH * h + 4 * g) = qv;
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
    float* urow = UZ + j * E3_LDU + h * D + 4 * g;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
      *reinterpret_cast<float4*>(urow + 16 * ct) = make_float4(acc[ct][0] * cq, acc[ct][1] * cq, acc[ct][2] * cq, acc[ct][3] * cq);
  }
  __syncthreads();

  // ---- phase 2: the edge loop; lane = (head eh, slice ei): columns 32 k + 4 ei .. + 3 (k = 0..3) of head eh
  {
    const int eh = lane >> 3, ei = lane & 7;
    const bool kv_once = a.kv_once != 0;
    const unsigned lo8 = 8u * (unsigned)lane;
    auto ld8 = [&](const float* base, bool nt) {
      return ea_ld(reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + lo8), nt);
    };
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      const int rl = pass ? rl2[1] : rl2[0];
      const int E = pass ? E2[1] : E2[0];
      const int e_base = pass ? eb2[1] : eb2[0];
      float* uz = UZ + rl * E3_LDU;
      // this lane's 16 columns of u_h (pairs of consecutive columns), q of its two K columns
      pk2 ux[8];
      {
        const float* up = uz + eh * D + 4 * ei;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 t = *reinterpret_cast<const float4*>(up + 32 * k);
          // (scores live in the log2 domain: u and q carry the factor log2 e, one multiply per row instead of one per edge)
          ux[2 * k] = pk2{t.x, t.y} * pk2{EA_LOG2E, EA_LOG2E};
          ux[2 * k + 1] = pk2{t.z, t.w} * pk2{EA_LOG2E, EA_LOG2E};
        }
      }
      const float2 q0 = *reinterpret_cast<const float2*>(AG + rl * E3_LDA + 2 * lane);
      const pk2 q = pk2{q0.x, q0.y} * pk2{EA_LOG2E, EA_LOG2E};
      // the slot is the ring from here on: the u reads above have returned before the first LDS-DMA write can land
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned ring = lds_addr(uz);
      const float* rbase = uz + 4 * ei;                   // + 128 s + 32 k: chunk k of this lane's 16 columns of ring row s
      pk2 zz[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) zz[k] = pk2{0.f, 0.f};
      pk2 ag = pk2{0.f, 0.f};
      float m = -INFINITY, lsum = 0.f;
      const int lhalf = lane >> 5;                        // which row of a DMA pair this lane copies 16 bytes of
      const unsigned lcol = 16u * (unsigned)(lane & 31);
      for (int c0 = 0; c0 < E; c0 += 64) {
        const int mc = min(64, E - c0);
        // source indices of up to 64 edges in one register: the first chunk's were requested at the top of the kernel
        int sv = pass ? sv2[1] : sv2[0];
        if (c0 > 0) sv = a.es.src[e_base + c0 + min(lane, mc - 1)];
        asm volatile("" : "+v"(sv));        // (a use in front of the trips: the wait for this load is taken here, not between a trip's LDS-DMA pieces)
        const char* rrows = reinterpret_cast<const char*>(a.es.rhat) + (size_t)(e_base + c0) * (D * sizeof(float));
        for (int i0 = 0; i0 < mc; i0 += G) {
          // rhat rows of the trip, two per LDS-DMA instruction (slots beyond the end of the list repeat the last edge: finite
          // data under a score of -inf), then the K / V rows
#pragma unroll
          for (int p = 0; p < G / 2; ++p) {
            const int ic = min(i0 + 2 * p + lhalf, mc - 1);
            lds_dma16(rrows + (size_t)ic * (D * sizeof(float)) + lcol, ring + 1024u * (unsigned)p);
          }
          pk2 kb[G], vb[G];
#pragma unroll
          for (int s = 0; s < G; ++s) {
            const int ic = min(i0 + s, mc - 1);
            const int sj = __builtin_amdgcn_readlane(sv, ic);
            kb[s] = ld8(a.Ksrc + (size_t)sj * D, kv_once);
            vb[s] = ld8(a.Vsrc + (size_t)sj * D, kv_once);
          }
          // the 2 G loads above are the only vector-memory operations younger than the LDS-DMA pieces: at most 2 G outstanding
          // means every piece has landed (vmcnt retires in order; the compiler barrier keeps the loads in front of the wait)
          asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * G) : "memory");
          // The loop is bound by the SIMD's issue slots (four waves x ~175 issue cycles per edge = the ~700 cycles per edge and wave
          // the cycle stamps show), so every instruction counts: q . k rides in the packed FMA chains of u . rhat, the log2 e factor
          // sits in u and q, the dead-slot mask is one scalar select + one v_min.
          {
#pragma unroll
            for (int s = 0; s < G; ++s) {
              pk2 r[8];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const float4 t = *reinterpret_cast<const float4*>(rbase + 128 * s + 32 * k);
                r[2 * k] = pk2{t.x, t.y};
                r[2 * k + 1] = pk2{t.z, t.w};
              }
              // score of head eh in the log2 domain: q_h . k_j + u_h . rhat_e, two chains of packed FMAs, then the head's 8-lane sum
              pk2 d2 = q * kb[s], d3 = ux[0] * r[0];
#pragma unroll
              for (int k = 1; k < 8; k += 2) d2 = pk_fma(ux[k], r[k], d2);
#pragma unroll
              for (int k = 2; k < 8; k += 2) d3 = pk_fma(ux[k], r[k], d3);
              d2 += d3;
              float val = sum8(d2[0] + d2[1]);
              // (-inf for the slots beyond the end of the list: the limit is wave-uniform - one scalar select, one v_min)
              {
                const unsigned lim = __builtin_amdgcn_readfirstlane(i0 + s < mc ? 0x7f800000u : 0xff800000u);
                asm("v_min_f32 %0, %1, %2" : "=v"(val) : "s"(lim), "v"(val));      // (fminf would add a quieting v_max of the limit)
              }
              const bool grow = val > m + EA_TAU;                // first edge: m = -inf
              if (__any(grow)) {
                const float mn = grow ? val : m;
                const float sc = __builtin_amdgcn_exp2f(m - mn);      // 0 on the first edge, 1 for heads that keep their reference
                lsum *= sc;
                const pk2 sc2 = bc_v(sc);
                ag *= sc2;
#pragma unroll
                for (int k = 0; k < 8; ++k) zz[k] *= sc2;
                m = mn;
              }
              const float pe = __builtin_amdgcn_exp2f(val - m);
              lsum += pe;
              const pk2 pe2 = bc_v(pe);
              ag = pk_fma(pe2, vb[s], ag);
#pragma unroll
              for (int k = 0; k < 8; ++k) zz[k] = pk_fma(pe2, r[k], zz[k]);
            }
          }
          // (the ring reads of this trip are consumed: the next trip's LDS-DMA may overwrite the slot)
        }
      }
      const float inv = 1.0f / (lsum + 1e-16f);
      *reinterpret_cast<float2*>(AG + rl * E3_LDA + 2 * lane) = make_float2(ag[0] * inv, ag[1] * inv);
      {
        float* zp = uz + eh * D + 4 * ei;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          *reinterpret_cast<float4*>(zp + 32 * k) = make_float4(zz[2 * k][0] * inv, zz[2 * k][1] * inv, zz[2 * k + 1][0] * inv, zz[2 * k + 1][1] * inv);
      }
      if (ei == 0) SG[rl * H + eh] = lsum * inv;
    }
  }
  // (phase 3's weight fragments are requested BEFORE the barrier: their L2 latency runs under the wait for the other waves)
  v8h p3h[4], p3l[4];
  {
    const unsigned short* Wv = reinterpret_cast<const unsigned short*>(a.pack + AH_POST) + (size_t)hp * QUARTER +
                               (size_t)(hh * 4) * 2 * 512 + lane * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      p3h[s] = *reinterpret_cast<const v8h*>(Wv + (s * 2) * 512);
      p3l[s] = *reinterpret_cast<const v8h*>(Wv + (s * 2 + 1) * 512);
    }
  }
  __syncthreads();

  // ---- phase 3: agg' = agg + W'_vr,h z_h + b'_h sigma_h  (k_attn_h's z-GEMM: |z| <= sqrt(127), static prescale 1024)
  {
    const float* zrow = UZ + j * E3_LDU + h * D + 8 * g;
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
      const float4 ag = *reinterpret_cast<const float4*>(AG + j * E3_LDA + DH * h + 4 * g);
      float4 o;
      o.x = ag.x + (acc[0] * zinv + bvr.x * sg);
      o.y = ag.y + (acc[1] * zinv + bvr.y * sg);
      o.z = ag.z + (acc[2] * zinv + bvr.z * sg);
      o.w = ag.w + (acc[3] * zinv + bvr.w * sg);
      *reinterpret_cast<float4*>(a.AGG + (size_t)row * D + DH * h + 4 * g) = o;
    }
  }
}

template __global__ void k_edge_fused3<4>(EdgeFusedArgs);
template __global__ void k_edge_fused3<6>(EdgeFusedArgs);
template __global__ void k_edge_fused3<8>(EdgeFusedArgs);

}  // namespace ig
