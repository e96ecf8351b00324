// k_edge_mfma: k_edge_fused (edge_fused.hip) with the SCORES of the relative-position term on the MATRIX pipe - the edge side of
// one AttentionLayer (reference infgen/modules/layers.py:78-92,109: propagate + PyG softmax) for a tile of 16 destination rows.
//
// Per edge the vector-pipe loop (edge_attn.cuh: EdgeAcc) spends ~100 instructions: ~40 on the eight scores u_h . r^ with the edge's
// normalised relative-position row r^ (128 values; products + a three-level cross-lane reduction), ~25 on the online softmax
// (a possible rescale of every accumulator per edge), ~25 on the eight accumulations z_h += p_h r^.  With sets of 20 - 60 edges
// per row (agent <-> agent, the map encoder's pt <-> pt) the launch is bound by that (profiles/r04f_pmc_sq_*: vector pipe 77 %
// active, matrix pipe 4 %).  Here a wave takes 16 edges of its row at a time:
//
//   scores   S^T[16 edges x 16] = R^[16 x 128] . U'[128 x 16]     (v_mfma_f32_16x16x32_f16, 4 k-steps, x 2 planes of R^)
//            columns of U' = the row's absorbed query u_h = q_h W'_kr,h as fp16 hi (columns 0..7, head = column) and fp16 lo
//            (columns 8..15): one DPP add per value (row_ror:8) gives hi.r + lo.r.  The A fragments - edge m's 8 features per
//            k-step - are 16 contiguous bytes of the edge's row: they come straight from global memory, no LDS.
//            q_h . k_src is added on the vector pipe (8 of the head's 16 dims x 4 edges per lane, the same DPP add joins the halves)
//   softmax  once per TRIP: 4 edges x 1 head per lane, log2 domain, reference moved only when exceeded by 2^8 (as EdgeAcc)
//   agg      sum_e p_h v_src in the same lane layout (8 dims x 4 edges), reduced over the four lane groups at the row's end
//   z        z_h += p_h r^ stays on the vector pipe, edge by edge in EdgeAcc's lane layout (lane l owns columns 2 l, 2 l + 1),
//            but with the trip's probabilities KNOWN: eight v_readlane from fixed lanes + eight packed FMAs per edge, no
//            maximum, no exponential, no rescale in the loop
//
// ~30 vector instructions per edge instead of ~100, in 128 registers and 75 KB of LDS: two 8-wave workgroups per CU like
// k_edge_fused.  (A first form with z on the matrix pipe as well - rhat tile through LDS-DMA and the transposing LDS reads of
// gfx950 - needed 255 registers and 6 KB of tile per wave; at 8 waves per CU it waited for memory: 358 us per agent-set launch
// against k_edge_fused's 308.  profiles/r05_edge_mfma_v1_full_matrix_loop.txt, commit 0529c2c.)
// R^ rows arrive in the "H8" format (kernels.h: fp16 of 2048 r^ + fp8 of the remainder, 384 B - k_fourier_h writes it); the z loop
// reads a row's two planes again (L1 / L2) and rebuilds 2048 r^ = hi + lo exactly in fp32.
// Phases 1 (u = q W'_kr) and 3 (agg + W'_vr z + b' sigma) are k_edge_fused's; phase 1 leaves U' as ready-made B fragments.
// Every reduction has a fixed order: results are bitwise reproducible; they differ from k_edge_fused's by rounding only.
#include "kernels.h"
#include "layout.h"
#include "tile.cuh"
#include "split.cuh"
#include "edge_attn.cuh"

namespace ig {

constexpr int EM_ROWS = 16;                 // destination rows per workgroup
constexpr int EM_WAVES = 8;
constexpr int EM_NT = 64 * EM_WAVES;
constexpr int EM_LDU = H * D + 4;           // floats per row image: U' fragments (4 KB), later the row's normalised z (fp32 [8][128])
constexpr int EM_LDA = D + 4;
constexpr int EM_ZG = 8;                    // rows of the z loop requested together

typedef _Float16 v2h __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v8h cvt8(uint2 b) {          // eight fp8 (e4m3) -> fp16, exact
  const v2h a0 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)b.x, 1.0f, false);
  const v2h a1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)b.x, 1.0f, true);
  const v2h a2 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)b.y, 1.0f, false);
  const v2h a3 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)b.y, 1.0f, true);
  return v8h{a0[0], a0[1], a1[0], a1[1], a2[0], a2[1], a3[0], a3[1]};
}
// columns 2 l, 2 l + 1 of an H8 row as 2048 r^ in fp32: fp16 pair + fp8 pair, both conversions and the sum exact
__device__ __forceinline__ pk2 h8_to_f32(unsigned hi, unsigned lo) {
  const v2h h = __builtin_bit_cast(v2h, hi);
  const pk2 l = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, false);
  return pk2{(float)h[0] + l[0], (float)h[1] + l[1]};
}

// timing experiment (INFGEN_EDGE_DBG bit 7): s_memtime of wave 0 of workgroup 2600 (a later round: warm instruction cache) at the marked points -> a.dbgbuf [64] x 64 bit
#define EM_STAMP(i) do { if ((a.dbg & 128) && blockIdx.x == 2600 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(a.dbgbuf)[i] = __builtin_readcyclecounter(); } while (0)
template <bool KV_ONCE>
__global__ __launch_bounds__(EM_NT, 4) void k_edge_mfma(EdgeFusedArgs a) {
  __shared__ __attribute__((aligned(16))) float UZ[EM_ROWS * EM_LDU];
  __shared__ __attribute__((aligned(16))) float AG[EM_ROWS * EM_LDA];     // q tile (phase 1 -> 2), then agg (phase 2 -> 3)
  __shared__ float SG[EM_ROWS * H];
  __shared__ float SCL[EM_ROWS * H];                                      // 1 / (scale of U' x 2048) per (row, head)
  __shared__ int M_CNT[EM_ROWS], M_OFF[EM_ROWS];
  __shared__ unsigned char row_order[EM_ROWS];
  const int ngroups = a.groups ? *a.n_groups : (a.rows + 15) / 16;
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
  const int h = w, hp = h >> 1, hh = h & 1;
  const int r0 = 16 * (a.groups ? a.groups[tile] : tile);
  const int row = r0 + j;
  const bool valid = row < a.rows;
  const float* hdr = a.pack + AH_HDR;
  // the tile's edge lists (count, first edge) -> LDS, and the order in which the waves take the rows: sorted by length, wave w takes
  // positions w and 15 - w (longest with shortest).  Only the ORDER changes: every row is summed trip by trip by one wave.
  if (w == EM_WAVES - 1) {
    const int rl = lane & 15;
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
  // ---- phase 1: u_h = q_h W'_kr,h (wave = head; k_edge_fused's arithmetic), left in LDS as the B fragments of the score product
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
    *reinterpret_cast<float4*>(AG + j * EM_LDA + DH * h + 4 * g) = qv;
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
    char* img = reinterpret_cast<char*>(UZ + j * EM_LDU);
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      unsigned h01, l01, h23, l23;
      split_pair(acc[ct][0] * su, acc[ct][1] * su, h01, l01);
      split_pair(acc[ct][2] * su, acc[ct][3] * su, h23, l23);
      const int ks = ct >> 1, kg = 2 * (ct & 1) + (g >> 1);
      char* p = img + ks * 1024 + (h + 16 * kg) * 16 + 8 * (g & 1);
      *reinterpret_cast<uint2*>(p) = make_uint2(h01, h23);
      *reinterpret_cast<uint2*>(p + 8 * 16) = make_uint2(l01, l23);
    }
    if (g == 0) SCL[j * H + h] = __uint_as_float((eu - 14u) << 23) * (1.0f / 2048.0f);
  }
  EM_STAMP(2);
  __syncthreads();
  EM_STAMP(3);
  if (a.dbg & 32) return;

  // ---- phase 2: the edge loop, one wave per destination row, 16 edges per trip
  if (!(a.dbg & 64)) {
    const int n = lane & 15;                 // column of S^T: head n & 7, fp16 hi (n < 8) or lo part of u; also: edge n of the trip (A fragments)
    const int hd = n & 7, half = n >> 3;
    const char* rh = reinterpret_cast<const char*>(a.es.rhat);
    EM_STAMP(4);
    for (int ri = 0; ri < 2; ++ri) {
      // the wave's rows: positions w and 15 - w of the sorted order
      const int rl = __builtin_amdgcn_readfirstlane((int)row_order[ri ? EM_ROWS - 1 - w : w]);
      const int E = __builtin_amdgcn_readfirstlane(M_CNT[rl]);
      const int e_base = __builtin_amdgcn_readfirstlane(M_OFF[rl]);
      int sv = E > 0 ? a.es.src[e_base + min(lane, E - 1)] : 0;            // source indices of up to 64 edges in one register
      const char* img = reinterpret_cast<const char*>(UZ + rl * EM_LDU);
      v8h ub[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) ub[ks] = *reinterpret_cast<const v8h*>(img + ks * 1024 + lane * 16);
      const float4 q0 = *reinterpret_cast<const float4*>(AG + rl * EM_LDA + DH * hd + 8 * half);
      const float4 q1 = *reinterpret_cast<const float4*>(AG + rl * EM_LDA + DH * hd + 8 * half + 4);
      const float cs = SCL[rl * H + hd];
      pk2 zz[H];                             // sum_e p_e,h 2048 r^_e: columns 2 lane, 2 lane + 1, every head
#pragma unroll
      for (int k = 0; k < H; ++k) zz[k] = pk2{0.f, 0.f};
      pk2 ag = {0.f, 0.f};                   // sum_e p_e,head(l) v_src: columns 2 lane, 2 lane + 1 (head lane >> 3)
      float m = -INFINITY, lsum = 0.f;

      for (int t0 = 0; t0 < E; t0 += 16) {
        const int ne = min(16, E - t0);
        if (t0 > 0 && (t0 & 63) == 0) sv = a.es.src[e_base + min(t0 + lane, E - 1)];       // lists beyond 64 edges: the next chunk of indices
        // ---- requests of the trip: A fragments of the score product (edge n: 4 x 16 bytes of the fp16 plane, 4 x 8 of the fp8
        // plane; slots beyond the list re-read its last edge), K and V rows of this lane's four edges (8 dims each)
        const char* arow = rh + (size_t)(e_base + min(t0 + n, E - 1)) * H8_ROW_BYTES + 16 * g;
        v8h ahi[4];
        uint2 alo[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          ahi[ks] = *reinterpret_cast<const v8h*>(arow + 64 * ks);
          alo[ks] = *reinterpret_cast<const uint2*>(arow + H8_LO_PLANE - 8 * g + 32 * ks);
        }
        f32x4 kf[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int sj = __shfl(sv, min(t0 + 4 * g + i, E - 1) & 63, 64);
          const float* kp = a.Ksrc + (size_t)sj * D + DH * hd + 8 * half;
          if constexpr (KV_ONCE) {
            kf[i][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(kp));
            kf[i][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(kp + 4));
          } else {
            kf[i][0] = *reinterpret_cast<const f32x4*>(kp);
            kf[i][1] = *reinterpret_cast<const f32x4*>(kp + 4);
          }
        }
        // ---- scores
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[ks], ub[ks], sacc, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(cvt8(alo[ks]), ub[ks], sacc, 0, 0, 0);
        float val[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float qk = q0.x * kf[i][0].x;
          qk = fmaf(q0.y, kf[i][0].y, qk); qk = fmaf(q0.z, kf[i][0].z, qk); qk = fmaf(q0.w, kf[i][0].w, qk);
          qk = fmaf(q1.x, kf[i][1].x, qk); qk = fmaf(q1.y, kf[i][1].y, qk); qk = fmaf(q1.z, kf[i][1].z, qk); qk = fmaf(q1.w, kf[i][1].w, qk);
          const float s = fmaf(sacc[i], cs, qk);               // this lane's part: (hi or lo of u) . r^  +  half of q . k
          const float tot = s + dpp_xor8(s);
          val[i] = (4 * g + i < ne) ? tot * EA_LOG2E : -INFINITY;
        }
        // ---- online softmax per trip (log2 domain; the reference only moves when a score exceeds it by more than EA_TAU)
        const float vmax = fmaxf(fmaxf(val[0], val[1]), fmaxf(val[2], val[3]));
        if (__any(vmax > m + EA_TAU)) {
          const float vm = xor_lanes_max(vmax);                // over the head's four lane groups (columns n and n ^ 8 already agree)
          const float mn = vm > m + EA_TAU ? vm : m;
          const float scl = __builtin_amdgcn_exp2f(m - mn);    // 0 at the first trip, 1 for heads that keep their reference
          lsum *= scl;
          {
            const float so = __shfl(scl, lane >> 3, 64);       // the factor of this lane's own head
            ag = pk2{ag[0] * so, ag[1] * so};
          }
#pragma unroll
          for (int k = 0; k < H; ++k) zz[k] *= bc_s(readlane_f(scl, k));      // lane k: column k = head k
          m = mn;
        }
        float p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { p[i] = __builtin_amdgcn_exp2f(val[i] - m); lsum += p[i]; }
        // ---- z and agg: edge by edge in the column layout (lane l: columns 2 l, 2 l + 1 of the edge's rhat and V rows).  The
        // probability of (edge e, head k) sits in lane k + 16 (e >> 2), register e & 3: eight v_readlane from fixed lanes for z,
        // one ds_bpermute for the lane's own head (agg).  EM_ZG edges' rows in flight.
        const char* zrow = rh + (size_t)(e_base + t0) * H8_ROW_BYTES;
#pragma unroll
        for (int e0 = 0; e0 < 16; e0 += EM_ZG) {
          if (e0 >= ne) break;
          unsigned rhi[EM_ZG], rlo[EM_ZG];
          pk2 vb[EM_ZG];
#pragma unroll
          for (int s_ = 0; s_ < EM_ZG; ++s_) {
            const int ec = min(e0 + s_, ne - 1);
            const char* rp = zrow + (size_t)ec * H8_ROW_BYTES;
            rhi[s_] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(rp + 4 * lane));      // (the row's last use)
            rlo[s_] = __builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(rp + H8_LO_PLANE + 2 * lane));
            const int sj = __builtin_amdgcn_readlane(sv, (t0 + ec) & 63);
            vb[s_] = ea_ld(a.Vsrc + (size_t)sj * D + 2 * lane, KV_ONCE);
          }
#pragma unroll
          for (int s_ = 0; s_ < EM_ZG; ++s_) {
            const int e = e0 + s_;               // (a slot beyond the list: its probability is exp2(-inf) = 0, its rows are the last edge's)
            const pk2 r2 = h8_to_f32(rhi[s_], rlo[s_]);
            const float pe = p[e & 3];
            const float po = __shfl(pe, (lane >> 3) + 16 * (e >> 2), 64);       // this lane's head: lane >> 3
            ag = pk2{fmaf(po, vb[s_][0], ag[0]), fmaf(po, vb[s_][1], ag[1])};
            float ph[H];
#pragma unroll
            for (int k = 0; k < H; ++k) ph[k] = readlane_f(pe, k + 16 * (e >> 2));
#pragma unroll
            for (int k = 0; k < H; ++k) zz[k] = pk_fma(bc_s(ph[k]), r2, zz[k]);
          }
        }
      }
      EM_STAMP(5 + 2 * ri);

      // ---- the row's results -> LDS (z over the row's own U' image)
      const float lt = xor_lanes(lsum);
      const float inv = 1.0f / (lt + 1e-16f);
      {
        const float io = __shfl(inv, lane >> 3, 64);
        *reinterpret_cast<float2*>(AG + rl * EM_LDA + 2 * lane) = make_float2(ag[0] * io, ag[1] * io);
      }
      if (g == 0 && half == 0) SG[rl * H + hd] = lt * inv;
      float* zr = UZ + rl * EM_LDU;
#pragma unroll
      for (int k = 0; k < H; ++k) {
        const float ih = readlane_f(inv, k) * (1.0f / 2048.0f);
        *reinterpret_cast<float2*>(zr + k * D + 2 * lane) = make_float2(zz[k][0] * ih, zz[k][1] * ih);
      }
      EM_STAMP(6 + 2 * ri);
    }
  }
  EM_STAMP(9);
  // (phase 3's weight fragments are requested before the barrier, as in k_edge_fused)
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
  EM_STAMP(10);
  __syncthreads();
  EM_STAMP(11);

  // ---- phase 3: agg' = agg + W'_vr,h z_h + b'_h sigma_h (k_edge_fused's)
  {
    const float* zrow = UZ + j * EM_LDU + h * D + 8 * g;
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

template __global__ void k_edge_mfma<false>(EdgeFusedArgs);
template __global__ void k_edge_mfma<true>(EdgeFusedArgs);

// fp32 rows -> H8 rows (operator-level entry infgen_rhat_to_h8: tests and callers that hold fp32 rows; the rollout's own rows are
// written in this form by k_fourier_h): one wave per row
__global__ __launch_bounds__(256) void k_rhat_to_h8(const float* __restrict__ in, int rows, char* __restrict__ out) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float2 v = *reinterpret_cast<const float2*>(in + (size_t)r * D + 2 * lane);
  unsigned hi, lo;
  h8_pair(v.x, v.y, hi, lo);
  *reinterpret_cast<unsigned*>(out + (size_t)r * H8_ROW_BYTES + 4 * lane) = hi;
  *reinterpret_cast<unsigned short*>(out + (size_t)r * H8_ROW_BYTES + H8_LO_PLANE + 2 * lane) = (unsigned short)lo;
}

}  // namespace ig
