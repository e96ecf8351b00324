// Edge-side kernels: per-step edge-set construction (temporal window / first-K map radius /
// agent-agent radius), the per-destination edge attention, contour integration + grid
// tokenisation, raw per-column feature gathers and the map-token radius graph.
#include "kernels.h"
#include "edge_attn.cuh"

namespace ig {

constexpr int INVALID = 0, VALID = 1, ENTER = 2, EXIT = 3;
constexpr int NUM_SEED_FEATURE = 10;     // reference agent_decoder.py:292
constexpr int A2A_MAX_NBR = 300;         // radius_graph(..., max_num_neighbors=300), agent_decoder.py:632-633
constexpr float MOTION_GAP = 1.0f, HEADING_GAP = 1.0f, INVALID_MOTION = -2.0f, INVALID_HEAD = -2.0f;

// ------------------------------------------------------------------------------------------
// k_edge_attn: one wavefront per destination row (4 rows per workgroup), single pass over the
// row's incoming edges with an online (running-max) softmax; no LDS, every global access is a
// coalesced 512-byte row.  Lane l owns columns 2l, 2l+1 (head l >> 3):
//   score_h = q_h . k_src,h + u_h . rhat_e      reduced with wave shuffles; the 8 per-head partial
//             sums are folded with a halving exchange (4 + 2 + 1 shuffles over lane bits 5,4,3) so
//             that lane l ends with the score of ITS head, then 3 more over bits 0..2
//   softmax   PyG semantics (layers.py:89): exp(s - max) / (sum + 1e-16); reference / sum kept per head, rescaled when a
//             score exceeds the reference by more than a factor 256 (edge_attn.cuh)
//   outputs   AGG = sum_e a_e v_src (own columns), Z_h = sum_e a_e,h rhat_e (all heads), SIG_h = sum_e a_e,h
// Rows without incoming edges produce exact zeros (0 / (0 + 1e-16)).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_edge_attn(EdgeAttnArgs a) {
  const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave_id());
  if (row >= a.rows) return;
  // padded layout (insertion): a row that holds no agent and cannot receive one in this step is left untouched, like in
  // the node kernels that consume agg / z / sigma (k_active_groups)
  if (a.n_agents && (row % a.A_cap) >= a.n_agents[row / a.A_cap] + a.margin + 15) return;
  const int E = __builtin_amdgcn_readfirstlane(a.es.cnt[row]);
  const int e_base = __builtin_amdgcn_readfirstlane(a.es.off[row]);
  const bool has_r = a.es.rhat != nullptr && a.U != nullptr;
  AttnState st;
  if (has_r) edge_attn_wave2<4, false, true>(a, row, E, e_base, 0, 1, st);
  else edge_attn_wave2<4, false, false>(a, row, E, e_base, 0, 1, st);
  edge_attn_write(a, row, st);
}

// Few destinations with long edge lists (the insertion seed node: up to 300 agents + 2048 map tokens):
// one 8-wave workgroup per destination, wave w takes edges w, w + 8, ...; the eight partial softmax
// states are merged through LDS (flash-decoding style split over the edge list).
constexpr int WIDE_WAVES = 8;
__global__ __launch_bounds__(64 * WIDE_WAVES) void k_edge_attn_wide(EdgeAttnArgs a) {
  __shared__ float sm[WIDE_WAVES][H];
  __shared__ float sl[WIDE_WAVES][H];
  __shared__ __attribute__((aligned(16))) float sag[WIDE_WAVES][D];
  __shared__ __attribute__((aligned(16))) float sz[WIDE_WAVES][H * D];
  const int row = blockIdx.x;
  const int w = wave_id(), lane = lane_id();
  const int E = (a.row_mask && !a.row_mask[row]) ? 0 : a.es.cnt[row];
  const int e_base = a.es.off[row];
  const bool has_r = a.es.rhat != nullptr && a.U != nullptr;
  AttnState st;
  if (has_r) edge_attn_wave2<4, false, true>(a, row, E, e_base, w, WIDE_WAVES, st);
  else edge_attn_wave2<4, false, false>(a, row, E, e_base, w, WIDE_WAVES, st);
  if ((lane & 7) == 0) { sm[w][lane >> 3] = st.m; sl[w][lane >> 3] = st.lsum; }
  __syncthreads();
  // rescale this wave's partial state to the global max of every head, publish, then wave 0 sums
  const int myh = lane >> 3;
  float mt = -INFINITY;
#pragma unroll
  for (int ww = 0; ww < WIDE_WAVES; ++ww) mt = fmaxf(mt, sm[ww][myh]);
  const float sc = (st.m == -INFINITY) ? 0.f : exp2f(st.m - mt);  // (references live in the log2 domain) waves that saw no edge contribute nothing
  *reinterpret_cast<float2*>(&sag[w][2 * lane]) = make_float2(st.ag.x * sc, st.ag.y * sc);
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const float sh = readlane_f(sc, 8 * h);
    *reinterpret_cast<float2*>(&sz[w][h * D + 2 * lane]) = make_float2(st.z[h].x * sh, st.z[h].y * sh);
  }
  if ((lane & 7) == 0) sl[w][myh] = st.lsum * sc;
  __syncthreads();
  if (w != 0) return;
  AttnState tot;
  tot.ag = make_float2(0.f, 0.f);
  tot.lsum = 0.f;
  tot.m = mt;
#pragma unroll
  for (int h = 0; h < H; ++h) tot.z[h] = make_float2(0.f, 0.f);
  for (int ww = 0; ww < WIDE_WAVES; ++ww) {
    const float2 g = *reinterpret_cast<const float2*>(&sag[ww][2 * lane]);
    tot.ag.x += g.x; tot.ag.y += g.y;
    tot.lsum += sl[ww][myh];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float2 zz = *reinterpret_cast<const float2*>(&sz[ww][h * D + 2 * lane]);
      tot.z[h].x += zz.x; tot.z[h].y += zz.y;
    }
  }
  edge_attn_write(a, row, tot);
}

// ------------------------------------------------------------------------------------------
// helpers on the column-major scene state
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t sidx(const SceneState& st, int s, int col, int a) {
  return ((size_t)s * st.T + col) * st.A_cap + a;
}

// block-wide exclusive scan of one int per thread (BT threads); returns exclusive prefix, total in *tot.  n (<= BT, uniform): threads
// n.. hold zeros - the doubling stops at n (two barriers per step: 64 rows in a 1024-thread workgroup take 6 steps, not 10)
template <int BT>
__device__ __forceinline__ int block_excl_scan(int v, int* sh /*[BT+1]*/, int* tot, int n = BT) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int o = 1; o < n; o <<= 1) {
    int x = (t >= o) ? sh[t - o] : 0;
    __syncthreads();
    sh[t] += x;
    __syncthreads();
  }
  const int incl = sh[t];
  *tot = sh[n - 1];
  __syncthreads();
  return incl - v;
}

// ------------------------------------------------------------------------------------------
// k_build_edges: one workgroup per scene; builds the three edge sets of column c.
//   temporal  reference agent_decoder.py:540-610   (window W, bos, tmask, NOT the last 10 rows)
//   map->agent  :683-758  first 5 map tokens in ascending index with d^2 < r^2 (torch_cluster.radius)
//   agent<->agent :612-681  all ordered pairs within r, both interact-valid
// Raw continuous edge features (before the Fourier embedding) are written as float4.
// src holds the row index into the K/V source array of that edge type:
//   temporal: (j % ring) * rows + row ; map: s * M_cap + m ; agent: s * A_cap + j
// ------------------------------------------------------------------------------------------
// one thread per agent row of the scene: BT = 256 threads for A_cap <= 256, 1024 beyond (long rollouts with insertion) - and for
// few scenes at any A_cap: the map and agent parts are one wave per destination row, 16 waves take a quarter of 4 waves' trips
// (same lists in the same order: the ranks come from ballots and the row offsets from the scan over the rows)
constexpr int MAP_LDS = 4096; // map-token positions staged in LDS by k_build_edges

template <int BT>
__global__ __launch_bounds__(BT) void k_build_edges(BuildEdgesArgs a) {
  __shared__ float px[BT], py[BT], hd[BT], hc[BT], hs[BT];
  __shared__ int stt[BT];
  __shared__ unsigned char im[BT];
  __shared__ int scan[BT + 1];
  __shared__ int mapidx[BT * 5];
  __shared__ int mapcnt[BT];
  __shared__ int base_t, base_m, base_a;
  extern __shared__ __attribute__((aligned(8))) float2 mxy[];   // min(M_cap, MAP_LDS) slots: sized at launch so that short maps leave room for more workgroups per CU
  const SceneState& st = a.st;
  const int s = blockIdx.x;
  const int t = threadIdx.x;
  const int A = st.n_agents[s];
  const int c = a.c;
  const int rows_total = a.rows;
  const int row = s * st.A_cap + t;
  const int part = blockIdx.y;        // the three sets of a scene are built by three workgroups (grid S x 3): independent lists
  if (a.clear_keys && part == 0 && t < st.A_cap) a.clear_keys[row] = 0ull;
  if (a.clear_sync && part == 0 && t == 0) a.clear_sync[s] = 0;
  if (a.edgeless) {
    if (part == 0 && t < st.A_cap) {
      a.t.off[row] = 0; a.t.cnt[row] = 0;
      a.m.off[row] = 0; a.m.cnt[row] = 0;
      a.a.off[row] = 0; a.a.cnt[row] = 0;
    }
    return;
  }
  if (t < st.A_cap) {
    const size_t i = sidx(st, s, c, t);
    const float h = st.head[i];
    px[t] = st.pos[2 * i]; py[t] = st.pos[2 * i + 1];
    hd[t] = h; hc[t] = cosf(h); hs[t] = sinf(h);
    if (st.first_new && t >= st.first_new[s] && t < A) { hc[t] = st.hv_ovr[2 * s]; hs[t] = st.hv_ovr[2 * s + 1]; }
    stt[t] = st.state[i];
    im[t] = (t < A) ? st.imask[i] : 0;
  }
  __syncthreads();

  // ---------------- temporal (window <= 16 columns: every slot of the row is fetched up front, the
  // loads are independent and overlap)
  if (part == 0) {
    constexpr int WMAX = 16;
    int cnt = 0;
    const int lo = max(c - st.W, 0);
    const bool dst_ok = (t < A) && (t < A - NUM_SEED_FEATURE);
    unsigned ok_mask = 0;
    float sx[WMAX], sy[WMAX], sh[WMAX];
    int sst[WMAX];
    if (dst_ok) {
      const int bos = st.bos[s * st.A_cap + t];
#pragma unroll
      for (int wq = 0; wq < WMAX; ++wq) {
        const int j = lo + wq;
        sx[wq] = 0.f; sy[wq] = 0.f; sh[wq] = 0.f; sst[wq] = 0;
        if (j < c && wq < st.W) {
          const size_t i = sidx(st, s, j, t);
          const bool ok = j >= bos && st.tmask[i];
          sx[wq] = st.pos[2 * i]; sy[wq] = st.pos[2 * i + 1]; sh[wq] = st.head[i]; sst[wq] = st.state[i];
          if (ok) ok_mask |= 1u << wq;
        }
      }
      cnt = __popc(ok_mask);
    }
    int tot;
    const int excl = block_excl_scan<BT>(cnt, scan, &tot, min(st.A_cap, BT));
    if (t == 0) { base_t = atomicAdd(a.t.total, tot); if (a.prof) atomicAdd(a.prof + 8, (unsigned long long)tot); }
    __syncthreads();
    if (t < st.A_cap) {
      int e = base_t + excl;
      a.t.off[row] = e;
      a.t.cnt[row] = cnt;
      if (cnt > 0) {
        const bool d_inv = stt[t] == INVALID;
#pragma unroll
        for (int wq = 0; wq < WMAX; ++wq) {
          if (!((ok_mask >> wq) & 1u)) continue;
          const int j = lo + wq;
          float dx = sx[wq] - px[t], dy = sy[wq] - py[t];
          float dth = wrap_angle(sh[wq] - hd[t]);
          const bool s_inv = sst[wq] == INVALID;
          if (s_inv && !d_inv) { dx = -MOTION_GAP; dy = -MOTION_GAP; dth = -HEADING_GAP; }
          if (!s_inv && d_inv) { dx = MOTION_GAP; dy = MOTION_GAP; }      // :598 is a no-op
          if (s_inv && d_inv) { dx = INVALID_MOTION; dy = INVALID_MOTION; dth = INVALID_HEAD; }
          if (e < a.t.cap) {
            a.t.src[e] = (j % st.ring) * rows_total + row;
            *reinterpret_cast<float4*>(a.t.raw + 4 * (size_t)e) =
                make_float4(norm2(dx, dy), angle_between(hc[t], hs[t], dx, dy), dth, (float)(j - c));
          }
          ++e;
        }
      }
    }
    __syncthreads();
  }

  // ---------------- map -> agent (first 5 within radius, ascending index): one wave per agent
  if (part == 1) {
    const int ms = st.map_scene ? st.map_scene[s] : s;        // the scene's slot in the map-side arrays (copies of a scene share one)
    const int M = st.n_map[ms];
    const float r2 = a.r_map * a.r_map;
    const float* mp = st.map_pos + (size_t)ms * st.M_cap * 2;
    const int lane = lane_id();
    const bool map_in_lds = M <= a.map_lds;
    if (map_in_lds) {
      for (int m = t; m < M; m += BT) mxy[m] = *reinterpret_cast<const float2*>(mp + 2 * m);
      __syncthreads();
    }
    for (int ag = wave_id(); ag < st.A_cap; ag += BT / 64) {
      int found = 0;
      if (ag < A && im[ag]) {
        const float ax = px[ag], ay = py[ag];
        for (int m0 = 0; m0 < M && found < 5; m0 += 64) {
          const int m = m0 + lane;
          bool in = false;
          if (m < M) {
            const float2 mc = map_in_lds ? mxy[m] : *reinterpret_cast<const float2*>(mp + 2 * m);
            const float dx = ax - mc.x, dy = ay - mc.y;
            in = (dx * dx + dy * dy) < r2;
          }
          const unsigned long long bal = __ballot(in);
          const int before = __popcll(bal & ((1ull << lane) - 1ull));
          if (in && found + before < 5) mapidx[ag * 5 + found + before] = m;
          found = min(5, found + (int)__popcll(bal));
        }
      }
      if (lane == 0) mapcnt[ag] = found;
    }
    __syncthreads();
    const int cnt = (t < st.A_cap) ? mapcnt[t] : 0;
    int tot;
    const int excl = block_excl_scan<BT>(cnt, scan, &tot, min(st.A_cap, BT));
    if (t == 0) { base_m = atomicAdd(a.m.total, tot); if (a.prof) atomicAdd(a.prof + 9, (unsigned long long)tot); }
    __syncthreads();
    if (t < st.A_cap) {
      int e = base_m + excl;
      a.m.off[row] = e;
      a.m.cnt[row] = cnt;
      const bool d_inv = stt[t] == INVALID;
      const float* mo = st.map_orient + (size_t)ms * st.M_cap;
      for (int k = 0; k < cnt; ++k, ++e) {
        const int m = mapidx[t * 5 + k];
        float dx = mp[2 * m] - px[t], dy = mp[2 * m + 1] - py[t];
        float dth = wrap_angle(mo[m] - hd[t]);
        if (d_inv) { dx = MOTION_GAP; dy = MOTION_GAP; dth = HEADING_GAP; }
        if (e < a.m.cap) {
          a.m.src[e] = ms * st.M_cap + m;
          *reinterpret_cast<float4*>(a.m.raw + 4 * (size_t)e) =
              make_float4(norm2(dx, dy), angle_between(hc[t], hs[t], dx, dy), dth, 0.f);
        }
      }
    }
    __syncthreads();
  }

  // ---------------- agent <-> agent
  if (part == 2) {
    const float r2 = a.r_agent * a.r_agent;
    // radius_graph(..., loop=False, max_num_neighbors=300) over ALL rows of the column, masked ones included, and only then
    // subgraph(mask) (agent_decoder.py:632-634): per destination the first 300 + 1 rows in ascending index within the radius
    // (itself among them) are candidates; the self pair and the masked sources are dropped afterwards.
    // One wave per destination, the sources across the lanes (64 per trip): ballots give the candidate rank and the slot of
    // every emitted edge, so a list is written by consecutive lanes in ascending source index.
    const int lane = lane_id();
    int* acnt = mapcnt;      // (the map part's arrays are free in this workgroup)
    int* aoff = mapidx;
    auto scan_sources = [&](int ag, int e0, bool emit) {
      int found = 0, cnt = 0;
      const float ax = px[ag], ay = py[ag];
      const bool d_inv = stt[ag] == INVALID;
      for (int j0 = 0; j0 < A && found < A2A_MAX_NBR + 1; j0 += 64) {
        const int j = j0 + lane;
        bool in = false;
        if (j < A) {
          const float ddx = ax - px[j], ddy = ay - py[j];
          in = ddx * ddx + ddy * ddy < r2;
        }
        const unsigned long long bal = __ballot(in);
        const unsigned long long lower = (1ull << lane) - 1ull;
        const bool cand = in && found + (int)__popcll(bal & lower) < A2A_MAX_NBR + 1;
        const bool out = cand && j != ag && im[j];
        const unsigned long long obal = __ballot(out);
        if (emit && out) {
          const int e = e0 + cnt + (int)__popcll(obal & lower);
          float dx = px[j] - ax, dy = py[j] - ay;
          float dth = wrap_angle(hd[j] - hd[ag]);
          const bool s_inv = stt[j] == INVALID;
          if (s_inv && !d_inv) { dx = -MOTION_GAP; dy = -MOTION_GAP; dth = -HEADING_GAP; }
          if (!s_inv && d_inv) { dx = MOTION_GAP; dy = MOTION_GAP; }      // :650 is a no-op
          if (s_inv && d_inv) { dx = INVALID_MOTION; dy = INVALID_MOTION; dth = INVALID_HEAD; }
          if (e < a.a.cap) {
            a.a.src[e] = s * st.A_cap + j;
            *reinterpret_cast<float4*>(a.a.raw + 4 * (size_t)e) =
                make_float4(norm2(dx, dy), angle_between(hc[ag], hs[ag], dx, dy), dth, 0.f);
          }
        }
        cnt += (int)__popcll(obal);
        found += (int)__popcll(bal);
      }
      return cnt;
    };
    for (int ag = wave_id(); ag < st.A_cap; ag += BT / 64) {
      const int cnt = (ag < A && im[ag]) ? scan_sources(ag, 0, false) : 0;
      if (lane == 0) acnt[ag] = cnt;
    }
    __syncthreads();
    const int cnt = (t < st.A_cap) ? acnt[t] : 0;
    int tot;
    const int excl = block_excl_scan<BT>(cnt, scan, &tot, min(st.A_cap, BT));
    if (t == 0) { base_a = atomicAdd(a.a.total, tot); if (a.prof) atomicAdd(a.prof + 10, (unsigned long long)tot); }
    __syncthreads();
    if (t < st.A_cap) {
      aoff[t] = base_a + excl;
      a.a.off[row] = base_a + excl;
      a.a.cnt[row] = cnt;
    }
    __syncthreads();
    for (int ag = wave_id(); ag < st.A_cap; ag += BT / 64)
      if (acnt[ag] > 0) scan_sources(ag, aoff[ag], true);
  }
}
template __global__ void k_build_edges<256>(BuildEdgesArgs);
template __global__ void k_build_edges<1024>(BuildEdgesArgs);

// ------------------------------------------------------------------------------------------
// k_map_graph: radius_graph(pos, r, loop=False, max_num_neighbors=K) of the map tokens of each
// scene (reference map_decoder.py:91-93): per centre the first K+1 tokens in ascending index
// with d^2 < r^2 (self included), self dropped.  One wave per centre token, 4 tokens per
// workgroup; edges are compacted (one atomicAdd per workgroup reserves the block's range).
// raw = (|d|, angle(orient_vec[dst], d), wrap(orient[src] - orient[dst]), 0)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_map_graph(MapGraphArgs a) {
  // MG_C centre tokens per wave, one after the other, and the scene's token positions / orientations staged in LDS once per
  // workgroup (its 4 x MG_C centres belong to one scene when M_cap is a multiple of that, as the engine's layouts are): a wave's
  // scan is 16 dependent trips over the scene's 1024 tokens, and with every trip a global load (rounds 1 - 3: one centre per wave,
  // 3.7 ms per 1024-scene launch) the launch was the sum of those latencies; one atomic per workgroup reserves the slots of all its lists.
  constexpr int MG_C = MAP_GRAPH_C;
  __shared__ int wcnt[4];
  __shared__ int wbase[4];
  constexpr int MG_LIST = 128;                                           // emitted sources of one centre (max_nbr <= 128: the list path)
  __shared__ int nbr_list[4][MG_LIST];
  extern __shared__ __attribute__((aligned(8))) float2 mg_xy[];          // [lds_tokens] positions, then [lds_tokens] orientations
  const int lane = lane_id(), w = wave_id();
  const int gw0 = (blockIdx.x * 4 + w) * MG_C;
  const float r2 = a.radius * a.radius;
  const bool in_lds = a.lds_tokens >= a.M_cap && a.M_cap % (4 * MG_C) == 0;
  float* mg_o = reinterpret_cast<float*>(mg_xy + a.lds_tokens);
  if (in_lds) {
    const int s_wg = (blockIdx.x * 4 * MG_C) / a.M_cap;
    if (s_wg < a.S) {
      const int M = a.n_map[s_wg];
      const float* mp = a.pos + (size_t)s_wg * a.M_cap * 2;
      const float* mo = a.orient + (size_t)s_wg * a.M_cap;
      for (int m = threadIdx.x; m < M; m += NT) { mg_xy[m] = *reinterpret_cast<const float2*>(mp + 2 * m); mg_o[m] = mo[m]; }
    }
    __syncthreads();
  }
  int kept_c[MG_C];
  unsigned long long chunks_c[MG_C];
  int kept_sum = 0;
#pragma unroll
  for (int c = 0; c < MG_C; ++c) {
    const int gw = gw0 + c;
    const int s = gw / a.M_cap, i = gw % a.M_cap;
    const bool live = s < a.S;
    const int M = live ? a.n_map[s] : 0;
    const float* mp = a.pos + (size_t)(live ? s : 0) * a.M_cap * 2;
    const bool centre = live && i < M;
    float cx = 0.f, cy = 0.f;
    if (centre) { cx = mp[2 * i]; cy = mp[2 * i + 1]; }
    // pass 1: count kept neighbours (self excluded); every lane remembers in which 64-token chunks it emits (bit per chunk: up to
    // 4096 tokens per scene), so that pass 2 neither reloads nor re-tests the tokens that are not emitted
    int found = 0, kept = 0;
    unsigned long long emit_chunks = 0;
    const bool masks_ok = M <= 4096;
    if (centre) {
      for (int m0 = 0; m0 < M && found < a.max_nbr + 1; m0 += 64) {
        const int m = m0 + lane;
        bool in = false;
        if (m < M) {
          const float2 q = in_lds ? mg_xy[m] : *reinterpret_cast<const float2*>(mp + 2 * m);
          const float dx = cx - q.x, dy = cy - q.y;
          in = (dx * dx + dy * dy) < r2;
        }
        const unsigned long long bal = __ballot(in);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        const bool emit = in && (found + before < a.max_nbr + 1) && (m != i);
        if (emit && masks_ok) emit_chunks |= 1ull << (m0 >> 6);
        kept += (int)__popcll(__ballot(emit));
        found += (int)__popcll(bal);
      }
    }
    kept_c[c] = kept; chunks_c[c] = emit_chunks;
    kept_sum += kept;
  }
  if (lane == 0) wcnt[w] = kept_sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    const int base = tot > 0 ? atomicAdd(a.e.total, tot) : 0;
    wbase[0] = base; wbase[1] = base + wcnt[0]; wbase[2] = wbase[1] + wcnt[1]; wbase[3] = wbase[2] + wcnt[2];
  }
  __syncthreads();
  int e0 = wbase[w];
#pragma unroll
  for (int c = 0; c < MG_C; ++c) {
    const int gw = gw0 + c;
    const int s = gw / a.M_cap, i = gw % a.M_cap;
    if (s >= a.S) break;
    const int M = a.n_map[s];
    const int row = s * a.M_cap + i;
    const float* mp = a.pos + (size_t)s * a.M_cap * 2;
    const float* mo = a.orient + (size_t)s * a.M_cap;
    const bool centre = i < M;
    const int kept = kept_c[c];
    const unsigned long long emit_chunks = chunks_c[c];
    const bool masks_ok = M <= 4096;
    const int e_row = e0;
    e0 += kept;
    if (lane == 0) { a.e.off[row] = e_row; a.e.cnt[row] = (e_row + kept <= a.e.cap) ? kept : 0; }
    if (!centre || kept == 0 || e_row + kept > a.e.cap) continue;   // overflow: total > cap is reported by the host
    const float cx = mp[2 * i], cy = mp[2 * i + 1], co = mo[i];
    const float ocs = cosf(co), osn = sinf(co);
    int found = 0;
    int written = 0;
    if (masks_ok && kept <= MG_LIST) {
      // the emitted sources are first compacted into a list (their ranks come from ballots, a few instructions per 64-token
      // chunk), then the features - a square root, an atan2 and an angle wrap per edge - are evaluated with the lanes full: the
      // ~22 neighbours of a centre sit in ~10 different chunks, and evaluating them chunk by chunk ran the ~90-instruction
      // feature block ten times per centre with two or three live lanes (most of the launch: the scan itself is ~20
      // instructions per chunk)
      for (int m0 = 0; m0 < M && written < kept; m0 += 64) {
        const bool emit = (emit_chunks >> (m0 >> 6)) & 1ull;
        const unsigned long long ebal = __ballot(emit);
        if (emit) nbr_list[w][written + __popcll(ebal & ((1ull << lane) - 1ull))] = m0 + lane;
        written += (int)__popcll(ebal);
      }
      // (one wave: its LDS writes are visible to its own later reads in program order)
      __builtin_amdgcn_wave_barrier();
      for (int k0 = 0; k0 < kept; k0 += 64) {
        const int k = k0 + lane;
        if (k < kept) {
          const int m = nbr_list[w][k];
          const float2 q = in_lds ? mg_xy[m] : *reinterpret_cast<const float2*>(mp + 2 * m);
          const float qo = in_lds ? mg_o[m] : mo[m];
          const float dx = q.x - cx, dy = q.y - cy;
          const int e = e_row + k;
          a.e.src[e] = s * a.M_cap + m;
          *reinterpret_cast<float4*>(a.e.raw + 4 * (size_t)e) =
              make_float4(norm2(dx, dy), angle_between(ocs, osn, dx, dy), wrap_angle(qo - co), 0.f);
        }
      }
      continue;
    }
    for (int m0 = 0; m0 < M && (masks_ok ? written < kept : found < a.max_nbr + 1); m0 += 64) {
      const int m = m0 + lane;
      bool emit;
      if (masks_ok) {
        emit = (emit_chunks >> (m0 >> 6)) & 1ull;
      } else {
        bool in = false;
        if (m < M) {
          const float2 q = in_lds ? mg_xy[m] : *reinterpret_cast<const float2*>(mp + 2 * m);
          const float dx = cx - q.x, dy = cy - q.y;
          in = (dx * dx + dy * dy) < r2;
        }
        const unsigned long long bal = __ballot(in);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        emit = in && (found + before < a.max_nbr + 1) && (m != i);
        found += (int)__popcll(bal);
      }
      const unsigned long long ebal = __ballot(emit);
      const int ebefore = __popcll(ebal & ((1ull << lane) - 1ull));
      if (emit) {
        const int e = e_row + written + ebefore;
        const float2 q = in_lds ? mg_xy[m] : *reinterpret_cast<const float2*>(mp + 2 * m);
        const float qo = in_lds ? mg_o[m] : mo[m];
        const float dx = q.x - cx, dy = q.y - cy;
        a.e.src[e] = s * a.M_cap + m;
        *reinterpret_cast<float4*>(a.e.raw + 4 * (size_t)e) =
            make_float4(norm2(dx, dy), angle_between(ocs, osn, dx, dy), wrap_angle(qo - co), 0.f);
      }
      written += (int)__popcll(ebal);
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_integrate: token -> contour -> next pose (reference agent_decoder.py:2168-2239), grid
// tokenisation of the new position (attr_tokenizer.py:77-89), invalid handling.
// One workgroup per scene.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void rawfeat_prep_item(const RawFeatArgs& a, int row, int slot, int c4);     // (defined below)
constexpr int GRID_LDS = 2048;   // grid cells staged in LDS (1961 for the 150 m / 3 m / 75 m grid)
// a.groups > 1 (few scenes): the scene's rows are dealt to `groups` workgroups (grid S x groups), A_cap / groups rows each - the
// grid-cell search is one wave per agent and 16 waves searching for 64 agents one after the other were most of the launch (41 us at
// 8 scenes on 8 of 256 CUs).  Every workgroup also integrates the ego's step for itself (the search is relative to the ego's NEW
// pose) without storing it; the arg-max keys are then cleared by the next launch (k_build_edges: clear_keys), not here - another
// workgroup may still have to read the ego's.
template <int BT>
__global__ __launch_bounds__(BT) void k_integrate(IntegrateArgs a) {
  __shared__ float npx[BT], npy[BT], nth[BT];
  __shared__ int nst[BT];
  __shared__ float ego[3];
  __shared__ __attribute__((aligned(8))) float2 gxy[GRID_LDS];
  const bool grid_in_lds = a.grid_size <= GRID_LDS;
  if (grid_in_lds)
    for (int g = threadIdx.x; g < a.grid_size; g += BT) gxy[g] = *reinterpret_cast<const float2*>(a.grid_xy + 2 * g);
  const SceneState& st = a.st;
  const int s = blockIdx.x, tl = threadIdx.x;
  const int groups = a.groups > 1 ? a.groups : 1;
  const int per = st.A_cap / groups;                 // rows of this workgroup: [a0, a0 + per)
  const int a0 = (int)blockIdx.y * per;
  const int A = st.n_agents[s];
  const int av = st.av_index[s];
  const int c = a.c, n = a.c + 1;
  const bool own = tl < per;                          // this thread integrates row a0 + tl ...
  const bool dup = groups > 1 && tl == per && (av < a0 || av >= a0 + per);     // ... or the ego of another workgroup, unstored
  const int t = own ? a0 + tl : av;
  // the split arg-max keys of k_heads (ordered 64-bit (max, first index) keys): decoded here instead of by k_heads_finish, and
  // cleared for the next decode step - for EVERY row of the scene (an appended row must not inherit an older maximum)
  int tok_dec = 0;
  if (a.heads_part && (own || dup)) {
    const int row = s * st.A_cap + t;
    tok_dec = (int)(0xffffffffu - (unsigned)(a.heads_part[row] & 0xffffffffull));
    if (own) {
      if (groups == 1) a.heads_part[row] = 0ull;
      a.next_token_w[row] = tok_dec;
    }
  }
  if (a.edge_totals && s == 0 && blockIdx.y == 0 && tl < 3) a.edge_totals[tl] = 0;       // the next column's k_build_edges starts from zero
  if (a.zero_sync && blockIdx.y == 0 && tl == 0) a.zero_sync[s] = 0;
  if ((own || dup) && t < A) {
    const int row = s * st.A_cap + t;
    int tok = a.heads_part ? tok_dec : a.next_token[row];
    int ns = a.next_state[row];
    if (ns == 2) ns = EXIT;                     // valid_state_type index 2 == 'exit'
    if (t == av) ns = VALID;                    // ego forced valid
    if (a.force_valid) ns = VALID;              // disable_insertion
    if (a.teacher_token) { tok = a.teacher_token[sidx(st, s, n, t)]; if (tok < 0) tok = 0; }
    if (a.teacher_state) ns = a.teacher_state[sidx(st, s, n, t)];
    const size_t ic = sidx(st, s, c, t);
    const float th = st.head[ic];
    const float cs = cosf(th), sn = sinf(th);
    const float bx = st.pos[2 * ic], by = st.pos[2 * ic + 1];
    const int ty = st.type[s * st.A_cap + t];
    const float* ct = a.vocab + ((size_t)ty * a.token_size + tok) * 48;
    float lx = 0.f, ly = 0.f, lth = 0.f;
    for (int k = 1; k < 6; ++k) {
      float cx[4], cy[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float x = ct[k * 8 + q * 2], y = ct[k * 8 + q * 2 + 1];
        // [x y] @ [[cos, sin], [-sin, cos]] + pos
        cx[q] = (x * cs + y * (-sn)) + bx;
        cy[q] = (x * sn + y * cs) + by;
      }
      const float mx = (((cx[0] + cx[1]) + cx[2]) + cx[3]) / 4.0f;
      const float my = (((cy[0] + cy[1]) + cy[2]) + cy[3]) / 4.0f;
      const float hh = atan2f(cy[0] - cy[3], cx[0] - cx[3]);
      if (own) {
        const size_t o = ((size_t)row * a.R + a.t * 5 + (k - 1));
        a.pred_traj[2 * o] = mx; a.pred_traj[2 * o + 1] = my;
        a.pred_head[o] = hh;
        a.pred_state[o] = (float)ns;
      }
      if (k == 5) { lx = mx; ly = my; lth = hh; }
    }
    if (a.teacher_pos) {        // the stored pose is the teacher's (pred_traj / pred_head above keep this step's own result)
      const size_t in_ = sidx(st, s, n, t);
      lx = a.teacher_pos[2 * in_]; ly = a.teacher_pos[2 * in_ + 1]; lth = a.teacher_head[in_];
    }
    if (own) { npx[tl] = lx; npy[tl] = ly; nth[tl] = lth; nst[tl] = ns; }
    if (t == av) { ego[0] = lx; ego[1] = ly; ego[2] = lth; }
  }
  __syncthreads();
  // grid token of every agent: one wave per agent, lanes over grid cells
  {
    const float ex = ego[0], ey = ego[1];
    const float phi = -(ego[2] - HALF_PI_F);
    const float cs = cosf(phi), sn = sinf(phi);
    const int lane = lane_id();
    const int a1 = min(A, a0 + per);
    for (int ag = a0 + wave_id(); ag < a1; ag += BT / 64) {
      const int al = ag - a0;
      const float dx = npx[al] - ex, dy = npy[al] - ey;
      const float rx = dx * cs + dy * (-sn);
      const float ry = dx * sn + dy * cs;
      float best = INFINITY;
      int bi = 0x7fffffff;
#pragma unroll 4
      for (int g = lane; g < a.grid_size; g += 64) {
        const float2 gc = grid_in_lds ? gxy[g] : *reinterpret_cast<const float2*>(a.grid_xy + 2 * g);
        const float ux = rx - gc.x, uy = ry - gc.y;
        const float d = sqrtf(ux * ux + uy * uy);
        if (d < best) { best = d; bi = g; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (lane == 0) {
        const size_t in_ = sidx(st, s, n, ag);
        const int ns = nst[al];
        const bool inv = ns == INVALID;
        st.state[in_] = ns;
        st.pos[2 * in_] = inv ? 0.f : npx[al];
        st.pos[2 * in_ + 1] = inv ? 0.f : npy[al];
        st.head[in_] = inv ? 0.f : nth[al];
        int cell = bi;
        if (a.teacher_grid && a.teacher_grid[in_] >= -1) cell = a.teacher_grid[in_];
        st.grid[in_] = inv ? -1 : cell;
        int tok = a.heads_part ? a.next_token_w[s * st.A_cap + ag] : a.next_token[s * st.A_cap + ag];
        if (a.teacher_token) { tok = a.teacher_token[in_]; }
        st.token[in_] = inv ? -1 : tok;
        if (inv) { st.imask[in_] = 0; st.catflag[in_] = 0; }
      }
    }
  }
  if (a.do_prep) {
    // the raw-feature gather of the column just written (k_rawfeat_prep), the rows of this workgroup: one launch less per step
    __syncthreads();
    for (int item = tl; item < per * 32; item += BT) {
      const int row = s * st.A_cap + a0 + (item >> 5);
      rawfeat_prep_item(a.prep, row, row, item & 31);
    }
  }
}
template __global__ void k_integrate<1024>(IntegrateArgs);

// ------------------------------------------------------------------------------------------
// k_rawfeat_prep: inputs of the raw per-column agent feature (reference agent_decoder.py:426-509,
// 2265-2287) for column `col`: motion/heading 2-vector for x_a_emb, the categorical
// embedding row, and the gathered token / state / grid embedding rows of the fusion input.
// One thread per float4 of a row (32 threads per row).
// ------------------------------------------------------------------------------------------
// one (row, 16-byte column group) item of the raw-feature gather: `row` = the state row read, `slot` = the row of the
// raw2 / cat / fus_in arrays written (differ only for row subsets)
__device__ __forceinline__ void rawfeat_prep_item(const RawFeatArgs& a, int row, int slot, int c4) {
  const SceneState& st = a.st;
  const int s = row / st.A_cap, ag = row % st.A_cap;
  const int j = a.col;
  const size_t i = sidx(st, s, j, ag);
  const int stj = st.state[i];
  if (c4 == 0) {
    float mx = 0.f, my = 0.f;
    const bool inv = stj == INVALID;
    if (j > 0) {
      const size_t ip = sidx(st, s, j - 1, ag);
      mx = st.pos[2 * i] - st.pos[2 * ip];
      my = st.pos[2 * i + 1] - st.pos[2 * ip + 1];
      if (inv) { mx = INVALID_MOTION; my = INVALID_MOTION; }
      const bool pinv = st.state[ip] == INVALID;
      if (pinv && !inv) { mx = MOTION_GAP; my = MOTION_GAP; }
      if (!pinv && inv) { mx = -MOTION_GAP; my = -MOTION_GAP; }
    } else {
      if (inv) { mx = INVALID_MOTION; my = INVALID_MOTION; }
      if (stj == ENTER) { mx = MOTION_GAP; my = MOTION_GAP; }
    }
    const float h = st.head[i];
    *reinterpret_cast<float4*>(a.raw2 + 4 * (size_t)slot) =
        make_float4(norm2(mx, my), angle_between(cosf(h), sinf(h), mx, my), 0.f, 0.f);
  }
  const float* catsrc = st.catflag[i] ? a.cat_agent + (size_t)row * D : a.cat_seed;
  *reinterpret_cast<float4*>(a.cat + (size_t)slot * D + 4 * c4) = *reinterpret_cast<const float4*>(catsrc + 4 * c4);
  int tok = st.token[i];
  if (tok < 0) tok += a.token_size + 2;             // python negative indexing: -1 no_token, -2 bos
  const int ty = st.type[row];
  const float* tsrc = a.tok_tab + ((size_t)ty * (a.token_size + 2) + tok) * D;
  int g = st.grid[i];
  if (g < 0) g += a.grid_size + 1;                  // -1 -> invalid row
  const float* gsrc = a.grid_tab + (size_t)g * D;
  const float* ssrc = a.state_emb + (size_t)stj * D;
  float* f = a.fus_in + (size_t)slot * 512;
  *reinterpret_cast<float4*>(f + 4 * c4) = *reinterpret_cast<const float4*>(tsrc + 4 * c4);
  *reinterpret_cast<float4*>(f + 256 + 4 * c4) = *reinterpret_cast<const float4*>(ssrc + 4 * c4);
  *reinterpret_cast<float4*>(f + 384 + 4 * c4) = *reinterpret_cast<const float4*>(gsrc + 4 * c4);
}

__global__ __launch_bounds__(NT) void k_rawfeat_prep(RawFeatArgs a) {
  const SceneState& st = a.st;
  const int gid = blockIdx.x * NT + threadIdx.x;
  const int slot = gid >> 5, c4 = gid & 31;
  const int rows = st.S * st.A_cap;
  if (slot >= (a.row_list ? a.n_list : rows)) return;
  // row subset (insertion: the rows appended in this iteration): read row_list[slot], write the compact slot
  const int row = a.row_list ? (a.row_mask[slot] ? a.row_list[slot] : 0) : slot;
  rawfeat_prep_item(a, row, slot, c4);
}

// dst[row_list[k]][:] = src[k][:] for the rows with row_mask[k] != 0 (128 floats per row, 32 threads per row)
__global__ __launch_bounds__(NT) void k_scatter_rows(const float* src, const int* row_list, const int* row_mask, int n, float* dst) {
  const int gid = blockIdx.x * NT + threadIdx.x;
  const int k = gid >> 5, c4 = gid & 31;
  if (k >= n || !row_mask[k]) return;
  *reinterpret_cast<float4*>(dst + (size_t)row_list[k] * D + 4 * c4) = *reinterpret_cast<const float4*>(src + (size_t)k * D + 4 * c4);
}

// the same for two arrays (K and V of the riders) in one launch
__global__ __launch_bounds__(NT) void k_scatter_rows2(const float* src0, const float* src1, const int* row_list, const int* row_mask,
                                                      int n, float* dst0, float* dst1) {
  const int gid = blockIdx.x * NT + threadIdx.x;
  const int k = gid >> 5, c4 = gid & 31;
  if (k >= n || !row_mask[k]) return;
  const size_t o = (size_t)row_list[k] * D + 4 * c4, i = (size_t)k * D + 4 * c4;
  *reinterpret_cast<float4*>(dst0 + o) = *reinterpret_cast<const float4*>(src0 + i);
  *reinterpret_cast<float4*>(dst1 + o) = *reinterpret_cast<const float4*>(src1 + i);
}

// the rows / flags of this iteration become the riders of the next seed chain (prev) and heading stage (pend)
__global__ __launch_bounds__(NT) void k_note_riders(const int* new_row, const int* inserted, int n, int* prev_row, int* prev_mask,
                                                    int* pend_row, int* pend_mask) {
  const int k = blockIdx.x * NT + threadIdx.x;
  if (k >= n) return;
  const int r = new_row[k], m = inserted[k];
  prev_row[k] = r; prev_mask[k] = m; pend_row[k] = r; pend_mask[k] = m;
}

// k_gather_rows: dst[k] = src[row_list[k]] (128 floats) where row_mask[k] != 0 (null: everywhere), zeros elsewhere; rows clamped
// to [0, limit).  With row_list == null: dst[k] = src[0] (a broadcast row).
__global__ __launch_bounds__(NT) void k_gather_rows(const float* src, const int* row_list, const int* row_mask, int n, int limit,
                                                    float* dst) {
  const int gid = blockIdx.x * NT + threadIdx.x;
  const int k = gid >> 5, c4 = gid & 31;
  if (k >= n) return;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!row_list) v = *reinterpret_cast<const float4*>(src + 4 * c4);
  else if (!row_mask || row_mask[k]) {
    const int r = min(max(row_list[k], 0), limit - 1);
    v = *reinterpret_cast<const float4*>(src + (size_t)r * D + 4 * c4);
  }
  *reinterpret_cast<float4*>(dst + (size_t)k * D + 4 * c4) = v;
}

// k_embedding_sum4: out[k] = tab0[i0[k]] + ((tab1[i1[k]] + tab2[i2[k]]) + tab3[i3[k]]) - the map-token table row plus its three
// categorical embeddings (map_decoder.py:87-89) in one pass over the rows instead of four gathers and three adds.  32 threads
// per row, one float4 each.
__global__ __launch_bounds__(NT) void k_embedding_sum4(EmbedSum4Args a) {
  const int gid = blockIdx.x * NT + threadIdx.x;
  const int k = gid >> 5, c4 = gid & 31;
  if (k >= a.rows) return;
  float4 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long long r = min(max(a.idx[i][k], 0ll), (long long)a.n[i] - 1);
    v[i] = *reinterpret_cast<const float4*>(a.tab[i] + (size_t)r * D + 4 * c4);
  }
  float4 o;
  o.x = v[0].x + ((v[1].x + v[2].x) + v[3].x);
  o.y = v[0].y + ((v[1].y + v[2].y) + v[3].y);
  o.z = v[0].z + ((v[1].z + v[2].z) + v[3].z);
  o.w = v[0].w + ((v[1].w + v[2].w) + v[3].w);
  *reinterpret_cast<float4*>(a.out + (size_t)k * D + 4 * c4) = o;
}

// k_insert_cat: categorical embedding / shape of the rows a sub-loop iteration appended (agent_decoder.py:1949-1950, :1993):
// cat_agent[new_row] = type_a_emb[type[new_row]] + shape_emb(new_shape)[s], shape_all[new_row] = new_shape[s]; and the new
// row's index inside its scene (the centre of the heading stage's edge search).  32 threads per scene.
__global__ __launch_bounds__(NT) void k_insert_cat(InsertCatArgs a) {
  const int gid = blockIdx.x * NT + threadIdx.x;
  const int s = gid >> 5, c4 = gid & 31;
  if (s >= a.S) return;
  const int row = a.new_row[s];
  if (c4 == 0) a.new_local[s] = min(max(row - s * a.A_cap, 0), a.A_cap - 1);
  if (!a.inserted[s] || a.inserted[s] < 0) return;
  const float4 te = *reinterpret_cast<const float4*>(a.type_emb + (size_t)a.type[row] * D + 4 * c4);
  const float4 se = *reinterpret_cast<const float4*>(a.shp + (size_t)s * D + 4 * c4);
  *reinterpret_cast<float4*>(a.cat_agent + (size_t)row * D + 4 * c4) = make_float4(te.x + se.x, te.y + se.y, te.z + se.z, te.w + se.w);
  if (c4 == 0) {
    a.shape_all[3 * (size_t)row] = a.new_shape[3 * s]; a.shape_all[3 * (size_t)row + 1] = a.new_shape[3 * s + 1];
    a.shape_all[3 * (size_t)row + 2] = a.new_shape[3 * s + 2];
  }
}

// ------------------------------------------------------------------------------------------
// Scenario insertion (reference agent_decoder.py:1773-2105)
// ------------------------------------------------------------------------------------------
// k_point_edges: _build_a2sa_edge / _build_map2sa_edge for ONE query point per scene (:760-904).
// wave 0: agents, wave 1: map tokens; two passes (count, reserve with one atomicAdd, write).
__global__ __launch_bounds__(128) void k_point_edges(PointEdgesArgs a) {
  const SceneState& st = a.st;
  const int s = blockIdx.x, lane = lane_id(), w = wave_id();
  const bool is_map = w == 1;
  if (!((a.which >> (is_map ? 1 : 0)) & 1)) return;
  EdgeBuf& eb = is_map ? a.em : a.ea;
  const int c = a.c;
  const bool on = a.active == nullptr || a.active[s] != 0;
  const int A = st.n_agents[s];
  const int ms = st.map_scene ? st.map_scene[s] : s;          // the scene's slot in the map-side arrays
  const int N = is_map ? st.n_map[ms] : A;
  const int K = is_map ? a.k_map : a.k_agent;
  const float r = is_map ? a.r_map : a.r_agent;
  const float r2 = r * r;
  const int crow = a.centre_row[s];
  const size_t ic = sidx(st, s, c, crow);
  const float cx = st.pos[2 * ic], cy = st.pos[2 * ic + 1], ch = st.head[ic];
  const float ccs = cosf(ch), csn = sinf(ch);
  const float* mp = st.map_pos + (size_t)ms * st.M_cap * 2;
  const float* mo = st.map_orient + (size_t)ms * st.M_cap;
  int kept = 0;
  for (int pass = 0; pass < 2; ++pass) {
    int found = 0, written = 0, e0 = 0;
    if (pass == 1) {
      e0 = kept > 0 ? __shfl(lane == 0 ? atomicAdd(eb.total, kept) : 0, 0, 64) : 0;
      if (lane == 0) { eb.off[s] = e0; eb.cnt[s] = (e0 + kept <= eb.cap) ? kept : 0; }
      if (kept == 0 || e0 + kept > eb.cap) break;
    }
    if (on) {
      for (int j0 = 0; j0 < N && found < K; j0 += 64) {
        const int j = j0 + lane;
        bool in = false;
        float px = 0.f, py = 0.f;
        if (j < N) {
          if (is_map) { px = mp[2 * j]; py = mp[2 * j + 1]; }
          else { const size_t ij = sidx(st, s, c, j); px = st.pos[2 * ij]; py = st.pos[2 * ij + 1]; }
          const float dx = cx - px, dy = cy - py;
          in = (dx * dx + dy * dy) < r2;
        }
        const unsigned long long bal = __ballot(in);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        bool emit = in && (found + before < K);
        if (emit && !is_map) {
          emit = st.imask[sidx(st, s, c, j)] != 0 && !(a.exclude_centre && j == crow);
        }
        const unsigned long long ebal = __ballot(emit);
        if (pass == 1 && emit) {
          const int e = e0 + written + __popcll(ebal & ((1ull << lane) - 1ull));
          const float dx = px - cx, dy = py - cy;
          const float oh = is_map ? mo[j] : st.head[sidx(st, s, c, j)];
          eb.src[e] = (is_map ? ms * st.M_cap : s * st.A_cap) + j;
          *reinterpret_cast<float4*>(eb.raw + 4 * (size_t)e) =
              make_float4(norm2(dx, dy), angle_between(ccs, csn, dx, dy), wrap_angle(oh - ch), 0.f);
        }
        written += (int)__popcll(ebal);
        found += (int)__popcll(bal);
      }
    }
    if (pass == 0) kept = written;
  }
}

// k_occupancy: one-hot sum of the grid tokens of column c (:1852-1854)
__global__ __launch_bounds__(NT) void k_occupancy(OccupancyArgs a) {
  const SceneState& st = a.st;
  const int s = blockIdx.x;
  float* o = a.occ + (size_t)s * a.grid_size;
  for (int g = threadIdx.x; g < a.grid_size; g += NT) o[g] = 0.f;
  __syncthreads();
  const int A = st.n_agents[s];
  for (int ag = threadIdx.x; ag < A; ag += NT) {
    const int g = st.grid[sidx(st, s, a.c, ag)];
    if (g >= 0) o[g] = 1.f;
  }
}

// k_occupancy_embed: k_occupancy + seed_agent_occ_embed (MLPLayer 1961 -> 128 -> 128, agent_decoder.py:1852-1856) of the 0 / 1
// occupancy vector in one launch: the first Linear of a 0 / 1 input is the sum of the weight columns of the occupied cells
// (ascending cell order), then LayerNorm, ReLU and the 128 x 128 Linear per scene - instead of two dependent row-tile GEMM
// launches (K = 1961) over 16 CUs.  One workgroup (256 threads) per scene; pack = packing.pack_mlp_layer layout.
__global__ __launch_bounds__(NT) void k_occupancy_embed(OccEmbedArgs a) {
  __shared__ unsigned char occ_b[2048];
  __shared__ unsigned short cells[2048];
  __shared__ int n_cells;
  __shared__ float hv[128];
  __shared__ float red[8];
  const SceneState& st = a.st;
  const int s = blockIdx.x, tid = threadIdx.x;
  const int G = a.grid_size;
  float* o = a.occ + (size_t)s * G;
  for (int g = tid; g < 2048; g += NT) occ_b[g] = 0;
  __syncthreads();
  const int A = st.n_agents[s];
  for (int ag = tid; ag < A; ag += NT) {
    const int g = st.grid[sidx(st, s, a.c, ag)];
    if (g >= 0 && g < G) occ_b[g] = 1;
  }
  __syncthreads();
  for (int g = tid; g < G; g += NT) o[g] = occ_b[g] ? 1.f : 0.f;
  if (a.active && !a.active[s]) return;             // (uniform over the workgroup)
  // the occupied cells in ascending order (wave 0: ballot + prefix count per 64 cells): the sum below keeps the order of the
  // dense product and touches only the ~A occupied columns
  if (tid < 64) {
    int base = 0;
    for (int g0 = 0; g0 < G; g0 += 64) {
      const int g = g0 + tid;
      const bool f = g < G && occ_b[g];
      const unsigned long long m = __ballot(f);
      if (f) cells[base + __popcll(m & ((1ull << tid) - 1ull))] = (unsigned short)g;
      base += __popcll(m);
    }
    if (tid == 0) n_cells = base;
  }
  __syncthreads();
  const int k0p = (G + 7) / 8 * 8;
  const float* W0 = a.pack;                         // P(k0p, 128): element (k, n) at ((k >> 3) * 128 + n) * 8 + (k & 7)
  const float* b0 = a.pack + (size_t)k0p * 128;
  const float* lg = b0 + 128;
  const float* lb = b0 + 256;
  const float* W3 = b0 + 384;                       // P(128, 128)
  const float* b3 = W3 + 128 * 128;
  float h = 0.f;
  if (tid < 128) {
    const int n = n_cells;
    for (int i = 0; i < n; ++i) {
      const int g = cells[i];
      h += W0[((size_t)(g >> 3) * 128 + tid) * 8 + (g & 7)];
    }
    h += b0[tid];
  }
  // LayerNorm over the 128 values (threads 0..127 = waves 0, 1), biased variance, eps 1e-5
  float v = tid < 128 ? h : 0.f;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  const float mean = (red[0] + red[1]) * (1.0f / 128.0f);
  const float d = tid < 128 ? h - mean : 0.f;
  float q = d * d;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = q;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((red[4] + red[5]) * (1.0f / 128.0f) + LN_EPS);
  if (tid < 128) hv[tid] = fmaxf(d * rstd * lg[tid] + lb[tid], 0.f);
  __syncthreads();
  if (tid < 128) {
    float acc = 0.f;
    for (int k = 0; k < 128; ++k) acc += hv[k] * W3[((size_t)(k >> 3) * 128 + tid) * 8 + (k & 7)];
    a.emb[(size_t)s * 128 + tid] = acc + b3[tid];
  }
}

// k_insert_decide: heads of the seed node -> enter? / type / shape / grid cell; occupied-cell
// rejection; append the new row (:1883-1999).  One wave per scene.
__global__ __launch_bounds__(64) void k_insert_decide(InsertDecideArgs a) {
  const SceneState& st = a.st;
  const int s = blockIdx.x, lane = threadIdx.x;
  const int c = a.c;
  if (!a.active[s]) { if (lane == 0) a.inserted[s] = 0; return; }
  // the position head's cell: arg-max (softmax is monotone; first index on ties) or, with sample_k > 1, the reference's
  // softmax -> topk(insert_beam_size) -> multinomial (:1900-1904) in its reproducible form: the k largest in (value desc,
  // index asc) order, inverse CDF over their probabilities with the caller's uniform (like k_sample_topk)
  const float* lp = a.lg_pos + (size_t)s * a.grid_size;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  {
    const int k = a.sample_k > 1 ? min(a.sample_k, 16) : 1;
    float topv[16];
    int topi[16];
    float prev_v = INFINITY;
    int prev_i = -1;
    for (int j = 0; j < k; ++j) {
      best = -INFINITY; bi = 0x7fffffff;
      for (int g = lane; g < a.grid_size; g += 64) {
        const float v = lp[g];
        const bool after = (v < prev_v) || (v == prev_v && g > prev_i);
        if (after && (v > best || (v == best && g < bi))) { best = v; bi = g; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      topv[j] = best; topi[j] = bi;
      prev_v = best; prev_i = bi;
    }
    if (k > 1) {
      float p[16], sum = 0.f;
      for (int j = 0; j < k; ++j) { p[j] = expf(topv[j] - topv[0]); sum += p[j]; }
      const float u = a.uniform[s] * sum;
      float cdf = 0.f;
      int pick = k - 1;
      for (int j = 0; j < k; ++j) { cdf += p[j]; if (u < cdf) { pick = j; break; } }
      bi = topi[pick];
    } else {
      bi = topi[0];
    }
  }
  if (lane != 0) return;
  const int cell = bi;
  const float* ls = a.lg_state + 2 * s;
  bool enter = ls[1] > ls[0];
  if (a.force_enter) enter = true;
  const float* lt = a.lg_type + 3 * s;
  int ty = 0;
  if (lt[1] > lt[ty]) ty = 1;
  if (lt[2] > lt[ty]) ty = 2;
  const bool occupied = a.occ[(size_t)s * a.grid_size + cell] != 0.f;
  const int A = a.n_agents[s];
  // :1906-1909: an occupied cell `continue`s - the iteration is spent, nothing is appended and, when cells are sampled, the
  // next iteration draws again (greedy: it would pick the same cell until the iterations run out, so the scene stops)
  if (occupied && a.sample_k > 1) { a.inserted[s] = 0; return; }
  const bool ok = enter && !occupied && a.n_new[s] + 1 <= a.max_new;
  // the reference would append a row here; if the scene's row head-room is used up that is reported (-1), never dropped
  if (ok && A >= st.A_cap) { a.inserted[s] = -1; a.active[s] = 0; return; }
  if (!ok) { a.inserted[s] = 0; a.active[s] = 0; return; }
  const int av = st.av_index[s];
  const size_t ie = sidx(st, s, c, av);
  const float ex = st.pos[2 * ie], ey = st.pos[2 * ie + 1], eh = st.head[ie];
  // decode_pos (attr_tokenizer.py:91-99): grid[cell] @ Rot(theta_ego - pi/2) + ego pos
  const float phi = eh - HALF_PI_F;
  const float cs = cosf(phi), sn = sinf(phi);
  const float gx = a.grid_xy[2 * cell], gy = a.grid_xy[2 * cell + 1];
  const float nx = (gx * cs + gy * (-sn)) + ex;
  const float ny = (gx * sn + gy * cs) + ey;
  const int row = s * st.A_cap + A;
  for (int j = 0; j < st.T; ++j) {
    const size_t i = sidx(st, s, j, A);
    st.pos[2 * i] = 0.f; st.pos[2 * i + 1] = 0.f; st.head[i] = 0.f;
    st.state[i] = INVALID; st.token[i] = -1; st.grid[i] = -1;
    st.tmask[i] = 1; st.imask[i] = j >= c ? 1 : 0; st.catflag[i] = j >= c ? 1 : 0;
  }
  const size_t in_ = sidx(st, s, c, A);
  st.pos[2 * in_] = nx; st.pos[2 * in_ + 1] = ny; st.head[in_] = eh;
  st.state[in_] = ENTER; st.token[in_] = -2; st.grid[in_] = cell;
  a.type[row] = ty;
  st.bos[row] = c;
  a.new_shape[3 * s] = a.shape[3 * s]; a.new_shape[3 * s + 1] = a.shape[3 * s + 1]; a.new_shape[3 * s + 2] = a.shape[3 * s + 2];
  a.new_cell[s] = cell;
  if (a.t > 0) {
    for (int k = 0; k < 5; ++k) {
      const size_t o = (size_t)row * a.R + (a.t - 1) * 5 + k;
      a.pred_traj[2 * o] = nx; a.pred_traj[2 * o + 1] = ny;
      a.pred_head[o] = eh; a.pred_state[o] = (float)ENTER;
    }
  }
  a.n_agents[s] = A + 1;
  a.n_new[s] += 1;
  a.new_row[s] = row;
  a.inserted[s] = 1;
}

// k_insert_finalize: heading token + xy offset of the new row (:2060-2074), head-vector override
__global__ __launch_bounds__(64) void k_insert_finalize(InsertFinalizeArgs a) {
  const SceneState& st = a.st;
  const int s = blockIdx.x;
  if (threadIdx.x != 0 || !a.inserted[s]) return;
  const int row = a.new_row[s];
  const int ag = row - s * st.A_cap;
  const float* lh = a.lg_heading + (size_t)s * a.n_heading;
  int bi = 0;
  for (int k = 1; k < a.n_heading; ++k) if (lh[k] > lh[bi]) bi = k;
  const size_t ie = sidx(st, s, a.c, st.av_index[s]);
  const float eh = st.head[ie];
  // decode_heading (attr_tokenizer.py:106-110): (idx * interval - 180) / 360 * 2 pi
  const float dec = ((float)bi * a.angle_interval - 180.0f) / 360.0f * TWO_PI_F;
  const float nh = wrap_angle(dec + eh);
  const size_t in_ = sidx(st, s, a.c, ag);
  st.head[in_] = nh;
  st.pos[2 * in_] += tanhf(a.offset[2 * s]) * 2.0f;
  st.pos[2 * in_ + 1] += tanhf(a.offset[2 * s + 1]) * 2.0f;
  a.hv_ovr[2 * s] = cosf(nh);
  a.hv_ovr[2 * s + 1] = sinf(nh);
}


// ------------------------------------------------------------------------------------------
// k_sample_topk: reproducible stand-in for the reference's stochastic decode
// (agent_decoder.py:2162-2163,2194-2195: softmax -> topk(motion_beam_size) -> torch.multinomial):
// the k most probable tokens in descending order, then inverse-CDF sampling over their
// (re-normalised) probabilities with a caller-supplied uniform.  One wave per row.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_sample_topk(SampleArgs a) {
  const int row = blockIdx.x * 4 + wave_id();
  if (row >= a.rows) return;
  const int lane = lane_id();
  const float* lg = a.logits + (size_t)row * a.n;
  float topv[16];
  int topi[16];
  float prev_v = INFINITY;
  int prev_i = -1;
  for (int j = 0; j < a.k; ++j) {
    // j-th largest: the largest element strictly after (prev_v, prev_i) in (value desc, index asc) order
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < a.n; c += 64) {
      const float v = lg[c];
      const bool after = (v < prev_v) || (v == prev_v && c > prev_i);
      if (after && (v > best || (v == best && c < bi))) { best = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    topv[j] = best; topi[j] = bi;
    prev_v = best; prev_i = bi;
  }
  if (lane != 0) return;
  float p[16], sum = 0.f;
  for (int j = 0; j < a.k; ++j) { p[j] = expf(topv[j] - topv[0]); sum += p[j]; }
  const float u = a.uniform[row] * sum;
  float cdf = 0.f;
  int pick = a.k - 1;
  for (int j = 0; j < a.k; ++j) {
    cdf += p[j];
    if (u < cdf) { pick = j; break; }
  }
  a.token[row] = topi[pick];
}

}  // namespace ig
