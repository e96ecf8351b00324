// Edge builders of the teacher-forced forward (reference infgen/modules/agent_decoder.py:1104-1603).
//
// The forward builds seven edge sets over all (row, column) nodes of a batch - temporal (:540-610), agent <-> agent (:612-681),
// agent -> seed (:760-849 mode 'insert'), map -> agent (:683-758), map -> seed (:851-904 mode 'insert'), and the refine stage's
// agent -> candidate / map -> candidate (mode 'feature') - plus the map encoder's token graph (map_decoder.py:91-114).  All of
// them are torch_cluster.radius semantics on some candidate range with an emit filter, so ONE kernel serves them:
//
// k_radius_edges: one wavefront per destination ("query").  Candidates are a contiguous index range of the candidate point
//   arrays, visited in ascending index; the first K with d^2 < r^2 (strict, fp32 dx*dx + dy*dy) count as found (torch_cluster's
//   order, tests/golden/_standins.py) whether or not they pass the emit filter (self, per-candidate flag, per-pair flag) - the
//   reference filters AFTER the radius call.  Two passes (count, fill) give a compact CSR by destination: one atomicAdd per
//   workgroup reserves its range.  raw = (|d|, angle(hv[dst], d), wrap(head[src] - head[dst]), candidate index - query index)
//   with the reference's gap / invalid overrides on both components of d.
#include "kernels.h"
#include "tile.cuh"

namespace ig {

constexpr int INVALID = 0, ENTER = 2;
constexpr float MOTION_GAP = 1.0f, HEADING_GAP = 1.0f, INVALID_MOTION = -2.0f, INVALID_HEAD = -2.0f;

__global__ __launch_bounds__(256) void k_radius_edges(RadiusEdgesArgs a) {
  __shared__ int wcnt[4];
  __shared__ int wbase[4];
  const int lane = lane_id(), w = wave_id();
  const int q = blockIdx.x * 4 + w;
  const bool live = q < a.n_q;
  int c0 = 0, c1 = 0, self = -1, qp = 0, node = 0;
  float qx = 0.f, qy = 0.f, qh = 0.f;
  bool q_inv = false;
  const unsigned char* pair = nullptr;
  if (live) {
    node = a.q_node[q]; qp = a.q_pt[q]; c0 = a.q_c0[q]; c1 = a.q_c1[q];
    qx = a.p_pos[2 * (size_t)qp]; qy = a.p_pos[2 * (size_t)qp + 1]; qh = a.p_head[qp];
    q_inv = a.p_inv ? a.p_inv[qp] != 0 : false;
    if (a.q_self) self = a.q_self[q];
    if (a.pair_ok && a.q_pair_off && a.q_pair_off[q] >= 0) pair = a.pair_ok + a.q_pair_off[q];
  }
  const float r2 = a.radius * a.radius;
  const unsigned long long lt = (1ull << lane) - 1ull;
  auto scan = [&](bool fill, int e0, float hc, float hs) {
    int found = 0, written = 0;
    for (int m0 = c0; m0 < c1 && found < a.K; m0 += 64) {
      const int m = m0 + lane;
      bool in = false;
      float cx = 0.f, cy = 0.f;
      if (m < c1) {
        cx = a.c_pos[2 * (size_t)m]; cy = a.c_pos[2 * (size_t)m + 1];
        const float dx = qx - cx, dy = qy - cy;
        in = (dx * dx + dy * dy) < r2;
      }
      const unsigned long long bal = __ballot(in);
      const int before = __popcll(bal & lt);
      bool emit = in && (found + before < a.K) && (m != self);
      if (emit && a.c_ok) emit = a.c_ok[m] != 0;
      if (emit && pair) emit = pair[m - c0] != 0;
      const unsigned long long ebal = __ballot(emit);
      if (fill && emit) {
        const int e = e0 + written + __popcll(ebal & lt);
        float dx = cx - qx, dy = cy - qy;
        float dth = wrap_angle(a.c_head[m] - qh);
        const bool s_inv = a.c_inv ? a.c_inv[m] != 0 : false;
        if (a.gap_rule == 1) {                       // agent_decoder.py:595-601, :647-653 (:598 / :650 are no-ops)
          if (s_inv && !q_inv) { dx = -MOTION_GAP; dy = -MOTION_GAP; dth = -HEADING_GAP; }
          if (!s_inv && q_inv) { dx = MOTION_GAP; dy = MOTION_GAP; }
          if (s_inv && q_inv) { dx = INVALID_MOTION; dy = INVALID_MOTION; dth = INVALID_HEAD; }
        } else if (a.gap_rule == 2) {                // :722-723
          if (q_inv) { dx = MOTION_GAP; dy = MOTION_GAP; dth = HEADING_GAP; }
        }
        a.e.src[e] = a.c_src ? a.c_src[m] : m;
        *reinterpret_cast<float4*>(a.e.raw + 4 * (size_t)e) =
            make_float4(norm2(dx, dy), angle_between(hc, hs, dx, dy), dth, a.index_diff ? (float)(m - qp) : 0.f);
      }
      written += (int)__popcll(ebal);
      found += (int)__popcll(bal);
    }
    return written;
  };
  const int kept = live ? scan(false, 0, 0.f, 0.f) : 0;
  if (lane == 0) wcnt[w] = kept;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    const int base = tot > 0 ? atomicAdd(a.e.total, tot) : 0;
    wbase[0] = base; wbase[1] = base + wcnt[0]; wbase[2] = wbase[1] + wcnt[1]; wbase[3] = wbase[2] + wcnt[2];
  }
  __syncthreads();
  if (!live) return;
  const int e0 = wbase[w];
  const bool fits = e0 + kept <= a.e.cap;                 // overflow: total > cap is reported by the host
  if (lane == 0) { a.e.off[node] = a.e_base + e0; a.e.cnt[node] = fits ? kept : 0; }
  if (kept == 0 || !fits) return;
  scan(true, e0, cosf(qh), sinf(qh));
}

// k_motion_features: the two continuous inputs of x_a_emb for every (row, column) of agent-major arrays
// (_build_vector_a + _build_agent_feature, agent_decoder.py:426-447, :480-484): |motion vector|, angle(head vector, motion
// vector) with the invalid / gap rules on both components; `gap_mask` (refine stage, :1327) forces the gap value.
__global__ __launch_bounds__(256) void k_motion_features(MotionFeatArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.rows * a.T) return;
  const int t = i % a.T;
  float mx = 0.f, my = 0.f;
  if (t > 0) { mx = a.pos[2 * (size_t)i] - a.pos[2 * (size_t)(i - 1)]; my = a.pos[2 * (size_t)i + 1] - a.pos[2 * (size_t)(i - 1) + 1]; }
  const int st = a.state[i];
  const bool inv = st == INVALID;
  if (inv) { mx = INVALID_MOTION; my = INVALID_MOTION; }
  const bool prev_inv = t > 0 ? a.state[i - 1] == INVALID : false;
  const bool last_inv = t > 0 ? (prev_inv && !inv) : (st == ENTER);
  if (last_inv) { mx = MOTION_GAP; my = MOTION_GAP; }
  if (t > 0 && !prev_inv && inv) { mx = -MOTION_GAP; my = -MOTION_GAP; }
  if (a.gap_mask && a.gap_mask[i]) { mx = MOTION_GAP; my = MOTION_GAP; }
  const float h = a.head[i];
  *reinterpret_cast<float4*>(a.out + 4 * (size_t)i) = make_float4(norm2(mx, my), angle_between(cosf(h), sinf(h), mx, my), 0.f, 0.f);
}

}  // namespace ig
