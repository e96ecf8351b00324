// Rollout-metric feature on the device: compute_distance_to_nearest_object
// (reference infgen/metrics/interact_features.py:19-95 with box_utils.py:77-113 and geometry_utils.py:10-129):
// signed distance of every evaluated object to the nearest other valid object at every step, boxes with rounded
// corners (shrink by 0.7 * min(l, w) / 2, measure, subtract both shrink radii).  Distance between two convex boxes =
// signed distance of the origin to the Minkowski sum of box 1 and the negated box 2 (8 vertices, merged by edge angle).
//   k_box_corners        one thread per (object, step): shrunk corners (+l,+w) (-l,+w) (-l,-w) (+l,-w) rotated, + shrink
//   k_nearest_distance   one thread per (evaluated object, step): loop over all objects of the scene
// Objects are expected "evaluated first" (the reference concatenates them in that order, :48-50); the self pair is the
// one with equal index.
#include "kernels.h"

namespace ig {

constexpr float MT_BIG = 1e10f;

__global__ __launch_bounds__(256) void k_box_corners(NearestArgs a) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)a.B * a.N * a.T) return;
  const float l = a.length[i], w = a.width[i], h = a.heading[i];
  const float sh = fminf(l, w) * a.rounding / 2.0f;
  const float l2 = (l - 2.0f * sh) * 0.5f, w2 = (w - 2.0f * sh) * 0.5f;
  const float c = cosf(h), s = sinf(h);
  const float lx[4] = {l2, -l2, -l2, l2}, ly[4] = {w2, w2, -w2, -w2};
  float* o = a.work + i * 9;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[2 * k] = (c * lx[k] - s * ly[k]) + a.cx[i];
    o[2 * k + 1] = (s * lx[k] + c * ly[k]) + a.cy[i];
  }
  o[8] = sh;
}

struct Box { float x[4], y[4]; };

__device__ __forceinline__ int downmost(const Box& b, float& dx, float& dy) {
  int i0 = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k) if (b.y[k] < b.y[i0]) i0 = k;          // first minimum
  const int i1 = (i0 + 1) & 3;
  const float ex = b.x[i1] - b.x[i0], ey = b.y[i1] - b.y[i0];
  const float len = sqrtf(ex * ex + ey * ey);
  dx = ex / len; dy = ey / len;
  return i0;
}

__device__ __forceinline__ float pair_distance(const Box& b1, const Box& b2n) {
  float d1x, d1y, d2x, d2y;
  const int s1 = downmost(b1, d1x, d1y), s2 = downmost(b2n, d2x, d2y);
  const bool cond = (d1x * d2y - d1y * d2x) >= 0.0f;
  float px[8], py[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int oa = k >> 1, ob = ((k + 1) >> 1) & 3;                   // orders [0,0,1,1,2,2,3,3] and [0,1,1,2,2,3,3,0]
    const int i1 = ((cond ? ob : oa) + s1) & 3, i2 = ((cond ? oa : ob) + s2) & 3;
    px[k] = b1.x[i1] + b2n.x[i2];
    py[k] = b1.y[i1] + b2n.y[i2];
  }
  // signed distance of the origin to the convex polygon p[0..7]
  float best = INFINITY;
  bool inside = true;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int kn = (k + 1) & 7;
    const float ex = px[kn] - px[k], ey = py[kn] - py[k];
    const float len = sqrtf(ex * ex + ey * ey);
    const float tx = ex / (len + 1.1920929e-07f), ty = ey / (len + 1.1920929e-07f);
    const float vx = -px[k], vy = -py[k];
    const float perp = (ty * vx) + (-tx * vy);                        // sum(-normal * v), normal = (-ty, tx)
    inside = inside && (perp <= 0.0f);
    const float prop = (tx * vx + ty * vy) / len;
    if (prop >= 0.0f && prop <= 1.0f) best = fminf(best, fabsf(perp));
    best = fminf(best, sqrtf(vx * vx + vy * vy));
  }
  return inside ? -best : best;
}

__global__ __launch_bounds__(128) void k_nearest_distance(NearestArgs a) {
  const int idx = blockIdx.x * 128 + threadIdx.x;
  if (idx >= a.B * a.n_eval * a.T) return;
  const int t = idx % a.T, e = (idx / a.T) % a.n_eval, b = idx / (a.T * a.n_eval);
  const size_t base = (size_t)b * a.N * a.T;
  const size_t ie = base + (size_t)e * a.T + t;
  float out = MT_BIG;
  if (a.valid[ie]) {
    Box b1;
    const float* w1 = a.work + ie * 9;
#pragma unroll
    for (int k = 0; k < 4; ++k) { b1.x[k] = w1[2 * k]; b1.y[k] = w1[2 * k + 1]; }
    const float sh1 = w1[8];
    for (int j = 0; j < a.N; ++j) {
      const size_t ij = base + (size_t)j * a.T + t;
      if (j == e || !a.valid[ij]) continue;
      const float* w2 = a.work + ij * 9;
      Box b2;
#pragma unroll
      for (int k = 0; k < 4; ++k) { b2.x[k] = -1.0f * w2[2 * k]; b2.y[k] = -1.0f * w2[2 * k + 1]; }
      const float d = (pair_distance(b1, b2) - sh1) - w2[8];
      out = fminf(out, d);
    }
  }
  a.out[((size_t)b * a.n_eval + e) * a.T + t] = out;
}

// ------------------------------------------------------------------------------------------
// k_kinematic: compute_kinematic_features (reference infgen/metrics/trajectory_features.py:37-51): central differences
// over steps, NaN at both ends; one thread per (object, step).  out: speed, accel, yaw rate, yaw accel, each [n][T].
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wrap_pm_pi(float a) {
  const float two_pi = 6.283185307179586f, pi = 3.141592653589793f;
  float m = fmodf(a + pi, two_pi);                 // python '%': result takes the sign of the divisor
  if (m < 0.f) m += two_pi;
  return m - pi;
}
__device__ __forceinline__ float speed_at(const KinematicArgs& a, size_t row, int t) {
  if (t < 1 || t > a.T - 2) return NAN;
  const float dx = (a.x[row + t + 1] - a.x[row + t - 1]) / 2.f, dy = (a.y[row + t + 1] - a.y[row + t - 1]) / 2.f;
  const float dz = a.z ? (a.z[row + t + 1] - a.z[row + t - 1]) / 2.f : 0.f;
  return sqrtf(dx * dx + dy * dy + dz * dz) / a.dt;
}
__device__ __forceinline__ float dh_step_at(const KinematicArgs& a, size_t row, int t) {
  if (t < 1 || t > a.T - 2) return NAN;
  return wrap_pm_pi((a.heading[row + t + 1] - a.heading[row + t - 1]) / 2.f * 2.f) / 2.f;
}
__global__ __launch_bounds__(256) void k_kinematic(KinematicArgs a) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)a.n * a.T) return;
  const int t = (int)(i % a.T);
  const size_t row = i - t;
  const bool inner = t >= 1 && t <= a.T - 2;
  a.speed[i] = speed_at(a, row, t);
  if (a.accel) a.accel[i] = inner ? (speed_at(a, row, t + 1) - speed_at(a, row, t - 1)) / 2.f / a.dt : NAN;
  if (a.yaw_rate) a.yaw_rate[i] = dh_step_at(a, row, t) / a.dt;
  if (a.yaw_accel)
    a.yaw_accel[i] = inner ? wrap_pm_pi((dh_step_at(a, row, t + 1) - dh_step_at(a, row, t - 1)) / 2.f * 2.f) / 2.f / (a.dt * a.dt) : NAN;
}

// ------------------------------------------------------------------------------------------
// k_ttc: compute_time_to_collision_with_object_in_front (reference infgen/metrics/interact_features.py:96-219): per
// (evaluated object, step) the nearest valid object it follows (ahead, laterally overlapping its trail, aligned) and
// distance / closing speed, capped at 5 s.  eval_idx lists the evaluated objects in their original order.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_ttc(TtcArgs a) {
  const int idx = blockIdx.x * 128 + threadIdx.x;
  if (idx >= a.B * a.n_eval * a.T) return;
  const int t = idx % a.T, e = (idx / a.T) % a.n_eval, b = idx / (a.T * a.n_eval);
  const size_t base = (size_t)b * a.N * a.T;
  const size_t ie = base + (size_t)a.eval_idx[e] * a.T + t;
  const float ex = a.cx[ie], ey = a.cy[ie], el = a.length[ie], ew = a.width[ie], eh = a.heading[ie], es = a.speed[ie];
  const float ce = cosf(-eh), se = sinf(-eh);
  const float max_diff = 1.3089969389957472f, max_diff_small = 0.17453292519943295f;     // 75 and 10 degrees
  float best = INFINITY, best_speed = 0.f;
  for (int j = 0; j < a.N; ++j) {
    const size_t ij = base + (size_t)j * a.T + t;
    if (!a.valid[ij]) continue;
    const float yd = fabsf(a.heading[ij] - eh);
    const float c = fabsf(cosf(yd)), s = fabsf(sinf(yd));
    const float hl = a.length[ij] / 2.0f, hw = a.width[ij] / 2.0f;
    const float long_off = hl * c + hw * s, lat_off = hl * s + hw * c;
    const float dx = a.cx[ij] - ex, dy = a.cy[ij] - ey;
    const float rx = ce * dx - se * dy, ry = se * dx + ce * dy;
    const float long_d = rx - el / 2.0f - long_off;
    const float lat_o = fabsf(ry) - ew / 2.0f - lat_off;
    const bool follow = long_d > 0.0f && yd <= max_diff && lat_o < 0.0f && (lat_o < -0.5f || yd <= max_diff_small);
    if (follow && long_d < best) { best = long_d; best_speed = a.speed[ij]; }      // first minimum
  }
  float ttc = 5.0f;
  if (best < INFINITY) {
    const float rel = es - best_speed;
    if (rel > 0.0f) ttc = fminf(best / rel, 5.0f);
  }
  a.out[((size_t)b * a.n_eval + e) * a.T + t] = ttc;
}

// ------------------------------------------------------------------------------------------
// k_placement: compute_num_placement + compute_distance_placement (reference infgen/metrics/placement_features.py:6-48):
// per step the number of agents entering / leaving (ego excluded) and per agent its distance to the ego where it does.
// One workgroup per (scene, step): threads over agents, counts reduced with wave ballots.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_placement(PlacementArgs a) {
  __shared__ int s_b[4], s_e[4];
  const int t = blockIdx.x % a.T, b = blockIdx.x / a.T;
  const size_t base = (size_t)b * a.N * a.T;
  const int av = a.av_index[b];
  const size_t iav = base + (size_t)av * a.T + t;
  const float ax = a.x[iav], ay = a.y[iav], az = a.z ? a.z[iav] : 0.f;
  int nb = 0, ne = 0;
  for (int n0 = 0; n0 < a.N; n0 += 256) {
    const int n = n0 + threadIdx.x;
    bool bos = false, eos = false;
    if (n < a.N) {
      const size_t i = base + (size_t)n * a.T + t;
      const int st = (n == av) ? -1 : a.state[i];
      bos = st == a.enter_state; eos = st == a.exit_state;
      const float dx = a.x[i] - ax, dy = a.y[i] - ay, dz = a.z ? a.z[i] - az : 0.f;
      const float d = sqrtf(dx * dx + dy * dy + dz * dz);
      a.bos_distance[i] = bos ? d : 0.f * d;                 // the reference multiplies by the mask (NaN stays NaN)
      a.eos_distance[i] = eos ? d : 0.f * d;
    }
    nb += __popcll(__ballot(bos));
    ne += __popcll(__ballot(eos));
  }
  if ((threadIdx.x & 63) == 0) { s_b[threadIdx.x >> 6] = nb; s_e[threadIdx.x >> 6] = ne; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a.num_bos[(size_t)b * a.T + t] = s_b[0] + s_b[1] + s_b[2] + s_b[3];
    a.num_eos[(size_t)b * a.T + t] = s_e[0] + s_e[1] + s_e[2] + s_e[3];
  }
}

// ------------------------------------------------------------------------------------------
// k_road_edge: compute_distance_to_road_edge (reference infgen/metrics/map_features.py:27-79, :139-349): per evaluated box
// and step the largest signed distance of its four bottom corners to the road edges; each corner takes the segment that
// is nearest in the z-stretched 3-D metric (first minimum over the flattened padded segment list) and signs the planar
// distance by the side of that segment, corrected at the ends by the neighbouring segment and the local convexity.
// One wave per box: lanes stride over the segments of the scene, wave arg-min, lanes 0..3 finish one corner each.
// ------------------------------------------------------------------------------------------
struct SegEval { float t, side, d2, d3; bool ok; float dx, dy; };

__device__ __forceinline__ SegEval seg_eval(const float4* __restrict__ poly, int p, int j, int L, float qx, float qy, float qz,
                                            float zs) {
  const float4 A = poly[(size_t)p * L + j], B = poly[(size_t)p * L + j + 1];
  SegEval r;
  r.ok = A.w != 0.f && B.w != 0.f;
  const float dx = B.x - A.x, dy = B.y - A.y, dz = B.z - A.z;
  const float wx = qx - A.x, wy = qy - A.y, wz = qz - A.z;
  const float den = dx * dx + dy * dy, num = wx * dx + wy * dy;
  r.t = den != 0.f ? num / den : 0.f;
  const float cr = wx * dy - wy * dx;
  r.side = (float)((cr > 0.f) - (cr < 0.f));
  const float tc = fminf(fmaxf(r.t, 0.f), 1.f);
  const float fx = wx - dx * tc, fy = wy - dy * tc, fz = (wz - dz * tc) * zs;
  const float pl = fx * fx + fy * fy;
  r.d2 = r.ok ? sqrtf(pl) : 1e10f;
  r.d3 = r.ok ? sqrtf(pl + fz * fz) : 1e10f;
  r.dx = dx; r.dy = dy;
  return r;
}

__global__ __launch_bounds__(256) void k_road_edge(RoadEdgeArgs a) {
  const int box = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (box >= a.B * a.n_eval * a.T) return;
  const int t = box % a.T, e = (box / a.T) % a.n_eval, b = box / (a.T * a.n_eval);
  const size_t i = ((size_t)b * a.N + a.eval_idx[b * a.n_eval + e]) * a.T + t;
  if (!a.valid[i]) { if (lane == 0) a.out[box] = -1e10f; return; }
  const float c = cosf(a.heading[i]), s = sinf(a.heading[i]);
  const float hl = 0.5f * a.length[i], hw = 0.5f * a.width[i];
  const float qz = (a.cz ? a.cz[i] : 0.f) - 0.5f * (a.height ? a.height[i] : 0.f);
  float qx[4], qy[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {                               // (+l,+w) (-l,+w) (-l,-w) (+l,-w)
    const float sl = (k == 0 || k == 3) ? hl : -hl, sw = (k < 2) ? hw : -hw;
    qx[k] = a.cx[i] + (c * sl - s * sw);
    qy[k] = a.cy[i] + (s * sl + c * sw);
  }
  const float4* poly = reinterpret_cast<const float4*>(a.poly);
  const int p0 = a.poly_off[b], S = a.L - 1, nseg = (a.poly_off[b + 1] - p0) * S;
  float best[4]; int bi[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { best[k] = INFINITY; bi[k] = 0x7fffffff; }
  for (int k = lane; k < nseg; k += 64) {
    const int p = p0 + k / S, j = k % S;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const SegEval r = seg_eval(poly, p, j, a.L, qx[q], qy[q], qz, a.z_stretch);
      if (r.d3 < best[q]) { best[q] = r.d3; bi[q] = k; }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int off = 32; off; off >>= 1) {
      const float ob = __shfl_xor(best[q], off);
      const int oi = __shfl_xor(bi[q], off);
      if (ob < best[q] || (ob == best[q] && oi < bi[q])) { best[q] = ob; bi[q] = oi; }
    }
  float res = -INFINITY;
  if (lane < 4 && nseg > 0) {
    const int q = lane;
    const float x = q == 0 ? qx[0] : q == 1 ? qx[1] : q == 2 ? qx[2] : qx[3];
    const float y = q == 0 ? qy[0] : q == 1 ? qy[1] : q == 2 ? qy[2] : qy[3];
    const int k = q == 0 ? bi[0] : q == 1 ? bi[1] : q == 2 ? bi[2] : bi[3];
    const int p = p0 + k / S, j = k % S;
    const bool cyc = a.cyclic[p] != 0;
    const int jw_prev = (j + S - 1) % S, jw_next = (j + 1) % S;               // convexity always wraps (padded list)
    const int jp = j > 0 ? j - 1 : (cyc ? S - 1 : 0), jn = j < S - 1 ? j + 1 : (cyc ? 0 : S - 1);
    const SegEval r = seg_eval(poly, p, j, a.L, x, y, qz, a.z_stretch);
    const SegEval rp = seg_eval(poly, p, jp, a.L, x, y, qz, a.z_stretch);
    const SegEval rn = seg_eval(poly, p, jn, a.L, x, y, qz, a.z_stretch);
    const SegEval wp = seg_eval(poly, p, jw_prev, a.L, x, y, qz, a.z_stretch);
    const SegEval wn = seg_eval(poly, p, jw_next, a.L, x, y, qz, a.z_stretch);
    const bool convex_in = wp.dx * r.dy - wp.dy * r.dx > 0.f, convex_out = r.dx * wn.dy - r.dy * wn.dx > 0.f;
    float sg = r.side;
    if (r.t < 0.f && rp.ok) sg = convex_in ? fmaxf(r.side, rp.side) : fminf(r.side, rp.side);
    else if (r.t > 1.f && rn.ok) sg = convex_out ? fmaxf(r.side, rn.side) : fminf(r.side, rn.side);
    res = sg * r.d2;
  }
  res = fmaxf(res, __shfl_xor(res, 1));
  res = fmaxf(res, __shfl_xor(res, 2));
  if (lane == 0) a.out[box] = res;
}

// ------------------------------------------------------------------------------------------
// k_window_loglik: the scoring step of LongMetric (reference infgen/metrics/compute_metrics.py:845-878 with :744-762) fused
// over the unfolded windows: a value is scored by the log-probability (under the logged distribution) of the histogram
// bin it falls into - edges[i] <= v < edges[i + 1], the last bin closed on the right, anything else (or NaN) bin 0, like
// torch.histogram followed by argmax -; per (row, window of `size` steps every `step`) the sum over the valid steps and
// their number.  One thread per (row, window); the <= 64 edges and log-probabilities sit in LDS.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_window_loglik(WindowLoglikArgs a) {
  __shared__ float edges[65], logp[64];
  for (int i = threadIdx.x; i <= a.nb; i += 256) edges[i] = a.edges[i];
  for (int i = threadIdx.x; i < a.nb; i += 256) logp[i] = a.logp[i];
  __syncthreads();
  const int W = (a.T - a.size) / a.step + 1;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.n * W) return;
  const int w = idx % W, r = idx / W;
  const float* v = a.values + (size_t)r * a.T + (size_t)w * a.step;
  const unsigned char* ok = a.valid ? a.valid + (size_t)r * a.T + (size_t)w * a.step : nullptr;
  float sum = 0.f;
  int cnt = 0;
  for (int k = 0; k < a.size; ++k) {
    if (ok && !ok[k]) continue;
    const float x = v[k];
    int b = 0;
    if (x >= edges[0] && x <= edges[a.nb]) {
      int lo = 0, hi = a.nb;                     // invariant: edges[lo] <= x < edges[hi] (or x == edges[nb])
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (x >= edges[mid]) lo = mid; else hi = mid; }
      b = lo;
    }
    sum += logp[b];
    ++cnt;
  }
  a.out_sum[idx] = sum;
  a.out_cnt[idx] = cnt;
}

}  // namespace ig
