// k_match_tokens: TokenProcessor._match_agent_token (reference infgen/datasets/preprocess.py:552-653, with
// cal_polygon_contour :24-54) - per agent, for every 0.5 s step: rotate / translate the last contour of all
// n_token motion tokens of the agent's type by the previous matched pose, pick the token whose four corners are
// closest (sum of corner distances, first minimum) to the agent's box at this step, then continue from the
// matched contour's pose (or from the logged pose where the step pair is not valid).
// One workgroup per agent: the 18 steps are sequential, the 2048 tokens of a step are spread over 256 threads
// (8 tokens each, held in registers for all steps - the vocabulary is read once); every thread carries the pose and
// repeats the uniform scalar work of a step, so a step costs one barrier.
// fp32 arithmetic restated from what torch executes on the CPU (checked bit for bit against torch ops):
//   bmm with K = 2:          w.x = fma(t.y, -sin, t.x * cos),  w.y = fma(t.y, cos, t.x * sin)
//   torch.norm(dim=-1):      sqrt(fma(dy, dy, dx * dx))
//   sum / mean over corners: sequential, mean = sum * 0.25
// cos / sin / atan2 are evaluated in fp64 and rounded once (correctly rounded fp32 results): the reference's sleef
// kernels are within 1 ulp of that and usually equal to it, so poses agree bit for bit on most steps and drift by
// ulps otherwise; a token can differ from the reference only where two tokens tie to within rounding.
#include "kernels.h"
#include "tile.cuh"

namespace ig {

constexpr int MT_THREADS = 256;
constexpr int MT_MAX_PER_THREAD = 8;     // n_token <= 2048

struct Contour { float x[4], y[4]; };

__device__ __forceinline__ float cos_cr(float x) { return (float)cos((double)x); }
__device__ __forceinline__ float sin_cr(float x) { return (float)sin((double)x); }

__device__ __forceinline__ Contour box_contour(float x, float y, float head, float width, float length) {
  const float hc = 0.5f * cos_cr(head), hs = 0.5f * sin_cr(head);
  const float lc = length * hc, ls = length * hs, wc = width * hc, ws = width * hs;
  Contour c;
  c.x[0] = (x + lc) - ws; c.y[0] = (y + ls) + wc;      // left front
  c.x[1] = (x + lc) + ws; c.y[1] = (y + ls) - wc;      // right front
  c.x[2] = (x - lc) + ws; c.y[2] = (y - ls) - wc;      // right back
  c.x[3] = (x - lc) - ws; c.y[3] = (y - ls) + wc;      // left back
  return c;
}

__global__ __launch_bounds__(MT_THREADS) void k_match_tokens(MatchTokensArgs a) {
  __shared__ float s_val[2][MT_THREADS / 64];
  __shared__ int s_idx[2][MT_THREADS / 64];
  const int ag = blockIdx.x, t = threadIdx.x;
  const float* V = a.tok + (a.type ? (size_t)a.type[ag] * a.n_token * 8 : (size_t)ag * a.tok_agent_stride);
  const int per = (a.n_token + MT_THREADS - 1) / MT_THREADS;
  float4 tk[MT_MAX_PER_THREAD][2];
#pragma unroll
  for (int j = 0; j < MT_MAX_PER_THREAD; ++j) {
    const int k = t + j * MT_THREADS;
    tk[j][0] = tk[j][1] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < per && k < a.n_token) {
      tk[j][0] = *reinterpret_cast<const float4*>(V + (size_t)k * 8);
      tk[j][1] = *reinterpret_cast<const float4*>(V + (size_t)k * 8 + 4);
    }
  }
  const float width = a.shape[2 * ag], length = a.shape[2 * ag + 1];
  const size_t row = (size_t)ag * a.T;
  // every thread carries the pose and repeats the (cheap, uniform) per-step scalar work: one barrier per step
  float ph = a.heading[row], px = a.pos[2 * row], py = a.pos[2 * row + 1];
  const int n_out = a.T / a.shift;
  for (int i = a.shift, o = 0; i < a.T; i += a.shift, ++o) {
    const float cs = cos_cr(ph), sn = sin_cr(ph);
    const float cxi = a.pos[2 * (row + i)], cyi = a.pos[2 * (row + i) + 1], chi = a.heading[row + i];
    const Contour cur = box_contour(cxi, cyi, chi, width, length);
    float best = INFINITY;
    int bidx = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < MT_MAX_PER_THREAD; ++j) {
      const int k = t + j * MT_THREADS;
      if (j < per && k < a.n_token) {
        const float tx[4] = {tk[j][0].x, tk[j][0].z, tk[j][1].x, tk[j][1].z};
        const float ty[4] = {tk[j][0].y, tk[j][0].w, tk[j][1].y, tk[j][1].w};
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float wx = __builtin_fmaf(ty[c], -sn, tx[c] * cs) + px;
          const float wy = __builtin_fmaf(ty[c], cs, tx[c] * sn) + py;
          const float dx = wx - cur.x[c], dy = wy - cur.y[c];
          const float n = sqrtf(__builtin_fmaf(dy, dy, dx * dx));
          sum = (c == 0) ? n : sum + n;
        }
        if (sum < best) { best = sum; bidx = k; }      // ascending k per thread: the first minimum stays
      }
    }
    // arg-min with the first-index tie rule: inside the wave by shuffles, across the four waves through LDS
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float v2 = __shfl_xor(best, off, 64);
      const int i2 = __shfl_xor(bidx, off, 64);
      if (v2 < best || (v2 == best && i2 < bidx)) { best = v2; bidx = i2; }
    }
    const int buf = o & 1;                               // double buffered: one barrier per step is enough
    if ((t & 63) == 0) { s_val[buf][t >> 6] = best; s_idx[buf][t >> 6] = bidx; }
    __syncthreads();
    best = s_val[buf][0]; bidx = s_idx[buf][0];
#pragma unroll
    for (int w = 1; w < MT_THREADS / 64; ++w) {
      const float v2 = s_val[buf][w];
      const int i2 = s_idx[buf][w];
      if (v2 < best || (v2 == best && i2 < bidx)) { best = v2; bidx = i2; }
    }
    const int k = bidx;
    const float4 p0 = *reinterpret_cast<const float4*>(V + (size_t)k * 8);
    const float4 p1 = *reinterpret_cast<const float4*>(V + (size_t)k * 8 + 4);
    const float tx[4] = {p0.x, p0.z, p1.x, p1.z}, ty[4] = {p0.y, p0.w, p1.y, p1.w};
    float wx[4], wy[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      wx[c] = __builtin_fmaf(ty[c], -sn, tx[c] * cs) + px;
      wy[c] = __builtin_fmaf(ty[c], cs, tx[c] * sn) + py;
    }
    if (t == 0) {
      a.token_index[(size_t)ag * n_out + o] = k;
      float* oc = a.token_contour + ((size_t)ag * n_out + o) * 8;
#pragma unroll
      for (int c = 0; c < 4; ++c) { oc[2 * c] = wx[c]; oc[2 * c + 1] = wy[c]; }
    }
    const bool ok = a.valid[row + i - a.shift] && a.valid[row + i];
    if (ok) {
      ph = (float)atan2((double)(wy[0] - wy[3]), (double)(wx[0] - wx[3]));
      px = (((wx[0] + wx[1]) + wx[2]) + wx[3]) * 0.25f;
      py = (((wy[0] + wy[1]) + wy[2]) + wy[3]) * 0.25f;
    } else {
      ph = chi; px = cxi; py = cyi;
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_match_map_tokens: InfGen.match_token_map's matching core (reference infgen/model/infgen.py:918-936): a
// three-point polyline piece, moved to its own frame, against the sample points of the n_token map tokens; sum of
// squared distances, first minimum.  One wave per piece, tokens spread over the lanes.
// Arithmetic as torch executes it on the CPU for these shapes (checked bit for bit): the (P,3,2) x (P,2,2) bmm is
// (dx * r00) + (dy * r10) WITHOUT fma; the six squared differences e0..e5 are summed as ((((e0+e4)+e5)+e1)+e2)+e3.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MT_THREADS) void k_match_map_tokens(MatchMapArgs a) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (p >= a.P) return;
  const float th = a.theta[p];
  const float cs = cos_cr(th), sn = sin_cr(th);
  const float* tp = a.traj_pos + (size_t)p * 6;
  float lx[3], ly[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float dx = tp[2 * i] - tp[0], dy = tp[2 * i + 1] - tp[1];
    lx[i] = (dx * cs) + (dy * sn);
    ly[i] = (dx * -sn) + (dy * cs);
  }
  float best = INFINITY;
  int bidx = 0x7fffffff;
  for (int k = lane; k < a.n_token; k += 64) {
    const float* sp = a.sample_pt + (size_t)k * 6;
    float e[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float ex = sp[2 * i] - lx[i], ey = sp[2 * i + 1] - ly[i];
      e[2 * i] = ex * ex; e[2 * i + 1] = ey * ey;
    }
    const float d = ((((e[0] + e[4]) + e[5]) + e[1]) + e[2]) + e[3];
    if (d < best) { best = d; bidx = k; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float v2 = __shfl_xor(best, off, 64);
    const int i2 = __shfl_xor(bidx, off, 64);
    if (v2 < best || (v2 == best && i2 < bidx)) { best = v2; bidx = i2; }
  }
  if (lane == 0) a.token_idx[p] = bidx;
}

// ------------------------------------------------------------------------------------------
// k_tokenize_prep / k_tokenize_state: the bookkeeping of TokenProcessor._tokenize_agent around the contour matching
// (reference infgen/datasets/preprocess.py:335-550).  One thread per agent; both are sequential scans over a track.
//   prep   clean_heading (:310-317: a heading that jumps by more than 1.5 rad between two valid steps keeps the previous
//          value), _extrapolate_agent_to_prev_token_step (:319-344: constant-velocity steps back to the token grid), the
//          per-type (width, length) of _get_agent_shape (:346-354)
//   state  token validity (both ends of a 0.5 s window valid), enter / exit / invalid states (:425-434), token position
//          and heading from the matched contour, the position of entering agents, token ids -1 / -2, the shape reset
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_tokenize_prep(TokenizeArgs a) {
  const int ag = blockIdx.x * 64 + threadIdx.x;
  if (ag >= a.A) return;
  unsigned char* valid = a.valid + (size_t)ag * a.T;
  float* head = a.heading + (size_t)ag * a.T;
  float2* pos = reinterpret_cast<float2*>(a.pos) + (size_t)ag * a.T;
  float2* vel = reinterpret_cast<float2*>(a.velocity) + (size_t)ag * a.T;
  const float PI_F = 3.14159274f, TWO_PI_F = 6.28318548f;
  float prev = head[0];
  int first = -1;
  for (int i = 0; i + 1 < a.T; ++i) {
    if (first < 0 && valid[i]) first = i;
    const float next = head[i + 1];
    float keep = next;
    if (valid[i] && valid[i + 1]) {
      float r = fmodf(__fadd_rn(__fsub_rn(prev, next), PI_F), TWO_PI_F);
      if (r < 0.f) r += TWO_PI_F;
      if (fabsf(__fadd_rn(-PI_F, r)) > 1.5f) { keep = prev; head[i + 1] = prev; }
    }
    prev = keep;
  }
  if (first < 0 && valid[a.T - 1]) first = a.T - 1;
  const int ty = a.type[ag];
  a.wl[2 * ag] = ty == 0 ? 2.f : 1.f;
  a.wl[2 * ag + 1] = ty == 0 ? 4.8f : (ty == 1 ? 2.f : 1.f);
  if (first < 0) return;
  int n = first % a.shift;
  if (first == a.current_step && !valid[a.current_step - a.shift]) n = a.shift;
  const float2 v = vel[first];
  const float h = head[first];
  const float sx = __fmul_rn(v.x, 0.1f), sy = __fmul_rn(v.y, 0.1f);
  float2 p = pos[first];
  for (int j = 1; j <= n; ++j) {
    const int k = first - j;
    p.x = __fsub_rn(p.x, sx); p.y = __fsub_rn(p.y, sy);
    pos[k] = p; vel[k] = v; head[k] = h; valid[k] = 1;
  }
}

__global__ __launch_bounds__(64) void k_tokenize_state(TokenizeArgs a) {
  const int ag = blockIdx.x * 64 + threadIdx.x;
  if (ag >= a.A) return;
  const unsigned char* valid = a.valid + (size_t)ag * a.T;
  const float2* pos = reinterpret_cast<const float2*>(a.pos) + (size_t)ag * a.T;
  const int n_tok = a.T / a.shift;
  int first = -1, last = -1;
  for (int k = 0; k < n_tok; ++k)
    if (valid[k * a.shift] && valid[k * a.shift + a.shift]) { if (first < 0) first = k; last = k; }
  if (first < 0) { first = 0; last = n_tok - 1; }
  for (int k = 0; k < n_tok; ++k) {
    const size_t o = (size_t)ag * n_tok + k;
    int st = a.valid_state;
    if (k == first) st = a.enter_state;
    if (k == last) st = a.exit_state;
    if (k < first || k > last) st = a.invalid_state;
    if (k == n_tok - 1 && st == a.exit_state) st = a.valid_state;
    const bool tv = valid[k * a.shift] && valid[k * a.shift + a.shift] && st != a.enter_state;
    const float* c = a.token_contour + o * 8;
    float px = (((c[0] + c[2]) + c[4]) + c[6]) * 0.25f, py = (((c[1] + c[3]) + c[5]) + c[7]) * 0.25f;
    float ph = atan2f(c[1] - c[7], c[0] - c[6]);
    if (st == a.invalid_state) { px = 0.f; py = 0.f; ph = 0.f; }
    if (st == a.enter_state) { const float2 p = pos[(k + 1) * a.shift]; px = p.x; py = p.y; }
    a.state_idx[o] = st;
    a.token_pos[2 * o] = px; a.token_pos[2 * o + 1] = py;
    a.token_heading[o] = ph;
    if (st == a.invalid_state) a.token_index[o] = -1;
    if (st == a.enter_state) a.token_index[o] = -2;
    a.raw_token_valid[o] = tv;
    a.token_valid[o] = a.predict_state ? 1 : tv;
  }
  if (a.shape_in) {                                          // every step takes the first fully non-zero (l, w, h) row
    const float* s = a.shape_in + (size_t)ag * a.T * 3;
    int k = 0;
    while (k < a.T && !(s[3 * k] != 0.f && s[3 * k + 1] != 0.f && s[3 * k + 2] != 0.f)) ++k;
    const float l = k < a.T ? s[3 * k] : 0.f, w = k < a.T ? s[3 * k + 1] : 0.f, h = k < a.T ? s[3 * k + 2] : 0.f;
    float* d = a.shape_out + (size_t)ag * a.T * 3;
    for (int t = 0; t < a.T; ++t) { d[3 * t] = l; d[3 * t + 1] = w; d[3 * t + 2] = h; }
  }
}

// ------------------------------------------------------------------------------------------
// k_fetch_enterings / k_pt_grid_cells: InfGen._fetch_enterings (reference infgen/model/infgen.py:1008-1128) - what the
// insertion branch needs per scene and token step: agents within pl2seed_radius of the ego, their cell of the polar-
// cropped grid in the ego frame (Attr_Tokenizer.encode_pos, attr_tokenizer.py:77-89) + offset, the relative heading and
// its bin (encode_heading :101-104), the entering agents ordered by bearing from the ego's heading (others -> ego), and
// the cell of every map token (predict_occ).  One workgroup per (scene, step): a wave per agent with lanes over the grid
// cells (first minimum), then a rank sort of the bearings in LDS (ties by row, like torch's stable CPU sort).
// ------------------------------------------------------------------------------------------
constexpr int FE_MAX_AGENTS = 2048;

struct EgoFrame { float ex, ey, cs, sn, hc, hs; };

__device__ __forceinline__ EgoFrame ego_frame(const float* pos, const float* head, size_t i) {
  EgoFrame f;
  f.ex = pos[2 * i]; f.ey = pos[2 * i + 1];
  const float th = head[i], phi = -(th - HALF_PI_F);
  f.cs = cos_cr(phi); f.sn = sin_cr(phi); f.hc = cos_cr(th); f.hs = sin_cr(th);
  return f;
}

// nearest grid cell of (rx, ry) over the lanes of a wave; returns the cell to every lane
__device__ __forceinline__ int nearest_cell(const float* __restrict__ grid, int n, float rx, float ry, int lane) {
  float best = INFINITY;
  int bi = 0x7fffffff;
  for (int g = lane; g < n; g += 64) {
    const float2 gc = *reinterpret_cast<const float2*>(grid + 2 * g);
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
  return bi;
}

__device__ __forceinline__ float wrap_pi(float a) {          // func.py:58-62 in fp32
  const float PI_F = 3.14159274f, TWO_PI_F = 6.28318548f;
  float r = fmodf(__fadd_rn(a, PI_F), TWO_PI_F);
  if (r < 0.f) r += TWO_PI_F;
  return __fadd_rn(-PI_F, r);
}

__global__ __launch_bounds__(256) void k_fetch_enterings(EnteringsArgs a) {
  __shared__ float key[FE_MAX_AGENTS];
  __shared__ int n_born;
  const int t = blockIdx.x % a.T, b = blockIdx.x / a.T;
  const int r0 = a.agent_ptr[b], n = a.agent_ptr[b + 1] - r0, av = a.av_index[b];
  const EgoFrame f = ego_frame(a.token_pos, a.token_heading, (size_t)(r0 + av) * a.T + t);
  const float ego_head = a.token_heading[(size_t)(r0 + av) * a.T + t];
  const int lane = lane_id();
  if (threadIdx.x == 0) n_born = 0;
  for (int ag = wave_id(); ag < n; ag += 4) {
    const size_t i = (size_t)(r0 + ag) * a.T + t;
    const int st = a.state_idx[i];
    const float dx = a.token_pos[2 * i] - f.ex, dy = a.token_pos[2 * i + 1] - f.ey;
    const bool near = sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy))) <= a.radius;
    const bool born = st == a.enter_state, use = near && st != a.invalid_state;
    const float rx = dx * f.cs + dy * (-f.sn), ry = dx * f.sn + dy * f.cs;
    int cell = -1;
    if (use) cell = nearest_cell(a.grid_xy, a.grid_size, rx, ry, lane);       // wave-uniform branch
    if (lane == 0) {
      a.grid_token_idx[i] = cell;
      a.grid_offset_xy[2 * i] = use ? rx - a.grid_xy[2 * cell] : 0.f;
      a.grid_offset_xy[2 * i + 1] = use ? ry - a.grid_xy[2 * cell + 1] : 0.f;
      a.pos_xy[2 * i] = use ? dx : 0.f;
      a.pos_xy[2 * i + 1] = use ? dy : 0.f;
      a.inrange_mask[i] = near; a.bos_mask[i] = born;
      const float w = wrap_pi(__fsub_rn(a.token_heading[i], ego_head));
      a.heading_theta[i] = w;
      // ((w + pi) / (2 pi) * 360) // angle_interval with torch's floor division of floats
      const float deg = __fmul_rn(__fdiv_rn(__fadd_rn(w, 3.14159274f), 6.28318548f), 360.f);
      const float mod = fmodf(deg, a.angle_interval);
      float div = __fdiv_rn(__fsub_rn(deg, mod), a.angle_interval);
      if (mod != 0.f && (a.angle_interval < 0.f) != (mod < 0.f)) div -= 1.f;
      float fl = floorf(div);
      if (div - fl > 0.5f) fl += 1.f;
      a.heading_token_idx[i] = (int)fl;
      float den = f.hc * dx + f.hs * dy;
      if (den == 0.f) den = 0.f;                             // torch's sum starts from +0: never -0 (atan2(0, -0) = pi)
      key[ag] = (born && near) ? (float)atan2((double)(f.hc * dy - f.hs * dx), (double)den) : INFINITY;
    }
  }
  __syncthreads();
  for (int ag = threadIdx.x; ag < n; ag += 256) {
    const float k = key[ag];
    if (k == INFINITY) continue;
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += (key[j] < k) || (key[j] == k && j < ag);
    a.sort_indices[(size_t)(r0 + rank) * a.T + t] = ag;
    atomicAdd(&n_born, 1);
  }
  __syncthreads();
  for (int r = n_born + threadIdx.x; r < n; r += 256) a.sort_indices[(size_t)(r0 + r) * a.T + t] = av;
}

__global__ __launch_bounds__(256) void k_pt_grid_cells(EnteringsArgs a) {
  const int w = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = lane_id();
  if (w >= a.T * a.M) return;
  const int t = w / a.M, m = w % a.M;
  int b = 0;
  while (m >= a.pt_ptr[b + 1]) ++b;                            // few scenes per call
  const EgoFrame f = ego_frame(a.token_pos, a.token_heading, (size_t)(a.agent_ptr[b] + a.av_index[b]) * a.T + t);
  const float dx = a.pt_pos[(size_t)m * a.pt_stride] - f.ex, dy = a.pt_pos[(size_t)m * a.pt_stride + 1] - f.ey;
  int cell = -1;
  if (sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy))) <= a.radius)
    cell = nearest_cell(a.grid_xy, a.grid_size, dx * f.cs + dy * (-f.sn), dx * f.sn + dy * f.cs, lane);
  if (lane == 0) a.pt_grid_token_idx[(size_t)t * a.M + m] = cell;
}

}  // namespace ig
