"""CPU restatement of one rollout-metric feature of the reference - TEST INFRASTRUCTURE ONLY (imported by tests/ only).
Pinned against a fixture produced by the reference's own function (tests/golden/make_golden_metrics.py).

  distance_to_nearest_object   infgen/metrics/interact_features.py:19-95
      box corners              infgen/metrics/box_utils.py:77-113 (first four corners, xy)
      Minkowski sum            infgen/metrics/geometry_utils.py:10-37, 69-83
      signed distance          infgen/metrics/geometry_utils.py:40-66, 94-129
"""
import math

import torch

BIG = 1e10


def _corners_xy(cx, cy, length, width, heading):
    """(..., 4, 2): (+l,+w), (-l,+w), (-l,-w), (+l,-w) halves rotated by heading, counter-clockwise"""
    c, s = torch.cos(heading), torch.sin(heading)
    l2, w2 = length * 0.5, width * 0.5
    lx = torch.stack([l2, -l2, -l2, l2], -1)
    ly = torch.stack([w2, w2, -w2, -w2], -1)
    return torch.stack([c[..., None] * lx - s[..., None] * ly + cx[..., None],
                        s[..., None] * lx + c[..., None] * ly + cy[..., None]], -1)


def _downmost(box):
    i0 = torch.argmin(box[..., 1], dim=-1)
    ar = torch.arange(box.shape[0])
    e = box[ar, (i0 + 1) % 4] - box[ar, i0]
    return i0, e / torch.norm(e, dim=-1, keepdim=True)


def _minkowski(b1, b2):
    o1 = torch.tensor([0, 0, 1, 1, 2, 2, 3, 3])
    o2 = torch.tensor([0, 1, 1, 2, 2, 3, 3, 0])
    s1, d1 = _downmost(b1)
    s2, d2 = _downmost(b2)
    cond = (d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0] >= 0.0)[:, None]
    i1 = (torch.where(cond, o2, o1) + s1[:, None]) % 4
    i2 = (torch.where(cond, o1, o2) + s2[:, None]) % 4
    ar = torch.arange(b1.shape[0])[:, None]
    return b1[ar, i1] + b2[ar, i2]


def _signed_distance_origin(poly):
    nxt = torch.roll(poly, -1, dims=1)
    e = nxt - poly
    ln = torch.norm(e, dim=-1)
    t = e / (ln[..., None] + torch.finfo(poly.dtype).eps)
    n = torch.stack([-t[..., 1], t[..., 0]], -1)
    v = -poly
    vd = torch.norm(v, dim=-1)
    perp = torch.sum(-n * v, dim=-1)
    inside = torch.all(perp <= 0, dim=-1)
    prop = torch.sum(t * v, dim=-1) / ln
    on = (prop >= 0.0) & (prop <= 1.0)
    ed = torch.where(on, perp.abs(), torch.tensor(float('inf')))
    md = torch.min(torch.cat([ed, vd], -1), dim=-1)[0]
    return torch.where(inside, -md, md)


@torch.no_grad()
def distance_to_nearest_object(cx, cy, length, width, heading, valid, eval_mask, corner_rounding_factor=0.7):
    """all inputs (N, T) (eval_mask (N,) bool) -> (n_eval, T): signed distance of every evaluated object to the
    nearest other valid object (rounded-corner boxes: shrink, measure, subtract the shrink radii), 1e10 if none"""
    N, T = cx.shape
    shrink = torch.minimum(length, width) * corner_rounding_factor / 2.0
    corners = _corners_xy(cx, cy, length - 2.0 * shrink, width - 2.0 * shrink, heading)      # (N, T, 4, 2)
    order = torch.cat([torch.nonzero(eval_mask)[:, 0], torch.nonzero(~eval_mask)[:, 0]])
    ne = int(eval_mask.sum())
    ev, al = corners[order[:ne]], corners[order]
    b1 = ev[:, None].expand(ne, N, T, 4, 2).reshape(-1, 4, 2)
    b2 = (-1.0 * al)[None].expand(ne, N, T, 4, 2).reshape(-1, 4, 2)
    d = _signed_distance_origin(_minkowski(b1, b2)).reshape(ne, N, T)
    sh = shrink[order]
    d = d - sh[:ne, None, :] - sh[None, :, :]
    d = d + torch.eye(ne, N)[:, :, None] * BIG
    ok = valid[order[:ne]][:, None, :] & valid[order][None, :, :]
    d = torch.where(ok, d, torch.tensor(BIG))
    return torch.min(d, dim=1).values


def _central_diff(t, pad):
    p = torch.full((*t.shape[:-1], 1), pad, dtype=t.dtype)
    return torch.cat([p, (t[..., 2:] - t[..., :-2]) / 2, p], dim=-1)


def _wrap(a):
    return (a + math.pi) % (2 * math.pi) - math.pi


@torch.no_grad()
def kinematic_features(x, y, z, heading, seconds_per_step):
    """infgen/metrics/trajectory_features.py:37-51 (central differences, NaN at both ends) ->
    linear speed, linear acceleration, yaw rate, yaw acceleration, each (..., T)"""
    dpos = _central_diff(torch.stack([x, y, z], dim=0), float('nan'))
    speed = torch.norm(dpos, p=2, dim=0) / seconds_per_step
    accel = _central_diff(speed, float('nan')) / seconds_per_step
    dh_step = _wrap(_central_diff(heading, float('nan')) * 2) / 2
    d2h_step = _wrap(_central_diff(dh_step, float('nan')) * 2) / 2
    return speed, accel, dh_step / seconds_per_step, d2h_step / (seconds_per_step ** 2)


@torch.no_grad()
def time_to_collision(cx, cy, length, width, heading, valid, eval_mask, seconds_per_step):
    """infgen/metrics/interact_features.py:96-219: for every evaluated object and step the closest valid object it is
    "following" (ahead, laterally overlapping its trail, yaw within 75 deg - 10 deg if the overlap is below 0.5 m) and
    distance / closing speed, capped at 5 s.  (N, T) inputs -> (n_eval, T)"""
    speed = kinematic_features(cx, cy, torch.zeros_like(cx), heading, seconds_per_step)[0]
    P = lambda a: a.permute(1, 0)                                  # (T, N)
    ex, ey, el, ew, eh, es = (P(a)[:, eval_mask] for a in (cx, cy, length, width, heading, speed))
    ox, oy, ol, ow, oh = (P(a) for a in (cx, cy, length, width, heading))
    yd = torch.abs(oh[:, None, :] - eh[:, :, None])                # (T, E, N)
    c, s = torch.cos(yd).abs(), torch.sin(yd).abs()
    long_off = ol[:, None] / 2.0 * c + ow[:, None] / 2.0 * s
    lat_off = ol[:, None] / 2.0 * s + ow[:, None] / 2.0 * c
    dx, dy = ox[:, None] - ex[:, :, None], oy[:, None] - ey[:, :, None]
    ce, se = torch.cos(-eh)[:, :, None], torch.sin(-eh)[:, :, None]
    rx, ry = ce * dx - se * dy, se * dx + ce * dy
    long_d = rx - el[:, :, None] / 2.0 - long_off
    lat_o = ry.abs() - ew[:, :, None] / 2.0 - lat_off
    follow = (long_d > 0.0) & (yd <= math.radians(75.0)) & (lat_o < 0.0) & ((lat_o < -0.5) | (yd <= math.radians(10.0)))
    ok = P(valid)[:, None] & follow
    masked = long_d + (1.0 - ok.float()) * BIG
    idx = masked.argmin(dim=-1)
    dist = torch.gather(masked, -1, idx[..., None])[..., 0]
    ahead_speed = torch.gather(P(speed)[:, None, :].expand_as(masked), -1, idx[..., None])[..., 0]
    rel = es - ahead_speed
    ttc = torch.where(rel > 0.0, torch.minimum(dist / rel, torch.tensor(5.0)), torch.tensor(5.0))
    return ttc.T


@torch.no_grad()
def placement_features(position, state, av_index, enter_state=2, exit_state=3):
    """infgen/metrics/placement_features.py:6-48 (compute_num_placement + compute_distance_placement; the ego row is
    excluded by setting its state to -1 as there).  position (N, T, 2|3), state (N, T) ->
    num_bos (T,), num_eos (T,), bos_distance (N, T), eos_distance (N, T)"""
    st = state.clone()
    st[av_index] = -1
    is_bos, is_eos = st == enter_state, st == exit_state
    dist = torch.norm(position - position[av_index:av_index + 1], p=2, dim=-1)
    return torch.sum(is_bos, dim=0), torch.sum(is_eos, dim=0), dist * is_bos, dist * is_eos


# ---------------------------------------------------------------------------------------------------------------
# compute_distance_to_road_edge (reference infgen/metrics/map_features.py:27-79 with :82-136 tensorisation and
# :139-349 signed distance to polylines)
# ---------------------------------------------------------------------------------------------------------------
def tensorize_polylines(polylines):
    """list of (n_i, 3) arrays -> padded (P, L, 4) float32 [x, y, z, valid] and cyclic (P,) bool; polylines with fewer
    than two points are dropped (map_features.py:82-136; cyclic = squared end gap below 1 m^2)."""
    keep = [torch.as_tensor(p, dtype=torch.float32).reshape(-1, 3) for p in polylines if len(p) >= 2]
    L = max(p.shape[0] for p in keep)
    out = torch.zeros(len(keep), L, 4)
    for i, p in enumerate(keep):
        out[i, :p.shape[0], :3] = p
        out[i, :p.shape[0], 3] = 1.0
    cyc = torch.stack([((p[0] - p[-1]) ** 2).sum() < 1.0 for p in keep])
    return out, cyc


def signed_distance_to_polylines(pts, poly, cyclic, z_stretch=3.0):
    """pts (Q, 3), poly (P, L, 4), cyclic (P,) -> (Q,) signed planar distance to the segment that is nearest in the
    z-stretched 3-D metric (first minimum); negative = port side (map_features.py:139-349).  Neighbour lookups wrap for
    cyclic polylines over the PADDED segment list, like the reference."""
    P, L, _ = poly.shape
    S = L - 1
    ok_pt = poly[..., 3] != 0
    ok = ok_pt[:, :-1] & ok_pt[:, 1:]                                   # (P, S)
    a, b = poly[:, :-1, :3], poly[:, 1:, :3]
    d = (b - a)[None]                                                   # (1, P, S, 3)
    w = pts[:, None, None, :] - a[None]                                 # (Q, P, S, 3)
    num = w[..., 0] * d[..., 0] + w[..., 1] * d[..., 1]
    den = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]
    t = torch.where(den != 0, num / den, torch.zeros_like(num))
    side = torch.sign(w[..., 0] * d[..., 1] - w[..., 1] * d[..., 0])
    foot = w - d * t.clamp(0.0, 1.0)[..., None]
    d3 = torch.sqrt(foot[..., 0] ** 2 + foot[..., 1] ** 2 + (foot[..., 2] * z_stretch) ** 2)
    d2 = torch.sqrt(foot[..., 0] ** 2 + foot[..., 1] ** 2)
    dxy = d[0, :, :, :2]                                                # (P, S, 2)
    prev_d = torch.roll(dxy, 1, dims=1)                                 # d[i-1], wrapping (also for open polylines)
    turn_in = (prev_d[..., 0] * dxy[..., 1] - prev_d[..., 1] * dxy[..., 0]) > 0      # convex at the start vertex of i
    turn_out = torch.roll(turn_in, -1, dims=1)                                     # convex at its end vertex
    idx = torch.arange(S)
    cyc = cyclic[:, None]
    i_prev = torch.where(cyc, (idx - 1) % S, (idx - 1).clamp(min=0))    # (P, S)
    i_next = torch.where(cyc, (idx + 1) % S, (idx + 1).clamp(max=S - 1))
    side_prev = torch.gather(side, 2, i_prev[None].expand_as(side))
    side_next = torch.gather(side, 2, i_next[None].expand_as(side))
    ok_prev = torch.gather(ok, 1, i_prev)
    ok_next = torch.gather(ok, 1, i_next)
    before = torch.where(turn_in[None], torch.maximum(side, side_prev), torch.minimum(side, side_prev))
    after = torch.where(turn_out[None], torch.maximum(side, side_next), torch.minimum(side, side_next))
    sgn = torch.where((t < 0) & ok_prev[None], before, torch.where((t > 1) & ok_next[None], after, side))
    big = torch.tensor(1e10)
    d3 = torch.where(ok[None], d3, big).reshape(pts.shape[0], -1)
    d2 = torch.where(ok[None], d2, big).reshape(pts.shape[0], -1)
    k = d3.argmin(-1, keepdim=True)
    return (sgn.reshape(pts.shape[0], -1).gather(1, k) * d2.gather(1, k))[:, 0]


def distance_to_road_edge(cx, cy, cz, length, width, height, heading, valid, eval_mask, poly, cyclic):
    """(N, T) boxes -> (n_eval, T): the most off-road bottom corner's signed distance; -1e10 where the box is invalid"""
    c, s = torch.cos(heading), torch.sin(heading)
    hl, hw = length * 0.5, width * 0.5
    zb = cz - height * 0.5
    cor = []
    for sl, sw in ((1, 1), (-1, 1), (-1, -1), (1, -1)):
        cor.append(torch.stack([cx + (c * (sl * hl) - s * (sw * hw)), cy + (s * (sl * hl) + c * (sw * hw)), zb], -1))
    cor = torch.stack(cor, -2)[eval_mask]                               # (n_eval, T, 4, 3)
    dist = signed_distance_to_polylines(cor.reshape(-1, 3), poly, cyclic).reshape(cor.shape[:3])
    return torch.where(valid[eval_mask], dist.max(-1)[0], torch.tensor(-1e10))
