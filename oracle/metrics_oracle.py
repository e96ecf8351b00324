"""CPU restatement of one rollout-metric feature of the reference - TEST INFRASTRUCTURE ONLY (imported by tests/ only).
Pinned against a fixture produced by the reference's own function (tests/golden/make_golden_metrics.py).

  distance_to_nearest_object   infgen/metrics/interact_features.py:19-95
      box corners              infgen/metrics/box_utils.py:77-113 (first four corners, xy)
      Minkowski sum            infgen/metrics/geometry_utils.py:10-37, 69-83
      signed distance          infgen/metrics/geometry_utils.py:40-66, 94-129
"""
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
