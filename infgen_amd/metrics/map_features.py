"""Mirror of the reference's infgen/metrics/map_features.py (compute_distance_to_road_edge, keyword-only like there);
the distance search runs in the HIP library (infgen_distance_to_road_edge).  No CPU fallback."""
from typing import Sequence, Tuple, Union

import numpy as np
import torch
from torch import Tensor

from .. import _lib

EXTREMELY_LARGE_DISTANCE = 1e10
OFFROAD_DISTANCE_THRESHOLD = 0.0
_CYCLIC_MAP_FEATURE_TOLERANCE_M2 = 1.0
_Z_STRETCH_FACTOR = 3.0


def tensorize_polylines(polylines: Sequence, device=None) -> Tuple[Tensor, Tensor]:
    """reference map_features.py:82-136: sequences of map points (objects with .x .y .z, or (n, 3) arrays) -> padded
    (P, L, 4) float32 [x, y, z, valid] and (P,) uint8 cyclic flags; polylines with fewer than two points are dropped.
    Host-side, once per scene: pass the result as `road_edge_polylines` to avoid repeating it."""
    keep = []
    for pl in polylines:
        if len(pl) < 2:
            continue
        if hasattr(pl[0], 'x'):
            pl = np.array([[pt.x, pt.y, pt.z] for pt in pl], dtype=np.float32)
        keep.append(np.asarray(pl, dtype=np.float32).reshape(-1, 3))
    L = max(p.shape[0] for p in keep)
    out = np.zeros((len(keep), L, 4), np.float32)
    cyc = np.zeros(len(keep), np.uint8)
    for i, p in enumerate(keep):
        out[i, :p.shape[0], :3] = p
        out[i, :p.shape[0], 3] = 1.0
        gap = p[0] - p[-1]
        cyc[i] = np.float32((gap * gap).sum(dtype=np.float32)) < _CYCLIC_MAP_FEATURE_TOLERANCE_M2
    return torch.from_numpy(out).to(device), torch.from_numpy(cyc).to(device)


@torch.no_grad()
def compute_distance_to_road_edge(*, center_x: Tensor, center_y: Tensor, center_z: Tensor, length: Tensor, width: Tensor,
                                  height: Tensor, heading: Tensor, valid: Tensor, evaluated_object_mask: Tensor,
                                  road_edge_polylines: Union[Sequence, Tuple[Tensor, Tensor]]) -> Tensor:
    """reference map_features.py:27-79.  (num_objects, num_steps) boxes -> (num_eval_objects, num_steps): signed distance
    of the most off-road bottom corner to the road edges (> 0 = off road), -1e10 where the box is invalid."""
    if road_edge_polylines is None or len(road_edge_polylines) == 0:
        raise ValueError('Missing road edges.')
    dev = center_x.device
    if dev.type != 'cuda':
        raise RuntimeError('compute_distance_to_road_edge runs on the GPU only (no CPU fallback)')
    if isinstance(road_edge_polylines, tuple) and torch.is_tensor(road_edge_polylines[0]):
        poly, cyc = road_edge_polylines
    else:
        poly, cyc = tensorize_polylines(road_edge_polylines)
    poly = poly.to(dev, torch.float32).contiguous()
    cyc = cyc.to(dev, torch.uint8).contiguous()
    c = lambda a: a.to(torch.float32).contiguous()
    cx, cy, cz, ln, wd, ht, hd = (c(a) for a in (center_x, center_y, center_z, length, width, height, heading))
    vd = valid.to(torch.uint8).contiguous()
    N, T = cx.shape
    eval_idx = torch.nonzero(evaluated_object_mask.to(dev).bool())[:, 0].to(torch.int32).contiguous()
    n_eval = int(eval_idx.numel())
    out = torch.empty(n_eval, T, device=dev, dtype=torch.float32)
    if n_eval:
        off = torch.tensor([0, poly.shape[0]], dtype=torch.int32, device=dev)
        _lib.check(_lib.load().infgen_distance_to_road_edge(
            _lib.ptr(cx), _lib.ptr(cy), _lib.ptr(cz), _lib.ptr(ln), _lib.ptr(wd), _lib.ptr(ht), _lib.ptr(hd), _lib.ptr(vd),
            _lib.ptr(eval_idx), 1, N, T, n_eval, _lib.ptr(poly), _lib.ptr(cyc), _lib.ptr(off), int(poly.shape[1]),
            _Z_STRETCH_FACTOR, _lib.ptr(out), torch.cuda.current_stream(dev).cuda_stream), 'infgen_distance_to_road_edge')
    return out
