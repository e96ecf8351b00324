"""Mirror of the reference's infgen/metrics/interact_features.py (same function name and arguments); the work runs
in the HIP library (infgen_distance_to_nearest_object).  No CPU fallback."""
import torch
from torch import Tensor

from .. import _lib

CORNER_ROUNDING_FACTOR = 0.7


@torch.no_grad()
def compute_distance_to_nearest_object(center_x: Tensor, center_y: Tensor, center_z: Tensor, length: Tensor, width: Tensor,
                                       height: Tensor, heading: Tensor, valid: Tensor, evaluated_object_mask: Tensor,
                                       corner_rounding_factor: float = CORNER_ROUNDING_FACTOR) -> Tensor:
    """reference interact_features.py:19-95.  (num_objects, num_steps) inputs - or (batch, num_objects, num_steps) with
    one (num_objects,) mask shared by the batch - -> (num_eval_objects, num_steps) [(batch, ...)].
    center_z / height do not enter the result (the reference only uses the xy corners)."""
    dev = center_x.device
    if dev.type != 'cuda':
        raise RuntimeError('compute_distance_to_nearest_object runs on the GPU only (no CPU fallback)')
    batched = center_x.dim() == 3
    mask = evaluated_object_mask.to(dev).bool()
    order = torch.cat([torch.nonzero(mask)[:, 0], torch.nonzero(~mask)[:, 0]])
    prep = lambda a: (a if batched else a[None]).index_select(1, order).to(torch.float32).contiguous()
    cx, cy, ln, wd, hd = (prep(a) for a in (center_x, center_y, length, width, heading))
    vd = (valid if batched else valid[None]).index_select(1, order).to(torch.uint8).contiguous()
    B, N, T = cx.shape
    n_eval = int(mask.sum())
    out = torch.empty(B, n_eval, T, device=dev, dtype=torch.float32)
    if n_eval == 0:
        return out if batched else out[0]
    work = torch.empty(B * N * T * 9, device=dev, dtype=torch.float32)
    _lib.check(_lib.load().infgen_distance_to_nearest_object(
        _lib.ptr(cx), _lib.ptr(cy), _lib.ptr(ln), _lib.ptr(wd), _lib.ptr(hd), _lib.ptr(vd), B, N, T, n_eval,
        float(corner_rounding_factor), _lib.ptr(work), _lib.ptr(out), torch.cuda.current_stream(dev).cuda_stream),
        'infgen_distance_to_nearest_object')
    return out if batched else out[0]


@torch.no_grad()
def compute_time_to_collision_with_object_in_front(*, center_x: Tensor, center_y: Tensor, length: Tensor, width: Tensor,
                                                   heading: Tensor, valid: Tensor, evaluated_object_mask: Tensor,
                                                   seconds_per_step: float) -> Tensor:
    """reference interact_features.py:96-219 (keyword-only like there).  (num_objects, num_steps) inputs, or with a
    leading batch of scenes sharing the mask -> (num_eval_objects, num_steps) seconds, capped at 5."""
    from .trajectory_features import compute_kinematic_features
    dev = center_x.device
    if dev.type != 'cuda':
        raise RuntimeError('compute_time_to_collision_with_object_in_front runs on the GPU only (no CPU fallback)')
    batched = center_x.dim() == 3
    prep = lambda a: (a if batched else a[None]).to(torch.float32).contiguous()
    cx, cy, ln, wd, hd = (prep(a) for a in (center_x, center_y, length, width, heading))
    vd = (valid if batched else valid[None]).to(torch.uint8).contiguous()
    speed = compute_kinematic_features(cx, cy, torch.zeros_like(cx), hd, seconds_per_step)[0].contiguous()
    B, N, T = cx.shape
    eval_idx = torch.nonzero(evaluated_object_mask.to(dev).bool())[:, 0].to(torch.int32).contiguous()
    n_eval = int(eval_idx.numel())
    out = torch.empty(B, n_eval, T, device=dev, dtype=torch.float32)
    if n_eval:
        _lib.check(_lib.load().infgen_time_to_collision(
            _lib.ptr(cx), _lib.ptr(cy), _lib.ptr(ln), _lib.ptr(wd), _lib.ptr(hd), _lib.ptr(speed), _lib.ptr(vd),
            _lib.ptr(eval_idx), B, N, T, n_eval, _lib.ptr(out), torch.cuda.current_stream(dev).cuda_stream),
            'infgen_time_to_collision')
    return out if batched else out[0]
