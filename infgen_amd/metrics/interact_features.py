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
