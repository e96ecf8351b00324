"""Mirror of the reference's infgen/metrics/trajectory_features.py (compute_kinematic_features) on the HIP library."""
from typing import Tuple

import torch
from torch import Tensor

from .. import _lib


@torch.no_grad()
def compute_kinematic_features(x: Tensor, y: Tensor, z: Tensor, heading: Tensor,
                               seconds_per_step: float) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """reference trajectory_features.py:37-51: (..., num_steps) inputs -> linear speed, linear acceleration, yaw rate,
    yaw acceleration of the same shape (central differences, NaN at both ends)."""
    dev = x.device
    if dev.type != 'cuda':
        raise RuntimeError('compute_kinematic_features runs on the GPU only (no CPU fallback)')
    shape = x.shape
    T = shape[-1]
    c = lambda a: a.to(torch.float32).reshape(-1, T).contiguous()
    xs, ys, zs, hs = c(x), c(y), c(z), c(heading)
    outs = [torch.empty_like(xs) for _ in range(4)]
    _lib.check(_lib.load().infgen_kinematic_features(_lib.ptr(xs), _lib.ptr(ys), _lib.ptr(zs), _lib.ptr(hs), xs.shape[0], T,
                                                     float(seconds_per_step), *(_lib.ptr(o) for o in outs),
                                                     torch.cuda.current_stream(dev).cuda_stream), 'infgen_kinematic_features')
    return tuple(o.reshape(shape) for o in outs)
