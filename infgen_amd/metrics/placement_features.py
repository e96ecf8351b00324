"""Mirror of the reference's infgen/metrics/placement_features.py on the HIP library (both functions in one launch)."""
from typing import List, Tuple

import torch
from torch import Tensor

from .. import _lib


@torch.no_grad()
def _placement(position: Tensor, state: Tensor, av_id: int, object_id: Tensor, agent_state: List[str]):
    dev = position.device
    if dev.type != 'cuda':
        raise RuntimeError('placement features run on the GPU only (no CPU fallback)')
    av_index = object_id.tolist().index(av_id)
    N, T = state.shape
    c = lambda a: a.to(torch.float32).contiguous()
    x, y = c(position[..., 0]), c(position[..., 1])
    z = c(position[..., 2]) if position.shape[-1] > 2 else None
    st = state.to(torch.int32).contiguous()
    av = torch.tensor([av_index], dtype=torch.int32, device=dev)
    nb = torch.empty(1, T, dtype=torch.int32, device=dev)
    ne = torch.empty_like(nb)
    db = torch.empty(1, N, T, dtype=torch.float32, device=dev)
    de = torch.empty_like(db)
    _lib.check(_lib.load().infgen_placement_features(
        _lib.ptr(x), _lib.ptr(y), _lib.ptr(z), _lib.ptr(st), _lib.ptr(av), 1, N, T, agent_state.index('enter'),
        agent_state.index('exit'), _lib.ptr(nb), _lib.ptr(ne), _lib.ptr(db), _lib.ptr(de),
        torch.cuda.current_stream(dev).cuda_stream), 'infgen_placement_features')
    return nb[0].long(), ne[0].long(), db[0], de[0]


def compute_num_placement(valid: Tensor, state: Tensor, av_id: int, object_id: Tensor, agent_state: List[str]) -> Tuple[Tensor, Tensor]:
    """reference placement_features.py:6-26 (needs the positions only for the shared kernel: pass them via
    compute_distance_placement if both are wanted).  Unlike the reference, `state` is not modified in place."""
    pos = torch.zeros(*state.shape, 2, device=state.device)
    nb, ne, _, _ = _placement(pos, state, av_id, object_id, agent_state)
    return nb, ne


def compute_distance_placement(position: Tensor, state: Tensor, valid: Tensor, av_id: int, object_id: Tensor,
                               agent_state: List[str]) -> Tuple[Tensor, Tensor]:
    """reference placement_features.py:29-48"""
    _, _, db, de = _placement(position, state, av_id, object_id, agent_state)
    return db, de
