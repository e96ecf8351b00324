"""Multi-GPU layout of the rollout: scenes are independent units, so rank r owns
scenes {r * per_rank ... (r+1) * per_rank - 1} (weak scaling) or {i : i mod W == r} of a fixed
list (the reference's DistributedSampler layout, infgen/datasets/scalable_dataset.py:266-269).
No data-path collective exists; one all-reduce of (seconds MAX, agent-steps SUM) closes a run —
the analogue of torchmetrics' dist_reduce_fx (infgen/metrics/compute_metrics.py:1199-1204).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch


def scenes_for_rank_weak(rank: int, per_rank: int) -> List[int]:
    return list(range(rank * per_rank, (rank + 1) * per_rank))


def scenes_for_rank_strided(rank: int, world: int, total: int) -> List[int]:
    return [i for i in range(total) if i % world == rank]


def reduce_run(seconds: float, agent_steps: float, device: torch.device) -> Tuple[float, float]:
    """MAX over ranks of the wall time, SUM over ranks of the agent-steps (no-op without a group)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds, agent_steps
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    c = torch.tensor([agent_steps], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), float(c.item())


def gather_metrics(values: Sequence[float], device: torch.device) -> List[List[float]]:
    """all_gather of a small per-rank metric vector (rank-major list)"""
    import torch.distributed as dist
    v = torch.tensor(list(values), dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [v.tolist()]
    out = [torch.zeros_like(v) for _ in range(dist.get_world_size())]
    dist.all_gather(out, v)
    return [o.tolist() for o in out]
