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


def scene_cost(n_agents: float, n_map: float, inserted: float = 0.0) -> float:
    """relative cost of one scene's rollout: the rows it decodes per step (initial agents + the agents insertion adds, counted
    by a pilot rollout or an estimate) weighted by its map size (map -> agent / map -> seed edge lists grow with it)"""
    return float(n_agents + inserted) * (1.0 + float(n_map) / 8192.0)


def scenes_for_rank_balanced(costs: Sequence[float], rank: int, world: int) -> List[int]:
    """Cost-sorted dealing for runs whose per-scene cost is data dependent (scenario insertion, SURVEY 8e: "balance by
    scene-cost estimate"): scenes in order of decreasing cost, each to the rank with the smallest load so far (longest
    processing time first; ties -> lower scene index, lower rank), so every rank computes the same partition from the same
    cost vector without talking.  The reference deals scene i to rank i mod W whatever it costs
    (infgen/datasets/scalable_dataset.py:266-269).  Returns this rank's scene indices in ascending order."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world
    count = [0] * world
    mine: List[int] = []
    for i in order:
        r = min(range(world), key=lambda k: (load[k], count[k], k))
        load[r] += float(costs[i])
        count[r] += 1
        if r == rank:
            mine.append(i)
    return sorted(mine)


def partition_spread(costs: Sequence[float], parts: Sequence[Sequence[int]]) -> float:
    """max / mean of the per-rank cost sums of a partition (1.0 = perfectly balanced)"""
    sums = [sum(float(costs[i]) for i in p) for p in parts]
    mean = sum(sums) / max(1, len(sums))
    return max(sums) / mean if mean > 0 else 1.0


def gather_costs(local: Sequence[Tuple[int, float]], total: int, device: torch.device) -> List[float]:
    """every rank contributes (scene index, cost) pairs of the scenes it ran; returns the full cost vector on all ranks
    (one all_reduce SUM of a dense vector - control-plane only, no rollout data moves)"""
    import torch.distributed as dist
    v = torch.zeros(total, dtype=torch.float64, device=device)
    for i, c in local:
        v[i] = float(c)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
    return v.tolist()


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
