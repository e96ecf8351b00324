"""CPU, world_size 2, gloo: the N > 1 path of bench.py / InfGenDecoder.inference_batch — scene
sharding with no data-path collective and the closing all-reduce / all-gather of counters."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from infgen_amd import dist as igd
    mine = igd.scenes_for_rank_weak(rank, 3)
    strided = igd.scenes_for_rank_strided(rank, world, 7)
    dist.barrier()
    secs, steps = igd.reduce_run(1.0 + rank, 100.0 * (rank + 1), torch.device('cpu'))
    gathered = igd.gather_metrics([float(rank), float(len(strided))], torch.device('cpu'))
    # insertion runs: a pilot on the strided deal measures every scene's cost, the cost vector is all-reduced, and every rank
    # computes the same balanced partition from it
    local = [(i, 10.0 + 7.0 * (i % 5) + (40.0 if i == 2 else 0.0)) for i in strided]
    costs = igd.gather_costs(local, 7, torch.device('cpu'))
    balanced = igd.scenes_for_rank_balanced(costs, rank, world)
    q.put((rank, mine, strided, secs, steps, gathered, costs, balanced))
    dist.destroy_process_group()


def test_two_rank_sharding_and_reduction():
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, s0, t0, c0, g0, k0, b0), (r1, m1, s1, t1, c1, g1, k1, b1) = res
    from infgen_amd import dist as igd
    assert k0 == k1 and len(k0) == 7 and min(k0) > 0                   # the same full cost vector on both ranks
    assert sorted(b0 + b1) == list(range(7)) and not set(b0) & set(b1)   # balanced: a partition too
    assert igd.partition_spread(k0, [b0, b1]) <= igd.partition_spread(k0, [s0, s1])
    assert igd.partition_spread(k0, [b0, b1]) < 1.05
    assert m0 == [0, 1, 2] and m1 == [3, 4, 5]                       # weak scaling: disjoint scene ids
    assert sorted(s0 + s1) == list(range(7)) and not set(s0) & set(s1)   # strided: a partition
    assert t0 == t1 == 2.0                                            # MAX over ranks
    assert c0 == c1 == 300.0                                          # SUM over ranks
    assert g0 == g1 == [[0.0, 4.0], [1.0, 3.0]]


def test_balanced_dealing_is_deterministic_and_tighter_than_strided():
    """LPT dealing on a skewed cost vector (a few scenes insert many agents): every scene dealt exactly once, ranks agree
    without communication, max / mean load below the strided deal's (reference layout: scalable_dataset.py:266-269)"""
    import numpy as np
    from infgen_amd import dist as igd
    rng = np.random.default_rng(5)
    costs = (64 + rng.gamma(1.2, 30.0, size=64)).tolist()            # 64 agents + a long-tailed number of inserted ones
    for world in (2, 4, 8):
        parts = [igd.scenes_for_rank_balanced(costs, r, world) for r in range(world)]
        assert sorted(i for p in parts for i in p) == list(range(64))
        strided = [igd.scenes_for_rank_strided(r, world, 64) for r in range(world)]
        assert igd.partition_spread(costs, parts) <= igd.partition_spread(costs, strided) + 1e-12
        assert igd.partition_spread(costs, parts) < (1.03 if world <= 4 else 1.08)       # (8 scenes per rank at world 8)
    assert igd.scenes_for_rank_balanced([1.0] * 6, 1, 3) == [1, 4]    # equal costs: ties by index / rank, like a strided deal
    assert igd.scene_cost(64, 1024, 22) > igd.scene_cost(64, 1024)


def test_single_process_is_a_noop():
    from infgen_amd import dist as igd
    assert igd.reduce_run(1.5, 10.0, torch.device('cpu')) == (1.5, 10.0)
    assert igd.gather_metrics([1.0], torch.device('cpu')) == [[1.0]]


def _run_bench(*argv, env_extra=None):
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), *argv], capture_output=True, text=True, env=env,
                       timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout            # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_bench_gpus_flag_spawns_the_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (torch.distributed.run on 127.0.0.1) and
    goes through the rank path of a real run - process group, sharding, barriers, MAX / SUM reductions, per-rank gather; the
    dry run swaps nccl for gloo and skips the kernels"""
    line = _run_bench('--gpus', '2', '--dry-run', '--steps', '3', '--scenes', '5')
    assert line['n_gpus'] == 2 and line['dry_run'] and line['backend'] == 'gloo'
    assert line['scenes_per_rank'] == [5, 5]
    assert line['c3_literal_scenes_per_rank'] == [32, 32]           # BASELINE C3: 64 scenes dealt to the ranks
    assert len(line['per_rank_ms']) == 2 and line['per_rank_ms'][1] > line['per_rank_ms'][0] * 0.5
    assert line['agent_steps_counted'] == 2 * 5 * 64 * 80 * 3       # SUM over the ranks
    assert line['ms_per_step'] >= max(line['per_rank_ms']) * 0.99   # MAX over the ranks


def test_bench_strong_scaling_deals_the_fixed_batch():
    line = _run_bench('--gpus', '2', '--dry-run', '--scaling', 'strong', '--total-scenes', '7', '--steps', '1')
    assert line['scenes_per_rank'] == [4, 3]


def test_bench_insertion_run_deals_scenes_by_cost():
    """`bench.py --gpus 2 --insertion`: per-scene costs measured by the pilot are exchanged (one all-reduce of a dense vector) and
    the scenes dealt again longest first - every scene exactly once, tighter than the initial deal"""
    line = _run_bench('--gpus', '2', '--dry-run', '--insertion', '--steps', '1', '--scenes', '9')
    b = line['insertion_balance']
    assert b['all_scenes_dealt_once'] and sum(b['scenes_per_rank']) == 18
    assert b['max_over_mean_after'] <= b['max_over_mean_before'] and b['max_over_mean_after'] < 1.02
    assert line['scenes_per_rank'] == b['scenes_per_rank']
    assert _run_bench('--gpus', '2', '--dry-run', '--steps', '1', '--scenes', '3')['insertion_balance'] is None


def test_bench_insertion_strong_scaling_two_ranks():
    """the first 8-GPU run will be the driver's - the rank path of `--insertion --scaling strong` (the
    BASELINE C4 batch dealt like the reference's DistributedSampler, then re-dealt by measured cost) on two gloo ranks: every
    rank reports, every scene of the fixed batch is dealt exactly once, the line names its ranks"""
    line = _run_bench('--gpus', '2', '--dry-run', '--insertion', '--scaling', 'strong', '--total-scenes', '11', '--steps', '2')
    assert line['n_gpus'] == 2 and line['ranks'] == 2 and line['scaling'] == 'strong' and line['backend'] == 'gloo'
    assert len(line['scenes_per_rank']) == 2 and sum(line['scenes_per_rank']) == 11 and min(line['scenes_per_rank']) >= 1
    b = line['insertion_balance']
    assert b['all_scenes_dealt_once'] and b['scenes_per_rank'] == line['scenes_per_rank']
    assert b['max_over_mean_after'] <= b['max_over_mean_before']
    assert len(line['per_rank_ms']) == 2 and line['c3_literal_scenes_per_rank'] == [32, 32]


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1')
    p = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2', '--dry-run'], capture_output=True,
                       text=True, env=env, timeout=120)
    assert p.returncode != 0 and 'launcher started 1 rank' in (p.stderr + p.stdout)


def _metric_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from infgen_amd.metrics.long_metric import LongMetric
    m = LongMetric.__new__(LongMetric)
    m.prefix, m.metrics_config, m.log_distributions = 'val', None, None
    m.reset()
    scal = {k: float(rank + 1) for k in m.field_names}
    m.update(metrics=(scal, {'metametric': torch.full((1, 4), float(rank + 1))}))
    a = m.synced_state()
    m.update(metrics=(scal, {'metametric': torch.full((1, 4), float(rank + 1))}))
    b = m.synced_state()                   # a second reduction must not count the first one's merge again
    q.put((rank, a['counters'][0], a['sums']['metametric'], b['counters'][0], b['sums']['metametric'], m.scenario_counter,
           len(b['longs']['metametric'])))
    dist.destroy_process_group()


def test_long_metric_reduction_leaves_the_local_state():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_metric_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, n1, s1, n2, s2, local_n, n_long in res:
        assert (n1, s1) == (2, 3.0)            # 1 scenario per rank, sums 1 + 2
        assert (n2, s2) == (4, 6.0)            # after one more update each: exactly twice, not merged-on-merged
        assert local_n == 2 and n_long == 4    # the object itself still holds only its own two scenarios
