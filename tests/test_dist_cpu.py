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
    q.put((rank, mine, strided, secs, steps, gathered))
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
    (r0, m0, s0, t0, c0, g0), (r1, m1, s1, t1, c1, g1) = res
    assert m0 == [0, 1, 2] and m1 == [3, 4, 5]                       # weak scaling: disjoint scene ids
    assert sorted(s0 + s1) == list(range(7)) and not set(s0) & set(s1)   # strided: a partition
    assert t0 == t1 == 2.0                                            # MAX over ranks
    assert c0 == c1 == 300.0                                          # SUM over ranks
    assert g0 == g1 == [[0.0, 4.0], [1.0, 3.0]]


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
