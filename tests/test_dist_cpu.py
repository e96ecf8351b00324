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
