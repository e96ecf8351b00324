"""The later widening steps on the device vs their CPU oracles (= the reference's torch code restated): the whole agent
tokeniser, _fetch_enterings, distance to the road edge, and rollouts -> MetricFeatures.  One JSON line per entry.
    python tools/bench_widening.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infgen_amd import synth  # noqa: E402
from infgen_amd.metrics import compute_distance_to_road_edge, compute_metric_features, output_to_rollouts, tensorize_polylines  # noqa: E402
from infgen_amd.modules import Attr_Tokenizer, TokenProcessor, fetch_enterings  # noqa: E402
from oracle import enterings_oracle as eo, metrics_oracle as mo, token_match_oracle as tm  # noqa: E402

dev = torch.device('cuda:0')
torch.set_num_threads(16)


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def tracks(A, T=91, seed=5):
    rng = np.random.default_rng(seed)
    atype = rng.integers(0, 3, size=A)
    speed = rng.uniform(0.0, 14.0, size=A) * np.where(atype == 1, 0.15, 1.0)
    t = np.arange(T) * 0.1
    head = (rng.uniform(-np.pi, np.pi, size=A)[:, None] + rng.uniform(-0.5, 0.5, size=A)[:, None] * t[None]).astype(np.float32)
    vel = (speed[:, None, None] * np.stack([np.cos(head), np.sin(head)], -1)).astype(np.float32)
    pos = (rng.uniform(-60, 60, size=(A, 1, 2)) + np.cumsum(vel, 1) * 0.1).astype(np.float32)
    pos3 = np.concatenate([pos, np.zeros((A, T, 1), np.float32)], -1)
    valid = rng.random((A, T)) > 0.03
    valid[:, :rng.integers(0, 30)] &= rng.random((A, 1)) > 0.3
    valid[:, 40] = True
    shape = (np.array([[4.8, 2.0, 1.6], [0.9, 0.9, 1.8], [1.9, 0.8, 1.7]], np.float32)[atype][:, None] * np.ones((1, T, 1), np.float32))
    return dict(valid_mask=valid, heading=head, position=pos3, velocity=vel, type=atype.astype(np.int64), shape=shape)


def bench_tokenize(A=32768, n_cpu=64):
    tr = tracks(A)
    vocab = synth.make_agent_vocab(2048)
    tp = TokenProcessor(2048, predict_state=True, agent_tokens=vocab).to(dev)
    g = {k: torch.from_numpy(v).to(dev) for k, v in tr.items()}

    tp.materialize_token_traj_all = False     # the (A, 2048, 6, 4, 2) per-agent copy is the reference's layout, not work

    def run():
        return tp._tokenize_agent({'agent': {k: v.clone() for k, v in g.items()}})
    dt = timed(run, n=5, warm=2)
    last = torch.stack([torch.from_numpy(vocab[k][:, -1]) for k in ('veh', 'ped', 'cyc')])
    c = {k: torch.from_numpy(v[:n_cpu]) for k, v in tr.items()}
    t0 = time.perf_counter()
    tm.tokenize_agent(c['valid_mask'], c['position'], c['heading'], c['velocity'], c['type'], c['shape'], last)
    dc = time.perf_counter() - t0
    return {'metric': 'agents tokenised / s (TokenProcessor._tokenize_agent, three launches)', 'value': A / dt, 'unit': 'agents/s',
            'agents': A, 'ms_per_call': dt * 1e3,
            'cpu_baseline': {'value': n_cpu / dc, 'unit': 'agents/s', 'cores': 16, 'kind': 'port',
                             'sample': f'{n_cpu} agents, {dc:.2f} s'}}


def bench_enterings(B=512, A=64, T=18, M=1024, n_cpu=4):
    rng = np.random.default_rng(9)
    pos = (rng.uniform(-90, 90, (B * A, 1, 2)) + np.cumsum(rng.normal(0, 2, (B * A, T, 2)), 1)).astype(np.float32)
    head = rng.uniform(-np.pi, np.pi, (B * A, T)).astype(np.float32)
    state = rng.choice([0, 1, 1, 1, 1, 2, 3], size=(B * A, T)).astype(np.int64)
    batch = np.repeat(np.arange(B), A)
    av = np.full(B, A - 1, np.int64)
    state[A - 1::A] = 1
    ptp = np.concatenate([rng.uniform(-120, 120, (B * M, 2)), np.zeros((B * M, 1))], -1).astype(np.float32)
    ptb = np.repeat(np.arange(B), M)
    tok = Attr_Tokenizer(grid_range=150., grid_interval=3., radius=75., angle_interval=3.)
    t = lambda a: torch.from_numpy(a).to(dev)

    class D(dict):
        num_graphs = B
    data = D(agent=dict(state_idx=t(state), token_pos=t(pos), token_heading=t(head), batch=t(batch), av_index=t(av)),
             pt_token=dict(token_idx=torch.zeros(B * M, dtype=torch.long, device=dev), position=t(ptp), batch=t(ptb)))
    dt = timed(lambda: fetch_enterings(data, tok, 75.0, predict_occ=True))
    c = lambda a, n: torch.from_numpy(a[:n])
    t0 = time.perf_counter()
    eo.fetch_enterings(c(pos, n_cpu * A), c(head, n_cpu * A), c(state, n_cpu * A), c(batch, n_cpu * A), c(av, n_cpu), tok.grid,
                       75.0, 3.0, pt_pos=c(ptp, n_cpu * M), pt_batch=c(ptb, n_cpu * M))
    dc = time.perf_counter() - t0
    cells = B * T * (A + M)
    return {'metric': 'positions gridded / s (InfGen._fetch_enterings, agents + map tokens, 1961 cells each)',
            'value': cells / dt, 'unit': 'positions/s', 'scenes': B, 'agents': A, 'map_tokens': M, 'ms_per_call': dt * 1e3,
            'cpu_baseline': {'value': n_cpu * T * (A + M) / dc, 'unit': 'positions/s', 'cores': 16, 'kind': 'port',
                             'sample': f'{n_cpu} scenes, {dc:.2f} s'}}


def roads(n, seed=3):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        m = int(rng.integers(10, 200))
        h = rng.uniform(-np.pi, np.pi) + np.cumsum(rng.normal(0, 0.1, m))
        xy = rng.uniform(-150, 150, 2) + np.cumsum(np.stack([np.cos(h), np.sin(h)], -1) * rng.uniform(0.5, 2.0, (m, 1)), 0)
        out.append(np.concatenate([xy, np.zeros((m, 1))], -1).astype(np.float32))
    return out


def bench_road(N=128, T=91, n_roads=300, n_cpu=8):
    rng = np.random.default_rng(2)
    head = rng.uniform(-np.pi, np.pi, (N, 1)) + 0.01 * np.arange(T)[None]
    cx = rng.uniform(-120, 120, (N, 1)) + np.cos(head) * np.arange(T)[None] * 0.8
    cy = rng.uniform(-120, 120, (N, 1)) + np.sin(head) * np.arange(T)[None] * 0.8
    f = lambda a: torch.from_numpy(np.ascontiguousarray(np.broadcast_to(a, (N, T)), dtype=np.float32))
    box = dict(center_x=f(cx), center_y=f(cy), center_z=f(0.0), length=f(4.8), width=f(2.0), height=f(1.6), heading=f(head),
               valid=torch.ones(N, T, dtype=torch.bool), evaluated_object_mask=torch.ones(N, dtype=torch.bool))
    rd = roads(n_roads)
    poly, cyc = tensorize_polylines(rd, dev)
    g = {k: v.to(dev) for k, v in box.items()}
    dt = timed(lambda: compute_distance_to_road_edge(road_edge_polylines=(poly, cyc), **g))
    segs = poly.shape[0] * (poly.shape[1] - 1)
    op, oc = mo.tensorize_polylines(rd)
    sub = {k: v[:n_cpu] for k, v in box.items()}
    t0 = time.perf_counter()
    mo.distance_to_road_edge(sub['center_x'], sub['center_y'], sub['center_z'], sub['length'], sub['width'], sub['height'],
                             sub['heading'], sub['valid'], sub['evaluated_object_mask'], op, oc)
    dc = time.perf_counter() - t0
    pairs = N * T * 4 * segs
    return {'metric': 'corner-segment pairs / s (compute_distance_to_road_edge)', 'value': pairs / dt, 'unit': 'pairs/s',
            'objects': N, 'steps': T, 'padded_segments': segs, 'ms_per_call': dt * 1e3,
            'cpu_baseline': {'value': n_cpu * T * 4 * segs / dc, 'unit': 'pairs/s', 'cores': 16, 'kind': 'port',
                             'sample': f'{n_cpu} objects, {dc:.2f} s'}}


def main():
    for fn in (bench_tokenize, bench_enterings, bench_road):
        print(json.dumps(fn()), flush=True)


if __name__ == '__main__':
    main()
