#!/usr/bin/env python
"""Closed-loop rollout throughput on MI355X (BASELINE.json metric: agent-steps/s).

    python bench.py --gpus N --steps K --warmup W

A "step" is ONE full pass of the hot path over the per-GPU batch of synthetic scenes: state
reset, map-encoder prologue, the edgeless column-0 chain and all R/5 decode steps
(reference InfGenDecoder.inference, infgen/modules/infgen_decoder.py:123-130).  Inputs
(scene arrays, packed weights) are resident in HBM before the timed region.

Workload (config.workload): BASELINE config C3 shapes — configs/ours_standard.yaml
hyper-parameters, 64 agents / 1024 map tokens per scene, R = 80 (16 decode steps), greedy
decoding, insertion disabled — with `--scenes` scenes per GPU (weak scaling: every rank owns
its own scenes, no data-path collective; one all-reduce of the timing/counters at the end).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the roofline arithmetic).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from infgen_amd import engine, synth, _lib  # noqa: E402
from infgen_amd import dist as igdist  # noqa: E402

FP32_MATRIX_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md (v_mfma_f32_32x32x2_f32)
# kernels that run on the fp16 matrix pipe with the three-term split: one algorithmic multiply-add costs three
# f16 MFMA multiply-adds, so the ceiling for ALGORITHMIC flops is the dense f16 peak (2.5 PFLOP/s) / 3
F16_SPLIT_PEAK_TFLOPS = 2500.0 / 3.0
SPLIT_KERNELS = ('k_fourier', 'k_attn_pre', 'k_attn_post')
HBM_PEAK_GBS = 8000.0                # HBM3E peak of the same guide


def pmc_traffic(kernel, args):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/traffic.json, written
    from tools/prof_round.sh output), if they were taken on this workload; None otherwise."""
    try:
        with open(os.path.join(REPO, 'profiles', 'traffic.json')) as f:
            t = json.load(f)
    except OSError:
        return None
    same = (t.get('scenes_per_gpu') == args.scenes and t.get('agents') == args.agents and
            t.get('map_tokens') == args.map_tokens and t.get('insertion') == bool(args.insertion))
    k = t.get('kernels', {}).get(kernel)
    return float(k['fetch_bytes_per_launch'] + k['write_bytes_per_launch']) if same and k else None


def load_shapes():
    with open(os.path.join(REPO, 'tests', 'golden', 'state_dict_shapes.json')) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


def build_scenes(cfg, n, agents, map_tokens, first_idx):
    vocab = synth.make_agent_vocab(cfg.token_size)
    map_vocab = synth.make_map_vocab()
    grid = synth.build_grid(cfg.grid_range, cfg.grid_interval, cfg.pl2seed_radius)
    scenes = [synth.make_scene(synth.scene_seed(3, first_idx + i), agents, map_tokens, cfg, half_extent=60.0,
                               ego_last=True, vocab=vocab, grid=grid) for i in range(n)]
    return scenes, vocab, map_vocab, grid


def cpu_baseline(cfg, sd, scene, vocab, map_vocab, grid, budget_s=20.0):
    """The CPU oracle (a port of the reference algorithm, column-wise) timed on this host's cores
    on a bounded sample: whole rollouts of ONE scene of the same workload until ~budget_s."""
    from oracle import rollout_oracle as ro
    if not cfg.disable_insertion:
        from oracle import insertion_oracle as io

        class _Shim:      # same call shape as rollout_oracle.run_scene
            @staticmethod
            def run_scene(tsd, scene, cfg_, vocab_, map_vocab_, grid_):
                return io.run_scene_with_insertion(tsd, scene, cfg_, vocab_, map_vocab_, grid_)
        ro = _Shim
    # torch's intra-op pool degrades badly beyond a few dozen threads on these small operators
    # (256 hardware threads on the GPU box): use 16 and say so in `cores`.
    ncores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(ncores)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    ro.run_scene(tsd, scene, cfg, vocab, map_vocab, grid)       # warm-up (thread pools, allocator)
    t0 = time.perf_counter()
    n = 0
    while True:
        out = ro.run_scene(tsd, scene, cfg, vocab, map_vocab, grid)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    agent_steps = out['pos_a'].shape[0] * cfg.num_recurrent_steps_val * n
    return dict(value=agent_steps / dt, unit='agent-steps/s', cores=ncores, kind='port',
                sample=f'{n} full rollout(s) of 1 scene (A={out["pos_a"].shape[0]}, M={len(scene["pt_token"]["orientation"])}, '
                       f'R={cfg.num_recurrent_steps_val}) incl. map encoder, {dt:.1f} s, torch {torch.get_num_threads()} threads')


def log(msg):
    if os.environ.get('BENCH_VERBOSE'):
        print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--scenes', type=int, default=512, help='scenes per GPU')
    ap.add_argument('--overlap', type=int, default=-1, help='infgen_set_overlap (default: library default)')
    ap.add_argument('--roofline-kernel', default='', help='report the roofline of this kernel instead of the dominant one')
    ap.add_argument('--agents', type=int, default=64)
    ap.add_argument('--map-tokens', type=int, default=1024)
    ap.add_argument('--insertion', action='store_true', help='scenario insertion on (configs/ours_long_term.yaml style)')
    ap.add_argument('--insert-headroom', type=int, default=None,
                    help='rows per scene reserved for inserted agents (default: min(10 per decode step, 96); A_cap <= 1024)')
    ap.add_argument('--rollout-steps', type=int, default=80, help='R = num_recurrent_steps_val (multiple of 5)')
    ap.add_argument('--streams', type=int, default=1, help='split the per-GPU batch over this many HIP streams')
    ap.add_argument('--gemm-terms', type=int, default=3, choices=(1, 3),
                    help='3: fp16 three-term split = fp32 accuracy (default); 1: plain fp16 operands, the reduced-precision mode '
                         'for BASELINE config C5 (outside the 1e-3 parity bar)')
    ap.add_argument('--edge-fuse', type=int, default=-1, choices=(-1, 0, 1, 2),
                    help='infgen_set_edge_fuse: 1 (library default) k_edge_fused from 257 rows, 0 the unfused sequence with U / Z in HBM')
    ap.add_argument('--edge-loop', type=int, default=-1, choices=(-1, 4, 6, 8))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=20.0)
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world,
                                device_id=torch.device('cuda', local_rank))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the product path has no CPU fallback)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    log(f'cpu_count={os.cpu_count()} device={torch.cuda.get_device_name(dev)}')
    cfg = synth.standard_config(disable_insertion=not args.insertion, num_recurrent_steps_val=args.rollout_steps)
    sd = synth.fill_state_dict(load_shapes(), seed=1, rich=True)
    first = igdist.scenes_for_rank_weak(rank, args.scenes)[0]
    scenes, vocab, map_vocab, grid = build_scenes(cfg, args.scenes, args.agents, args.map_tokens, first)
    log('scenes built')
    w = engine.PackedWeights(sd, cfg, dev)
    ns = max(1, args.streams)
    per = (len(scenes) + ns - 1) // ns
    engines = [engine.RolloutEngine(w, scenes[i * per:(i + 1) * per], vocab, map_vocab, grid, store_logits=False,
                                    insert_headroom=args.insert_headroom)
               for i in range(ns) if scenes[i * per:(i + 1) * per]]
    streams = [torch.cuda.Stream(device=dev) for _ in engines] if ns > 1 else [None]

    class _Multi:
        """the per-GPU batch split over several HIP streams (memory-bound edge attention of one group
        overlaps the MFMA-bound kernels of another)"""
        def rollout(self):
            if ns == 1:
                engines[0].rollout()
                return
            cur = torch.cuda.current_stream(dev)
            for e, st in zip(engines, streams):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    e.rollout()
            for st in streams:
                cur.wait_stream(st)

        def agent_steps(self):
            return sum(e.agent_steps() for e in engines)

        @property
        def n_agents(self):
            return torch.cat([e.n_agents for e in engines])

        @property
        def hosts(self):
            return [h for e in engines for h in e.hosts]
    eng = _Multi()

    log('engine built')
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, sd, scenes[0], vocab, map_vocab, grid, args.cpu_budget)

    log(f'cpu baseline done: {cpu}')
    lib = _lib.load()
    if args.overlap >= 0:
        _lib.check(lib.infgen_set_overlap(args.overlap))
    _lib.check(lib.infgen_set_gemm_terms(args.gemm_terms))
    if args.edge_loop >= 0:
        _lib.check(lib.infgen_set_edge_loop(args.edge_loop))
    if args.edge_fuse >= 0:
        _lib.check(lib.infgen_set_edge_fuse(args.edge_fuse))
    for _ in range(args.warmup):
        eng.rollout()
        torch.cuda.synchronize(dev)
        log('warmup rollout done')
    # roofline leg 1 (untimed): one rollout with HIP events around EVERY kernel -> which kernel dominates
    _lib.prof_enable((1 << len(_lib.KERNEL_IDS)) - 1)
    eng.rollout()
    per_kernel = _lib.prof_collect()
    dominant = args.roofline_kernel or max(per_kernel, key=lambda k: per_kernel[k]['ms'])
    log('per-kernel ms of one rollout: ' + ', '.join(f'{k}={v["ms"]:.2f}' for k, v in per_kernel.items()))
    # roofline leg 2: events only around the dominant kernel's launches, inside the timed region
    _lib.prof_enable(1 << _lib.KERNEL_IDS.index(dominant))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.rollout()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    dom = _lib.prof_collect()[dominant]
    _lib.prof_enable(0)
    roof = None
    common = {'launches': dom['calls'],
              'share_of_gpu_time': per_kernel[dominant]['ms'] / max(1e-9, sum(v['ms'] for v in per_kernel.values())),
              'per_kernel_ms_one_rollout': {k: round(v['ms'], 3) for k, v in per_kernel.items()}}
    if dom['calls'] > 0 and dominant == 'k_edge_attn':
        # HBM-bound gather kernel: compulsory bytes (DESIGN.md "Measurement"): 512 B of rhat per edge; per destination
        # row q 512 + u 4096 in, agg 512 + z 4096 + sigma 32 out; K and V (1 KB) once per DISTINCT source row of a
        # launch: every temporal edge has its own source, the agent set re-reads the rows of its own launch, the map
        # set at most the map tokens of the batch.
        L = cfg.num_agent_layers
        ed = {k: v * L for k, v in dom['edges_built'].items()}      # each set feeds one launch per layer
        rows = args.scenes * engines[0].A_cap
        rows_total = float(dom['calls']) * rows
        launches_per_kind = dom['calls'] / 3.0
        e_all = sum(ed.values())
        kv = 1024.0 * (ed['temporal'] + launches_per_kind * rows
                       + min(ed['map'], launches_per_kind * args.scenes * args.map_tokens))
        nbytes = 512.0 * e_all + 9248.0 * rows_total + kv
        avg_s = dom['ms'] * 1e-3 / dom['calls']
        ach = nbytes / (dom['ms'] * 1e-3) / 1e9
        roof = {'bound': 'hbm', 'kernel': dominant, 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': ach / HBM_PEAK_GBS, 'traffic': None, 'avg_launch_us': avg_s * 1e6,
                'bytes_per_launch': nbytes / dom['calls'], 'edges_per_launch': e_all / dom['calls'], **common}
    elif dom['calls'] > 0 and dom['macs'] > 0:
        avg_s = dom['ms'] * 1e-3 / dom['calls']
        flops_per_launch = 2.0 * dom['macs'] / dom['calls']
        ach = flops_per_launch / avg_s / 1e12
        split = dominant in SPLIT_KERNELS
        peak = (F16_SPLIT_PEAK_TFLOPS if args.gemm_terms == 3 else 2500.0) if split else FP32_MATRIX_PEAK_TFLOPS
        roof = {'bound': 'mfma', 'kernel': dominant, 'achieved': ach, 'peak': peak,
                'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': None,
                'arithmetic': 'fp16 MFMA, three-term hi/lo split (peak = 2500 / 3)' if split else 'fp32-input MFMA',
                'avg_launch_us': avg_s * 1e6, 'flops_per_launch': flops_per_launch, **common}

    if roof is not None:
        roof['traffic'] = pmc_traffic(dominant, args)

    agent_steps = float(eng.agent_steps() * args.steps)
    inserted = int((eng.n_agents.sum().item() - sum(h['A'] for h in eng.hosts))) if args.insertion else 0
    dt, agent_steps = igdist.reduce_run(dt, agent_steps, dev)
    if rank == 0:
        line = {
            'metric': 'agent-steps/sec (closed-loop rollout)',
            'value': agent_steps / dt,
            'unit': 'agent-steps/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32' if args.gemm_terms == 3 else 'f16',
            'data': 'synthetic',
            'config': {
                'workload': f'C3 shapes: configs/ours_standard.yaml, {args.agents} agents / {args.map_tokens} map tokens '
                            f'per scene, R={args.rollout_steps} ({cfg.num_decode_steps} decode steps), greedy, '
                            f'insertion {"on" if args.insertion else "disabled"}, '
                            f'{args.scenes} scenes per GPU, one step = reset + map encoder + full rollout',
                'scenes_per_gpu': args.scenes, 'streams': ns, 'gemm_terms': args.gemm_terms, 'agents': args.agents, 'map_tokens': args.map_tokens,
                'decode_steps': cfg.num_decode_steps, 'insertion': bool(args.insertion), 'rows_per_scene': engines[0].A_cap, 'agents_inserted_last_rollout': inserted, 'scenes_at_row_cap': sum(e.scenes_at_row_cap() for e in engines), 'agent_token_steps_per_s': agent_steps / dt / cfg.shift,
                'parallelism': f'scenes sharded over {world} rank(s), no data-path collective',
            },
            'roofline': roof,
            'cpu_baseline': cpu,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
