#!/usr/bin/env python
"""Closed-loop rollout throughput on MI355X (BASELINE.json metric: agent-steps/s).

    python bench.py --gpus N --steps K --warmup W

A "step" is ONE full pass of the hot path over the per-GPU batch of synthetic scenes: state
reset, map-encoder prologue, the edgeless column-0 chain and all R/5 decode steps
(reference InfGenDecoder.inference, infgen/modules/infgen_decoder.py:123-130).  Inputs
(scene arrays, packed weights) are resident in HBM before the timed region.

Workload (config.workload): BASELINE config C3 shapes by default — configs/ours_standard.yaml
hyper-parameters, 64 agents / 1024 map tokens per scene, R = 80 (16 decode steps), greedy
decoding, insertion disabled — with `--scenes` scenes per GPU (weak scaling: every rank owns
its own scenes) or, with `--scaling strong`, the literal BASELINE batch of `--total-scenes` (64)
scenes dealt to the ranks like the reference's DistributedSampler.  No data-path collective; one
all-reduce of the timing / counters at the end.

Multi-GPU: `--gpus N` with N > 1 and no WORLD_SIZE in the environment re-launches itself through
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU, backend nccl =
RCCL); under an external torchrun the ranks are taken from the environment and must equal `--gpus`.  `--dry-run` drives the
same rank path on CPU with the gloo backend and no kernels (tests/test_dist_cpu.py).

Besides the headline value the line carries `config.c3_literal`: BASELINE C3's literal batch (64 scenes in total) dealt to
the ranks like the reference's DistributedSampler, and - at N = 1 - the 8-scene leg one GPU of an 8-way shard would run.

Prints ONE JSON line on rank 0.  `roofline` follows SURVEY section 8d: algorithmic bytes and FLOPs
(the figures of 8d, evaluated on the edge counts the device reports) over measured time, for the
dominant kernel and for the whole step; DESIGN.md "Measurement" has the arithmetic.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from infgen_amd import engine, synth, _lib  # noqa: E402
from infgen_amd import dist as igdist  # noqa: E402

# peaks: /opt/skills/guides/MI355X_MICROARCH.md
PROF_STRIDE = 5            # of the dominant kernel's decode-step launches in the timed region every fifth carries HIP events
HBM_PEAK_GBS = 8000.0
FP32_MATRIX_PEAK_TFLOPS = 157.3     # v_mfma_f32_32x32x2_f32
F16_DENSE_PEAK_TFLOPS = 2500.0
# kernels on the fp16 matrix pipe with the three-term split: one algorithmic multiply-add costs three f16 MFMA
# multiply-adds, so the ceiling for ALGORITHMIC flops at fp32 accuracy is the dense f16 peak / 3
F16_SPLIT_PEAK_TFLOPS = F16_DENSE_PEAK_TFLOPS / 3.0
# what "f32" means on the line: every tensor in HBM is fp32 (weights, activations, K / V, the relative-position rows), every
# accumulation and every vector operation is fp32; inside the hot GEMM kernels an fp32 operand x enters the f16 matrix pipe as
# the exact pair hi = rne16(x), lo = rne16(x - hi) (|x - hi - lo| <= 2^-23 |x|) and a product is hi hi + hi lo + lo hi in fp32
ARITHMETIC_F32 = ('fp32 storage, fp32 accumulation; GEMM operands enter the f16 matrix pipe as round-to-nearest hi + lo fp16 pairs '
                  '(<= 2^-23 per operand), three MFMA terms per product; error against fp64 measured 0.5 - 1.1 x (bar 1.25 x) the fp32-input MFMA '
                  'kernels\' on every operator (tests/test_precision_gpu.py, profiles/r06_precision.json); fp32 rhat rows')

# ---- SURVEY section 8d, per decode step of one scene (1 MAC = 2 FLOP, D = 128) ----------------------------------
NODE_MAC_PER_ROW = 4_702_848        # 18 x 212,992 + 12 x 32,768 + 82,176 + 98,304 + 278,528 + 16,768
EDGE_MAC_TEMPORAL = 346_112         # 147,968 (Fourier, 4 dims) + 6 x 33,024
EDGE_MAC_OTHER = 313_216            # 115,072 (Fourier, 3 dims) + 6 x 33,024
EDGE_MAC_PER_LAYER = 33_024         # the reference's per-edge W_kr / W_vr projections + the two dot products of one layer
GRID_FLOP_PER_ROW = 1961 * 4
KV_ROW_BYTES = 1024                 # one source row's K and V (fp32)
ROW_FIXED_BYTES = 6144 + 12288 + 512 + 2048 + 192 + 80     # K/V written (temporal), K/V written + read (agent set), state, embedding rows, template, outputs
WEIGHT_BYTES_PER_STEP = 23.7e6      # motion-path weights, read once per decode step per GPU


def c3_shapes(args):
    return args.agents == 64 and args.map_tokens == 1024 and args.rollout_steps == 80


def load_shapes():
    with open(os.path.join(REPO, 'tests', 'golden', 'state_dict_shapes.json')) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


def build_scenes(cfg, indices, agents, map_tokens):
    vocab = synth.make_agent_vocab(cfg.token_size)
    map_vocab = synth.make_map_vocab()
    grid = synth.build_grid(cfg.grid_range, cfg.grid_interval, cfg.pl2seed_radius)
    scenes = [synth.make_scene(synth.scene_seed(3, i), agents, map_tokens, cfg, half_extent=60.0,
                               ego_last=True, vocab=vocab, grid=grid) for i in indices]
    return scenes, vocab, map_vocab, grid


# ---------------------------------------------------------------------------------------- CPU baseline
def _cpu_worker(job):
    """one process of the CPU baseline: whole rollouts of ONE scene with the oracle until the budget is used"""
    idx, agents, map_tokens, R, insertion, budget_s, threads, all_columns = job
    import torch as th
    th.set_num_threads(threads)
    from oracle import rollout_oracle as ro
    cfg = synth.standard_config(disable_insertion=not insertion, num_recurrent_steps_val=R)
    sd = synth.fill_state_dict(load_shapes(), seed=1, rich=True)
    scenes, vocab, map_vocab, grid = build_scenes(cfg, [idx], agents, map_tokens)
    tsd = {k: th.from_numpy(v) for k, v in sd.items()}
    if insertion:
        from oracle import insertion_oracle as io
        run = lambda: io.run_scene_with_insertion(tsd, scenes[0], cfg, vocab, map_vocab, grid)
    else:
        run = lambda: ro.run_scene(tsd, scenes[0], cfg, vocab, map_vocab, grid, all_columns=all_columns)
    if not all_columns:
        run()                               # warm-up (thread pools, allocator)
    t0 = time.perf_counter()
    n, steps = 0, 0
    while True:
        out = run()
        n += 1
        steps += out['pos_a'].shape[0] * R
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    return steps, dt, n


def cpu_baseline(args, budget_s):
    """The CPU oracle (a column-wise port of the reference algorithm) on the host's cores.  Scenes are independent, so - like
    the reference under DDP - several processes run whole rollouts of their own scene of the same workload.  Two layouts are
    timed (half of the budget each) and the faster one is reported with the threads it used: one process with 16 torch threads,
    and one process per 16 hardware threads with 8 torch threads each (torch's intra-op pool does not scale on these small
    operators, and 256 busy threads on the box's 256 hardware threads ran 2x slower than 16)."""
    import multiprocessing as mp
    ncpu = os.cpu_count() or 1
    ctx = mp.get_context('spawn')
    layouts = [(1, min(16, ncpu))]
    if ncpu >= 32:
        layouts.append((ncpu // 16, 8))
    best, notes = None, []
    # "reference-shaped" variant: the same port with the reference's control flow - all A*T nodes through the 18
    # layers at every step (agent_decoder.py:2133-2158) - one process x 16 threads, whole rollouts for a fifth of the budget
    ref_shaped = None
    if not args.insertion:
        with ctx.Pool(1) as pool:
            steps, busy, n = pool.map(_cpu_worker, [(0, args.agents, args.map_tokens, args.rollout_steps, False, budget_s / 5,
                                                     min(16, ncpu), True)])[0]
        ref_shaped = dict(value=steps / busy, unit='agent-steps/s', cores=min(16, ncpu), kind='reference-shaped',
                          sample=f'{n} full rollout(s) of one scene, {busy:.1f} s: the port with the reference\'s all-columns '
                                 f'recompute per step (agent_decoder.py:2133-2158)')
        budget_s *= 0.8
    for procs, threads in layouts:
        jobs = [(i, args.agents, args.map_tokens, args.rollout_steps, bool(args.insertion), budget_s / len(layouts), threads, False)
                for i in range(procs)]
        with ctx.Pool(procs) as pool:
            res = pool.map(_cpu_worker, jobs)
        steps, busy, n = sum(r[0] for r in res), max(r[1] for r in res), sum(r[2] for r in res)
        notes.append(f'{procs} process(es) x {threads} threads: {steps / busy:.0f} agent-steps/s ({n} rollouts, {busy:.1f} s)')
        if best is None or steps / busy > best[0]:
            best = (steps / busy, procs * threads)
    return dict(value=best[0], unit='agent-steps/s', cores=best[1], kind='port',
                sample=f'full rollouts incl. map encoder of one scene per process (A={args.agents}, M={args.map_tokens}, '
                       f'R={args.rollout_steps}), windowed port of the reference algorithm (oracle/); host has {ncpu} hardware '
                       f'threads; ' + '; '.join(notes), reference_shaped=ref_shaped)


# ---------------------------------------------------------------------------------------- parity gate
PARITY_COPIES = 336      # x 32 rows per scene = 10,752 rows: above every by-size switch of the library (k_edge_fused<6,false,1,8> from
                         # 4,097 rows, k_attn_h / k_heads_h / k_mlpemb_h from 10,241 rows) - the kernels the number is measured on


def parity_gate(dev):
    """parity reported with every perf number - the C1 fixture and the A = 24 edge-case fixture (outputs of the
    reference's own InfGenDecoder.inference) free-running through the library that is about to be timed: once as the single
    scene (the small-launch kernels) and once as a batch of PARITY_COPIES copies of the scene, which takes the launches
    through the kernels of the timed region (large-batch variants, fp32 rhat rows); every copy must reproduce the
    fixture"""
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from conftest import load_case
    res = {}
    for name in ('c1_a8_m128', 'a24_m256_edge'):
        c = load_case(name)
        z = c['z']
        w = engine.PackedWeights(c['sd'], c['cfg'], dev)
        tol = 1e-3                                  # (flat, also on the x64-sharpened heads of these fixtures)
        r = {}
        ref_tok = torch.from_numpy(z['next_token_idx'].astype(np.int64)).to(dev)
        ref_st = torch.from_numpy(z['next_state_idx'].astype(np.int64)).to(dev)
        ref_lg = torch.from_numpy(z['logits']).to(dev)
        for tag, copies in (('single', 1), ('timed_kernels', PARITY_COPIES)):
            eng = engine.RolloutEngine(w, [c['scene']] * copies, c['vocab'], c['map_vocab'], c['grid'], store_logits=True,
                                       use_graph=False)
            eng.rollout()
            outs = eng.outputs_device()
            r[tag] = dict(rows=eng.rows,
                          tokens_exact=all(bool(torch.equal(o['next_token_idx'], ref_tok)) for o in outs),
                          states_exact=all(bool(torch.equal(o['next_state_idx'], ref_st)) for o in outs),
                          logits_max_abs_err=max(float((o['logits'] - ref_lg).abs().max().item()) for o in outs),
                          logits_tol=tol)
            del eng, outs
        res[name] = r
    res['ok'] = all(v['tokens_exact'] and v['states_exact'] and v['logits_max_abs_err'] <= v['logits_tol']
                    for r in res.values() for v in r.values())
    return res


def pmc_traffic(kernel, args):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/traffic.json, written
    from tools/prof_round.sh output), if they were taken on this workload; None otherwise."""
    try:
        with open(os.path.join(REPO, 'profiles', 'traffic.json')) as f:
            t = json.load(f)
    except OSError:
        return None
    same = (t.get('scenes_per_gpu') == args.scenes and t.get('agents') == args.agents and
            t.get('map_tokens') == args.map_tokens and t.get('insertion') == bool(args.insertion) and
            t.get('rollout_steps', 80) == args.rollout_steps and args.scaling == 'weak')
    k = t.get('kernels', {}).get(kernel)
    return float(k['fetch_bytes_per_launch'] + k['write_bytes_per_launch']) if same and k else None


def log(msg):
    if os.environ.get('BENCH_VERBOSE'):
        print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------------------- ranks
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`--gpus N` without an external launcher: one rank per GPU through torch.distributed.run (the reference's
    `--devices N` -> Lightning DDP, run.py:72,130); returns the launcher's exit code"""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    return subprocess.call(cmd, env=env)


class Ranks:
    """the process group of a run: nccl (= RCCL) on GPUs, gloo for --dry-run; a single process has no group"""

    def __init__(self, args):
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        if self.world != args.gpus:
            raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started {self.world} rank(s)')
        self.dry = bool(args.dry_run)
        self.dist = None
        if self.dry:
            self.dev = torch.device('cpu')
        else:
            assert torch.cuda.is_available(), 'bench.py needs a GPU (the product path has no CPU fallback)'
            assert torch.cuda.device_count() > self.local_rank, \
                f'rank {self.rank}: cuda:{self.local_rank} does not exist ({torch.cuda.device_count()} device(s) visible)'
            torch.cuda.set_device(self.local_rank)
            self.dev = torch.device('cuda', self.local_rank)
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if self.dry:
                dist.init_process_group('gloo', rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group('nccl', rank=self.rank, world_size=self.world, device_id=self.dev)
            assert dist.get_world_size() == args.gpus
            self.dist = dist
        self.backend = 'none' if self.dist is None else 'gloo' if self.dry else 'nccl (RCCL)'

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def sync(self):
        if not self.dry:
            torch.cuda.synchronize(self.dev)

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def timed(ranks, fn, steps):
    """EXACTLY `steps` calls of fn between barrier + synchronize on both sides; returns this rank's seconds"""
    ranks.barrier()
    ranks.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    ranks.sync()
    ranks.barrier()
    return time.perf_counter() - t0


def dry_run(args, ranks):
    """the rank path of a run without kernels: scene sharding, barriers, the closing reductions - on gloo"""
    mine = (igdist.scenes_for_rank_strided(ranks.rank, ranks.world, args.total_scenes) if args.scaling == 'strong'
            else igdist.scenes_for_rank_weak(ranks.rank, args.scenes))
    literal = igdist.scenes_for_rank_strided(ranks.rank, ranks.world, 64)
    balance = None
    if args.insertion and ranks.world > 1 and not args.no_balance:
        # the pilot's cost exchange and the cost-sorted re-deal of a real insertion run, on synthetic per-scene costs
        total = args.total_scenes if args.scaling == 'strong' else ranks.world * args.scenes
        local = [(i, igdist.scene_cost(args.agents + (37 * (i + 1)) % 53, args.map_tokens)) for i in mine]
        costs = igdist.gather_costs(local, total, ranks.dev)
        dealt = [igdist.scenes_for_rank_strided(r, ranks.world, total) if args.scaling == 'strong'
                 else igdist.scenes_for_rank_weak(r, args.scenes) for r in range(ranks.world)]
        parts = [igdist.scenes_for_rank_balanced(costs, r, ranks.world) for r in range(ranks.world)]
        balance = {'max_over_mean_before': igdist.partition_spread(costs, dealt),
                   'max_over_mean_after': igdist.partition_spread(costs, parts), 'scenes_per_rank': [len(p_) for p_ in parts],
                   'all_scenes_dealt_once': sorted(i for p_ in parts for i in p_) == list(range(total))}
        mine = parts[ranks.rank]
    dt = timed(ranks, lambda: time.sleep(0.002 * (ranks.rank + 1)), args.steps)
    steps_local = float(len(mine) * args.agents * args.rollout_steps * args.steps)
    per_rank = igdist.gather_metrics([1e3 * dt / args.steps, float(len(mine)), float(len(literal))], ranks.dev)
    dt, agent_steps = igdist.reduce_run(dt, steps_local, ranks.dev)
    if ranks.rank == 0:
        print(json.dumps({'metric': 'agent-steps/sec (closed-loop rollout)', 'value': None, 'unit': 'agent-steps/s',
                          'n_gpus': ranks.world, 'rccl_ranks': 0, 'ranks': ranks.world, 'backend': ranks.backend, 'dry_run': True, 'steps': args.steps,
                          'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'scaling': args.scaling,
                          'agent_steps_counted': agent_steps, 'per_rank_ms': [r[0] for r in per_rank],
                          'scenes_per_rank': [int(r[1]) for r in per_rank],
                          'c3_literal_scenes_per_rank': [int(r[2]) for r in per_rank], 'insertion_balance': balance}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--scenes', type=int, default=1024, help='scenes per GPU (weak scaling; 512 until round 4: 25.7 M, 1024: 27.2 M)')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help='strong: a fixed batch of --total-scenes scenes dealt to the ranks (scene i -> rank i mod N)')
    ap.add_argument('--total-scenes', type=int, default=64, help='batch size of --scaling strong (BASELINE C3: 64)')
    ap.add_argument('--overlap', type=int, default=-1, help='infgen_set_overlap (default: library default)')
    ap.add_argument('--roofline-kernel', default='', help='report the roofline of this kernel instead of the dominant one')
    ap.add_argument('--agents', type=int, default=64)
    ap.add_argument('--map-tokens', type=int, default=1024)
    ap.add_argument('--insertion', action='store_true', help='scenario insertion on (configs/ours_long_term.yaml style)')
    ap.add_argument('--insert-headroom', type=int, default=None,
                    help='rows per scene reserved for inserted agents (default: min(10 per decode step, 96), doubled until the '
                         'rollout fits; A_cap <= 1024)')
    ap.add_argument('--rollout-steps', type=int, default=80, help='R = num_recurrent_steps_val (multiple of 5)')
    ap.add_argument('--streams', type=int, default=1, help='split the per-GPU batch over this many HIP streams')
    ap.add_argument('--gemm-terms', type=int, default=3, choices=(1, 2, 3),
                    help='3: round-to-nearest hi + lo fp16 operand split, three MFMA terms = fp32 arithmetic (default); the reduced-precision '
                         'modes for BASELINE config C5 (outside the 1e-3 parity bar): 2 = bf16 operands (weights and activations rounded to '
                         '8 significant bits, fp32 accumulation), 1 = fp16 operands')
    ap.add_argument('--attn-mode', type=int, default=-1, choices=(-1, 0, 1, 2, 3),
                    help='infgen_set_attn_mode: 2 (library default) node kernels by launch size, 1 the split kernels on 64-row tiles always')
    ap.add_argument('--edge-fuse', type=int, default=-1, choices=(-1, 0, 1, 2),
                    help='infgen_set_edge_fuse: 1 (library default) k_edge_fused from 257 rows, 0 the unfused sequence with U / Z in HBM')
    ap.add_argument('--edge-loop', type=int, default=-1, choices=(-1, 4, 6, 8))
    ap.add_argument('--edge-kernel', type=int, default=-1, choices=(-1, 0, 1, 2),
                    help='infgen_set_edge_kernel: 0 k_edge_fused, 1 k_edge_fused3 for launches beyond 4 k rows, 2 k_edge_fused3 always')
    ap.add_argument('--graph', type=int, default=-1, choices=(-1, 0, 1, 2),
                    help='replay the decode steps of a rollout from a captured HIP graph: 1 always, 0 never, -1 (default) the '
                         'engine\'s rule (off unless INFGEN_GRAPH=1)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--no-literal', action='store_true', help='skip the config.c3_literal legs')
    ap.add_argument('--no-strict', action='store_true', help='skip the secondary legs (fp32_mfma, rhat24, two_streams, multi_rollout, prologue)')
    ap.add_argument('--no-balance', action='store_true',
                    help='insertion on several ranks: keep the initial deal instead of the cost-sorted one (dist.scenes_for_rank_balanced)')
    ap.add_argument('--cpu-budget', type=float, default=20.0)
    ap.add_argument('--dry-run', action='store_true',
                    help='the rank path only (launcher, sharding, barriers, reductions) on CPU with the gloo backend; no kernels')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    ranks = Ranks(args)
    rank, world, dev = ranks.rank, ranks.world, ranks.dev
    if args.dry_run:
        dry_run(args, ranks)
        ranks.close()
        return

    log(f'cpu_count={os.cpu_count()} device={torch.cuda.get_device_name(dev)}')
    # the CPU baseline runs first (rank 0, N = 1 only), before this process holds GPU work
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, args.cpu_budget)
        log(f'cpu baseline done: {cpu}')

    cfg = synth.standard_config(disable_insertion=not args.insertion, num_recurrent_steps_val=args.rollout_steps)
    sd = synth.fill_state_dict(load_shapes(), seed=1, rich=True)
    if args.scaling == 'strong':
        mine = igdist.scenes_for_rank_strided(rank, world, args.total_scenes)
    else:
        mine = igdist.scenes_for_rank_weak(rank, args.scenes)
    scenes, vocab, map_vocab, grid = build_scenes(cfg, mine, args.agents, args.map_tokens)
    log(f'{len(scenes)} scenes built')
    lib = _lib.load()
    if args.overlap >= 0:
        _lib.check(lib.infgen_set_overlap(args.overlap))
    _lib.check(lib.infgen_set_gemm_terms(args.gemm_terms))
    if args.attn_mode >= 0:
        _lib.check(lib.infgen_set_attn_mode(args.attn_mode))
    if args.edge_loop >= 0:
        _lib.check(lib.infgen_set_edge_loop(args.edge_loop))
    if args.edge_fuse >= 0:
        _lib.check(lib.infgen_set_edge_fuse(args.edge_fuse))
    if args.edge_kernel >= 0:
        _lib.check(lib.infgen_set_edge_kernel(args.edge_kernel))

    parity = None
    if rank == 0 and not args.no_parity and args.gemm_terms == 3:
        parity = parity_gate(dev)
        log(f'parity: {parity}')

    w = engine.PackedWeights(sd, cfg, dev, operand_bits=8 if args.gemm_terms == 2 else 11)
    ns = max(1, args.streams)
    per = (len(scenes) + ns - 1) // ns
    use_graph = None if args.graph < 0 else ('all' if args.graph == 2 else bool(args.graph))       # 2: the whole rollout as one graph

    def make_engines(headroom):
        return [engine.RolloutEngine(w, scenes[i * per:(i + 1) * per], vocab, map_vocab, grid, store_logits=False,
                                     insert_headroom=headroom, use_graph=use_graph)
                for i in range(ns) if scenes[i * per:(i + 1) * per]]
    engines = make_engines(args.insert_headroom)
    streams = [torch.cuda.Stream(device=dev) for _ in engines] if ns > 1 else [None]

    class _Multi:
        """the per-GPU batch split over several HIP streams (memory-bound edge attention of one group
        overlaps the MFMA-bound kernels of another)"""
        def rollout(self):
            engine.rollout_many(engines, streams if ns > 1 else None)

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

    # warm-up; with insertion the row head-room is doubled until the (deterministic) rollout fits - an engine never drops
    # an insertion silently
    def warm_up():
        done = 0
        while done < max(args.warmup, 1 if args.insertion else 0):
            try:
                eng.rollout()
                torch.cuda.synchronize(dev)
                done += 1
            except engine.InsertionHeadroomError:
                limit = lib.infgen_layout_query(_lib.Q_MAX_AGENTS)
                if engines[0].A_cap >= limit:
                    raise
                engines[:] = make_engines(min(2 * engines[0].A_cap, limit) - args.agents)
                log(f'insertion head-room exhausted: rows per scene -> {engines[0].A_cap}')
                done = 0
            log('warmup rollout done')
    warm_up()
    # insertion on several ranks: per-scene cost is data dependent.  The warm-up rollout is the pilot: every rank
    # contributes the agents its scenes ended with, the cost vector is all-reduced, and the scenes are dealt again longest
    # first (dist.scenes_for_rank_balanced) - a control-plane exchange before the timed region, none inside it
    balance = None
    if args.insertion and world > 1 and not args.no_balance:
        total = args.total_scenes if args.scaling == 'strong' else world * args.scenes
        fin = eng.n_agents.tolist()
        local = [(i, igdist.scene_cost(n, args.map_tokens)) for i, n in zip(mine, fin)]
        costs = igdist.gather_costs(local, total, dev)
        dealt = [igdist.scenes_for_rank_strided(r, world, total) if args.scaling == 'strong' else igdist.scenes_for_rank_weak(r, args.scenes)
                 for r in range(world)]
        parts = [igdist.scenes_for_rank_balanced(costs, r, world) for r in range(world)]
        balance = {'cost': 'agents at the end of the pilot rollout x (1 + map tokens / 8192)',
                   'max_over_mean_before': igdist.partition_spread(costs, dealt),
                   'max_over_mean_after': igdist.partition_spread(costs, parts), 'scenes_per_rank': [len(p_) for p_ in parts]}
        log(f'balanced dealing: {balance}')
        mine = parts[rank]
        del engines[:]
        torch.cuda.empty_cache()
        scenes, vocab, map_vocab, grid = build_scenes(cfg, mine, args.agents, args.map_tokens)
        per = (len(scenes) + ns - 1) // ns
        engines[:] = make_engines(args.insert_headroom)
        warm_up()
    # roofline leg 1 (untimed): one rollout with HIP events around EVERY kernel -> which kernel dominates
    # (an engine that replays a HIP graph runs this rollout eagerly: events are not part of the captured graph)
    _lib.prof_enable((1 << len(_lib.KERNEL_IDS)) - 1)
    eng.rollout()
    per_kernel = _lib.prof_collect()
    dominant = args.roofline_kernel or max(per_kernel, key=lambda k: per_kernel[k]['ms'])
    log('per-kernel ms of one rollout: ' + ', '.join(f'{k}={v["ms"]:.2f}' for k, v in per_kernel.items()))
    # roofline leg 2: events only around the dominant kernel's launches, inside the timed region - and only around every
    # PROF_STRIDE-th of its decode-step launches (an event pair costs ~5 us of launch-stream time: ~3 ms per rollout with all 306
    # bracketed).  5 is coprime to the 18 launches of a step (6 layers x temporal, map, agent): every position is sampled equally often
    _lib.prof_enable(1 << _lib.KERNEL_IDS.index(dominant))
    _lib.prof_set_stride(PROF_STRIDE)
    dt_local = timed(ranks, eng.rollout, args.steps)
    log(f'timed region done: {1e3 * dt_local / args.steps:.2f} ms per step')
    dt = dt_local
    timed_k = _lib.prof_collect()
    seen_k = _lib.prof_seen()[dominant]
    dom = timed_k[dominant]
    _lib.prof_enable(0)

    # ---- algorithmic work of the timed rollouts (this rank), from the edge totals the device counted
    L = cfg.num_agent_layers
    steps_dec = cfg.num_decode_steps
    ed1 = per_kernel['k_edge_attn']['edges_built']                       # one rollout (leg 1)
    rows_dec = float(sum(h['A'] for h in eng.hosts)) * steps_dec          # decoded rows (agent-token-steps) of one rollout
    if args.insertion:
        rows_dec = float(eng.n_agents.sum().item()) * steps_dec           # upper bound: the final agent count on every step
    f_alg = 2.0 * (rows_dec * NODE_MAC_PER_ROW + ed1['temporal'] * EDGE_MAC_TEMPORAL +
                   (ed1['map'] + ed1['agent']) * EDGE_MAC_OTHER) + rows_dec * GRID_FLOP_PER_ROW
    b_alg = (L * KV_ROW_BYTES * (ed1['temporal'] + ed1['map']) + rows_dec * ROW_FIXED_BYTES +
             steps_dec * (WEIGHT_BYTES_PER_STEP + 8.0 * args.map_tokens * len(scenes)))
    t_roll = dt / args.steps
    mm_peak = F16_SPLIT_PEAK_TFLOPS if args.gemm_terms == 3 else F16_DENSE_PEAK_TFLOPS
    step_roof = {'hbm_fraction': b_alg / t_roll / (HBM_PEAK_GBS * 1e9),
                 'mfma_fraction': f_alg / t_roll / (mm_peak * 1e12),
                 'mfma_fraction_vs_fp32_matrix_peak': f_alg / t_roll / (FP32_MATRIX_PEAK_TFLOPS * 1e12),
                 'algorithmic_gbytes_per_rollout': b_alg / 1e9, 'algorithmic_gflop_per_rollout': f_alg / 1e9,
                 'edges_built_per_rollout': {k: float(ed1[k]) for k in ('temporal', 'map', 'agent')},
                 'note': 'SURVEY 8d figures (windowed K/V, every datum moved once; reference per-edge FLOPs) over the measured '
                         'rollout time incl. the map-encoder prologue'}
    step_roof['bound'] = 'hbm' if step_roof['hbm_fraction'] >= step_roof['mfma_fraction'] else 'mfma'

    roof = None
    # the other GEMM kernels of the rollout by the same rule (leg 1's events around every launch: algorithmic FLOPs of the launches -
    # DESIGN.md 5.2's per-unit figures x the units the device counted - over their summed duration, against the peak of their arithmetic)
    total_ms = max(1e-9, sum(v['ms'] for v in per_kernel.values()))
    others = {}
    big = len(scenes) * engines[0].A_cap > 10240              # (the by-size rule of the node-side kernels: split arithmetic beyond 10,240 rows)
    for k in ('k_fourier', 'k_attn_post', 'k_attn_pre', 'k_heads'):
        v = per_kernel.get(k)
        if not v or v['calls'] <= 0 or v['macs'] <= 0 or k == dominant:
            continue
        split = k == 'k_fourier' or big
        peak = mm_peak if split else FP32_MATRIX_PEAK_TFLOPS
        tf = 2.0 * v['macs'] / (v['ms'] * 1e-3) / 1e12
        others[k] = {'bound': 'mfma', 'achieved': tf, 'peak': peak, 'unit': 'TFLOP/s', 'frac': tf / peak, 'launches': v['calls'],
                     'avg_launch_us': 1e3 * v['ms'] / v['calls'], 'share_of_gpu_time': v['ms'] / total_ms}
    common = {'launches': dom['calls'], 'avg_launch_us': 1e3 * dom['ms'] / max(1, dom['calls']), 'other_kernels': others,
              'share_of_gpu_time': per_kernel[dominant]['ms'] / max(1e-9, sum(v['ms'] for v in per_kernel.values())),
              'per_kernel_ms_one_rollout': {k: round(v['ms'], 3) for k, v in per_kernel.items()}, 'step': step_roof}
    if dom['calls'] > 0 and dominant == 'k_edge_attn':
        # the edge kernel (k_edge_fused), one launch per sublayer: HBM-bound.  SURVEY 8d's bytes of a decode-step
        # launch: 1 KB of K / V per temporal or map edge (every edge has its own source row), 1 KB per decoded row for the agent
        # set (a scene's K / V rows are shared by its rows).  Only the launches issued INSIDE decode steps are priced - the
        # profiler tags them (infgen_prof_collect_steps); the map encoder's pt <-> pt launches and the edgeless column-0 chain of
        # the prologue are reported next to them, their edges are not in the numerator.
        ed = timed_k['k_edge_attn']['edges_built']                        # all timed rollouts (decode steps only: the edgeless
        rows_t = rows_dec * args.steps                                    # column-0 chain builds none)
        nedge = ed['temporal'] + ed['map'] + ed['agent']
        nbytes = L * KV_ROW_BYTES * (ed['temporal'] + ed['map'] + rows_t)
        calls, secs = max(1, dom['step_calls']), max(1e-9, dom['step_ms'] * 1e-3)
        # (calls / secs: the bracketed launches; nedge / nbytes / rows_t are scaled to them - the sample is uniform over the step's
        # positions, so bytes per bracketed launch = bytes per launch)
        in_region = max(calls, seen_k['seen_step'])
        sampled = calls / in_region
        nedge, nbytes, rows_t = nedge * sampled, nbytes * sampled, rows_t * sampled
        hbm_frac = nbytes / secs / (HBM_PEAK_GBS * 1e9)
        # what this design has to move for the same launches: + the fp32 rhat row per edge, + q in / agg out per row
        rhat_b = 512.0                     # fp32 rows (InfgenOptions.rhat_format = 0, the default)
        model = nbytes + L * rhat_b * nedge + 3 * L * 1024.0 * rows_t
        # the REFERENCE's per-edge projections (33,024 MAC per edge and layer) that these launches stand for, and what the absorbed
        # form executes instead: 2,304 per edge and layer + two 128 x 128 products per row and sublayer
        nflop_ref = 2.0 * L * EDGE_MAC_PER_LAYER * nedge
        nflop_exec = 2.0 * L * (2304.0 * nedge + 3 * 2 * 16384.0 * rows_t)
        alg_per_launch = nbytes / calls
        traffic = pmc_traffic('k_edge_attn_step', args)
        roof = {'bound': 'hbm', 'kernel': 'k_edge_fused (the launches inside decode steps)',
                'achieved': nbytes / secs / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': hbm_frac, 'traffic': traffic,
                'traffic_ratio': (traffic / alg_per_launch) if traffic else None,
                'hbm_fraction': hbm_frac,
                'algorithmic_bytes_per_launch': alg_per_launch,
                'traffic_model_bytes_per_launch': model / calls,
                'traffic_model_frac': model / secs / (HBM_PEAK_GBS * 1e9),
                'edges_per_launch': nedge * L / calls,
                'launches': dom['step_calls'], 'launches_in_timed_region': in_region, 'avg_launch_us': 1e6 * secs / calls,
                'other_launches': {'what': 'map encoder pt <-> pt sublayers + edgeless column-0 chain (prologue)',
                                   'launches': dom['calls'] - dom['step_calls'],
                                   'avg_launch_us': 1e3 * (dom['ms'] - dom['step_ms']) / max(1, dom['calls'] - dom['step_calls'])},
                'reference_flop_equivalent': {
                    'tflops': nflop_ref / secs / 1e12, 'of_fp16_split_peak': nflop_ref / secs / (mm_peak * 1e12),
                    'of_fp32_matrix_peak': nflop_ref / secs / (FP32_MATRIX_PEAK_TFLOPS * 1e12),
                    'executed_tflops': nflop_exec / secs / 1e12,
                    'note': 'the reference computes W_kr / W_vr per edge (33,024 MAC per edge and layer); the absorbed form does '
                            'not execute that work (14x fewer multiply-adds), so this is NOT a roofline fraction of the kernel'},
                'note': 'SURVEY 8d bytes (1 KB of K / V per temporal / map edge and per decoded row of the agent set) of the '
                        'decode-step launches / their summed HIP-event duration / 8 TB/s; traffic = HBM-side bytes per such launch from '
                        'the rocprofv3 PMC passes (profiles/traffic.json), traffic_ratio = traffic / algorithmic bytes',
                **{k: v for k, v in common.items() if k not in ('launches', 'avg_launch_us')}}
    elif dom['calls'] > 0 and dom['macs'] > 0:
        secs = dom['ms'] * 1e-3
        flops = 2.0 * dom['macs']
        split = dominant in ('k_fourier', 'k_attn_pre', 'k_attn_post', 'k_heads')
        peak = mm_peak if split else FP32_MATRIX_PEAK_TFLOPS
        frac = flops / secs / (peak * 1e12)
        roof = {'bound': 'mfma', 'kernel': dominant, 'achieved': flops / secs / 1e12, 'peak': peak,
                'unit': 'TFLOP/s', 'frac': frac, 'traffic': None, 'mfma_fraction': frac, 'hbm_fraction': None,
                'arithmetic': 'fp16 MFMA, three-term hi/lo split (peak = 2500 / 3)' if split else 'fp32-input MFMA',
                'algorithmic_flop_per_launch': flops / dom['calls'], **common}
    if roof is not None and roof.get('traffic') is None and dominant != 'k_edge_attn':
        roof['traffic'] = pmc_traffic(dominant, args)

    agent_steps = float(eng.agent_steps() * args.steps)       # (with insertion: the rows decoded at every step of the last rollout)
    inserted = int((eng.n_agents.sum().item() - sum(h['A'] for h in eng.hosts))) if args.insertion else 0
    n_scenes_local = len(scenes)
    rows_per_scene = engines[0].A_cap
    graph_used = False      # (the timed region carries HIP events around the dominant kernel: launches are issued eagerly)
    per_rank = igdist.gather_metrics([1e3 * dt_local / args.steps, float(n_scenes_local)], dev)
    dt, agent_steps = igdist.reduce_run(dt, agent_steps, dev)

    # ---- config.fp32_mfma: the same batch through the fp32-input MFMA kernels (v_mfma_f32_32x32x2_f32 in every GEMM kernel, the
    # unsplit operands) - the arithmetic tests/test_precision_gpu.py measures the headline's split kernels against.  A per-engine
    # option block (InfgenOptions), nothing process-wide is edited.
    # ---- config.rhat24: the headline's kernels with the relative-position rows between the Fourier and the edge kernels packed
    # into 24 bits (InfgenOptions.rhat_format = 1: 2^-17 per value - a reduced-precision mode, never the headline)
    strict = r24leg = None
    if not args.no_strict and args.gemm_terms == 3 and not args.insertion and ns == 1:
        def opt_leg(options, what):
            e = engine.RolloutEngine(w, scenes, vocab, map_vocab, grid, store_logits=False, use_graph=False, options=options)
            e.rollout()
            torch.cuda.synchronize(dev)
            ssteps = max(1, min(3, args.steps))
            t = timed(ranks, e.rollout, ssteps)
            t, n = igdist.reduce_run(t, float(e.agent_steps() * ssteps), dev)
            del e
            torch.cuda.empty_cache()
            return {'value': n / t, 'ms_per_step': 1e3 * t / ssteps, 'steps': ssteps, 'options': options, 'arithmetic': what}
        log('fp32-MFMA leg')
        strict = opt_leg(dict(fourier_mode=0, attn_mode=0, rhat_format=0),
                         'v_mfma_f32_32x32x2_f32 (fp32 operands) in every GEMM kernel, fp32 rhat rows')
        log('24-bit rhat leg')
        r24leg = opt_leg(dict(rhat_format=1), 'the headline\'s kernels with packed 24-bit rhat rows (2^-17 per value; reduced precision)')

    # ---- config.two_streams: the same batch as two engines (half the scenes each) on two HIP streams, sequenced by one host thread
    # (engine.rollout_many): one engine's HBM-bound edge launches run under the other's matrix / vector-bound ones.  Reported next to
    # the headline, not as it: with two streams the per-kernel HIP-event durations overlap, and the roofline of the line is the
    # single-stream kernel's.
    two = None
    if not args.no_strict and args.gemm_terms == 3 and not args.insertion and ns == 1 and len(scenes) >= 128:
        log('two-stream leg')
        half = (len(scenes) + 1) // 2
        es = [engine.RolloutEngine(w, scenes[i * half:(i + 1) * half], vocab, map_vocab, grid, store_logits=False, use_graph=False)
              for i in range(2)]
        st2 = [torch.cuda.Stream(device=dev) for _ in es]
        run2 = lambda: engine.rollout_many(es, st2)
        run2()
        torch.cuda.synchronize(dev)
        tsteps = max(1, min(3, args.steps))
        t = timed(ranks, run2, tsteps)
        t, n = igdist.reduce_run(t, float(sum(e.agent_steps() for e in es) * tsteps), dev)
        two = {'value': n / t, 'ms_per_step': 1e3 * t / tsteps, 'steps': tsteps, 'engines': 2, 'scenes_per_engine': half,
               'note': 'the same scenes as two engines on two HIP streams (bench.py --streams 2 times exactly this as the headline)'}
        del es
        torch.cuda.empty_cache()

    # ---- config.multi_rollout: the reference's validation workload - n_rollout_close_val sampled rollouts per scene
    # (infgen/model/infgen.py:704-706; top-5 token draws) - as 32 scenes x 32 rollouts: RolloutEngine(copies=32) encodes every
    # scene's map ONCE (one pt <-> pt graph, one map encoder pass, one set of map K / V rows per scene: what the reference's
    # inference_no_map(data, map_enc), infgen_decoder.py:132-134, is for), next to the same 1024 rollouts run as independent scenes
    multi = None
    if c3_shapes(args) and not args.insertion and not args.no_strict and args.gemm_terms == 3 and ns == 1 and len(scenes) >= 32:
        log('multi-rollout leg')
        n_sc, n_ro = 32, 32
        rng = np.random.default_rng(11)
        u = rng.random((cfg.num_decode_steps, n_sc * n_ro, args.agents)).astype(np.float32)
        msteps = max(1, min(3, args.steps))
        res = {}
        for tag in ('shared_map', 'independent'):
            if tag == 'shared_map':
                e = engine.RolloutEngine(w, scenes[:n_sc], vocab, map_vocab, grid, store_logits=False, use_graph=False,
                                         sample_k=5, sample_uniforms=u, copies=n_ro)
            else:
                e = engine.RolloutEngine(w, [sc for sc in scenes[:n_sc] for _ in range(n_ro)], vocab, map_vocab, grid,
                                         store_logits=False, use_graph=False, sample_k=5, sample_uniforms=u)
            e.rollout()
            torch.cuda.synchronize(dev)
            t = timed(ranks, e.rollout, msteps)
            t, n = igdist.reduce_run(t, float(e.agent_steps() * msteps), dev)
            tok = e.token.clone()
            mem = sum(x.numel() * 4 for x in e.mapK + e.mapV) + e.x_pt.numel() * 4
            res[tag] = {'value': n / t, 'ms_per_step': 1e3 * t / msteps, 'map_side_bytes': int(mem), 'tokens': tok}
            del e
            torch.cuda.empty_cache()
        same = bool(torch.equal(res['shared_map']['tokens'], res['independent']['tokens']))
        multi = {'scenes': n_sc, 'rollouts_per_scene': n_ro, 'sample_k': 5, 'steps': msteps,
                 'value': res['shared_map']['value'], 'ms_per_step': res['shared_map']['ms_per_step'],
                 'independent_scenes_value': res['independent']['value'], 'independent_scenes_ms_per_step': res['independent']['ms_per_step'],
                 'speedup': res['shared_map']['value'] / res['independent']['value'],
                 'map_side_bytes': res['shared_map']['map_side_bytes'],
                 'independent_map_side_bytes': res['independent']['map_side_bytes'],
                 'same_tokens_as_independent_scenes': same,
                 'note': '32 scenes x 32 top-5 sampled rollouts (caller-supplied uniforms, the same for both legs): one map encoding '
                         'per scene (RolloutEngine(copies=32)) vs the same 1024 rollouts as independent scenes'}

    # ---- prologue_ms / decode_ms (throughput includes the map-encoder prologue, "report it separately too"): the
    # prologue (state reset, map encoder, map K / V, edgeless column-0 chain) of the same batch timed on its own; the decode
    # steps are the rest of the timed rollout
    prologue = None
    if ns == 1 and engines and not args.insertion and not args.no_strict:      # (--no-strict: the timed region only, e.g. under rocprofv3)
        log('prologue leg')
        e0 = engines[0]
        psteps = max(1, min(3, args.steps))
        e0.prologue()
        torch.cuda.synchronize(dev)
        tp = timed(ranks, e0.prologue, psteps) / psteps
        tm = timed(ranks, lambda: e0.prologue(map_only=True), psteps) / psteps
        prologue = {'prologue_ms': 1e3 * tp, 'map_encoder_ms': 1e3 * tm, 'steps': psteps,
                    'what': 'reset + categorical embeddings + map encoder (map_decoder.py:70-130) + map K / V of the six map -> agent '
                            'layers + edgeless column-0 chain; map_encoder_ms: reset + embeddings + map encoder only'}

    # ---- config.c3_literal: BASELINE C3 as written - 64 scenes in total, dealt to the ranks like the reference's
    # DistributedSampler (scene i -> rank i mod N); at N = 1 also the 8 scenes one GPU of an 8-way shard owns
    c3 = args.agents == 64 and args.map_tokens == 1024 and args.rollout_steps == 80 and not args.insertion
    literal = None
    if c3 and not args.no_literal and args.gemm_terms == 3:
        del engines[:]
        eng = None
        torch.cuda.empty_cache()

        def leg(total, ids, steps):
            log(f'c3_literal leg: {len(ids)} of {total} scenes')
            sc, _, _, _ = build_scenes(cfg, ids, args.agents, args.map_tokens)
            e = engine.RolloutEngine(w, sc, vocab, map_vocab, grid, store_logits=False, use_graph=use_graph)
            for _ in range(3):                  # (a graph engine: eager, capture, first replay)
                e.rollout()
            torch.cuda.synchronize(dev)
            t = timed(ranks, e.rollout, steps)
            t, n = igdist.reduce_run(t, float(e.agent_steps() * steps), dev)
            return {'total_scenes': total, 'scenes_per_gpu': len(ids), 'value': n / t, 'ms_per_step': 1e3 * t / steps,
                    'steps': steps, 'hip_graph': bool(e.use_graph)}
        lsteps = max(args.steps, 10)
        literal = leg(64, igdist.scenes_for_rank_strided(rank, world, 64), lsteps)
        literal['note'] = ('BASELINE C3 literal batch: 64 scenes in total, scene i -> rank i mod N (strong scaling of the fixed batch); '
                           'value = 64 x 64 x 80 agent-steps / max-over-ranks time of one rollout of the rank\'s share')
        if world == 1:
            shard = leg(8, igdist.scenes_for_rank_strided(0, 8, 64), lsteps)
            literal['one_gpu_of_8way_shard'] = shard
            literal['projected_8gpu_value'] = 8.0 * shard['value']
            literal['projected_8gpu_speedup_over_1gpu'] = 8.0 * shard['value'] / literal['value']
            literal['note'] += (f"; projected strong-scaling factor of this fixed batch at 8 GPUs: {literal['projected_8gpu_speedup_over_1gpu']:.2f} x "
                                '(latency-bound per GPU) - north_star\'s >= 7.5 x at 8 GPUs is to be read against WEAK scaling (scenes per GPU fixed)')
            literal['note'] += ('; one_gpu_of_8way_shard: the 8 scenes rank 0 of an 8-way shard owns, run here on one GPU - scenes are '
                                'independent and there is no data-path collective, so 8 x its value projects the 8-GPU figure of the '
                                'literal batch (latency-bound at 8 scenes per GPU; the weak-scaling headline does not have this limit)')

    if rank == 0:
        c4 = args.agents == 64 and args.map_tokens == 1024 and args.rollout_steps == 800 and args.insertion
        c5 = args.agents == 256 and args.map_tokens == 4096 and args.rollout_steps == 800
        shape = 'C3 shapes' if c3 else 'C4 shapes' if c4 else 'C5 shapes' if c5 else 'custom shapes'
        batch = (f'{args.total_scenes} scenes dealt to {world} rank(s) (strong scaling)' if args.scaling == 'strong'
                 else f'{args.scenes} scenes per GPU (weak scaling)')
        line = {
            'metric': 'agent-steps/sec (closed-loop rollout)',
            'value': agent_steps / dt,
            'unit': 'agent-steps/s',
            'n_gpus': world,
            'rccl_ranks': world if ranks.dist is not None else 0,
            'ranks': world,
            'backend': ranks.backend,
            'per_rank_ms': [r[0] for r in per_rank],
            'per_rank_ms_spread': max(r[0] for r in per_rank) / (sum(r[0] for r in per_rank) / len(per_rank)),
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True,
            'scaling': args.scaling,
            'vs_baseline': None,
            'dtype': {3: 'f32', 2: 'bf16', 1: 'f16'}[args.gemm_terms],
            'arithmetic': (ARITHMETIC_F32 if args.gemm_terms == 3 else
                           'bf16-precision MFMA operands (weights and activations rounded to nearest even at 8 significant bits, carried '
                           'on the f16 pipe), fp32 accumulate; outside the 1e-3 parity bar (BASELINE C5 "bf16")' if args.gemm_terms == 2 else
                           'fp16 MFMA operands (hi term only), fp32 accumulate; outside the 1e-3 parity bar (BASELINE C5 reduced mode)'),
            'data': 'synthetic',
            'config': {
                'workload': f'{shape}: configs/ours_{"long_term" if args.insertion else "standard"}.yaml hyper-parameters, '
                            f'{args.agents} agents / {args.map_tokens} map tokens per scene, R={args.rollout_steps} '
                            f'({cfg.num_decode_steps} decode steps), greedy, insertion {"on" if args.insertion else "disabled"}, '
                            f'{batch}, one step = reset + map encoder + full rollout',
                'scenes_per_gpu': n_scenes_local, 'scenes_per_rank': [int(r[1]) for r in per_rank], 'streams': ns,
                'gemm_terms': args.gemm_terms, 'agents': args.agents,
                'map_tokens': args.map_tokens, 'decode_steps': cfg.num_decode_steps, 'insertion': bool(args.insertion),
                'rows_per_scene': rows_per_scene, 'agents_inserted_last_rollout': inserted, 'hip_graph': graph_used,
                'agent_steps_counted': 'rows decoded at every step incl. inserted agents x 5 (SURVEY 8d)' if args.insertion else 'agents x R',
                'agent_token_steps_per_s': agent_steps / dt / cfg.shift,
                'parallelism': f'scenes sharded over {world} rank(s), no data-path collective',
                'c3_literal': literal,
                'fp32_mfma': strict,
                'rhat24': r24leg,
                'two_streams': two,
                'multi_rollout': multi,
                'insertion_balance': balance,
            },
            'roofline': roof,
            'cpu_baseline': cpu,
            'parity': parity,
        }
        # the nested legs again as FLAT scalar keys (a consumer that keeps only top-level scalars still sees the conservative
        # figures next to the headline)
        ms = 1e3 * dt / args.steps
        flat = {
            'fp32_mfma_value': strict['value'] if strict else None,
            'strict_fp32_value': strict['value'] if strict else None,       # (the key earlier rounds' lines used for the same leg)
            'rhat24_value': r24leg['value'] if r24leg else None,
            'two_streams_value': two['value'] if two else None,
            'multi_rollout_value': multi['value'] if multi else None,
            'multi_rollout_speedup_over_independent_scenes': multi['speedup'] if multi else None,
            'c3_literal_value': literal['value'] if literal else None,
            'c3_literal_ms': literal['ms_per_step'] if literal else None,
            'c3_8scene_value': literal['one_gpu_of_8way_shard']['value'] if literal and 'one_gpu_of_8way_shard' in literal else None,
            'c3_8scene_ms': literal['one_gpu_of_8way_shard']['ms_per_step'] if literal and 'one_gpu_of_8way_shard' in literal else None,
            'c3_literal_projected_8gpu_speedup': literal.get('projected_8gpu_speedup_over_1gpu') if literal else None,
            'prologue_ms': prologue['prologue_ms'] if prologue else None,
            'map_encoder_ms': prologue['map_encoder_ms'] if prologue else None,
            'decode_ms': (ms - prologue['prologue_ms']) if prologue else None,
            'decode_only_value': (agent_steps / args.steps / ((ms - prologue['prologue_ms']) * 1e-3)) if prologue else None,
            'roofline_frac': roof['frac'] if roof else None,
            'roofline_traffic_ratio': roof.get('traffic_ratio') if roof else None,
            'scenes_per_rank_min': min(int(r[1]) for r in per_rank),
        }
        line.update(flat)
        line['config']['prologue'] = prologue
        print(json.dumps(line))
    ranks.close()


if __name__ == '__main__':
    main()
