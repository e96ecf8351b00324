"""k_layers_p (one launch per decode step) against the per-sublayer launches: first-step X / logits differences, free-running
tokens vs the reference fixture, bitwise re-run, and rollout time at 8 / 64 scenes.  usage: python tools/probe_layers_p.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from conftest import load_case
from infgen_amd import engine, synth, _lib

dev = torch.device('cuda:0')
lib = _lib.load()


def run(case, mode, scenes=None, steps=None):
    _lib.check(lib.infgen_set_layers_p(mode))
    c = load_case(case)
    w = engine.PackedWeights(c['sd'], c['cfg'], dev)
    sc = scenes or [c['scene']]
    e = engine.RolloutEngine(w, sc, c['vocab'], c['map_vocab'], c['grid'], store_logits=True, use_graph=False)
    e.prologue()
    if steps is None:
        e.run()
    else:
        for t in range(steps):
            e.step(t)
    torch.cuda.synchronize()
    return c, e


for case in ([] if os.environ.get('PROBE_S') else ['c1_a8_m128', 'a24_m256_edge', 'c3_a64_m1024']):
    c, a = run(case, 0, steps=1)
    _, b = run(case, 1, steps=1)
    dx = (a.X - b.X).abs().max().item()
    n = a.hosts[0]['A']
    dl = (a.logits[0, :n] - b.logits[0, :n]).abs().max().item()
    print(f'{case}: rows {a.rows} first step |dX| {dx:.3e} (|X| {a.X.abs().max().item():.2f})  |dlogits| {dl:.3e}', flush=True)
    c, a = run(case, 0)
    _, b = run(case, 1)
    _, b2 = run(case, 1)
    z = c['z']
    ta, tb = a.outputs()[0]['next_token_idx'], b.outputs()[0]['next_token_idx']
    print(f'   free-running tokens: old == fixture {np.array_equal(ta, z["next_token_idx"])}, new == fixture '
          f'{np.array_equal(tb, z["next_token_idx"])}; new re-run bitwise {torch.equal(b.X, b2.X) and torch.equal(b.logits, b2.logits)}', flush=True)
    if 'logits' in z.files:
        lg = b.outputs()[0]['logits']
        k = min(lg.shape[0], z['logits'].shape[0])
        print(f'   logits vs fixture: new {np.abs(lg[:k] - z["logits"][:k]).max():.3e}  old {np.abs(a.outputs()[0]["logits"][:k] - z["logits"][:k]).max():.3e}')

cfg = synth.standard_config(disable_insertion=True, num_recurrent_steps_val=80)
import json
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'state_dict_shapes.json'))).items()}
sd = synth.fill_state_dict(shapes, seed=1, rich=True)
vocab, map_vocab = synth.make_agent_vocab(cfg.token_size), synth.make_map_vocab()
grid = synth.build_grid(cfg.grid_range, cfg.grid_interval, cfg.pl2seed_radius)
w = engine.PackedWeights(sd, cfg, dev)
for S in [int(v) for v in os.environ.get('PROBE_S', '8,64').split(',')]:
    scenes = [synth.make_scene(synth.scene_seed(3, i), 64, 1024, cfg, vocab=vocab, grid=grid) for i in range(S)]
    toks = {}
    for mode in (0, 1):
        _lib.check(lib.infgen_set_layers_p(mode))
        e = engine.RolloutEngine(w, scenes, vocab, map_vocab, grid, store_logits=False, use_graph=False)
        for _ in range(3):
            e.rollout()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            e.rollout()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 100
        toks[mode] = e.token.clone()
        print(f'S={S} layers_p={mode}: {ms:.2f} ms per rollout = {S * 64 * 80 / ms / 1e3:.2f} M agent-steps/s', flush=True)
    print(f'   tokens equal between the modes: {(toks[0] == toks[1]).float().mean().item():.4f}')
