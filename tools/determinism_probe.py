"""where does a repeated rollout first differ?  prologue (map encoder, column-0 chain) and the first decode steps of the bench
workload, run twice, intermediate arrays compared bitwise.  usage: determinism_probe.py [scenes] [edge_fuse]"""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from infgen_amd import engine, synth, _lib
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
fuse = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = _lib.load()
_lib.check(lib.infgen_set_edge_fuse(fuse))
dev = torch.device('cuda:0')
cfg = synth.standard_config()
sd = synth.fill_state_dict(bench.load_shapes(), seed=1, rich=True, head_gain=1.0)
vocab, map_vocab = synth.make_agent_vocab(cfg.token_size), synth.make_map_vocab()
grid = synth.build_grid(cfg.grid_range, cfg.grid_interval, cfg.pl2seed_radius)
scenes = [synth.make_scene(synth.scene_seed(3, i), 64, 1024, cfg, vocab=vocab, grid=grid, slip=0.2) for i in range(S)]
w = engine.PackedWeights(sd, cfg, dev)
eng = engine.RolloutEngine(w, scenes, vocab, map_vocab, grid)


def snap():
    torch.cuda.synchronize()
    return {k: getattr(eng, k).clone() for k in ('x_pt', 'X', 'Q', 'AGG', 'Ka', 'Va', 'next_token')} | \
        {'mapK0': eng.mapK[0].clone(), 'ringK0': eng.ringK[0].clone(), 'ringK5': eng.ringK[5].clone()}


def run(nsteps):
    out = []
    eng.prologue(map_only=True); out.append(('map_only', snap()))
    eng.prologue(); out.append(('prologue', snap()))
    for t in range(nsteps):
        eng.step(t); out.append((f'step{t}', snap()))
    return out


NS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
a = run(NS)
for rep in range(3):
    b = run(NS)
    for (na, sa), (nb, sb) in zip(a, b):
        for k in (sa if NS > 0 else ('x_pt',)):
            x, y = sa[k].view(torch.int32), sb[k].view(torch.int32)
            if not torch.equal(x, y):
                d = (x != y)
                rows = d.reshape(d.shape[0], -1).any(1).nonzero().flatten()
                print(f'rep {rep} stage {na} tensor {k}: {int(d.sum())} elements differ in {rows.numel()} rows, first rows {rows[:8].tolist()}')
print('done fuse', fuse, 'scenes', S)
