import sys, time, json, os, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from infgen_amd import engine, synth, _lib
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device('cuda:0')
shapes = bench.load_shapes()
cfg = synth.standard_config()
sd = synth.fill_state_dict(shapes, seed=1, rich=True, head_gain=64.0)
vocab, map_vocab = synth.make_agent_vocab(cfg.token_size), synth.make_map_vocab()
grid = synth.build_grid(cfg.grid_range, cfg.grid_interval, cfg.pl2seed_radius)
scenes = [synth.make_scene(synth.scene_seed(3, i), 64, 1024, cfg, vocab=vocab, grid=grid) for i in range(S)]
w = engine.PackedWeights(sd, cfg, dev)
eng = engine.RolloutEngine(w, scenes, vocab, map_vocab, grid)
for _ in range(2): eng.rollout()
torch.cuda.synchronize()
def t(f, n=3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def pro():
    eng.reset(); eng.prologue()
print('scenes', S, 'rollout ms', t(eng.rollout), 'reset+prologue ms', t(pro))
_lib.prof_enable((1 << len(_lib.KERNEL_IDS)) - 1)
pro()
pk = _lib.prof_collect()
_lib.prof_enable(0)
print({k: (round(v['ms'], 2), v['calls']) for k, v in pk.items() if v['calls']})
print('map graph edges', int(eng._map_graph['total'].cpu()[0]) if hasattr(eng, '_map_graph') else '?')
