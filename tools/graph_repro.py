import os, sys, time, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
from infgen_amd import _lib
if os.environ.get('EXP_LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['EXP_LIB'])
import bench
from infgen_amd import engine, synth
dev = torch.device('cuda:0')
cfg = synth.standard_config()
sd = synth.fill_state_dict(bench.load_shapes(), seed=1, rich=True)
w = engine.PackedWeights(sd, cfg, dev)
S = int(sys.argv[1]); mode = {'0': False, '1': True, 'all': 'all'}[sys.argv[2]]
scenes, vocab, map_vocab, grid = bench.build_scenes(cfg, range(S), 64, 1024)
e = engine.RolloutEngine(w, scenes, vocab, map_vocab, grid, use_graph=mode)
for i in range(4):
    e.rollout()
torch.cuda.synchronize(); print('warm ok', flush=True)
t0 = time.perf_counter()
for i in range(20):
    e.rollout()
torch.cuda.synchronize()
print('scenes', S, 'graph', mode, round(1e3 * (time.perf_counter() - t0) / 20, 3), 'ms', flush=True)
