"""magnitude and pattern of the run-to-run differences of the map encoder output (x_pt) with k_edge_fused, and its distance
to the unfused sequence"""
import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from infgen_amd import engine, synth, _lib
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda:0')
cfg = synth.standard_config()
sd = synth.fill_state_dict(bench.load_shapes(), seed=1, rich=True, head_gain=1.0)
vocab, map_vocab = synth.make_agent_vocab(cfg.token_size), synth.make_map_vocab()
grid = synth.build_grid(cfg.grid_range, cfg.grid_interval, cfg.pl2seed_radius)
scenes = [synth.make_scene(synth.scene_seed(3, i), 64, 1024, cfg, vocab=vocab, grid=grid, slip=0.2) for i in range(S)]
w = engine.PackedWeights(sd, cfg, dev)
eng = engine.RolloutEngine(w, scenes, vocab, map_vocab, grid)
os.environ['INFGEN_MAP_FUSE'] = '0'
eng.prologue(map_only=True); torch.cuda.synchronize()
ref = eng.x_pt.clone()
eng.prologue(map_only=True); torch.cuda.synchronize()
print('unfused repeat identical:', torch.equal(ref.view(torch.int32), eng.x_pt.view(torch.int32)))
os.environ['INFGEN_MAP_FUSE'] = '1'
runs = []
for _ in range(4):
    eng.prologue(map_only=True); torch.cuda.synchronize()
    runs.append(eng.x_pt.clone())
for i, r in enumerate(runs):
    err = (r - ref).abs().amax(1)
    bad = (err > 1e-4).nonzero().flatten()
    print(f'run {i}: max |fused - unfused| {float(err.max()):.3e}, rows > 1e-4: {bad.numel()}, first {bad[:10].tolist()}, '
          f'median err {float(err.median()):.2e}')
d = (runs[0] - runs[1]).abs().amax(1)
nz = (d > 0).nonzero().flatten()
print('rows differing between fused runs 0 and 1:', nz.numel(), 'max diff', float(d.max()), 'median of nonzero', float(d[nz].median()) if nz.numel() else 0)
# pattern inside 16-row tiles
if nz.numel():
    print('row index mod 16 histogram of differing rows:', torch.bincount(nz % 16, minlength=16).tolist())
    tiles = torch.unique(nz // 16)
    print('tiles touched', tiles.numel(), 'of', ref.shape[0] // 16, 'first', tiles[:10].tolist())
