"""determinism soak with insertion on: N rollouts of the same batch, states / tokens / poses / row counts compared bitwise"""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from infgen_amd import engine, synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device('cuda:0')
cfg = synth.standard_config(disable_insertion=False)
sd = synth.fill_state_dict(bench.load_shapes(), seed=1, rich=True, head_gain=64.0)
vocab, map_vocab = synth.make_agent_vocab(cfg.token_size), synth.make_map_vocab()
grid = synth.build_grid(cfg.grid_range, cfg.grid_interval, cfg.pl2seed_radius)
scenes = [synth.make_scene(synth.scene_seed(3, i), 64, 1024, cfg, vocab=vocab, grid=grid, slip=0.2) for i in range(S)]
w = engine.PackedWeights(sd, cfg, dev)
eng = engine.RolloutEngine(w, scenes, vocab, map_vocab, grid)
names = ('n_agents', 'state', 'token', 'pos', 'head', 'X', 'pred_traj')
ref, bad = None, 0
for it in range(N):
    eng.rollout(); torch.cuda.synchronize()
    snap = [getattr(eng, n).clone() for n in names]
    if ref is None:
        ref = snap
        print('inserted', int(eng.n_agents.sum()) - 64 * S)
    else:
        for n, a, b in zip(names, snap, ref):
            ai = a.view(torch.int32) if a.dtype == torch.float32 else a
            bi = b.view(torch.int32) if b.dtype == torch.float32 else b
            if not torch.equal(ai, bi):
                bad += 1
                print('rollout', it, 'tensor', n, 'differs in', int((ai != bi).sum()), 'elements')
print('scenes', S, 'rollouts', N, 'mismatching tensors', bad)
