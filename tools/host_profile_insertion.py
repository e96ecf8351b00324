"""cProfile of the host side of one rollout with insertion on: where the Python sequencing of the insertion sub-loop
spends its time, next to the wall time of the rollout.
    python tools/host_profile_insertion.py [scenes] [rollout_steps] [insert_headroom]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from infgen_amd import engine, synth  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
R = int(sys.argv[2]) if len(sys.argv) > 2 else 80
HEAD = int(sys.argv[3]) if len(sys.argv) > 3 else None
dev = torch.device('cuda:0')
cfg = synth.standard_config(disable_insertion=False, num_recurrent_steps_val=R)
sd = synth.fill_state_dict(bench.load_shapes(), seed=1, rich=True)
scenes, vocab, map_vocab, grid = bench.build_scenes(cfg, S, 64, 1024, 0)
w = engine.PackedWeights(sd, cfg, dev)
eng = engine.RolloutEngine(w, scenes, vocab, map_vocab, grid, store_logits=False, insert_headroom=HEAD)
eng.rollout(); torch.cuda.synchronize()
t0 = time.perf_counter(); eng.rollout(); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f'rollout wall {1e3 * (t1 - t0):.1f} ms, agents inserted {int(eng.n_agents.sum()) - 64 * S}')
pr = cProfile.Profile()
pr.enable(); eng.rollout(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
