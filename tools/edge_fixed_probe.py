"""k_edge_fused on a workload that does not depend on its own numerics: the map encoder's three pt<->pt sublayers (edge lists from
the map tokens alone) and, with --teacher, a teacher-forced rollout (tokens and states forced: the agents move identically whatever
the kernel computes).  Used to judge experiments that change the kernel's arithmetic (results may be wrong, the work is the same).

    python tools/edge_fixed_probe.py [scenes] [reps]
"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from infgen_amd import engine, synth, _lib

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda:0')
cfg = synth.standard_config()
sd = synth.fill_state_dict(bench.load_shapes(), seed=1, rich=True, head_gain=1.0)
vocab, map_vocab = synth.make_agent_vocab(cfg.token_size), synth.make_map_vocab()
grid = synth.build_grid(cfg.grid_range, cfg.grid_interval, cfg.pl2seed_radius)
scenes = [synth.make_scene(synth.scene_seed(3, i), 64, 1024, cfg, vocab=vocab, grid=grid, slip=0.2) for i in range(S)]
w = engine.PackedWeights(sd, cfg, dev)
eng = engine.RolloutEngine(w, scenes, vocab, map_vocab, grid)
eng.prologue(map_only=True); torch.cuda.synchronize()
_lib.prof_enable((1 << len(_lib.KERNEL_IDS)) - 1)
for _ in range(reps):
    eng.prologue(map_only=True)
out = _lib.prof_collect()
_lib.prof_enable(0)
print('map encoder,', S, 'scenes:', ', '.join(f'{k} {v["ms"] / reps:.3f} ms / {v["calls"] // reps}' for k, v in out.items() if v['calls']))
# teacher-forced rollout: force the tokens / states of a first free run
import os, pickle
TF = f'/tmp/edge_probe_teacher_{S}.pkl'          # (kept across invocations: library variants are compared on the same teacher)
if os.path.exists(TF):
    teacher = pickle.load(open(TF, 'rb'))
else:
    eng.rollout(); torch.cuda.synchronize()
    teacher = [(o['next_token_idx'], o['next_state_idx']) for o in eng.outputs()]
    pickle.dump(teacher, open(TF, 'wb'))
eng2 = engine.RolloutEngine(w, scenes, vocab, map_vocab, grid, teacher=teacher)
eng2.rollout(); torch.cuda.synchronize()
_lib.prof_enable((1 << len(_lib.KERNEL_IDS)) - 1)
for _ in range(reps):
    eng2.rollout()
out = _lib.prof_collect()
_lib.prof_enable(0)
print('teacher-forced rollout:', ', '.join(f'{k} {v["ms"] / reps:.3f} ms / {v["calls"] // reps}' for k, v in out.items() if v['calls']))
print('edges built', out['k_edge_attn'].get('edges_built'))
