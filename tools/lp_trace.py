"""INFGEN_LP_TRACE=<n>: dump the s_memtime stamps of workgroup 0 of the n-th k_layers_p launch (8 scenes of the bench family)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infgen_amd import engine, synth, _lib
dev = torch.device('cuda:0')
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = synth.standard_config(disable_insertion=True, num_recurrent_steps_val=80)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(root, 'tests', 'golden', 'state_dict_shapes.json'))).items()}
sd = synth.fill_state_dict(shapes, seed=1, rich=True)
vocab, map_vocab = synth.make_agent_vocab(cfg.token_size), synth.make_map_vocab()
grid = synth.build_grid(cfg.grid_range, cfg.grid_interval, cfg.pl2seed_radius)
w = engine.PackedWeights(sd, cfg, dev)
scenes = [synth.make_scene(synth.scene_seed(3, i), 64, 1024, cfg, vocab=vocab, grid=grid) for i in range(S)]
e = engine.RolloutEngine(w, scenes, vocab, map_vocab, grid, store_logits=False, use_graph=False)
for _ in range(3):
    e.rollout()
torch.cuda.synchronize()
