import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import load_case, make_weights
from infgen_amd import engine, synth
from oracle import rollout_oracle as ro
c = load_case('c1_a8_m128')
cfg = synth.standard_config(num_recurrent_steps_val=10)
sd = make_weights(seed=8, head_gain=64.0)
A, M, L = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
scene = synth.make_scene(41, A, M, cfg, half_extent=L, vocab=c['vocab'], grid=c['grid'], slip=float(sys.argv[4]) if len(sys.argv) > 4 else 0.0)
tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
ref = ro.run_scene(tsd, scene, cfg, c['vocab'], c['map_vocab'], c['grid'])
dev = torch.device('cuda:0')
w = engine.PackedWeights(sd, cfg, dev)
teacher = [(ref['next_token_idx'].numpy(), ref['next_state_idx'].numpy())]
eng = engine.RolloutEngine(w, [scene], c['vocab'], c['map_vocab'], c['grid'], store_logits=True, teacher=teacher)
eng.rollout()
o = eng.outputs()[0]
print('A', A, 'M', M, 'L', L, 'x_pt err', np.abs(o['x_pt'] - ref['x_pt'].numpy()).max(), 'map edges', int(eng._mg['total'].item()), 'cap', eng._mg['cap'],
      'max map deg', int(eng._mg['cnt'].max().item()))
print('  logits err per step', [float(np.abs(o['logits'][t] - ref['logits'].numpy()[t]).max()) for t in range(2)], 'edges ref', ref['edge_count'].tolist(),
      'edges eng', [int(eng.edges[k]['total'].item()) for k in 'tam'])
