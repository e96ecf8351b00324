"""Reduced-precision mode (infgen_set_gemm_terms(1): plain fp16 operands in the GEMM kernels) against the default fp32-accurate
split on the same scenes: teacher-forced logits error and arg-max agreement, free-running token agreement and pose drift.
    python tools/agreement_fp16.py [agents] [map_tokens] [scenes] [rollout_steps]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from infgen_amd import _lib, engine, synth  # noqa: E402

A, M, S, R = (int(x) for x in (sys.argv[1:5] + ['256', '4096', '16', '80'][len(sys.argv) - 1:]))
dev = torch.device('cuda:0')
lib = _lib.load()
cfg = synth.standard_config(num_recurrent_steps_val=R)
sd = synth.fill_state_dict(bench.load_shapes(), seed=1, rich=True)
vocab, map_vocab = synth.make_agent_vocab(cfg.token_size), synth.make_map_vocab()
grid = synth.build_grid(cfg.grid_range, cfg.grid_interval, cfg.pl2seed_radius)
scenes = [synth.make_scene(synth.scene_seed(5, i), A, M, cfg, vocab=vocab, grid=grid, slip=0.2) for i in range(S)]
w = engine.PackedWeights(sd, cfg, dev)
_lib.check(lib.infgen_set_attn_mode(1))          # the split kernels at every size, so that the mode under test is what runs


def run(terms, teacher=None):
    _lib.check(lib.infgen_set_gemm_terms(terms))
    eng = engine.RolloutEngine(w, scenes, vocab, map_vocab, grid, store_logits=True, teacher=teacher)
    eng.rollout()
    return eng.outputs()


ref = run(3)
free = run(1)
teacher = [(o['next_token_idx'], o['next_state_idx']) for o in ref]
tf = run(1, teacher)
_lib.check(lib.infgen_set_gemm_terms(3))
tok_ref = np.stack([o['next_token_idx'][:, 2:] for o in ref])
tok_free = np.stack([o['next_token_idx'][:, 2:] for o in free])
lg_ref = np.stack([o['logits'] for o in ref])
lg_tf = np.stack([o['logits'] for o in tf])
part = np.partition(lg_ref, -2, axis=-1)
margin = part[..., -1] - part[..., -2]
print(json.dumps({
    'workload': f'{S} scenes x {A} agents x {M} map tokens, R={R}, random-init weights of the reference architecture',
    'teacher_forced': {'max_abs_logit_err': float(np.abs(lg_tf - lg_ref).max()), 'mean_abs_logit_err': float(np.abs(lg_tf - lg_ref).mean()),
                       'logit_scale_rms': float(np.sqrt((lg_ref ** 2).mean())),
                       'argmax_agreement': float((lg_tf.argmax(-1) == lg_ref.argmax(-1)).mean()),
                       'median_top1_top2_margin': float(np.median(margin))},
    'free_running': {'token_agreement': float((tok_free == tok_ref).mean()),
                     'agreement_first_decode_step': float((tok_free[..., 0] == tok_ref[..., 0]).mean()),
                     'final_pose_drift_m_median': float(np.median(np.linalg.norm(
                         np.stack([o['pos_a'][:, -1] for o in free]) - np.stack([o['pos_a'][:, -1] for o in ref]), axis=-1)))}}))
