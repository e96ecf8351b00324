"""Golden vectors for the whole agent tokeniser: the REFERENCE's own
`TokenProcessor._tokenize_agent` (infgen/datasets/preprocess.py:335-550) on seeded synthetic tracks with the synthetic
token tables of infgen_amd.synth.  Build container only.

    PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION=python python tests/golden/make_golden_tokenize.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from make_golden_tokens import make_tracks, TokenProcessor  # noqa: E402  (installs the stand-ins, imports the reference)

from infgen_amd import synth  # noqa: E402


def make_scene(seed, A):
    tr = make_tracks(seed, A)
    rng = np.random.default_rng(seed + 3)
    valid = tr['valid']
    # the entry patterns the extrapolation distinguishes: first seen at the current step with / without step 5 (always
    # unseen then), first seen off the token grid, first seen on it, never seen, seen for one step only
    for a, t0 in zip(range(min(A, 6)), (10, 10, 13, 15, 91, 47)):
        valid[a] = True
        valid[a, :t0] = False
    if A > 5:
        valid[5, 48:] = False
    head = tr['heading']
    for a in range(A):                                  # heading flips the cleaner has to undo
        if rng.random() < 0.3:
            t = rng.integers(1, 90)
            head[a, t] += np.float32(np.pi)
    dt = np.float32(0.1)
    vel = np.zeros((A, 91, 2), np.float32)
    vel[:, 1:] = (tr['pos'][:, 1:, :2] - tr['pos'][:, :-1, :2]) / dt
    vel[:, 0] = vel[:, 1]
    pos = tr['pos'].copy()
    pos[..., 2] = rng.normal(0.5, 0.2, (A, 1)).astype(np.float32)
    lwh = np.array([[4.8, 2.0, 1.6], [0.9, 0.9, 1.8], [1.9, 0.8, 1.7]], np.float32)[tr['type']]
    shape = lwh[:, None, :] * rng.uniform(0.9, 1.1, (A, 1, 1)).astype(np.float32) * valid[:, :, None]
    never = ~valid.any(1)
    shape[never, 0] = lwh[never]                         # the reference needs one non-zero shape row per agent
    return dict(valid_mask=valid, heading=head, position=pos, velocity=vel, type=tr['type'], shape=shape.astype(np.float32),
                category=np.zeros(A, np.int64))


def main():
    cfg = synth.standard_config()
    vocab = synth.make_agent_vocab(cfg.token_size)
    tp = object.__new__(TokenProcessor)
    torch.nn.Module.__init__(tp)
    tp.shift, tp.noise, tp.training, tp.current_step, tp.disable_invalid = 5, False, False, 10, False
    tp.invalid_state, tp.valid_state, tp.enter_state, tp.exit_state = 0, 1, 2, 3
    for k, v in vocab.items():
        tp.register_buffer(f'agent_token_all_{k}', torch.from_numpy(v), persistent=False)
    for case, (seed, A) in {'tokenize_a40': (7501, 40), 'tokenize_a6': (7502, 6)}.items():
        sc = make_scene(seed, A)
        data = {'agent': {k: torch.from_numpy(v.copy()) for k, v in sc.items()}}
        with torch.no_grad():
            out = tp._tokenize_agent(data)['agent']
        keep = {}
        for k in ('token_idx', 'state_idx', 'token_contour', 'token_pos', 'token_heading', 'agent_valid_mask',
                  'raw_agent_valid_mask', 'shape', 'valid_mask', 'heading', 'velocity'):
            keep['out_' + k] = out[k].numpy()
        keep['out_raw_height'] = np.array([float(out['raw_height'][k]) for k in ('veh', 'ped', 'cyc')], np.float32)
        assert out['traj_pos'] is None and out['token_traj'].shape == (A, cfg.token_size, 4, 2)
        np.savez_compressed(os.path.join(HERE, case + '.npz'), seed=seed, **{'in_' + k: v for k, v in sc.items()}, **keep)
        st = out['state_idx'].numpy()
        print(case, st.shape, 'enter', int((st == 2).sum()), 'exit', int((st == 3).sum()), 'invalid', int((st == 0).sum()),
              'heights', keep['out_raw_height'])


if __name__ == '__main__':
    main()
