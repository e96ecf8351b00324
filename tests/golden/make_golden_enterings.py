"""Golden vectors for InfGen._fetch_enterings: the REFERENCE's own method
(infgen/model/infgen.py:1008-1128) with its own Attr_Tokenizer on the tokenised agents of tests/golden/tokenize_a40.npz
split into two scenes, plus random map tokens.  Build container only.

    PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION=python python tests/golden/make_golden_enterings.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import _standins  # noqa: E402

_standins.install()
sys.path.insert(0, '/root/reference')
from infgen.model.infgen import InfGen  # noqa: E402
from infgen.modules.attr_tokenizer import Attr_Tokenizer  # noqa: E402

_standins.assert_reference(InfGen), _standins.assert_reference(Attr_Tokenizer)


class _Data(dict):
    num_graphs = 2


def main():
    z = np.load(os.path.join(HERE, 'tokenize_a40.npz'))
    rng = np.random.default_rng(7601)
    A = z['out_state_idx'].shape[0]
    batch = np.where(np.arange(A) < 25, 0, 1).astype(np.int64)
    token_pos = z['out_token_pos'].copy()
    # pull the agents of each scene around its ego (the tracks of the tokeniser fixture are spread over +-80 m)
    av = np.array([24, 14], np.int64)
    for b, a0 in ((0, 0), (1, 25)):
        rows = np.nonzero(batch == b)[0]
        ego0 = token_pos[a0 + av[b], 2]
        shift = (ego0 - token_pos[rows, 2]) * rng.uniform(0.2, 0.9, (len(rows), 1)).astype(np.float32)
        token_pos[rows] += shift[:, None, :] * (z['out_state_idx'][rows] != 0)[..., None]
    M = 300
    pt_batch = np.sort(rng.integers(0, 2, M)).astype(np.int64)
    pt_pos = (token_pos[[av[0], 25 + av[1]], 2][pt_batch] + rng.uniform(-110, 110, (M, 2))).astype(np.float32)
    pt_pos = np.concatenate([pt_pos, np.zeros((M, 1), np.float32)], -1)
    tok = Attr_Tokenizer(grid_range=150., grid_interval=3., radius=75., angle_interval=3.)
    fake = types.SimpleNamespace(predict_occ=True, enter_state=2, invalid_state=0, pl2seed_radius=75., attr_tokenizer=tok,
                                 save_path='')
    data = _Data(agent=dict(state_idx=torch.from_numpy(z['out_state_idx']), token_pos=torch.from_numpy(token_pos),
                            token_heading=torch.from_numpy(z['out_token_heading']), batch=torch.from_numpy(batch),
                            av_index=torch.from_numpy(av)),
                 pt_token=dict(token_idx=torch.zeros(M, dtype=torch.long), position=torch.from_numpy(pt_pos),
                               batch=torch.from_numpy(pt_batch)))
    with torch.no_grad():
        out = InfGen._fetch_enterings(fake, data)['agent']
    keys = ('grid_token_idx', 'grid_offset_xy', 'heading_token_idx', 'sort_indices', 'inrange_mask', 'bos_mask', 'pos_xy',
            'heading_theta', 'pt_grid_token_idx')
    np.savez_compressed(os.path.join(HERE, 'enterings_a40.npz'), token_pos=token_pos, token_heading=z['out_token_heading'],
                        state_idx=z['out_state_idx'], batch=batch, av_index=av, pt_pos=pt_pos, pt_batch=pt_batch,
                        grid=tok.grid.numpy(), **{'out_' + k: out[k].numpy() for k in keys})
    g = out['grid_token_idx'].numpy()
    print('cells set', int((g >= 0).sum()), 'of', g.size, 'in range', int(out['inrange_mask'].sum()), 'entering in range',
          int((out['bos_mask'] & out['inrange_mask']).sum()), 'pt cells', int((out['pt_grid_token_idx'] >= 0).sum()))


if __name__ == '__main__':
    main()
