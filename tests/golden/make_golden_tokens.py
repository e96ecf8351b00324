"""Golden vectors for the agent tokeniser: runs the REFERENCE's own
`TokenProcessor._match_agent_token` (infgen/datasets/preprocess.py:552-653, with cal_polygon_contour :24-54)
on seeded synthetic trajectories and stores inputs + outputs.  Build container only (imports /root/reference).

    PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION=python python tests/golden/make_golden_tokens.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
import _standins  # noqa: E402

_standins.install()
sys.path.insert(0, '/root/reference')
from infgen.datasets.preprocess import TokenProcessor  # noqa: E402

_standins.assert_reference(TokenProcessor)

from infgen_amd import synth  # noqa: E402


def make_tracks(seed, A, n_step=91):
    """agents on noisy unicycle tracks at 10 Hz, with dropouts in the validity mask"""
    rng = np.random.default_rng(seed)
    atype = rng.integers(0, 3, size=A)
    speed = rng.uniform(0.0, 14.0, size=A) * np.where(atype == 1, 0.15, 1.0)
    yaw_rate = rng.uniform(-0.5, 0.5, size=A)
    head0 = rng.uniform(-np.pi, np.pi, size=A)
    pos0 = rng.uniform(-80, 80, size=(A, 2))
    t = np.arange(n_step) * 0.1
    head = head0[:, None] + yaw_rate[:, None] * t[None, :] + rng.normal(0, 0.01, size=(A, n_step))
    vel = speed[:, None, None] * np.stack([np.cos(head), np.sin(head)], -1)
    pos = pos0[:, None, :] + np.cumsum(vel, 1) * 0.1 + rng.normal(0, 0.02, size=(A, n_step, 2))
    pos3 = np.concatenate([pos, np.zeros((A, n_step, 1))], -1).astype(np.float32)
    valid = np.ones((A, n_step), bool)
    for a in range(A):
        if rng.random() < 0.4:
            s = rng.integers(0, n_step - 10)
            valid[a, s:s + rng.integers(1, 30)] = False
        if rng.random() < 0.2:
            valid[a, :rng.integers(1, 40)] = False
    # (width, length) per type exactly as the caller builds it (preprocess.py:346-354: veh 2 x 4.8, ped 1 x 2, cyc 1 x 1)
    shape = np.array([[2.0, 4.8], [1.0, 2.0], [1.0, 1.0]], np.float32)[atype]
    return dict(valid=valid, pos=pos3, heading=head.astype(np.float32), shape=shape, type=atype.astype(np.int64))


def main():
    cfg = synth.standard_config()
    vocab = synth.make_agent_vocab(cfg.token_size)
    tp = object.__new__(TokenProcessor)
    torch.nn.Module.__init__(tp)
    tp.shift, tp.noise, tp.training = 5, False, False
    names = ['veh', 'ped', 'cyc']
    for case, (seed, A) in {'tok_a48': (7001, 48), 'tok_a7': (7002, 7)}.items():
        tr = make_tracks(seed, A)
        token_traj = torch.stack([torch.from_numpy(vocab[names[k]][:, -1]) for k in tr['type']])   # (A, 2048, 4, 2)
        with torch.no_grad():
            idx, contour, _ = tp._match_agent_token(torch.from_numpy(tr['valid']), torch.from_numpy(tr['pos'][..., :2].copy()),
                                                    torch.from_numpy(tr['heading']), torch.from_numpy(tr['shape']),
                                                    token_traj, None)
        np.savez_compressed(os.path.join(HERE, case + '.npz'), seed=seed, token_index=idx.numpy(),
                            token_contour=contour.numpy(), **tr)
        print(case, idx.shape, contour.shape, 'unique tokens', len(np.unique(idx.numpy())))


def make_polylines(seed, P):
    """P three-point polyline pieces (start, middle, end) as `map_save.traj_pos`, with the direction of the piece"""
    rng = np.random.default_rng(seed)
    theta = rng.uniform(-np.pi, np.pi, size=P)
    start = rng.uniform(-100, 100, size=(P, 2))
    length = rng.uniform(1.0, 6.0, size=P)
    curv = rng.uniform(-0.2, 0.2, size=P)
    s = np.linspace(0, 1, 3)[None, :] * length[:, None]
    th = curv[:, None] * s
    lx = np.where(np.abs(curv[:, None]) < 1e-9, s, np.sin(th) / np.where(curv[:, None] == 0, 1, curv[:, None]))
    ly = np.where(np.abs(curv[:, None]) < 1e-9, 0 * s, (1 - np.cos(th)) / np.where(curv[:, None] == 0, 1, curv[:, None]))
    c, sn = np.cos(theta)[:, None], np.sin(theta)[:, None]
    pos = np.stack([start[:, :1] + lx * c - ly * sn, start[:, 1:] + lx * sn + ly * c], -1)
    pos += rng.normal(0, 0.05, size=pos.shape)
    return dict(traj_pos=pos.astype(np.float32), traj_theta=theta.astype(np.float32),
                pl_idx_list=np.sort(rng.integers(0, max(1, P // 6), size=P)).astype(np.int64),
                side=rng.integers(0, 3, size=P).astype(np.int64))


def main_map():
    """InfGen.match_token_map (infgen/model/infgen.py:918-936: polyline piece -> nearest of the 1024 map tokens)"""
    import types
    from infgen.model.infgen import InfGen
    mv = synth.make_map_vocab()                        # (1024, 11, 2) like map_traj_token5.pkl['traj_src']
    sample_pt = np.ascontiguousarray(mv[:, ::5]).astype(np.float32)     # (1024, 3, 2) like ['sample_pt']
    fake = types.SimpleNamespace(map_token={'sample_pt': torch.from_numpy(sample_pt), 'traj_src': torch.from_numpy(mv)},
                                 noise=False)
    for case, (seed, P) in {'maptok_p500': (7101, 500), 'maptok_p3': (7102, 3)}.items():
        pl = make_polylines(seed, P)
        data = {'map_save': {'traj_pos': torch.from_numpy(pl['traj_pos']), 'traj_theta': torch.from_numpy(pl['traj_theta']),
                             'pl_idx_list': torch.from_numpy(pl['pl_idx_list'])},
                'pt_token': {'side': torch.from_numpy(pl['side']), 'num_nodes': P}}
        with torch.no_grad():
            out = InfGen.match_token_map(fake, data)
        np.savez_compressed(os.path.join(HERE, case + '.npz'), seed=seed, token_idx=out['pt_token']['token_idx'].numpy(),
                            position=out['pt_token']['position'].numpy(), **pl)
        print(case, out['pt_token']['token_idx'].shape, 'unique', len(np.unique(out['pt_token']['token_idx'].numpy())))


if __name__ == '__main__':
    main()
    main_map()
