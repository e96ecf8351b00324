"""Golden vectors for a rollout-metric feature: the REFERENCE's own
compute_distance_to_nearest_object (infgen/metrics/interact_features.py:19-95) on seeded boxes.  Build container only.

    PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION=python python tests/golden/make_golden_metrics.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import _standins  # noqa: E402

_standins.install()
sys.path.insert(0, '/root/reference')
from infgen.metrics.interact_features import compute_distance_to_nearest_object  # noqa: E402


def make_boxes(seed, N, T, extent):
    rng = np.random.default_rng(seed)
    head0 = rng.uniform(-np.pi, np.pi, N)
    speed = rng.uniform(0, 10, N)
    t = np.arange(T) * 0.1
    heading = head0[:, None] + rng.uniform(-0.3, 0.3, N)[:, None] * t[None]
    pos0 = rng.uniform(-extent, extent, (N, 2))
    vel = speed[:, None, None] * np.stack([np.cos(heading), np.sin(heading)], -1)
    pos = pos0[:, None] + np.cumsum(vel, 1) * 0.1
    kind = rng.integers(0, 3, N)
    length = np.array([4.8, 1.0, 2.0])[kind][:, None] * rng.uniform(0.8, 1.2, (N, 1)) * np.ones((1, T))
    width = np.array([2.0, 1.0, 0.8])[kind][:, None] * rng.uniform(0.8, 1.2, (N, 1)) * np.ones((1, T))
    valid = rng.random((N, T)) > 0.1
    eval_mask = np.zeros(N, bool)
    eval_mask[rng.choice(N, max(1, N // 3), replace=False)] = True
    f = lambda a: a.astype(np.float32)
    return dict(cx=f(pos[..., 0]), cy=f(pos[..., 1]), length=f(length), width=f(width), heading=f(heading), valid=valid,
                eval_mask=eval_mask)


def main():
    for case, (seed, N, T, ext) in {'dist_n24_t30': (7201, 24, 30, 25.0), 'dist_n5_t4': (7202, 5, 4, 6.0)}.items():
        b = make_boxes(seed, N, T, ext)
        tt = {k: torch.from_numpy(v) for k, v in b.items()}
        z = torch.zeros_like(tt['cx'])
        with torch.no_grad():
            out = compute_distance_to_nearest_object(tt['cx'], tt['cy'], z, tt['length'], tt['width'], z + 1.5, tt['heading'],
                                                     tt['valid'], tt['eval_mask'])
        np.savez_compressed(os.path.join(HERE, case + '.npz'), seed=seed, distance=out.numpy(), **b)
        o = out.numpy()
        print(case, o.shape, 'collisions', int((o < 0).sum()), 'none-valid', int((o > 1e9).sum()), 'min', o.min())


if __name__ == '__main__':
    main()
