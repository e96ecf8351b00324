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
from infgen.metrics.interact_features import (compute_distance_to_nearest_object,  # noqa: E402
                                               compute_time_to_collision_with_object_in_front)
from infgen.metrics.trajectory_features import compute_kinematic_features  # noqa: E402
from infgen.metrics.placement_features import compute_num_placement, compute_distance_placement  # noqa: E402

for _f in (compute_distance_to_nearest_object, compute_kinematic_features, compute_num_placement):
    _standins.assert_reference(_f)


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


def make_platoon(seed, N, T):
    """vehicles on three lanes driving the same way at different speeds (followers closing in on leaders)"""
    rng = np.random.default_rng(seed)
    lane = rng.integers(0, 3, N)
    s0 = rng.uniform(0, 120, N)
    speed = rng.uniform(2, 15, N)
    th = 0.3 + rng.normal(0, 0.03, N)
    t = np.arange(T) * 0.1
    s = s0[:, None] + speed[:, None] * t[None]
    off = (lane - 1) * 3.5 + rng.normal(0, 0.3, N)
    heading = th[:, None] + rng.normal(0, 0.01, (N, T))
    cx = s * np.cos(0.3) - off[:, None] * np.sin(0.3)
    cy = s * np.sin(0.3) + off[:, None] * np.cos(0.3)
    length = rng.uniform(4.0, 5.5, (N, 1)) * np.ones((1, T))
    width = rng.uniform(1.8, 2.2, (N, 1)) * np.ones((1, T))
    valid = rng.random((N, T)) > 0.05
    eval_mask = rng.random(N) < 0.6
    eval_mask[0] = True
    f = lambda a: a.astype(np.float32)
    return dict(cx=f(cx), cy=f(cy), length=f(length), width=f(width), heading=f(heading), valid=valid, eval_mask=eval_mask)


def main():
    for case, (seed, N, T, ext) in {'dist_n24_t30': (7201, 24, 30, 25.0), 'dist_n5_t4': (7202, 5, 4, 6.0),
                                    'ttc_platoon_n20_t30': (7203, 20, 30, None)}.items():
        b = make_boxes(seed, N, T, ext) if ext else make_platoon(seed, N, T)
        tt = {k: torch.from_numpy(v) for k, v in b.items()}
        z = torch.zeros_like(tt['cx'])
        with torch.no_grad():
            out = compute_distance_to_nearest_object(tt['cx'], tt['cy'], z, tt['length'], tt['width'], z + 1.5, tt['heading'],
                                                     tt['valid'], tt['eval_mask'])
            ttc = compute_time_to_collision_with_object_in_front(
                center_x=tt['cx'], center_y=tt['cy'], length=tt['length'], width=tt['width'], heading=tt['heading'],
                valid=tt['valid'], evaluated_object_mask=tt['eval_mask'], seconds_per_step=0.1)
            kin = compute_kinematic_features(tt['cx'], tt['cy'], z, tt['heading'], 0.1)
            rs = np.random.default_rng(seed + 1)
            state = torch.from_numpy(rs.choice([0, 1, 1, 1, 2, 3], size=tt['cx'].shape).astype(np.int64))
            pos3 = torch.stack([tt['cx'], tt['cy'], z], -1)
            oid = torch.arange(100, 100 + N)
            names = ['invalid', 'valid', 'enter', 'exit']
            nb, ne = compute_num_placement(tt['valid'], state.clone(), 100 + N - 1, oid, names)
            db, de = compute_distance_placement(pos3, state.clone(), tt['valid'], 100 + N - 1, oid, names)
        b = dict(b, state=state.numpy(), num_bos=nb.numpy(), num_eos=ne.numpy(), bos_distance=db.numpy(), eos_distance=de.numpy())
        np.savez_compressed(os.path.join(HERE, case + '.npz'), seed=seed, distance=out.numpy(), ttc=ttc.numpy(),
                            speed=kin[0].numpy(), accel=kin[1].numpy(), yaw_rate=kin[2].numpy(), yaw_accel=kin[3].numpy(), **b)
        print('   ttc < 5 s cells', int((ttc.numpy() < 5).sum()))
        o = out.numpy()
        print(case, o.shape, 'collisions', int((o < 0).sum()), 'none-valid', int((o > 1e9).sum()), 'min', o.min())


if __name__ == '__main__':
    main()
