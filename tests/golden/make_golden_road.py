"""Golden vectors for compute_distance_to_road_edge: the REFERENCE's own function
(infgen/metrics/map_features.py:27-79) on seeded boxes and synthetic road edges.  Build container only.

    PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION=python python tests/golden/make_golden_road.py
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import _standins  # noqa: E402
from make_golden_metrics import make_boxes  # noqa: E402  (imports the reference too)

from infgen.metrics.map_features import compute_distance_to_road_edge  # noqa: E402

_standins.assert_reference(compute_distance_to_road_edge)


def make_roads(seed, extent, n_open):
    """a counter-clockwise closed outer boundary (cyclic), a clockwise island, open wiggly edges of different lengths,
    one edge on an overpass (z = 8 m), one with a repeated point (zero-length segment) and one degenerate (1 point)"""
    rng = np.random.default_rng(seed)
    roads = []
    th = np.linspace(0, 2 * np.pi, 41)
    r = extent * (1.2 + 0.1 * np.sin(3 * th))
    roads.append(np.stack([r * np.cos(th), r * np.sin(th), np.zeros_like(th)], -1))          # closes on itself
    th = np.linspace(0, -2 * np.pi, 13)[:-1]
    roads.append(np.stack([4 * np.cos(th) + 3, 3 * np.sin(th) - 2, np.zeros_like(th)], -1))   # gap > 1 m: not cyclic
    for k in range(n_open):
        n = int(rng.integers(2, 30))
        p0 = rng.uniform(-extent, extent, 2)
        h = rng.uniform(-np.pi, np.pi) + np.cumsum(rng.normal(0, 0.25, n))
        step = rng.uniform(0.5, 4.0, n)
        xy = p0 + np.cumsum(np.stack([np.cos(h), np.sin(h)], -1) * step[:, None], 0)
        z = np.full(n, 8.0 if k == 0 else rng.normal(0, 0.2))
        roads.append(np.concatenate([xy, z[:, None]], -1))
    dup = roads[-1].copy()
    if len(dup) > 3:
        dup[2] = dup[1]
        roads.append(dup + np.array([1.5, -2.5, 0.0]))
    roads.append(np.array([[0.0, 0.0, 0.0]]))                                                  # dropped by the reference
    return [p.astype(np.float32) for p in roads]


def main():
    for case, (seed, N, T, ext, n_open) in {'road_n24_t30': (7301, 24, 30, 25.0, 9), 'road_n5_t4': (7302, 5, 4, 6.0, 2)}.items():
        b = make_boxes(seed, N, T, ext)
        rng = np.random.default_rng(seed + 5)
        b['cz'] = rng.normal(0, 0.3, (N, 1)).astype(np.float32) * np.ones((1, T), np.float32)
        b['cz'][0] += 8.0                                                  # one object on the overpass level
        b['height'] = rng.uniform(1.4, 2.0, (N, 1)).astype(np.float32) * np.ones((1, T), np.float32)
        roads = make_roads(seed + 9, ext, n_open)
        msgs = [[SimpleNamespace(x=float(p[0]), y=float(p[1]), z=float(p[2])) for p in road] for road in roads]
        tt = {k: torch.from_numpy(v) for k, v in b.items()}
        with torch.no_grad():
            out = compute_distance_to_road_edge(center_x=tt['cx'], center_y=tt['cy'], center_z=tt['cz'], length=tt['length'],
                                                width=tt['width'], height=tt['height'], heading=tt['heading'],
                                                valid=tt['valid'], evaluated_object_mask=tt['eval_mask'],
                                                road_edge_polylines=msgs)
        flat = np.concatenate(roads, 0)
        np.savez_compressed(os.path.join(HERE, case + '.npz'), seed=seed, road_points=flat,
                            road_lengths=np.array([len(r) for r in roads]), distance=out.numpy(), **b)
        o = out.numpy()
        print(case, o.shape, 'roads', len(roads), 'off-road cells', int((o > 0).sum()), 'invalid', int((o < -1e9).sum()),
              'range', o[o > -1e9].min(), o.max())


if __name__ == '__main__':
    main()
