"""GPU: rollout-metric feature through the C ABI against the reference's own output and the oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', ['dist_n24_t30', 'dist_n5_t4', 'ttc_platoon_n20_t30'])
def test_distance_to_nearest_object_golden(case):
    from infgen_amd.metrics import compute_distance_to_nearest_object
    dev = torch.device('cuda:0')
    z = np.load(os.path.join(GOLDEN, case + '.npz'))
    t = {k: torch.from_numpy(z[k]).to(dev) for k in ('cx', 'cy', 'length', 'width', 'heading', 'valid', 'eval_mask')}
    zero = torch.zeros_like(t['cx'])
    d = compute_distance_to_nearest_object(t['cx'], t['cy'], zero, t['length'], t['width'], zero + 1.5, t['heading'], t['valid'],
                                           t['eval_mask']).cpu().numpy()
    ref = z['distance']
    assert np.array_equal(d > 1e9, ref > 1e9)                       # "no valid other object" cells
    fin = ref < 1e9
    assert np.abs(d[fin] - ref[fin]).max() <= 2e-5                  # fp32, positions of tens of metres
    assert np.array_equal(d[fin] < 0, ref[fin] < 0)                 # same collisions


def test_distance_to_nearest_object_batched_vs_oracle():
    """64 scenes x 48 objects x 80 steps in one launch vs the oracle scene by scene"""
    from infgen_amd.metrics import compute_distance_to_nearest_object
    from oracle import metrics_oracle as mo
    rng = np.random.default_rng(17)
    B, N, T = 64, 48, 80
    head = rng.uniform(-np.pi, np.pi, (B, N, 1)) + rng.uniform(-0.3, 0.3, (B, N, 1)) * (np.arange(T) * 0.1)
    vel = rng.uniform(0, 10, (B, N, 1, 1)) * np.stack([np.cos(head), np.sin(head)], -1)
    pos = rng.uniform(-40, 40, (B, N, 1, 2)) + np.cumsum(vel, 2) * 0.1
    length = rng.uniform(0.8, 5.5, (B, N, 1)) * np.ones((1, 1, T))
    width = rng.uniform(0.5, 2.2, (B, N, 1)) * np.ones((1, 1, T))
    valid = rng.random((B, N, T)) > 0.1
    mask = np.zeros(N, bool); mask[rng.choice(N, 16, replace=False)] = True
    f = lambda a: torch.from_numpy(a.astype(np.float32))
    cx, cy, ln, wd, hd = f(pos[..., 0]), f(pos[..., 1]), f(length), f(width), f(head)
    vt, mt = torch.from_numpy(valid), torch.from_numpy(mask)
    dev = torch.device('cuda:0')
    d = compute_distance_to_nearest_object(cx.to(dev), cy.to(dev), cx.to(dev) * 0, ln.to(dev), wd.to(dev), ln.to(dev), hd.to(dev),
                                           vt.to(dev), mt.to(dev)).cpu().numpy()
    for b in range(0, B, 7):
        ref = mo.distance_to_nearest_object(cx[b], cy[b], ln[b], wd[b], hd[b], vt[b], mt).numpy()
        fin = ref < 1e9
        assert np.array_equal(d[b] > 1e9, ~fin)
        assert np.abs(d[b][fin] - ref[fin]).max() <= 5e-5


@pytest.mark.parametrize('case', ['dist_n24_t30', 'dist_n5_t4', 'ttc_platoon_n20_t30'])
def test_kinematics_and_time_to_collision_golden(case):
    from infgen_amd.metrics import compute_kinematic_features, compute_time_to_collision_with_object_in_front
    dev = torch.device('cuda:0')
    z = np.load(os.path.join(GOLDEN, case + '.npz'))
    t = {k: torch.from_numpy(z[k]).to(dev) for k in ('cx', 'cy', 'length', 'width', 'heading', 'valid', 'eval_mask')}
    kin = compute_kinematic_features(t['cx'], t['cy'], torch.zeros_like(t['cx']), t['heading'], 0.1)
    for a, n, tol in zip(kin, ('speed', 'accel', 'yaw_rate', 'yaw_accel'), (1e-4, 2e-3, 1e-5, 2e-4)):
        a = a.cpu().numpy()
        assert np.array_equal(np.isnan(a), np.isnan(z[n]))
        assert np.nanmax(np.abs(a - z[n]), initial=0.0) <= tol, n          # differences of fp32 positions / 0.1 s (/ 0.01 s^2)
    ttc = compute_time_to_collision_with_object_in_front(center_x=t['cx'], center_y=t['cy'], length=t['length'], width=t['width'],
                                                         heading=t['heading'], valid=t['valid'],
                                                         evaluated_object_mask=t['eval_mask'], seconds_per_step=0.1).cpu().numpy()
    assert np.array_equal(ttc < 5.0, z['ttc'] < 5.0)
    assert np.abs(ttc - z['ttc']).max() <= 1e-3


@pytest.mark.parametrize('case', ['dist_n24_t30', 'dist_n5_t4', 'ttc_platoon_n20_t30'])
def test_placement_features_golden(case):
    from infgen_amd.metrics import compute_num_placement, compute_distance_placement
    dev = torch.device('cuda:0')
    z = np.load(os.path.join(GOLDEN, case + '.npz'))
    N = z['cx'].shape[0]
    pos = torch.stack([torch.from_numpy(z['cx']), torch.from_numpy(z['cy']), torch.zeros(z['cx'].shape)], -1).to(dev)
    state = torch.from_numpy(z['state']).to(dev)
    oid = torch.arange(100, 100 + N)
    names = ['invalid', 'valid', 'enter', 'exit']
    nb, ne = compute_num_placement(torch.from_numpy(z['valid']).to(dev), state, 100 + N - 1, oid, names)
    db, de = compute_distance_placement(pos, state, torch.from_numpy(z['valid']).to(dev), 100 + N - 1, oid, names)
    assert np.array_equal(nb.cpu().numpy(), z['num_bos']) and np.array_equal(ne.cpu().numpy(), z['num_eos'])
    assert np.abs(db.cpu().numpy() - z['bos_distance']).max() <= 1e-4
    assert np.abs(de.cpu().numpy() - z['eos_distance']).max() <= 1e-4
    assert np.array_equal(db.cpu().numpy() > 0, z['bos_distance'] > 0)
