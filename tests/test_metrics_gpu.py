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


@pytest.mark.parametrize('case', ['road_n24_t30', 'road_n5_t4'])
def test_distance_to_road_edge_golden(case):
    """device vs the reference's own output: 1e-4 m (device cos/sin and fused multiply-adds), same off-road set"""
    from infgen_amd.metrics import compute_distance_to_road_edge, tensorize_polylines
    dev = torch.device('cuda:0')
    z = np.load(os.path.join(GOLDEN, case + '.npz'))
    t = {k: torch.from_numpy(z[k]).to(dev) for k in z.files if z[k].ndim > 0}
    roads = np.split(z['road_points'], np.cumsum(z['road_lengths'])[:-1])
    kw = dict(center_x=t['cx'], center_y=t['cy'], center_z=t['cz'], length=t['length'], width=t['width'], height=t['height'],
              heading=t['heading'], valid=t['valid'], evaluated_object_mask=t['eval_mask'])
    out = compute_distance_to_road_edge(road_edge_polylines=roads, **kw)
    ref = t['distance']
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() <= 1e-4
    assert torch.equal(out > 0, ref > 0)
    again = compute_distance_to_road_edge(road_edge_polylines=tensorize_polylines(roads, dev), **kw)
    assert torch.equal(out, again)
    with pytest.raises(ValueError):
        compute_distance_to_road_edge(road_edge_polylines=[], **kw)


def test_distance_to_road_edge_vs_oracle_large():
    """a bigger seeded scene against the oracle (64 boxes x 80 steps, 60 road edges of up to 120 points)"""
    from oracle import metrics_oracle as mo
    from infgen_amd.metrics import compute_distance_to_road_edge
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(77)
    N, T = 64, 80
    r = lambda *s: torch.rand(*s, generator=g)
    head = (r(N, 1) * 2 - 1) * 3.14159 + 0.02 * torch.arange(T)[None]
    cx = (r(N, 1) * 2 - 1) * 80 + torch.cos(head) * torch.arange(T)[None] * 0.5
    cy = (r(N, 1) * 2 - 1) * 80 + torch.sin(head) * torch.arange(T)[None] * 0.5
    cz = r(N, 1).expand(N, T).contiguous()
    ln, wd, ht = 4 + r(N, 1).expand(N, T), 1.8 + 0.4 * r(N, 1).expand(N, T), 1.5 + r(N, 1).expand(N, T)
    valid = r(N, T) > 0.1
    mask = r(N) > 0.5
    roads = []
    for k in range(60):
        n = int(torch.randint(2, 120, (1,), generator=g))
        h = (r(1) * 6.28 + torch.cumsum((r(n) - 0.5) * 0.3, 0))
        xy = (r(1, 2) * 2 - 1) * 90 + torch.cumsum(torch.stack([torch.cos(h), torch.sin(h)], -1) * (0.5 + 2 * r(n, 1)), 0)
        roads.append(torch.cat([xy, torch.full((n, 1), 6.0 if k % 7 == 0 else 0.0)], -1).numpy())
    loop = torch.linspace(0, 6.2831853, 90)
    roads.append(torch.stack([100 * torch.cos(loop), 100 * torch.sin(loop), torch.zeros(90)], -1).numpy())
    poly, cyc = mo.tensorize_polylines(roads)
    want = mo.distance_to_road_edge(cx, cy, cz, ln, wd, ht, head, valid, mask, poly, cyc)
    d = lambda a: a.to(dev)
    got = compute_distance_to_road_edge(center_x=d(cx), center_y=d(cy), center_z=d(cz), length=d(ln), width=d(wd),
                                        height=d(ht), heading=d(head), valid=d(valid), evaluated_object_mask=d(mask),
                                        road_edge_polylines=(poly, cyc.to(torch.uint8))).cpu()
    err = (got - want).abs()
    # a corner equidistant (to rounding) from two segments of different polylines may pick the other one: allow 0.1 %
    assert (err > 1e-3).float().mean().item() <= 1e-3, err.max()


def test_rollout_sink_features_golden():
    """rollouts dict -> output_to_rollouts -> compute_metric_features, all arrays on the GPU, against the REFERENCE's
    MetricFeatures of the same dict (tests/golden/make_golden_features.py)"""
    from infgen_amd.metrics import compute_metric_features, output_to_rollouts
    from test_oracle_golden import _features_fixture
    dev = torch.device('cuda:0')
    z, scen, roads = _features_fixture()
    scen = {k: v.to(dev) if torch.is_tensor(v) else v for k, v in scen.items()}

    class _F:
        def __init__(self, pts, edge=True):
            from types import SimpleNamespace as NS
            self.road_edge, self._e = NS(polyline=[NS(x=p[0], y=p[1], z=p[2]) for p in pts.tolist()]), edge

        def HasField(self, name):
            return self._e and name == 'road_edge'

    from types import SimpleNamespace
    log = SimpleNamespace(map_features=[_F(r) for r in roads] + [_F(roads[0] + 5, False)])
    sim = output_to_rollouts(scen)[0].joint_scenes[0]
    f = compute_metric_features(sim, evaluate_agent_ids=None, scenario_log=log)
    tol = dict(linear_speed=1e-4, linear_acceleration=2e-3, angular_speed=1e-4, angular_acceleration=2e-3,
               distance_to_nearest_object=1e-4, time_to_collision=1e-3, distance_to_road_edge=1e-4,
               distance_placement=1e-4, distance_removement=1e-4)
    for name, t in tol.items():
        got, want = getattr(f, name).cpu().numpy(), z['f_' + name]
        assert got.shape == want.shape, name
        assert np.array_equal(np.isnan(got), np.isnan(want)), name
        assert np.nanmax(np.abs(got - want)) <= t, (name, np.nanmax(np.abs(got - want)))
    for name in ('object_id', 'valid', 'collision_per_step', 'offroad_per_step', 'num_placement', 'num_removement'):
        assert np.array_equal(getattr(f, name).cpu().numpy(), z['f_' + name]), name
    sub = compute_metric_features(sim, evaluate_agent_ids=torch.from_numpy(z['eval_ids']), road_edge_polylines=None)
    assert sub.distance_to_road_edge is None and sub.linear_speed.shape == (4, 80)
    assert np.array_equal(sub.valid.cpu().numpy(), z['s_valid'])
    assert np.nanmax(np.abs(sub.linear_speed.cpu().numpy() - z['s_linear_speed'])) <= 1e-4


def test_rollout_to_features_end_to_end():
    """InfGenDecoder.inference -> format_rollouts -> output_to_rollouts -> compute_metric_features without leaving the
    GPU; checked against the oracle functions on the same rollout"""
    from conftest import load_case
    from oracle import metrics_oracle as mo
    from test_boundary_cpu import _decoder
    from test_modules_gpu import _load, _to_data
    from infgen_amd.metrics import compute_metric_features, format_rollouts, output_to_rollouts
    c = load_case('a24_m256_edge')
    dev = torch.device('cuda:0')
    dec = _decoder(c['cfg'])
    _load(dec, c['sd'])
    dec = dec.to(dev).eval()
    data = _to_data(c['scene'], dev)
    if 'tfrecord_path' not in data:
        data['tfrecord_path'] = ['none']
    out = dec.inference(data)
    roll = format_rollouts(data, [out])
    assert roll['pred_traj'].is_cuda and roll['pred_traj'].shape[1] == 1 and roll['scenario_id'].shape == (1, 16)
    assert roll['av_id'] == int(out['agent_id'][int(out['ego_index'])])
    sr = output_to_rollouts(roll)
    sim = sr[0].joint_scenes[0]
    f = compute_metric_features(sim)
    A, T = sim.x.shape
    assert f.linear_speed.shape == (A, T - 11) and f.num_placement.shape[0] == 1
    cpu = lambda a: a.detach().cpu()
    every = torch.ones(A, dtype=torch.bool)
    d = mo.distance_to_nearest_object(cpu(sim.x), cpu(sim.y), cpu(sim.length), cpu(sim.width), cpu(sim.heading),
                                      cpu(sim.valid), every)[:, 11:]
    assert (cpu(f.distance_to_nearest_object) - d).abs().max() <= 1e-3
    nb, ne, db, de = mo.placement_features(torch.cat([cpu(sim.token_pos), torch.zeros(A, sim.state.shape[1], 1)], -1),
                                           cpu(sim.state), int(out['ego_index']))
    assert torch.equal(cpu(f.num_placement)[0], nb[2:]) and torch.equal(cpu(f.num_removement)[0], ne[2:])
    pickled = format_rollouts(data, [out], to_cpu=True)
    assert not pickled['pred_traj'].is_cuda


def test_window_log_likelihood_vs_oracle():
    """the fused scoring kernel against the oracle's bin lookup and windows: values on the bin edges, outside the range and
    NaN included; sums to 2e-6 relative (fp32 summation order), counts exact"""
    from oracle import scores_oracle as so
    from infgen_amd.metrics import window_log_likelihood
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(3)
    n, T, nb, lo, hi = 37, 203, 11, -12.0, 12.0
    v = (torch.rand(n, T, generator=g) * 30 - 15)
    edges = torch.linspace(lo, hi, nb + 1).float()
    v[0, :12] = edges
    v[1, :5] = torch.tensor([float('nan'), float('inf'), -float('inf'), hi, lo])
    ok = torch.rand(n, T, generator=g) > 0.2
    ok[2] = False
    logp = torch.log_softmax(torch.randn(nb, generator=g), 0)
    ll = so.windows(logp[so.bin_index(v, lo, hi, nb)], 80, 5)
    w = so.windows(ok, 80, 5)
    s, c = window_log_likelihood(v.to(dev), ok.to(dev), lo, hi, nb, logp.to(dev), 80, 5)
    assert torch.equal(c.cpu().long(), w.sum(-1))
    want = torch.where(w, ll, torch.zeros_like(ll)).sum(-1)
    assert float((s.cpu() - want).abs().max()) <= 2e-6 * float(want.abs().max())
    s2, c2 = window_log_likelihood(v.to(dev), None, lo, hi, nb, logp.to(dev), 16, 1)
    assert int(c2.min()) == 16 and float((s2.cpu() - so.windows(logp[so.bin_index(v, lo, hi, nb)], 16, 1).sum(-1)).abs().max()) <= 1e-4


def test_scenario_scores_golden():
    """rollouts dict -> features -> likelihoods and meta-metric, all on the GPU, against the REFERENCE's
    compute_scenario_metrics_for_bundle (tests/golden/make_golden_scores.py).  2e-3: a feature value that differs in its
    last bits from the reference's can fall into the neighbouring histogram bin"""
    from infgen_amd.metrics import compute_metric_features, compute_scenario_metrics, output_to_rollouts
    from test_oracle_golden import _scores_fixture
    dev = torch.device('cuda:0')
    z, scen, fields, cfg, logp = _scores_fixture()
    scen = {k: v.to(dev) if torch.is_tensor(v) else v for k, v in scen.items()}
    feats = compute_metric_features(output_to_rollouts(scen)[0].joint_scenes[0])
    config = {f: dict(histogram=dict(min_val=c[0], max_val=c[1], num_bins=int(c[2])), metametric_weight=c[4]) for f, c in cfg.items()}
    out, long = compute_scenario_metrics(config, logp, feats)
    for f in fields:
        assert abs(out[f + '_likelihood'] - float(z['m_' + f + '_likelihood'])) <= 2e-3, f
        want = z['l_' + f + '_likelihood']
        assert long[f + '_likelihood'].shape == want.shape, f
        assert np.abs(long[f + '_likelihood'].cpu().numpy() - want).max() <= 5e-3, f
    assert abs(out['metametric'] - float(z['metametric'])) <= 1e-3
    assert np.abs(long['metametric'].cpu().numpy() - z['l_metametric']).max() <= 2e-3
    assert abs(out['simulated_collision_rate'] - float(z['simulated_collision_rate'])) <= 1e-6


def test_widening_entries_edge_cases():
    """empty and degenerate inputs of the widening entry points: no evaluated object, a single object (no partner: the
    reference's 1e10 sentinel), one polyline of two points, all-invalid boxes, windows as long as the series, and the
    C ABI's error returns for impossible sizes"""
    from infgen_amd import _lib
    from infgen_amd.metrics import (compute_distance_to_nearest_object, compute_distance_to_road_edge,
                                    compute_kinematic_features, window_log_likelihood)
    from oracle import metrics_oracle as mo
    dev = torch.device('cuda:0')
    lib = _lib.load()
    T = 7
    one = lambda v: torch.full((1, T), float(v), device=dev)
    x = torch.arange(T, device=dev, dtype=torch.float32)[None] * 0.5
    valid = torch.ones(1, T, dtype=torch.bool, device=dev)
    none = torch.zeros(1, dtype=torch.bool, device=dev)
    every = torch.ones(1, dtype=torch.bool, device=dev)
    assert compute_distance_to_nearest_object(x, one(0), one(0), one(4), one(2), one(1.5), one(0), valid, none).shape == (0, T)
    alone = compute_distance_to_nearest_object(x, one(0), one(0), one(4), one(2), one(1.5), one(0), valid, every)
    want = mo.distance_to_nearest_object(x.cpu(), one(0).cpu(), one(4).cpu(), one(2).cpu(), one(0).cpu(), valid.cpu(), every.cpu())
    assert torch.equal(alone.cpu(), want)
    kin = compute_kinematic_features(x, one(0), one(0), one(0), 0.1)
    assert torch.isnan(kin[0][:, 0]).all() and abs(float(kin[0][0, 3]) - 5.0) < 1e-4
    road = [np.array([[0.0, -3.0, 0.0], [10.0, -3.0, 0.0]], np.float32)]
    kw = dict(center_x=x, center_y=one(0), center_z=one(0), length=one(4), width=one(2), height=one(1.5), heading=one(0))
    d = compute_distance_to_road_edge(valid=valid, evaluated_object_mask=every, road_edge_polylines=road, **kw)
    assert abs(float(d[0, 2]) + 2.0) < 1e-5                       # left of a +x edge at y = -3: on the road, 2 m from it
    d0 = compute_distance_to_road_edge(valid=~valid, evaluated_object_mask=every, road_edge_polylines=road, **kw)
    assert (d0 == -1e10).all()
    assert compute_distance_to_road_edge(valid=valid, evaluated_object_mask=none, road_edge_polylines=road, **kw).shape == (0, T)
    logp = torch.log(torch.tensor([0.25, 0.75], device=dev))
    s, c = window_log_likelihood(x, valid, 0.0, 4.0, 2, logp, T, 5)            # one window covering the whole series
    assert s.shape == (1, 1) and int(c) == T
    assert abs(float(s) - float(4 * logp[0] + 3 * logp[1])) < 1e-5             # 0, .5, 1, 1.5 | 2, 2.5, 3
    p = _lib.ptr
    assert lib.infgen_window_log_likelihood(p(x), None, 1, T, T + 1, 1, p(logp), p(logp), 2, p(s), p(c), None) != 0
    assert lib.infgen_window_log_likelihood(p(x), None, 1, T, 2, 1, p(logp), p(logp), 65, p(s), p(c), None) != 0
    assert lib.infgen_window_log_likelihood(p(x), None, 0, T, 2, 1, p(logp), p(logp), 2, p(s), p(c), None) == 0
    assert b'num_bins' in lib.infgen_last_error()
