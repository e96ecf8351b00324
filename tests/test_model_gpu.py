"""GPU: the caller of the hot path, `infgen_amd.model.InfGen` (mirror of infgen/model/infgen.py for the close-loop
validation branch): raw scene -> TokenProcessor -> match_token_map -> _fetch_enterings -> InfGenDecoder.inference ->
rollouts pickle -> MetricFeatures, every stage on the device; the rollout is checked against the CPU oracle run on the
scene the device pre-processing produced."""
import os
import pickle
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import make_weights

pytestmark = pytest.mark.gpu


def _model_config(cfg):
    dec = SimpleNamespace(num_future_steps=80, pl2seed_radius=cfg.pl2seed_radius, token_size=cfg.token_size,
                          seed_size=cfg.seed_size, num_map_layers=cfg.num_map_layers, num_agent_layers=cfg.num_agent_layers,
                          pl2pl_radius=cfg.pl2pl_radius, pl2a_radius=cfg.pl2a_radius, a2a_radius=cfg.a2a_radius,
                          a2sa_radius=cfg.a2sa_radius, pl2sa_radius=cfg.pl2sa_radius, time_span=cfg.time_span,
                          buffer_size=cfg.buffer_size)
    return SimpleNamespace(dataset='waymo', input_dim=2, hidden_dim=128, output_dim=2, output_head=False,
                           num_historical_steps=11, num_freq_bands=64, num_heads=8, head_dim=16, dropout=0.1,
                           decoder_type='agent_decoder', predict_motion=True, predict_state=True, predict_map=False,
                           predict_occ=True, state_token=cfg.state_token, grid_range=cfg.grid_range,
                           grid_interval=cfg.grid_interval, angle_interval=cfg.angle_interval, disable_insertion=True,
                           num_recurrent_steps_val=cfg.num_recurrent_steps_val, loss_weight=None, val_open_loop=False,
                           val_close_loop=True, n_rollout_close_val=1, warmup_steps=0, lr=0.0, total_steps=0, decoder=dec)


def _raw_scene(seed, A, P, dev):
    """a logged scene before tokenisation: 91 steps of agent tracks at 10 Hz and P three-point polyline pieces"""
    rng = np.random.default_rng(seed)
    atype = rng.integers(0, 3, A)
    atype[-1] = 0
    t = np.arange(91) * 0.1
    head = rng.uniform(-np.pi, np.pi, (A, 1)) + rng.uniform(-0.2, 0.2, (A, 1)) * t[None]
    speed = rng.uniform(1.0, 9.0, (A, 1)) * np.where(atype == 1, 0.2, 1.0)[:, None]
    slip = rng.uniform(0.05, 0.15, (A, 1)) * rng.choice([-1.0, 1.0], (A, 1))
    vel = speed[..., None] * np.stack([np.cos(head + slip), np.sin(head + slip)], -1)
    pos0 = rng.uniform(-40, 40, (A, 1, 2))
    pos0[-1] = 0.0
    pos = pos0 + np.cumsum(vel, 1) * 0.1
    valid = np.ones((A, 91), bool)
    valid[0, :23] = False                                    # enters later, off the token grid
    valid[1, 60:] = False                                    # leaves
    lwh = np.array([[4.8, 2.0, 1.6], [0.9, 0.9, 1.8], [1.9, 0.8, 1.7]], np.float32)[atype]
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    agent = dict(num_nodes=A, av_idx=A - 1, id=torch.arange(A, device=dev), type=torch.from_numpy(atype).to(dev).to(torch.uint8),
                 category=torch.full((A,), 2, dtype=torch.uint8, device=dev), valid_mask=torch.from_numpy(valid).to(dev),
                 position=f(np.concatenate([pos, np.zeros((A, 91, 1))], -1)), heading=f(head), velocity=f(vel),
                 shape=f(lwh[:, None, :] * valid[..., None]))
    theta = rng.uniform(-np.pi, np.pi, P)
    start = rng.uniform(-60, 60, (P, 2))
    s = np.linspace(0, 1, 3)[None] * rng.uniform(1.0, 5.0, (P, 1))
    tp = start[:, None, :] + s[..., None] * np.stack([np.cos(theta), np.sin(theta)], -1)[:, None, :]
    npoly = max(1, P // 8)
    pl_idx = np.sort(rng.integers(0, npoly, P))
    pt = dict(num_nodes=P, side=torch.from_numpy(rng.integers(0, 3, P)).to(dev),
              type=torch.from_numpy(rng.integers(0, 17, P)).to(dev).to(torch.uint8),
              pl_type=torch.from_numpy(rng.integers(0, 4, P)).to(dev).to(torch.uint8))
    return {'agent': agent, 'pt_token': pt, 'city': 'synthetic', 'scenario_id': ['raw_%d' % seed], 'tfrecord_path': ['none'],
            'map_save': dict(traj_pos=f(tp), traj_theta=f(theta), pl_idx_list=torch.from_numpy(pl_idx).to(dev)),
            'map_polygon': dict(num_nodes=npoly, light_type=torch.from_numpy(rng.integers(0, 4, npoly)).to(dev).to(torch.uint8))}


def test_infgen_validation_step_end_to_end(tmp_path):
    from infgen_amd import synth
    from infgen_amd.model import InfGen
    from infgen_amd.modules.infgen_decoder import scene_from_data
    from oracle import rollout_oracle as ro
    dev = torch.device('cuda:0')
    cfg = synth.standard_config()
    vocab, map_vocab = synth.make_agent_vocab(cfg.token_size), synth.make_map_vocab()
    grid = synth.build_grid(cfg.grid_range, cfg.grid_interval, cfg.pl2seed_radius)
    model = InfGen(_model_config(cfg), save_path=str(tmp_path), map_token_traj=map_vocab, agent_tokens=vocab)
    sd = make_weights(seed=1, head_gain=64.0)
    full = {k: torch.from_numpy(sd[k[len('encoder.'):]]) if k.startswith('encoder.') and k[len('encoder.'):] in sd else v
            for k, v in model.state_dict().items()}
    model.load_state_dict(full, strict=True)
    torch.save({'state_dict': model.state_dict(), 'epoch': 3}, tmp_path / 'ckpt.pt')
    assert model.load_state_from_file(str(tmp_path / 'ckpt.pt'))[1] == 3
    model = model.to(dev).eval()
    model.set('validation')
    model.noise = False                       # the reference's random top-8 map-token resampling is not reproducible
    model.on_validation_start()
    data = _raw_scene(4242, 12, 160, dev)
    out = model.validation_step(data, 0)
    A = 11          # agent 0 is not there yet in the two history columns: the rollout drops such rows (agent_decoder.py:1609)
    assert out['next_token_idx'].shape == (A, 18) and 'city' not in data
    assert out['agent_id'].tolist() == list(range(1, 12))
    ag = data['agent']
    assert ag['state_idx'][0, :4].tolist() == [0, 0, 0, 0] and int(ag['state_idx'][0, 4]) == 2      # enters at step 25
    assert (ag['grid_token_idx'][-1] == 980).all()                                                  # the ego's own cell
    # the rollouts file has the reference's layout (infgen.py:819-835)
    with open(tmp_path / 'idx_0_0_rollouts.pkl', 'rb') as f:
        roll = pickle.load(f)
    assert set(roll) == {'_scenario_id', 'scenario_id', 'av_id', 'agent_id', 'agent_batch', 'pred_traj', 'pred_z', 'pred_head',
                         'pred_shape', 'pred_type', 'pred_state', 'pred_valid', 'token_pos', 'token_head', 'tfrecord_path'}
    assert roll['pred_traj'].shape == (A, 1, 91, 2) and not roll['pred_traj'].is_cuda and roll['av_id'] == 11
    assert model.validation_step(data, 0) is None            # an existing rollouts file is skipped, like the reference
    feats = model.scenario_features[0]
    assert feats.linear_speed.shape == (A, 80) and feats.distance_to_nearest_object.is_cuda
    # scoring: a LongMetric whose logged distributions come from this scenario's own features (stand-in for the WOMD logs)
    from infgen_amd.metrics import LongMetric
    hist = lambda lo, hi, nb: dict(histogram=dict(min_val=lo, max_val=hi, num_bins=nb, additive_smoothing_pseudocount=0.1),
                                   bernoulli=dict(additive_smoothing_pseudocount=0.001), metametric_weight=0.1)
    mcfg = dict(linear_speed=hist(0., 25., 10), linear_acceleration=hist(-12., 12., 11), angular_speed=hist(-0.628, 0.628, 11),
                angular_acceleration=hist(-3.14, 3.14, 11), distance_to_nearest_object=hist(-5., 40., 10),
                collision_indication=hist(-0.5, 0.5, 2), time_to_collision=hist(0., 5., 10), num_placement=hist(0., 10., 10),
                num_removement=hist(0., 10., 10), distance_placement=hist(0., 100., 10), distance_removement=hist(0., 100., 10))
    model._long_metrics = LongMetric('val_close_long', mcfg, log_features=feats)
    model._long_metrics.update(features=feats)
    res = model._long_metrics.compute()
    assert res['val_close_long/wosac/scenario_counter'] == 1
    assert 0.0 < res['val_close_long/wosac/kinematic_metrics'] <= 1.0 and 0.0 < res['val_close_long/wosac/realism_meta_metric'] <= 1.1
    # the rollout equals the CPU oracle's on the scene the device pre-processing produced
    scene = scene_from_data(data)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    ref = ro.run_scene(tsd, scene, cfg, vocab, map_vocab, grid, live_state=False)
    assert np.array_equal(out['next_token_idx'].cpu().numpy(), ref['next_token_idx'].numpy())
    assert np.abs(out['pred_traj'].cpu().numpy() - ref['pred_traj'].numpy()).max() <= 1e-3
