"""Golden vectors for the scoring stage of LongMetric: the REFERENCE's own compute_scenario_metrics_for_bundle
(infgen/metrics/compute_metrics.py:880-1103: windows of 80 steps every 5, histogram log-likelihoods under the logged
distributions, validity-weighted averages, meta-metric) with its own metric_config.textproto, on a seeded 200-step
rollouts dict.  Logged distributions: the reference's _get_log_distributions (:1105-1163) on a second seeded scene.
Build container only.

    PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION=python python tests/golden/make_golden_scores.py
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from make_golden_metrics import make_platoon  # noqa: E402  (installs the stand-ins, imports the reference)

import infgen.metrics.compute_metrics as cm  # noqa: E402
import _standins  # noqa: E402

_standins.assert_reference(cm)
from google.protobuf import text_format  # noqa: E402

cm.submission_specs = SimpleNamespace(CURRENT_TIME_INDEX=10, STEP_DURATION_SECONDS=0.1, N_SIMULATION_STEPS=80)
cm.degree = lambda index, num_nodes=None, dtype=None: torch.bincount(index).to(dtype or torch.long)
cm.tqdm = lambda it, **k: it

FIELDS = ('linear_speed', 'linear_acceleration', 'angular_speed', 'angular_acceleration', 'distance_to_nearest_object',
          'collision_indication', 'time_to_collision', 'num_placement', 'num_removement', 'distance_placement',
          'distance_removement')


def rollouts_dict(seed, N, R):
    T10, T2 = 11 + R, (11 + R) // 5
    b = make_platoon(seed, N, T10)
    rng = np.random.default_rng(seed + 1)
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    state = torch.from_numpy(rng.choice([0, 1, 1, 1, 1, 1, 2, 3], size=(N, T2)).astype(np.int64))
    return dict(
        scenario_id=cm.get_scenario_id_int_tensor(['s%d' % seed]), av_id=100 + N - 1,
        agent_id=torch.arange(100, 100 + N)[:, None], agent_batch=torch.zeros(N, dtype=torch.long),
        pred_traj=f32(np.stack([b['cx'], b['cy']], -1))[:, None], pred_z=torch.zeros(N, 1, T10),
        pred_head=f32(b['heading'])[:, None],
        pred_shape=f32(np.stack([b['length'][:, 0], b['width'][:, 0], np.full(N, 1.6)], -1))[:, None],
        pred_type=torch.zeros(N, 1, dtype=torch.long), pred_state=state[:, None],
        pred_valid=torch.from_numpy(b['valid'])[:, None],
        token_pos=f32(np.stack([b['cx'][:, ::5][:, :T2], b['cy'][:, ::5][:, :T2]], -1))[:, None],
        token_head=f32(b['heading'][:, ::5][:, :T2])[:, None])


def main():
    with open('/root/reference/infgen/metrics/metric_config.textproto') as f:
        config = text_format.Parse(f.read(), cm.long_metrics_pb2.SimAgentMetricsConfig())
    with torch.no_grad():
        # logged distributions from another scene's features
        log_sim = cm.output_to_rollouts(rollouts_dict(7701, 16, 80))[0].joint_scenes[0]
        lf = cm.compute_metric_features(log_sim)
        vals = dict(linear_speed=lf.linear_speed, linear_acceleration=lf.linear_acceleration, angular_speed=lf.angular_speed,
                    angular_acceleration=lf.angular_acceleration, distance_to_nearest_object=lf.distance_to_nearest_object,
                    collision_indication=torch.any(torch.where(lf.valid, lf.collision_per_step, False), dim=1, keepdim=True),
                    time_to_collision=lf.time_to_collision, num_placement=lf.num_placement.float(),
                    num_removement=lf.num_removement.float(), distance_placement=lf.distance_placement,
                    distance_removement=lf.distance_removement)
        dists = {k: cm._get_log_distributions(k, getattr(config, k), torch.nan_to_num(v, nan=0.0),
                                              'bernoulli' if k == 'collision_indication' else 'histogram')
                 for k, v in vals.items()}
        log_d = cm.LogDistributions(**dists)
        scen = rollouts_dict(7702, 20, 200)
        sr = cm.output_to_rollouts(scen)[0]
        metrics, long = cm.compute_scenario_metrics_for_bundle(config, log_d, None, sr)
    out = {'in_' + k: v.numpy() for k, v in scen.items() if torch.is_tensor(v)}
    out.update({'logv_' + k: torch.nan_to_num(v, nan=0.0).numpy() for k, v in vals.items()})     # what the distributions were built from
    # LongMetric.compute (:1401-1447) for this one scenario: means = the scenario's values, then the bucket aggregation
    with torch.no_grad():
        buck = cm.LongMetric.aggregate_metrics_to_buckets(config, metrics)
        long_mean = {k: cm._reduce_mean(v, dim=0) for k, v in long.items() if torch.is_tensor(v)}
        buck_long = cm.LongMetric.aggregate_metrics_long_to_buckets(config, long_mean)
    for k in ('realism_meta_metric', 'kinematic_metrics', 'interactive_metrics', 'map_based_metrics', 'placement_based_metrics'):
        out['b_' + k] = np.float32(getattr(buck, k))
        out['bl_' + k] = buck_long[k].numpy()
    out.update({'logp_' + k: d.logits.numpy()[0] for k, d in dists.items()})
    out.update({'m_' + k + '_likelihood': np.float32(getattr(metrics, k + '_likelihood')) for k in FIELDS})
    out.update({'l_' + k: v.numpy() for k, v in long.items() if torch.is_tensor(v)})
    cfgd = {}
    for k in FIELDS:
        fc = getattr(config, k)
        if fc.HasField('histogram'):
            h = fc.histogram
            cfgd[k] = [h.min_val, h.max_val, h.num_bins, h.additive_smoothing_pseudocount, fc.metametric_weight]
        else:
            cfgd[k] = [-0.5, 0.5, 2, fc.bernoulli.additive_smoothing_pseudocount, fc.metametric_weight]
    np.savez_compressed(os.path.join(HERE, 'scores_platoon_n20_r200.npz'), av_id=scen['av_id'], metametric=np.float32(metrics.metametric),
                        simulated_collision_rate=np.float32(metrics.simulated_collision_rate),
                        config=np.array([cfgd[k] for k in FIELDS], np.float64), fields=np.array(FIELDS), **out)
    print('metametric', metrics.metametric, 'collision rate', metrics.simulated_collision_rate)
    for k in FIELDS:
        print(k, getattr(metrics, k + '_likelihood'), tuple(long[k + '_likelihood'].shape))
    print('metametric long', tuple(long['metametric'].shape))


if __name__ == '__main__':
    main()
