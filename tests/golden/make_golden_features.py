"""Golden vectors for the rollout sink: the REFERENCE's own output_to_rollouts + compute_metric_features
(infgen/metrics/compute_metrics.py:360-463, :560-707) on a seeded rollouts dict (layout of infgen/model/infgen.py:819-835)
with synthetic road edges.  Build container only.

    PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION=python python tests/golden/make_golden_features.py
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from make_golden_road import make_roads  # noqa: E402  (installs the stand-ins, imports the reference)
from make_golden_metrics import make_platoon  # noqa: E402

import infgen.metrics.compute_metrics as cm  # noqa: E402
import _standins  # noqa: E402

_standins.assert_reference(cm)

# waymo_open_dataset.utils.sim_agents.submission_specs is absent here (a mock): its published constants
cm.submission_specs = SimpleNamespace(CURRENT_TIME_INDEX=10, STEP_DURATION_SECONDS=0.1)
# torch_geometric.utils.degree (absent): number of occurrences of every index
cm.degree = lambda index, num_nodes=None, dtype=None: torch.bincount(index).to(dtype or torch.long)


class _Feature:
    def __init__(self, pts, is_edge=True):
        self.road_edge = SimpleNamespace(polyline=[SimpleNamespace(x=float(p[0]), y=float(p[1]), z=float(p[2])) for p in pts])
        self._edge = is_edge

    def HasField(self, name):
        return self._edge and name == 'road_edge'


def main():
    seed, N, R = 7401, 20, 80
    T10, T2 = 11 + R, 2 + R // 5 + 1
    b = make_platoon(seed, N, T10)
    rng = np.random.default_rng(seed + 1)
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    state = torch.from_numpy(rng.choice([0, 1, 1, 1, 1, 2, 3], size=(N, T2)).astype(np.int64))
    token_pos = f32(np.stack([b['cx'][:, ::5][:, :T2], b['cy'][:, ::5][:, :T2]], -1))
    scen = dict(
        scenario_id=cm.get_scenario_id_int_tensor(['a1b2c3d4e5f6']), av_id=100 + N - 1,
        agent_id=torch.arange(100, 100 + N)[:, None], agent_batch=torch.zeros(N, dtype=torch.long),
        pred_traj=f32(np.stack([b['cx'], b['cy']], -1))[:, None], pred_z=torch.zeros(N, 1, T10),
        pred_head=f32(b['heading'])[:, None], pred_shape=f32(np.stack([b['length'][:, 0], b['width'][:, 0],
                                                                      np.full(N, 1.6)], -1))[:, None],
        pred_type=torch.zeros(N, 1, dtype=torch.long), pred_state=state[:, None], pred_valid=torch.from_numpy(b['valid'])[:, None],
        token_pos=token_pos[:, None], token_head=f32(b['heading'][:, ::5][:, :T2])[:, None])
    roads = make_roads(seed + 9, 60.0, 6)
    roads.append(np.stack([np.linspace(-10, 150, 30) * np.cos(0.3) + 6 * np.sin(0.3),
                           np.linspace(-10, 150, 30) * np.sin(0.3) - 6 * np.cos(0.3), np.zeros(30)], -1).astype(np.float32))
    log = SimpleNamespace(map_features=[_Feature(r) for r in roads] + [_Feature(roads[0] + 5, is_edge=False)])
    with torch.no_grad():
        sr = cm.output_to_rollouts(scen)
        sim = sr[0].joint_scenes[0]
        eval_ids = torch.tensor([100 + N - 1, 103, 111, 104])
        feats = cm.compute_metric_features(sim, evaluate_agent_ids=None, scenario_log=log)      # the reference's own call
        # with a subset (the path its callers leave unused): needs a (n_agent,) object_type, which output_to_rollouts
        # does not produce (its repeat of the 2-D pred_type gives (1, n_rollout * n_step, 1))
        import dataclasses
        sub = cm.compute_metric_features(dataclasses.replace(sim, object_type=torch.zeros(N, dtype=torch.long)),
                                         evaluate_agent_ids=eval_ids, scenario_log=None)
    out = {'in_' + k: v.numpy() for k, v in scen.items() if torch.is_tensor(v)}
    out.update({'f_' + f: getattr(feats, f).numpy() for f in feats.__dataclass_fields__})
    out.update({'s_' + f: getattr(sub, f).numpy() for f in ('valid', 'linear_speed', 'linear_acceleration', 'angular_speed',
                                                            'angular_acceleration')})
    np.savez_compressed(os.path.join(HERE, 'features_platoon_n20.npz'), av_id=scen['av_id'], eval_ids=eval_ids.numpy(),
                        scenario_str=sr[0].scenario_id, road_points=np.concatenate(roads, 0),
                        road_lengths=np.array([len(r) for r in roads]), **out)
    for f in feats.__dataclass_fields__:
        v = getattr(feats, f)
        print(f, tuple(v.shape), v.dtype)
    print('collisions', int(feats.collision_per_step.sum()), 'offroad', int(feats.offroad_per_step.sum()),
          'ttc<5', int((feats.time_to_collision < 5).sum()), 'placed', feats.num_placement.sum().item())


if __name__ == '__main__':
    main()
