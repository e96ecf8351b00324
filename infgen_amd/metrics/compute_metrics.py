"""Rollout sink and metric features: mirror of the reference's formatting of close-loop rollouts
(infgen/model/infgen.py:788-835), `output_to_rollouts` (infgen/metrics/compute_metrics.py:360-463) and
`compute_metric_features` (:560-707).  The rollout arrays stay on the GPU from `InfGenDecoder.inference` to the features:
every feature is one launch of the HIP library (see the sibling modules); nothing is computed on the CPU.

Constants of the Waymo sim-agents submission spec the reference takes from `waymo_open_dataset` (a third-party package,
`submission_specs`): CURRENT_TIME_INDEX 10, STEP_DURATION_SECONDS 0.1; SHIFT 5 is the reference's token stride (:38).
"""
import dataclasses
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch
from torch import Tensor

from . import interact_features, map_features, placement_features, trajectory_features

CURRENT_TIME_INDEX = 10
STEP_DURATION_SECONDS = 0.1
SHIFT = 5
AGENT_STATE = ['invalid', 'valid', 'enter', 'exit']
COLLISION_DISTANCE_THRESHOLD = 0.0


def get_scenario_id_int_tensor(scenario_id: List[str], device=torch.device('cpu')) -> Tensor:
    """reference compute_metrics.py:348-357: characters as int32, -1 padded to 16"""
    out = torch.full((len(scenario_id), 16), -1, dtype=torch.int32)
    for i, sid in enumerate(scenario_id):
        out[i, :len(sid)] = torch.tensor([ord(ch) for ch in sid], dtype=torch.int32)
    return out.to(device)


def format_rollouts(data, rollouts: Sequence[Dict[str, Tensor]], to_cpu: bool = False) -> Dict:
    """reference infgen.py:788-835: stack the per-rollout outputs of `InfGenDecoder.inference` on a new dim 1 into the dict
    that is pickled / handed to the metrics.  The reference moves every array to the CPU here; by default this keeps them
    where they are (`to_cpu=True` reproduces the pickled layout)."""
    keys = dict(pred_valid='pred_valid', token_pos='pos_a', token_head='head_a', pred_traj='pred_traj', pred_head='pred_head',
                pred_z='pred_z', pred_shape='eval_shape', pred_type='pred_type', pred_state='next_state_idx',
                agent_id='agent_id')
    out = {k: torch.stack([r[src] for r in rollouts], dim=1) for k, src in keys.items()}
    first = rollouts[0]
    out = dict(_scenario_id=data['scenario_id'], scenario_id=get_scenario_id_int_tensor(data['scenario_id']),
               av_id=int(first['agent_id'][int(first['ego_index'])]),
               agent_batch=torch.zeros(out['pred_traj'].shape[0], dtype=torch.long, device=out['pred_traj'].device),
               tfrecord_path=data['tfrecord_path'] if 'tfrecord_path' in data else None, **out)
    if to_cpu:
        out = {k: v.cpu() if torch.is_tensor(v) else v for k, v in out.items()}
    return out


@dataclass(frozen=True)
class ObjectTrajectories:
    """reference compute_metrics.py:142-163 (fields and meaning)"""
    x: Tensor
    y: Tensor
    z: Tensor
    heading: Tensor
    length: Tensor
    width: Tensor
    height: Tensor
    valid: Tensor
    object_id: Tensor
    object_type: Tensor
    state: Optional[Tensor] = None
    token_pos: Optional[Tensor] = None
    token_heading: Optional[Tensor] = None
    token_valid: Optional[Tensor] = None
    processed_object_id: Optional[Tensor] = None
    av_id: Optional[int] = None
    processed_av_id: Optional[int] = None

    def gather_objects_by_id(self, object_ids: Tensor) -> 'ObjectTrajectories':
        """:187-213: rows of the given ids (10 Hz fields only; the token-rate fields stay whole, like the reference)"""
        hit = self.object_id[None, :] == object_ids.to(self.object_id.device)[:, None]
        if not bool(hit.any(1).all()):
            raise ValueError('Some items in `reference_tensor` are missing from `tensor`: '
                             f'\n{object_ids} \nvs. \n{self.object_id}.')
        idx = hit.int().argmax(1)
        rows = {f: getattr(self, f).index_select(-2, idx) for f in ('x', 'y', 'z', 'heading', 'length', 'width', 'height', 'valid')}
        return dataclasses.replace(self, object_id=self.object_id[idx], object_type=self.object_type[idx], **rows)


@dataclass(frozen=True)
class ScenarioRollouts:
    joint_scenes: List[ObjectTrajectories]
    scenario_id: str


def output_to_rollouts(scenario: Dict) -> List[ScenarioRollouts]:
    """reference compute_metrics.py:360-463: the rollouts dict -> per scenario, per rollout trajectories with the shape
    broadcast over the steps.  `object_type` is (n_agent,) here (the reference's repeat of a 2-D tensor yields an
    unusable shape, and no feature reads it)."""
    sid = scenario['scenario_id'].cpu()
    batch = scenario['agent_batch']
    n_scen = sid.shape[0]
    n_step = scenario['pred_traj'].shape[2]
    state = scenario['pred_state'] if 'pred_state' in scenario else torch.zeros_like(scenario['pred_z']).long()
    out = []
    for s in range(n_scen):
        rows = torch.nonzero(batch == s)[:, 0]
        g = lambda k: scenario[k].index_select(0, rows)
        traj, shape, ids = g('pred_traj'), g('pred_shape'), g('agent_id')
        st = state.index_select(0, rows)
        scenes = []
        for r in range(traj.shape[1]):
            sh = shape[:, r, None, :].expand(-1, n_step, -1)
            scenes.append(ObjectTrajectories(
                x=traj[:, r, :, 0], y=traj[:, r, :, 1], z=g('pred_z')[:, r], heading=g('pred_head')[:, r],
                length=sh[..., 0], width=sh[..., 1], height=sh[..., 2], valid=g('pred_valid')[:, r], state=st[:, r],
                object_id=ids[:, r], processed_object_id=ids[:, r], object_type=g('pred_type')[:, r],
                token_pos=g('token_pos')[:, r, :, :2], token_heading=g('token_head')[:, r],
                av_id=scenario.get('av_id', -1), processed_av_id=scenario.get('av_id', -1)))
        out.append(ScenarioRollouts(joint_scenes=scenes, scenario_id=''.join(chr(c) for c in sid[s].tolist() if c > 0)))
    return out


@dataclass(frozen=True)
class MetricFeatures:
    """reference compute_metrics.py:500-516"""
    object_id: Tensor
    valid: Tensor
    linear_speed: Tensor
    linear_acceleration: Tensor
    angular_speed: Tensor
    angular_acceleration: Tensor
    distance_to_nearest_object: Tensor
    collision_per_step: Tensor
    time_to_collision: Tensor
    distance_to_road_edge: Optional[Tensor]
    offroad_per_step: Optional[Tensor]
    num_placement: Tensor
    num_removement: Tensor
    distance_placement: Tensor
    distance_removement: Tensor


def compute_metric_features(simulate_trajectories: ObjectTrajectories, evaluate_agent_ids: Optional[Tensor] = None,
                            scenario_log=None, road_edge_polylines=None) -> MetricFeatures:
    """reference compute_metrics.py:560-707.  Road edges come from `scenario_log.map_features[*].road_edge.polyline`
    like there, or directly as `road_edge_polylines` (a list of polylines or the (padded, cyclic) pair of
    `map_features.tensorize_polylines`).  Without either the two map features are None (the reference leaves them
    uninitialised)."""
    sim = simulate_trajectories
    ev = sim.gather_objects_by_id(evaluate_agent_ids) if evaluate_agent_ids is not None else sim
    cut = CURRENT_TIME_INDEX + 1
    kin = trajectory_features.compute_kinematic_features(ev.x, ev.y, ev.z, ev.heading, seconds_per_step=STEP_DURATION_SECONDS)
    speed, accel, yaw_rate, yaw_accel = (k[:, cut:] for k in kin)
    every = torch.ones(sim.object_id.shape[0], dtype=torch.bool, device=sim.x.device)
    boxes = dict(center_x=sim.x, center_y=sim.y, length=sim.length, width=sim.width, heading=sim.heading, valid=sim.valid,
                 evaluated_object_mask=every)
    dist = interact_features.compute_distance_to_nearest_object(center_z=sim.z, height=sim.height, **boxes)[:, cut:]
    ttc = interact_features.compute_time_to_collision_with_object_in_front(seconds_per_step=STEP_DURATION_SECONDS,
                                                                           **boxes)[:, cut:]
    if road_edge_polylines is None and scenario_log is not None:
        road_edge_polylines = [f.road_edge.polyline for f in scenario_log.map_features if f.HasField('road_edge')]
    road = offroad = None
    if road_edge_polylines is not None:
        road = map_features.compute_distance_to_road_edge(center_z=sim.z, height=sim.height,
                                                          road_edge_polylines=road_edge_polylines, **boxes)[:, cut:]
        offroad = road > map_features.OFFROAD_DISTANCE_THRESHOLD
    if sim.av_id == sim.processed_av_id == -1:
        n_agent, n10 = speed.shape
        z1 = torch.zeros(n10 // SHIFT, device=sim.x.device)
        num_in, num_out = z1, z1.clone()
        d_in = torch.zeros(n_agent, n10 // SHIFT, device=sim.x.device)
        d_out = d_in.clone()
    else:
        assert sim.av_id == sim.processed_av_id, f'Got duplicated av_id: {sim.av_id} and {sim.processed_av_id}'
        c2 = CURRENT_TIME_INDEX // SHIFT
        num_in, num_out = (n[c2:] for n in placement_features.compute_num_placement(
            state=sim.state, valid=sim.token_valid, av_id=sim.processed_av_id, object_id=sim.processed_object_id,
            agent_state=AGENT_STATE))
        d_in, d_out = (d[:, c2:] for d in placement_features.compute_distance_placement(
            position=sim.token_pos, state=sim.state, valid=sim.valid, av_id=sim.processed_av_id,
            object_id=sim.processed_object_id, agent_state=AGENT_STATE))
    return MetricFeatures(object_id=sim.object_id, valid=ev.valid[:, cut:], linear_speed=speed, linear_acceleration=accel,
                          angular_speed=yaw_rate, angular_acceleration=yaw_accel, distance_to_nearest_object=dist,
                          collision_per_step=dist < COLLISION_DISTANCE_THRESHOLD, time_to_collision=ttc,
                          distance_to_road_edge=road, offroad_per_step=offroad, num_placement=num_in[None],
                          num_removement=num_out[None], distance_placement=d_in, distance_removement=d_out)


def __getattr__(name):
    # the reference keeps LongMetric in this module (infgen/metrics/compute_metrics.py:1105); here it lives in long_metric.py,
    # which imports from this file - resolved on first use
    if name in ('LongMetric', 'compute_log_distributions', 'get_log_distributions'):
        from . import long_metric
        return getattr(long_metric, name)
    raise AttributeError(name)
