"""Mirror of the reference's LongMetric (infgen/metrics/compute_metrics.py:1165-1515) without torchmetrics: the logged
distributions (`_get_log_distributions`, :1105-1163), the per-scenario accumulation (`update`, :1309-1399) and `compute`
with the bucket aggregation (:1401-1515).  Scenario scores come from `scores.compute_scenario_metrics` (device); what is
accumulated here are a few floats and (1, n_window) tensors per scenario.  Across ranks the sums add up and the lists
concatenate (dist_reduce_fx 'sum' / 'cat' in the reference): `state()` / `merge()` carry that over torch.distributed."""
from typing import Dict, List, Optional

import torch
from torch import Tensor

from .compute_metrics import MetricFeatures
from .scores import FIELDS, _field, _hist, compute_scenario_metrics

BUCKETS = {'kinematic': ['linear_speed', 'linear_acceleration', 'angular_speed', 'angular_acceleration'],
           'interactive': ['distance_to_nearest_object', 'collision_indication', 'time_to_collision'],
           'map_based': [],
           'placement_based': ['num_placement', 'num_removement', 'distance_placement', 'distance_removement']}


def get_log_distributions(field: str, config, log_values: Tensor, estimate_method: str = 'histogram') -> torch.distributions.Categorical:
    """reference :1105-1163: histogram of the logged feature values (clamped to the range; `distance_*` keep only the values
    strictly inside it, `num_placement` drops the last two steps) + pseudo-count -> Categorical"""
    lo, hi, nb, _ = _hist(config, field)
    fc = _field(config, field)
    if estimate_method == 'bernoulli' or field == 'collision_indication':
        pseudo = float(_field(_field(fc, 'bernoulli'), 'additive_smoothing_pseudocount'))
        log_values = log_values.float()
    else:
        pseudo = float(_field(_field(fc, 'histogram'), 'additive_smoothing_pseudocount'))
    v = log_values
    if 'distance_' in field:
        v = v[(v > lo) & (v < hi)]
    if field == 'num_placement':
        v = v[:, :-2]
    v = v.reshape(-1).float().clamp(lo, hi)
    edges = torch.linspace(lo, hi, nb + 1).float().to(v.device)
    idx = torch.bucketize(v.contiguous(), edges, right=True) - 1
    idx = torch.where(v == edges[-1], torch.full_like(idx, nb - 1), idx)         # torch.histogram: last bin closed
    counts = torch.bincount(idx.clamp(0, nb - 1), minlength=nb).float()[None] + pseudo
    return torch.distributions.Categorical(probs=counts)


def compute_log_distributions(config, log_features: MetricFeatures) -> Dict[str, torch.distributions.Categorical]:
    """reference LongMetric._compute_distributions (:1214-1262)"""
    f = log_features
    hit = torch.any(torch.where(f.valid, f.collision_per_step, torch.zeros_like(f.collision_per_step)), dim=1)[..., None]
    vals = dict(linear_speed=f.linear_speed, linear_acceleration=f.linear_acceleration, angular_speed=f.angular_speed,
                angular_acceleration=f.angular_acceleration, distance_to_nearest_object=f.distance_to_nearest_object,
                collision_indication=hit, time_to_collision=f.time_to_collision, num_placement=f.num_placement.float(),
                num_removement=f.num_removement.float(), distance_placement=f.distance_placement,
                distance_removement=f.distance_removement)
    return {k: get_log_distributions(k, config, v, 'bernoulli' if k == 'collision_indication' else 'histogram')
            for k, v in vals.items()}


def _reduce_mean(t: Tensor, dim=None) -> Tensor:
    ok = (t > 0) & (t <= 1)
    z = torch.where(ok, t, torch.zeros_like(t))
    return z.sum() / ok.sum().clamp(min=1) if dim is None else z.sum(0) / ok.sum(0).clamp(min=1)


class LongMetric:
    field_names = ['metametric', 'average_displacement_error', 'min_average_displacement_error'] + \
                  [k + '_likelihood' for k in FIELDS[:7]] + ['simulated_collision_rate'] + [k + '_likelihood' for k in FIELDS[7:]]

    def __init__(self, prefix: str = '', metrics_config=None, log_distributions=None, log_features: Optional[MetricFeatures] = None):
        self.prefix, self.metrics_config = prefix, metrics_config
        if log_distributions is None:
            if log_features is None:
                raise ValueError('LongMetric needs log_distributions or log_features (the reference loads total_features.pkl)')
            log_distributions = compute_log_distributions(metrics_config, log_features)
        self.log_distributions = log_distributions
        self.reset()

    def reset(self):
        self.sums = {k: 0.0 for k in self.field_names}
        self.longs: Dict[str, List[Tensor]] = {k: [] for k in self.field_names}
        self.scenario_counter = self.placement_valid_scenario_counter = self.removement_valid_scenario_counter = 0

    def update(self, features: Optional[MetricFeatures] = None, metrics=None) -> None:
        """one scenario: its MetricFeatures (scored here) or the (scalars, per-window) pair of compute_scenario_metrics"""
        scal, long = metrics if metrics is not None else compute_scenario_metrics(self.metrics_config, self.log_distributions, features)
        self.scenario_counter += 1
        self.placement_valid_scenario_counter += scal['distance_placement_likelihood'] > 0
        self.removement_valid_scenario_counter += scal['distance_removement_likelihood'] > 0
        for k in self.field_names:
            self.sums[k] += float(scal.get(k, 0.0))
            if k in long:
                self.longs[k].append(long[k].detach().cpu())

    def state(self) -> Dict:
        """a copy of the accumulated state (safe to pickle / merge elsewhere)"""
        return dict(sums=dict(self.sums), longs={k: list(v) for k, v in self.longs.items()},
                    counters=(self.scenario_counter, self.placement_valid_scenario_counter,
                              self.removement_valid_scenario_counter))

    def synced_state(self) -> Dict:
        """the reference's torchmetrics reduction at compute() (dist_reduce_fx 'sum' for the scalars, 'cat' for the per-window
        lists, compute_metrics.py:1199-1204): the state of ALL ranks, returned as a new dict - this object keeps its local state
        (torchmetrics restores it after compute() too), so update() / compute() may be repeated without double counting.
        A COLLECTIVE when a process group is up: every rank must call it (hence compute())."""
        import torch.distributed as dist
        st = self.state()
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return st
        states = [None] * dist.get_world_size()
        dist.all_gather_object(states, st)
        tot = dict(sums={k: 0.0 for k in self.field_names}, longs={k: [] for k in self.field_names}, counters=(0, 0, 0))
        for o in states:                      # rank order: the 'cat' lists are rank-major like torchmetrics' gather
            for k in self.field_names:
                tot['sums'][k] += o['sums'][k]
                tot['longs'][k] += o['longs'][k]
            tot['counters'] = tuple(a + b for a, b in zip(tot['counters'], o['counters']))
        return tot

    def merge(self, other_state: Dict) -> None:
        """add another rank's state (sum / cat)"""
        for k in self.field_names:
            self.sums[k] += other_state['sums'][k]
            self.longs[k] += other_state['longs'][k]
        c = other_state['counters']
        self.scenario_counter += c[0]
        self.placement_valid_scenario_counter += c[1]
        self.removement_valid_scenario_counter += c[2]

    def compute(self) -> Dict:
        """reference :1401-1447, on the state of all ranks (``synced_state``: a collective under a process group; the local
        state is left as it is)"""
        st = self.synced_state()
        sums, longs, (n_all, n_place, n_remove) = st['sums'], st['longs'], st['counters']
        mean, mean_long = {}, {}
        for k in self.field_names:
            den = n_all
            if k == 'distance_placement_likelihood':
                den = n_place
            if k == 'distance_removement_likelihood':
                den = n_remove
            mean[k] = sums[k] / max(den, 1)
            if longs[k]:
                mean_long[k] = _reduce_mean(torch.cat(longs[k]), dim=0)
        w = {f: _hist(self.metrics_config, f)[3] for f in FIELDS}
        out = {f'{self.prefix}/wosac/realism_meta_metric': mean['metametric'], f'{self.prefix}/wosac/min_ade':
               mean['min_average_displacement_error'], f'{self.prefix}/wosac/scenario_counter': int(n_all)}
        n_win = next(iter(mean_long.values())).shape[0] if mean_long else 0
        long_b = {'realism_meta_metric': mean_long.get('metametric')}
        for b, fields in BUCKETS.items():
            ws = sum(w[f] for f in fields) or 1
            out[f'{self.prefix}/wosac/{b}_metrics'] = sum(w[f] * mean[f + '_likelihood'] for f in fields) / ws
            if mean_long:
                acc = torch.zeros(n_win)
                for f in fields:
                    acc = acc + w[f] * mean_long[f + '_likelihood']
                long_b[f'{b}_metrics'] = acc / ws
        for k in self.field_names:
            out[f'{self.prefix}/wosac_likelihood/{k}'] = float(mean[k])
        for k, v in long_b.items():
            if v is not None:
                out[f'{self.prefix}/wosac_long/{k}'] = [round(x, 4) for x in v.tolist()]
        for k, v in mean_long.items():
            out[f'{self.prefix}/wosac_long_likelihood/{k}'] = [round(x, 4) for x in v.tolist()]
        self._last_long = long_b
        return out
