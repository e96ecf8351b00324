"""Scoring stage of the reference's LongMetric on the device: mirror of compute_scenario_metrics_for_bundle
(infgen/metrics/compute_metrics.py:880-1103) from MetricFeatures to the per-feature likelihoods and the meta-metric.
The windowed histogram log-likelihoods (the reference's unfold + vmap(torch.histogram) + Categorical.log_prob + masked mean)
are one launch of `infgen_window_log_likelihood` per feature; what remains is arithmetic on (objects x windows) arrays.

`config`: the reference's SimAgentMetricsConfig (protobuf) or any object / dict with the same fields per feature
(`histogram.{min_val, max_val, num_bins}` or `bernoulli`, `metametric_weight`).  `log_distributions`: per feature a
torch.distributions.Categorical (as the reference's LogDistributions holds) or a tensor of log-probabilities."""
from typing import Dict, Tuple

import torch
from torch import Tensor

from .. import _lib
from .compute_metrics import SHIFT, MetricFeatures

N_SIMULATION_STEPS = 80          # waymo_open_dataset submission_specs
KINEMATIC = ('linear_speed', 'linear_acceleration', 'angular_speed', 'angular_acceleration')
FIELDS = KINEMATIC + ('distance_to_nearest_object', 'collision_indication', 'time_to_collision', 'num_placement',
                      'num_removement', 'distance_placement', 'distance_removement')


def _field(obj, name):
    return obj[name] if isinstance(obj, dict) else getattr(obj, name)


def _hist(config, field) -> Tuple[float, float, int, float]:
    fc = _field(config, field)
    if field == 'collision_indication':
        return -0.5, 0.5, 2, float(_field(fc, 'metametric_weight'))
    h = _field(fc, 'histogram')
    return float(_field(h, 'min_val')), float(_field(h, 'max_val')), int(_field(h, 'num_bins')), float(_field(fc, 'metametric_weight'))


def _logp(log_distributions, field, dev) -> Tensor:
    d = _field(log_distributions, field)
    lp = d.logits if hasattr(d, 'logits') else torch.as_tensor(d)
    return lp.reshape(-1).to(dev, torch.float32).contiguous()


def window_log_likelihood(values: Tensor, valid, lo: float, hi: float, num_bins: int, logp: Tensor, size: int, step: int):
    """values / valid (n, T) on the GPU -> (sum of log-probabilities over the valid steps, number of valid steps) per
    (n, window)"""
    dev = values.device
    if dev.type != 'cuda':
        raise RuntimeError('window_log_likelihood runs on the GPU only (no CPU fallback)')
    v = values.to(torch.float32).contiguous()
    ok = valid.to(torch.uint8).contiguous() if valid is not None else None
    n, T = v.shape
    W = (T - size) // step + 1
    edges = torch.linspace(lo, hi, num_bins + 1).float().to(dev)          # the reference's edges, computed the same way
    s = torch.empty(n, W, dtype=torch.float32, device=dev)
    c = torch.empty(n, W, dtype=torch.int32, device=dev)
    _lib.check(_lib.load().infgen_window_log_likelihood(_lib.ptr(v), _lib.ptr(ok), n, T, size, step, _lib.ptr(edges),
                                                       _lib.ptr(logp), num_bins, _lib.ptr(s), _lib.ptr(c),
                                                       torch.cuda.current_stream(dev).cuda_stream),
               'infgen_window_log_likelihood')
    return s, c


def _masked_mean(t: Tensor, dim=None) -> Tensor:
    """reference _reduce_mean (:765-774): mean over the entries in (0, 1]"""
    ok = (t > 0) & (t <= 1)
    z = torch.where(ok, t, torch.zeros_like(t))
    return z.sum() / ok.sum().clamp(min=1) if dim is None else z.sum(0) / ok.sum(0).clamp(min=1)


@torch.no_grad()
def compute_scenario_metrics(config, log_distributions, features: MetricFeatures, size: int = N_SIMULATION_STEPS,
                             step: int = SHIFT) -> Tuple[Dict[str, float], Dict[str, Tensor]]:
    """reference compute_metrics.py:880-1103 for one rollout's MetricFeatures -> (the SimAgentMetrics fields as floats:
    `<feature>_likelihood`, `metametric`, `simulated_collision_rate`; the per-window tensors (1, n_window) of the second
    return value of the reference)."""
    f = features
    valid = f.valid
    dev = valid.device
    sv = torch.zeros_like(valid)
    sv[:, 1:-1] = valid[:, 2:] & valid[:, :-2]                           # compute_kinematic_validity
    av = torch.zeros_like(valid)
    av[:, 1:-1] = sv[:, 2:] & sv[:, :-2]

    def score(field, values, ok, sz, stp):
        lo, hi, nb, _ = _hist(config, field)
        s, c = window_log_likelihood(values, ok, lo, hi, nb, _logp(log_distributions, field, dev), sz, stp)
        return s, c

    def likelihood(field, values, ok, sz=size, stp=step):
        s, c = score(field, values, ok, sz, stp)
        if int(c.sum()) == 0:
            return torch.zeros_like(s)                                  # exp(-inf), :759-760
        return torch.exp(s / c)                                          # 0 / 0 = NaN for a window without a valid step

    per = {}
    for k, ok in zip(KINEMATIC, (sv, av, sv, av)):
        per[k] = likelihood(k, getattr(f, k), ok)
    d = f.distance_to_nearest_object
    lo, hi, _, _ = _hist(config, 'distance_to_nearest_object')
    per['distance_to_nearest_object'] = likelihood('distance_to_nearest_object', d, valid & (d >= lo) & (d <= hi))
    per['time_to_collision'] = likelihood('time_to_collision', f.time_to_collision, valid)
    tok_valid = valid[:, ::SHIFT]
    for k in ('distance_placement', 'distance_removement'):
        d = getattr(f, k)
        lo, hi, _, _ = _hist(config, k)
        per[k] = likelihood(k, d, tok_valid[:, :d.shape[1]] & (d > lo) & (d < hi), size // SHIFT, step // SHIFT)
    scal = {k: _masked_mean(v) for k, v in per.items()}
    long = {k: _masked_mean(v, 0)[None] for k, v in per.items()}
    # collision indication per (object, window): any valid colliding step; bernoulli = two bins around 0 and 1
    hit_cnt = score('collision_indication', f.collision_per_step.float(), valid & f.collision_per_step, size, step)[1]
    hit = (hit_cnt > 0).float()
    ll_hit, _ = score('collision_indication', hit.reshape(-1, 1), None, 1, 1)
    ll_hit = ll_hit.reshape(hit.shape)
    scal['collision_indication'] = _masked_mean(torch.exp(ll_hit.mean()))
    long['collision_indication'] = _masked_mean(torch.exp(ll_hit), 0)[None]
    for k in ('num_placement', 'num_removement'):
        s, c = score(k, getattr(f, k).float(), None, size // SHIFT, step // SHIFT)
        scal[k] = _masked_mean(torch.exp(s.sum() / c.sum()))
        long[k] = torch.exp(s / c)
    weights = {k: _hist(config, k)[3] for k in FIELDS}
    out = {k + '_likelihood': float(scal[k]) for k in FIELDS}
    out['metametric'] = sum(weights[k] * out[k + '_likelihood'] for k in FIELDS)
    out['simulated_collision_rate'] = float(hit.mean())
    meta_long = sum(weights[k] * long[k][0] for k in FIELDS)
    for k in FIELDS:
        meta_long = torch.where(long[k][0] == 0, torch.zeros_like(meta_long), meta_long)
    long_out = {k + '_likelihood': long[k] for k in FIELDS}
    long_out['metametric'] = meta_long[None]
    return out, long_out
