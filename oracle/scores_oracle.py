"""CPU restatement of the scoring stage of the reference's LongMetric - TEST INFRASTRUCTURE ONLY (imported by tests/,
never by the product path).  Follows infgen/metrics/compute_metrics.py: log_likelihood_estimate_timeseries (:845-878),
_reduce_average_with_validity (:744-762), _reduce_mean (:765-774), compute_scenario_metrics_for_bundle (:880-1103),
_compute_metametric(_long) (:469-498).  Pinned on the reference's own output with its own metric_config.textproto
(tests/golden/make_golden_scores.py -> scores_platoon_n20_r200.npz; tests/test_oracle_golden.py).

A feature value is scored by the bin of a fixed histogram it falls into (values outside the range or NaN land in bin 0,
like torch.histogram + argmax there); its log-probability comes from the logged distribution of that feature.  A window
of 80 steps (16 token steps) every 5 (1) is scored by the exponential of its average log-probability over valid steps."""
import torch

KINEMATIC = ('linear_speed', 'linear_acceleration', 'angular_speed', 'angular_acceleration')
FIELDS = KINEMATIC + ('distance_to_nearest_object', 'collision_indication', 'time_to_collision', 'num_placement',
                      'num_removement', 'distance_placement', 'distance_removement')


def bin_index(v, lo, hi, nb):
    edges = torch.linspace(lo, hi, nb + 1).float()
    v = v.contiguous()
    idx = torch.bucketize(v, edges, right=True) - 1            # edges[i] <= v < edges[i + 1]
    idx = torch.where(v == edges[-1], torch.full_like(idx, nb - 1), idx)      # the last bin is closed on the right
    return torch.where((idx < 0) | (idx >= nb) | torch.isnan(v), torch.zeros_like(idx), idx)


def windows(t, size, step):
    return torch.stack([t[:, s:s + size] for s in range(0, t.shape[1] - size + 1, step)], 1)


def window_score(values, valid, logp, cfg, size, step):
    """values, valid (n, T) -> exp(average log-probability over the valid steps) per (n, window); 0/0 = NaN"""
    ll = windows(logp[bin_index(values, cfg[0], cfg[1], int(cfg[2]))], size, step)
    v = windows(valid, size, step)
    if v.sum() == 0:
        return torch.zeros(v.shape[:2])
    return torch.exp(torch.where(v, ll, torch.zeros_like(ll)).sum(-1) / v.sum(-1))


def masked_mean(t, dim=None):
    ok = (t > 0) & (t <= 1)
    if dim is None:
        return torch.where(ok, t, torch.zeros_like(t)).sum() / ok.sum().clamp(min=1)
    return torch.where(ok, t, torch.zeros_like(t)).sum(0) / ok.sum(0).clamp(min=1)


def scenario_scores(feat, logp, cfg, size=80, step=5, shift=5):
    """feat: MetricFeatures-like dict of one rollout (valid, the 11 feature arrays, collision_per_step); logp / cfg: per
    field the log-probabilities of the logged distribution and (min, max, bins, pseudocount, weight).
    -> (scalars per field + metametric + collision rate, per-window values per field + metametric)"""
    valid = feat['valid']
    sv = torch.zeros_like(valid)
    sv[:, 1:-1] = valid[:, 2:] & valid[:, :-2]
    av = torch.zeros_like(valid)
    av[:, 1:-1] = sv[:, 2:] & sv[:, :-2]
    per = {}
    for k, v in zip(KINEMATIC, (sv, av, sv, av)):
        per[k] = window_score(feat[k], v, logp[k], cfg[k], size, step)
    d = feat['distance_to_nearest_object']
    c = cfg['distance_to_nearest_object']
    per['distance_to_nearest_object'] = window_score(d, valid & (d >= c[0]) & (d <= c[1]), logp['distance_to_nearest_object'], c,
                                                     size, step)
    per['time_to_collision'] = window_score(feat['time_to_collision'], valid, logp['time_to_collision'],
                                            cfg['time_to_collision'], size, step)
    hit = windows(valid & feat['collision_per_step'], size, step).any(-1)               # (n, W)
    ll_hit = logp['collision_indication'][bin_index(hit.float(), -0.5, 0.5, 2)]
    tok_valid = valid[:, ::shift]
    for k in ('distance_placement', 'distance_removement'):
        d, c = feat[k], cfg[k]
        per[k] = window_score(d, tok_valid[:, :d.shape[1]] & (d > c[0]) & (d < c[1]), logp[k], c, size // shift, step // shift)
    scal = {k: masked_mean(v) for k, v in per.items()}
    long = {k: masked_mean(v, 0)[None] for k, v in per.items()}
    scal['collision_indication'] = masked_mean(torch.exp(ll_hit.mean()))
    long['collision_indication'] = masked_mean(torch.exp(ll_hit[..., None].mean(-1)), 0)[None]
    for k in ('num_placement', 'num_removement'):
        c = cfg[k]
        ll = windows(logp[k][bin_index(feat[k].float(), c[0], c[1], int(c[2]))], size // shift, step // shift)
        scal[k] = masked_mean(torch.exp(ll.mean()))
        long[k] = torch.exp(ll.mean(-1))
    meta = sum(cfg[k][4] * float(scal[k]) for k in FIELDS)
    meta_long = sum(cfg[k][4] * long[k][0] for k in FIELDS)
    for k in FIELDS:
        meta_long = torch.where(long[k][0] == 0, torch.zeros_like(meta_long), meta_long)
    scal = {k: float(v) for k, v in scal.items()}
    scal.update(metametric=meta, simulated_collision_rate=float(hit.float().mean()))
    long['metametric'] = meta_long[None]
    return scal, long
