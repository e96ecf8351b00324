"""CPU: the oracle restatement is pinned against fixtures produced by the REFERENCE's own
modules (tests/golden/make_golden.py): tokens/states exact, poses bit-identical, hooked logits
to fp32 round-off (scaled by the sharpening gain of the fixture)."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES, load_case
from oracle import rollout_oracle as ro


@pytest.mark.parametrize('case', GOLDEN_CASES)
def test_oracle_matches_reference_fixture(case):
    c = load_case(case)
    z, m = c['z'], c['meta']
    sd = {k: torch.from_numpy(v) for k, v in c['sd'].items()}
    torch.set_num_threads(8)
    out = ro.run_scene(sd, c['scene'], c['cfg'], c['vocab'], c['map_vocab'], c['grid'], live_state=m['live_state'])
    assert np.abs(out['x_pt'].numpy() - z['x_pt']).max() <= 1e-5
    assert np.array_equal(out['next_token_idx'].numpy(), z['next_token_idx'])
    assert np.array_equal(out['next_state_idx'].numpy(), z['next_state_idx'])
    nl = z['logits'].shape[0]                   # (the C3-sized fixture keeps the full logits of its first steps only ...)
    assert np.abs(out['logits'].numpy()[:nl] - z['logits']).max() <= 1e-5 * max(1.0, m['head_gain']) * 4
    if 'logit_max' in z.files:                  # (... and every row's maximum / arg-max for all steps)
        assert np.abs(out['logits'].numpy().max(-1) - z['logit_max']).max() <= 1e-5 * max(1.0, m['head_gain']) * 4
        assert np.array_equal(out['logits'].numpy().argmax(-1), z['logit_argmax'])
    for k in ('pos_a', 'head_a', 'pred_traj', 'pred_head', 'pred_state'):
        assert np.abs(out[k].numpy() - z[k]).max() <= 1e-5, k
    assert np.array_equal(out['pred_valid'].numpy(), z['pred_valid'])
    assert np.array_equal(out['agent_id'].numpy(), z['agent_id'])
    assert out['ego_index'] == int(z['ego_index'])
    assert np.array_equal(out['edge_count'], z['edge_count'])


@pytest.mark.parametrize('case', ['c1_a8_m128', 'a24_m256_edge'])
def test_reference_shaped_oracle_matches_reference_fixture(case):
    """the all-columns control flow of agent_decoder.py:2133-2158 (bench.py's "reference-shaped" CPU baseline) gives the
    reference's tokens / logits too - it only re-does the node-side work of every column"""
    c = load_case(case)
    z, m = c['z'], c['meta']
    sd = {k: torch.from_numpy(v) for k, v in c['sd'].items()}
    torch.set_num_threads(8)
    out = ro.run_scene(sd, c['scene'], c['cfg'], c['vocab'], c['map_vocab'], c['grid'], live_state=m['live_state'],
                       all_columns=True)
    assert np.array_equal(out['next_token_idx'].numpy(), z['next_token_idx'])
    assert np.array_equal(out['next_state_idx'].numpy(), z['next_state_idx'])
    assert np.abs(out['logits'].numpy() - z['logits']).max() <= 1e-5 * max(1.0, m['head_gain']) * 4
    assert np.array_equal(out['edge_count'], z['edge_count'])


def _sorted_triples(step, dst, src):
    tri = np.stack([np.full(len(dst), step, np.int64), np.asarray(dst, np.int64), np.asarray(src, np.int64)], 1).reshape(-1, 3)
    return tri[np.lexsort((tri[:, 2], tri[:, 1]))].astype(np.int32)


@pytest.mark.parametrize('case', ['c1_a8_m128', 'a24_m256_edge', 'c3_a64_m1024'])
def test_oracle_edge_lists_and_triple_outputs_match_the_reference(case):
    """below the logits level: the edge LISTS of every decode step - sorted (step, destination agent, source
    column | source agent | source map token) of the temporal / agent / map sets, agent_decoder.py:540-758 - are the reference's
    own (tests/golden/make_golden_internals.py hooks its three builders), and so is the residual stream of the current column
    after the first and the last layer triple of decode steps 0..2 (hooks on a2a_attn_layers[0] / [L-1], :2133-2158)"""
    import os
    from conftest import GOLDEN
    c = load_case(case)
    zi = np.load(os.path.join(GOLDEN, case + '_internals.npz'))
    sd = {k: torch.from_numpy(v) for k, v in c['sd'].items()}
    torch.set_num_threads(8)
    tr = {}
    out = ro.run_scene(sd, c['scene'], c['cfg'], c['vocab'], c['map_vocab'], c['grid'], live_state=c['meta']['live_state'], trace=tr)
    assert np.array_equal(out['next_token_idx'].numpy(), zi['next_token_idx'])
    assert np.array_equal(zi['edge_count'], c['z']['edge_count'])             # (both generators saw the same run)
    for kind, key in (('t', 'edges_t'), ('a', 'edges_a'), ('m', 'edges_m')):
        mine = [_sorted_triples(s, e[kind][1].numpy(), e[kind][0].numpy()) for s, e in enumerate(tr['edges'])]
        mine = np.concatenate(mine) if sum(len(x) for x in mine) else np.zeros((0, 3), np.int32)
        assert np.array_equal(mine, zi[key]), kind
    L = c['cfg'].num_agent_layers
    for s in range(zi['act_first'].shape[0]):
        assert np.abs(tr['x'][s][0].numpy() - zi['act_first'][s]).max() <= 2e-5, s
        assert np.abs(tr['x'][s][L - 1].numpy() - zi['act_last'][s]).max() <= 2e-5, s


def test_quirk_last_ten_rows_have_no_temporal_edges():
    """SURVEY a-Q1: with A <= 10 there are no temporal edges at all (fixture c1 has A = 8)."""
    z = load_case('c1_a8_m128')['z']
    assert (z['edge_count'][:, 0] == 0).all()
    z = load_case('a24_m256_edge')['z']
    assert (z['edge_count'][:, 0] > 0).all()


INSERTION_CASES = ['ins_forced_a16_m256', 'ins_natural_a20_m256', 'ins_sampled_a16_m256']


@pytest.mark.parametrize('case', INSERTION_CASES)
def test_insertion_oracle_matches_reference_fixture(case):
    """scenario insertion (agent_decoder.py:1773-2105): forced enter (DEBUG=1), the natural seed head, and the stochastic cell
    choice (top-10 multinomial, :1900-1904) replayed from the fixture's uniforms - 119 of its 150 draws hit an occupied cell and
    `continue` (:1906-1909)"""
    from oracle import insertion_oracle as io
    c = load_case(case)
    z, m = c['z'], c['meta']
    cfg = c['cfg']
    cfg.disable_insertion = False
    sd = {k: torch.from_numpy(v) for k, v in c['sd'].items()}
    out = io.run_scene_with_insertion(sd, c['scene'], cfg, c['vocab'], c['map_vocab'], c['grid'],
                                      force_enter=(m['insertion'] == 'forced'), insert_k=m.get('insert_k', 1),
                                      insert_uniforms=z['insert_uniforms'] if 'insert_uniforms' in z.files else None)
    assert out['n_agents'].tolist() == z['n_agents_step'].tolist()          # same insertions at the same steps
    assert out['n_agents'][-1] > out['n_agents'][0]
    assert np.array_equal(out['next_token_idx'].numpy(), z['next_token_idx'])
    assert np.array_equal(out['next_state_idx'].numpy(), z['next_state_idx'])
    assert np.array_equal(out['agent_id'].numpy(), z['agent_id'])
    assert np.array_equal(out['pred_type'].numpy(), z['pred_type'])
    assert np.abs(out['pred_shape'].numpy() - z['pred_shape']).max() <= 1e-5
    for i, l in enumerate(out['logits']):
        assert np.abs(l.numpy() - z['logits'][i, :l.shape[0]]).max() <= 1e-5 * m['head_gain'] * 4
    for k in ('pos_a', 'head_a', 'pred_traj', 'pred_head', 'pred_state'):
        assert np.abs(out[k].numpy() - z[k]).max() <= 1e-4, k
    assert np.array_equal(out['edge_count'], z['edge_count'])


@pytest.mark.parametrize('case', ['tok_a48', 'tok_a7'])
def test_token_match_oracle_reproduces_the_reference(case):
    """oracle/token_match_oracle.py vs TokenProcessor._match_agent_token run by tests/golden/make_golden_tokens.py:
    token indices and matched contours bit-identical (48 / 7 agents, 18 steps, validity dropouts)"""
    import os
    import torch
    from conftest import GOLDEN
    from infgen_amd import synth
    from oracle import token_match_oracle as tm
    z = np.load(os.path.join(GOLDEN, case + '.npz'))
    vocab = synth.make_agent_vocab(synth.standard_config().token_size)
    tt = torch.stack([torch.from_numpy(vocab[('veh', 'ped', 'cyc')[k]][:, -1]) for k in z['type']])
    idx, con = tm.match_agent_token(torch.from_numpy(z['valid']), torch.from_numpy(z['pos'][..., :2].copy()),
                                    torch.from_numpy(z['heading']), torch.from_numpy(z['shape']), tt)
    assert np.array_equal(idx.numpy(), z['token_index'])
    assert np.array_equal(con.numpy(), z['token_contour'])


@pytest.mark.parametrize('case', ['maptok_p500', 'maptok_p3'])
def test_map_token_match_oracle_reproduces_the_reference(case):
    """oracle match_token_map vs InfGen.match_token_map (tests/golden/make_golden_tokens.py): identical token ids"""
    import os
    import torch
    from conftest import GOLDEN
    from infgen_amd import synth
    from oracle import token_match_oracle as tm
    z = np.load(os.path.join(GOLDEN, case + '.npz'))
    sample_pt = torch.from_numpy(np.ascontiguousarray(synth.make_map_vocab()[:, ::5]).astype(np.float32))
    idx = tm.match_token_map(torch.from_numpy(z['traj_pos']), torch.from_numpy(z['traj_theta']), sample_pt)
    assert np.array_equal(idx.numpy(), z['token_idx'])
    assert np.array_equal(z['position'][:, :2], z['traj_pos'][:, 0])


@pytest.mark.parametrize('case', ['dist_n24_t30', 'dist_n5_t4', 'ttc_platoon_n20_t30'])
def test_nearest_object_distance_oracle_reproduces_the_reference(case):
    """oracle/metrics_oracle.py vs compute_distance_to_nearest_object run by tests/golden/make_golden_metrics.py: bit-identical"""
    import os
    import torch
    from conftest import GOLDEN
    from oracle import metrics_oracle as mo
    z = np.load(os.path.join(GOLDEN, case + '.npz'))
    t = {k: torch.from_numpy(z[k]) for k in ('cx', 'cy', 'length', 'width', 'heading', 'valid', 'eval_mask')}
    d = mo.distance_to_nearest_object(t['cx'], t['cy'], t['length'], t['width'], t['heading'], t['valid'], t['eval_mask'])
    assert np.array_equal(d.numpy(), z['distance'])
    ttc = mo.time_to_collision(t['cx'], t['cy'], t['length'], t['width'], t['heading'], t['valid'], t['eval_mask'], 0.1)
    assert np.array_equal(ttc.numpy(), z['ttc'])
    kin = mo.kinematic_features(t['cx'], t['cy'], torch.zeros_like(t['cx']), t['heading'], 0.1)
    for a, n in zip(kin, ('speed', 'accel', 'yaw_rate', 'yaw_accel')):
        assert np.array_equal(a.numpy(), z[n], equal_nan=True)
    pos = torch.stack([t['cx'], t['cy'], torch.zeros_like(t['cx'])], -1)
    pl = mo.placement_features(pos, torch.from_numpy(z['state']), z['cx'].shape[0] - 1)
    for a, n in zip(pl, ('num_bos', 'num_eos', 'bos_distance', 'eos_distance')):
        assert np.array_equal(a.numpy(), z[n])


@pytest.mark.parametrize('case', ['road_n24_t30', 'road_n5_t4'])
def test_road_edge_oracle_matches_reference(case):
    """oracle/metrics_oracle.distance_to_road_edge vs the reference's compute_distance_to_road_edge (fixture made by
    tests/golden/make_golden_road.py); 1e-5 m: the corner rotation is not restated bit for bit (observed 4.8e-7)"""
    import os
    from conftest import GOLDEN
    from oracle import metrics_oracle as mo
    z = np.load(os.path.join(GOLDEN, case + '.npz'))
    t = {k: torch.from_numpy(z[k]) for k in z.files if z[k].ndim > 0}
    roads = np.split(z['road_points'], np.cumsum(z['road_lengths'])[:-1])
    poly, cyc = mo.tensorize_polylines(roads)
    assert poly.shape[0] == sum(n >= 2 for n in z['road_lengths']) and bool(cyc[0]) and not bool(cyc[1])
    out = mo.distance_to_road_edge(t['cx'], t['cy'], t['cz'], t['length'], t['width'], t['height'], t['heading'], t['valid'],
                                   t['eval_mask'], poly, cyc)
    assert (out - t['distance']).abs().max() <= 1e-5
    assert torch.equal(out > 0, t['distance'] > 0)


def _features_fixture():
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'features_platoon_n20.npz'))
    scen = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('in_')}
    scen['av_id'] = int(z['av_id'])
    roads = np.split(z['road_points'], np.cumsum(z['road_lengths'])[:-1])
    return z, scen, roads


def test_rollout_sink_layout_and_oracle_features():
    """host side of the rollout sink (output_to_rollouts, infgen/metrics/compute_metrics.py:360-463) on the fixture's
    rollouts dict, and the oracle functions composed like compute_metric_features (:560-707) against the REFERENCE's
    features of the same dict (tests/golden/make_golden_features.py)"""
    from infgen_amd.metrics import compute_metrics as cmx
    from oracle import metrics_oracle as mo
    z, scen, roads = _features_fixture()
    assert torch.equal(cmx.get_scenario_id_int_tensor(['a1b2c3d4e5f6']), scen['scenario_id'])
    sr = cmx.output_to_rollouts(scen)
    assert len(sr) == 1 and sr[0].scenario_id == str(z['scenario_str']) and len(sr[0].joint_scenes) == 1
    sim = sr[0].joint_scenes[0]
    N, T = sim.x.shape
    assert (N, T) == (20, 91) and sim.length.shape == (N, T) and sim.state.shape == (N, 19) and sim.token_pos.shape == (N, 19, 2)
    assert sim.av_id == sim.processed_av_id == 119 and torch.equal(sim.object_id, torch.arange(100, 120))
    sub = sim.gather_objects_by_id(torch.from_numpy(z['eval_ids']))
    assert torch.equal(sub.x, sim.x[[19, 3, 11, 4]]) and sub.state is sim.state
    with pytest.raises(ValueError):
        sim.gather_objects_by_id(torch.tensor([5]))
    every = torch.ones(N, dtype=torch.bool)
    kin = mo.kinematic_features(sim.x, sim.y, sim.z, sim.heading, 0.1)
    for a, n in zip(kin, ('linear_speed', 'linear_acceleration', 'angular_speed', 'angular_acceleration')):
        assert np.array_equal(a[:, 11:].numpy(), z['f_' + n], equal_nan=True), n
    d = mo.distance_to_nearest_object(sim.x, sim.y, sim.length, sim.width, sim.heading, sim.valid, every)[:, 11:]
    assert np.array_equal(d.numpy(), z['f_distance_to_nearest_object'])
    assert np.array_equal((d < 0).numpy(), z['f_collision_per_step'])
    ttc = mo.time_to_collision(sim.x, sim.y, sim.length, sim.width, sim.heading, sim.valid, every, 0.1)[:, 11:]
    assert np.array_equal(ttc.numpy(), z['f_time_to_collision'])
    poly, cyc = mo.tensorize_polylines(roads)
    road = mo.distance_to_road_edge(sim.x, sim.y, sim.z, sim.length, sim.width, sim.height, sim.heading, sim.valid, every,
                                    poly, cyc)[:, 11:]
    assert (road - torch.from_numpy(z['f_distance_to_road_edge'])).abs().max() <= 1e-5
    assert np.array_equal((road > 0).numpy(), z['f_offroad_per_step'])
    pos3 = torch.cat([sim.token_pos, torch.zeros(N, 19, 1)], -1)
    nb, ne, db, de = mo.placement_features(pos3, sim.state, 19)
    assert np.array_equal(nb[None, 2:].numpy(), z['f_num_placement']) and np.array_equal(ne[None, 2:].numpy(), z['f_num_removement'])
    assert np.array_equal(db[:, 2:].numpy(), z['f_distance_placement']) and np.array_equal(de[:, 2:].numpy(), z['f_distance_removement'])


@pytest.mark.parametrize('case', ['tokenize_a40', 'tokenize_a6'])
def test_tokenize_agent_oracle_matches_reference(case):
    """oracle/token_match_oracle.tokenize_agent vs the reference's TokenProcessor._tokenize_agent
    (tests/golden/make_golden_tokenize.py): everything bit-identical except token_heading (atan2 of a strided view takes
    torch's scalar path: 2.4e-7)"""
    import os
    from conftest import GOLDEN
    from infgen_amd import synth
    from oracle import token_match_oracle as tm
    z = np.load(os.path.join(GOLDEN, case + '.npz'))
    vocab = synth.make_agent_vocab(synth.standard_config().token_size)
    last = torch.stack([torch.from_numpy(vocab[k][:, -1]) for k in ('veh', 'ped', 'cyc')])
    i = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('in_')}
    o = tm.tokenize_agent(i['valid_mask'], i['position'], i['heading'], i['velocity'], i['type'], i['shape'], last)
    for k in ('token_idx', 'state_idx', 'token_contour', 'token_pos', 'agent_valid_mask', 'raw_agent_valid_mask', 'shape',
              'valid_mask', 'heading', 'velocity'):
        assert np.array_equal(o[k].numpy(), z['out_' + k]), k
    assert np.abs(o['token_heading'].numpy() - z['out_token_heading']).max() <= 1e-6
    h = np.array([float(o['raw_height'][k]) for k in ('veh', 'ped', 'cyc')], np.float32)
    assert np.array_equal(h, z['out_raw_height'], equal_nan=True)


def test_fetch_enterings_oracle_matches_reference():
    """oracle/enterings_oracle.fetch_enterings vs the reference's InfGen._fetch_enterings with its own Attr_Tokenizer
    (tests/golden/make_golden_enterings.py): all nine outputs bit-identical (two scenes, 40 agents, 300 map tokens)"""
    import os
    from conftest import GOLDEN
    from oracle import enterings_oracle as eo
    z = np.load(os.path.join(GOLDEN, 'enterings_a40.npz'))
    t = lambda k: torch.from_numpy(z[k])
    o = eo.fetch_enterings(t('token_pos'), t('token_heading'), t('state_idx'), t('batch'), t('av_index'), t('grid'), 75.0,
                           3.0, pt_pos=t('pt_pos'), pt_batch=t('pt_batch'))
    assert len(o) == 9
    for k, v in o.items():
        assert np.array_equal(v.numpy(), z['out_' + k]), k


def _scores_fixture():
    """inputs of tests/golden/make_golden_scores.py -> (rollouts dict, feature dict by the oracle functions, logp, cfg, z)"""
    import os
    from conftest import GOLDEN
    from oracle import metrics_oracle as mo, scores_oracle as so
    z = np.load(os.path.join(GOLDEN, 'scores_platoon_n20_r200.npz'))
    scen = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('in_')}
    scen['av_id'] = int(z['av_id'])
    fields = [str(f) for f in z['fields']]
    cfg = {f: z['config'][i].tolist() for i, f in enumerate(fields)}
    logp = {f: torch.from_numpy(z['logp_' + f]) for f in fields}
    return z, scen, fields, cfg, logp


def test_scoring_oracle_matches_reference():
    """oracle/scores_oracle.scenario_scores on features from the oracle functions vs the reference's
    compute_scenario_metrics_for_bundle with its own metric_config.textproto (25 windows of a 200-step rollout): the eleven
    likelihoods, their per-window values, the meta-metric and the collision rate"""
    from oracle import metrics_oracle as mo, scores_oracle as so
    z, scen, fields, cfg, logp = _scores_fixture()
    x, y = scen['pred_traj'][:, 0, :, 0], scen['pred_traj'][:, 0, :, 1]
    N, T = x.shape
    hd, valid = scen['pred_head'][:, 0], scen['pred_valid'][:, 0]
    ln, wd = scen['pred_shape'][:, 0, 0:1].expand(N, T), scen['pred_shape'][:, 0, 1:2].expand(N, T)
    every = torch.ones(N, dtype=torch.bool)
    feat = dict(valid=valid[:, 11:])
    for k, a in zip(so.KINEMATIC, mo.kinematic_features(x, y, torch.zeros_like(x), hd, 0.1)):
        feat[k] = a[:, 11:]
    d = mo.distance_to_nearest_object(x, y, ln, wd, hd, valid, every)[:, 11:]
    feat.update(distance_to_nearest_object=d, collision_per_step=d < 0,
                time_to_collision=mo.time_to_collision(x, y, ln, wd, hd, valid, every, 0.1)[:, 11:])
    pos3 = torch.cat([scen['token_pos'][:, 0], torch.zeros(N, scen['token_pos'].shape[2], 1)], -1)
    nb, ne, db, de = mo.placement_features(pos3, scen['pred_state'][:, 0], N - 1)
    feat.update(num_placement=nb[None, 2:], num_removement=ne[None, 2:], distance_placement=db[:, 2:], distance_removement=de[:, 2:])
    scal, long = so.scenario_scores(feat, logp, cfg)
    for f in fields:
        assert abs(scal[f] - float(z['m_' + f + '_likelihood'])) <= 1e-7, f
        assert torch.equal(long[f], torch.from_numpy(z['l_' + f + '_likelihood'])), f
    assert abs(scal['metametric'] - float(z['metametric'])) <= 1e-6
    assert torch.equal(long['metametric'], torch.from_numpy(z['l_metametric']))
    assert abs(scal['simulated_collision_rate'] - float(z['simulated_collision_rate'])) <= 1e-7


def test_long_metric_host_side_matches_reference():
    """infgen_amd.metrics.long_metric (host side of LongMetric): the logged distributions equal the reference's
    _get_log_distributions on the same logged values; update + compute of one scenario reproduce the reference's bucket
    aggregation (tests/golden/make_golden_scores.py)"""
    from infgen_amd.metrics.long_metric import LongMetric, get_log_distributions
    z, scen, fields, cfg, logp = _scores_fixture()
    config = {f: dict(histogram=dict(min_val=c[0], max_val=c[1], num_bins=int(c[2]), additive_smoothing_pseudocount=c[3]),
                      bernoulli=dict(additive_smoothing_pseudocount=c[3]), metametric_weight=c[4]) for f, c in cfg.items()}
    for f in fields:
        d = get_log_distributions(f, config, torch.from_numpy(z['logv_' + f]))
        assert torch.equal(d.logits[0], logp[f]), f
    scal = {f + '_likelihood': float(z['m_' + f + '_likelihood']) for f in fields}
    scal.update(metametric=float(z['metametric']), simulated_collision_rate=float(z['simulated_collision_rate']))
    long = {f + '_likelihood': torch.from_numpy(z['l_' + f + '_likelihood']) for f in fields}
    long['metametric'] = torch.from_numpy(z['l_metametric'])
    lm = LongMetric('val', config, log_distributions=logp)
    lm.update(metrics=(scal, long))
    out = lm.compute()
    assert out['val/wosac/scenario_counter'] == 1
    for b in ('kinematic', 'interactive', 'map_based', 'placement_based'):
        assert abs(out[f'val/wosac/{b}_metrics'] - float(z[f'b_{b}_metrics'])) <= 1e-6, b
        assert np.abs(lm._last_long[f'{b}_metrics'].numpy() - z[f'bl_{b}_metrics']).max() <= 1e-6, b
    assert abs(out['val/wosac/realism_meta_metric'] - float(z['b_realism_meta_metric'])) <= 1e-6
    assert np.abs(lm._last_long['realism_meta_metric'].numpy() - z['bl_realism_meta_metric']).max() <= 1e-6
    other = LongMetric('val', config, log_distributions=logp)
    other.update(metrics=(scal, long))
    lm.merge(other.state())
    assert lm.compute()['val/wosac/scenario_counter'] == 2
    assert abs(lm.compute()['val/wosac/kinematic_metrics'] - float(z['b_kinematic_metrics'])) <= 1e-6
