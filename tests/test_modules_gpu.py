"""GPU: the reference-shaped module API (infgen_amd.modules) — same constructor/forward
signatures as infgen/modules/{layers,infgen_decoder}.py — against the oracle and the golden
fixtures.  These read like tests of the reference's own modules would."""
import numpy as np
import pytest
import torch

from conftest import load_case, make_weights
from test_boundary_cpu import _decoder

pytestmark = pytest.mark.gpu


def _to_data(scene, dev):
    d = {}
    for k, v in scene.items():
        if isinstance(v, dict):
            d[k] = {kk: (torch.from_numpy(vv).to(dev) if isinstance(vv, np.ndarray) else vv) for kk, vv in v.items()}
        else:
            d[k] = torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v
    d[('pt_token', 'to', 'map_polygon')] = d.pop('pt_token__to__map_polygon')
    a = d['agent']
    for k in ('agent_valid_mask', 'category', 'valid_mask', 'av_index', 'shape'):
        d[k] = a[k]
    return d


def _load(dec, sd):
    full = {k: torch.from_numpy(sd[k]) if k in sd else v for k, v in dec.state_dict().items()}
    dec.load_state_dict(full, strict=True)


@pytest.mark.parametrize('name', ['c1_a8_m128', 'a24_m256_edge'])
def test_infgen_decoder_inference_drop_in(name):
    c = load_case(name)
    z = c['z']
    dev = torch.device('cuda:0')
    dec = _decoder(c['cfg'])
    _load(dec, c['sd'])
    dec = dec.to(dev).eval()
    data = _to_data(c['scene'], dev)
    bs_before = int(data['batch_size_a'][0])
    out = dec.inference(data)
    for k in ('ego_index', 'agent_id', 'valid_mask', 'pos_a', 'head_a', 'gt_traj', 'pred_traj', 'pred_head', 'pred_type',
              'pred_state', 'pred_z', 'pred_shape', 'eval_shape', 'pred_valid', 'next_state_prob_seed',
              'next_pos_rel_prob_seed', 'next_token_idx', 'next_state_idx', 'grid_agent_occ_seed', 'grid_pt_occ_seed',
              'grid_agent_occ_gt_seed', 'agent_labels', 'log_message', 'x_pt', 'scenario_id', 'av_index'):
        assert k in out, k
    assert np.array_equal(out['next_token_idx'].cpu().numpy(), z['next_token_idx'])
    assert np.array_equal(out['next_state_idx'].cpu().numpy(), z['next_state_idx'])
    assert np.abs(out['pos_a'].cpu().numpy() - z['pos_a']).max() <= 1e-3
    assert np.abs(out['pred_traj'].cpu().numpy() - z['pred_traj']).max() <= 1e-3
    assert np.array_equal(out['pred_valid'].cpu().numpy(), z['pred_valid'])
    assert np.abs(out['x_pt'].cpu().numpy() - z['x_pt']).max() <= 1e-4
    # reference side effect: batch_size_a reduced by the rows filtered before the ego (agent_decoder.py:1649)
    removed = c['meta']['A'] - z['pos_a'].shape[0]
    assert int(data['batch_size_a'][0]) == bs_before - removed
    # inference_no_map with the map encoder's own output reproduces the same rollout
    me = dec.map_encoder(data)
    assert np.abs(me['x_pt'].cpu().numpy() - z['x_pt']).max() <= 1e-4
    out2 = dec.inference_no_map(_to_data(c['scene'], dev), me)
    assert np.array_equal(out2['next_token_idx'].cpu().numpy(), z['next_token_idx'])


def test_inference_batch_equals_single():
    c = load_case('c1_a8_m128')
    dev = torch.device('cuda:0')
    dec = _decoder(c['cfg'])
    _load(dec, c['sd'])
    dec = dec.to(dev).eval()
    from infgen_amd import synth
    scenes = [c['scene'], synth.make_scene(77, 11, 90, c['cfg'], vocab=c['vocab'], grid=c['grid'])]
    outs = dec.inference_batch([_to_data(s, dev) for s in scenes])
    single = dec.inference(_to_data(scenes[1], dev))
    assert np.array_equal(outs[0]['next_token_idx'].cpu().numpy(), c['z']['next_token_idx'])
    assert torch.equal(outs[1]['next_token_idx'], single['next_token_idx'])
    assert set(outs[1]) == set(single), set(outs[1]) ^ set(single)       # the batch entry returns inference's key set


def test_inference_rollouts_carry_the_seed_outputs(monkeypatch):
    """the n-copies batch of inference_rollouts (reference loop infgen/model/infgen.py:704-706) returns for every
    rollout what inference() returns for it - key set, and (greedy) the same seed-node arrays instead of zero placeholders"""
    c = load_case('ins_forced_a16_m256')
    monkeypatch.setenv('DEBUG', '1')
    dev = torch.device('cuda:0')
    dec = _decoder(c['cfg'])
    dec.agent_encoder.disable_insertion = False
    _load(dec, c['sd'])
    dec = dec.to(dev).eval()
    single = dec.inference(_to_data(c['scene'], dev))
    outs = dec.inference_rollouts(_to_data(c['scene'], dev), 2)
    assert len(outs) == 2
    for o in outs:
        assert set(o) == set(single), set(o) ^ set(single)
        assert torch.equal(o['next_token_idx'], single['next_token_idx'])
        for k in ('next_state_prob_seed', 'next_pos_rel_prob_seed', 'grid_agent_occ_seed', 'grid_pt_occ_seed',
                  'grid_agent_occ_gt_seed'):
            assert float(single[k].abs().sum()) > 0 and torch.allclose(o[k], single[k], atol=1e-5), k


def test_infgen_decoder_inference_with_insertion(monkeypatch):
    """disable_insertion=False through the module API; DEBUG=1 forces 'enter' like the reference"""
    c = load_case('ins_forced_a16_m256')
    z = c['z']
    monkeypatch.setenv('DEBUG', '1')
    dev = torch.device('cuda:0')
    cfg = c['cfg']
    dec = _decoder(cfg)
    dec.agent_encoder.disable_insertion = False
    _load(dec, c['sd'])
    dec = dec.to(dev).eval()
    out = dec.inference(_to_data(c['scene'], dev))
    assert out['pos_a'].shape[0] == z['pos_a'].shape[0] == 36
    assert np.array_equal(out['next_token_idx'].cpu().numpy(), z['next_token_idx'])
    assert np.array_equal(out['agent_id'].cpu().numpy(), z['agent_id'])
    assert 'inserted' in out['log_message']
    # the decoder reuses its engine for a second call of the same layout - nothing a call returned may
    # alias engine-owned buffers.  A second call on a DIFFERENT scene of the same shape must leave the first call's tensors alone
    keys = ('pos_a', 'head_a', 'next_token_idx', 'next_state_prob_seed', 'next_pos_rel_prob_seed', 'grid_agent_occ_seed',
            'grid_pt_occ_seed', 'grid_agent_occ_gt_seed')
    snap = {k: out[k].clone() for k in keys}
    assert float(snap['next_pos_rel_prob_seed'].abs().sum()) > 0
    from infgen_amd import synth
    other = synth.make_scene(99321, 16, 256, cfg, ego_last=True, vocab=c['vocab'], grid=c['grid'])
    out2 = dec.inference(_to_data(other, dev))
    for k in keys:
        assert torch.equal(out[k], snap[k]), k
    assert not torch.equal(out2['next_pos_rel_prob_seed'], snap['next_pos_rel_prob_seed'])


@pytest.mark.parametrize('insertion', [False, True])
def test_inference_batch_device_side_reload_equals_the_host_path(insertion, monkeypatch):
    """the drop-in entry sets a one-shape batch of device tensors up ON the device when it reuses an engine
    (RolloutEngine.reload_device: no device -> host -> device round trip of the scene arrays).  Every returned array must equal
    what the host setup gives for the same batch - here the first call (new engine: host setup) against the second (reload on
    the device) and against a decoder that is kept on the host path."""
    from infgen_amd import synth
    from infgen_amd.modules import infgen_decoder as idm
    c = load_case('ins_forced_a16_m256' if insertion else 'c1_a8_m128')
    cfg = c['cfg']
    if insertion:
        monkeypatch.setenv('DEBUG', '1')
    dev = torch.device('cuda:0')
    dec = _decoder(cfg)
    dec.agent_encoder.disable_insertion = not insertion
    _load(dec, c['sd'])
    dec = dec.to(dev).eval()
    A, M = (16, 256) if insertion else (11, 90)
    mk = lambda seeds: [synth.make_scene(s, A, M, cfg, ego_last=(s % 2 == 0), vocab=c['vocab'], grid=c['grid']) for s in seeds]
    first, second = mk(range(500, 509)), mk(range(600, 609))
    keep = [sc for sc in first + second if (np.asarray(sc['agent']['state_idx'])[:, cfg.hist_columns - 1] != 0).all()]
    assert len(keep) >= 16                                    # (no filtered rows: the batches stay on the device path)
    first, second = keep[:8], keep[8:16]
    dec.inference_batch([_to_data(s, dev) for s in first])             # builds the engine (host setup)
    eng = next(iter(dec._engines.values()))
    assert not getattr(eng, '_hosts_light', False)
    outs = dec.inference_batch([_to_data(s, dev) for s in second])     # reuses it: set up on the device
    assert next(iter(dec._engines.values())) is eng and eng._hosts_light is True
    outs = [dict(o) for o in outs]
    monkeypatch.setattr(idm, 'stack_datas', lambda datas, **kw: None)
    ref = dec.inference_batch([_to_data(s, dev) for s in second])      # the same engine through the host-side reload
    assert eng._hosts_light is False
    assert len(outs) == len(ref) == 8
    for o, r in zip(outs, ref):
        assert set(o) == set(r)
        for k in r:
            if isinstance(r[k], torch.Tensor):
                assert o[k].dtype == r[k].dtype and o[k].shape == r[k].shape and torch.equal(o[k], r[k]), k
            else:
                assert o[k] == r[k], k
    # ... and the host-side outputs() of an engine that was reloaded on the device rebuilds its per-scene host dicts
    monkeypatch.undo()
    if insertion:
        monkeypatch.setenv('DEBUG', '1')
    dec.inference_batch([_to_data(s, dev) for s in first])
    assert eng._hosts_light is True
    host = eng.outputs()
    assert eng._hosts_light is False and len(host) == 8 and host[0]['pos_a'].shape[0] >= A


def test_operator_modules_match_oracle():
    """AttentionLayer / FourierEmbedding / MLPEmbedding / MLPLayer forward(...) with the reference's
    argument conventions (edge_index = [src; dst] COO in arbitrary order)."""
    from infgen_amd.modules import AttentionLayer, FourierEmbedding, MLPEmbedding, MLPLayer
    from oracle import rollout_oracle as ro
    dev = torch.device('cuda:0')
    sd = make_weights(seed=4)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    rng = np.random.default_rng(0)

    def load(m, prefix):
        st = {k[len(prefix) + 1:]: torch.from_numpy(v) for k, v in sd.items() if k.startswith(prefix + '.')}
        m.load_state_dict(st, strict=True)
        return m.to(dev)

    p = 'agent_encoder.a2a_attn_layers.1'
    layer = load(AttentionLayer(128, 8, 16, 0.1, bipartite=False, has_pos_emb=True), p)
    n, e = 30, 200
    x = torch.from_numpy(rng.standard_normal((n, 128)).astype(np.float32))
    r = torch.from_numpy(rng.standard_normal((e, 128)).astype(np.float32))
    ei = torch.from_numpy(np.stack([rng.integers(0, n, e), rng.integers(0, n - 3, e)]))   # rows n-3.. get no edges
    with torch.no_grad():
        ref = ro.attention_layer(tsd, p, x, r, ei[0], ei[1])
    out = layer(x.to(dev), r.to(dev), ei.to(dev))
    assert (out.cpu() - ref).abs().max() <= 1e-4

    p = 'agent_encoder.pt2a_attn_layers.4'
    layer = load(AttentionLayer(128, 8, 16, 0.1, bipartite=True, has_pos_emb=True), p)
    xs = torch.from_numpy(rng.standard_normal((50, 128)).astype(np.float32))
    ei = torch.from_numpy(np.stack([rng.integers(0, 50, e), rng.integers(0, n, e)]))
    with torch.no_grad():
        ref = ro.attention_layer(tsd, p, x, r, ei[0], ei[1], x_src_raw=xs)
    out = layer((xs.to(dev), x.to(dev)), r.to(dev), ei.to(dev))
    assert (out.cpu() - ref).abs().max() <= 1e-4

    p = 'agent_encoder.r_t_emb'
    fe = load(FourierEmbedding(4, 128, 64), p)
    ci = torch.from_numpy(rng.uniform(-3, 3, (41, 4)).astype(np.float32))
    with torch.no_grad():
        ref = ro.fourier_embedding(tsd, p, ci)
    assert (fe(continuous_inputs=ci.to(dev), categorical_embs=None).cpu() - ref).abs().max() <= 5e-5

    p = 'agent_encoder.fusion_emb'
    me = load(MLPEmbedding(512, 128), p)
    xi = torch.from_numpy(rng.standard_normal((19, 512)).astype(np.float32))
    with torch.no_grad():
        ref = ro.mlp_embedding(tsd, p, xi)
    assert (me(xi.to(dev)).cpu() - ref).abs().max() <= 5e-5

    p = 'agent_encoder.seed_heading_rel_token_predict_head'
    ml = load(MLPLayer(128, 128, 120), p)
    xi = torch.from_numpy(rng.standard_normal((19, 128)).astype(np.float32))
    with torch.no_grad():
        ref = ro.mlp_layer(tsd, p, xi)
    assert (ml(xi.to(dev)).cpu() - ref).abs().max() <= 5e-5


def test_rollout_precision_attribute_selects_the_arithmetic():
    """InfGenDecoder.rollout_precision (the module-level counterpart of the trainer's precision flag): '32' is the default and
    reproduces the fixture; 'bf16' repacks the weights at 8 bits and runs the rollout's engines with gemm_terms = 2 - the map
    encoding moves by bf16-level amounts, the first decode step still mostly agrees; anything else is refused"""
    c = load_case('c2_a32_m512')
    z = c['z']
    dev = torch.device('cuda:0')
    dec = _decoder(c['cfg'])
    _load(dec, c['sd'])
    dec = dec.to(dev).eval()
    assert dec.rollout_precision == '32'
    full = dec.inference(_to_data(c['scene'], dev))
    assert np.array_equal(full['next_token_idx'].cpu().numpy(), z['next_token_idx'])
    assert dec._last_w.operand_bits == 11
    dec.rollout_precision = 'bf16'
    low = dec.inference(_to_data(c['scene'], dev))
    assert dec._last_w.operand_bits == 8
    eng = next(iter(dec._engines.values()))
    assert int(eng._ctx.opts.gemm_terms) == 2 and int(eng._ctx.opts.use) == 1
    d = np.abs(low['x_pt'].cpu().numpy() - full['x_pt'].cpu().numpy())
    print(f'map encoding, bf16 operands vs fp32: max {d.max():.2e} mean {d.mean():.2e}')
    assert 1e-5 < d.max() < 0.2 and d.mean() < 2e-2
    t0, t1 = full['next_token_idx'].cpu().numpy(), low['next_token_idx'].cpu().numpy()
    assert t0.shape == t1.shape and (t0[:, 0] == t1[:, 0]).mean() >= 0.8
    assert np.isfinite(low['pos_a'].cpu().numpy()).all()
    dec.rollout_precision = '32'
    again = dec.inference(_to_data(c['scene'], dev))
    assert torch.equal(again['next_token_idx'], full['next_token_idx']) and dec._last_w.operand_bits == 11
    dec.rollout_precision = 'fp8'
    with pytest.raises(ValueError, match='rollout_precision'):
        dec.inference(_to_data(c['scene'], dev))
