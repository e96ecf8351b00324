"""Teacher-forced forward on the GPU (infgen_amd/forward_engine.py) against the reference's own output
(tests/golden/forward_a40.npz) and against the CPU oracle on the same batch."""
import numpy as np
import pytest
import torch

from forward_case import load_forward_case
from test_forward_oracle import EXACT

pytestmark = pytest.mark.gpu

# reference tolerance: logits within 1e-3 (fp32, unsharpened heads); the kernels sit two orders below
TOL = {'x_pt': 2e-4, 'x_a': 1e-3, 'next_token_prob': 1e-3, 'next_state_prob': 1e-3, 'next_state_prob_seed': 1e-3,
       'raw_next_state_prob_seed': 1e-3, 'next_type_prob_seed': 1e-3, 'next_pos_rel_prob_seed': 1e-3,
       'next_head_rel_prob_seed': 1e-3, 'next_offset_xy_seed': 1e-3, 'next_shape_seed': 1e-3}


@pytest.fixture(scope='module')
def run():
    from infgen_amd import engine, forward_engine
    c = load_forward_case()
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], c['cfg'], dev)
    eng = forward_engine.ForwardEngine(w, c['batch'], c['vocab'], c['map_vocab'], c['grid'])
    torch.manual_seed(c['meta']['rng_seed'])
    out = eng.run()
    torch.cuda.synchronize()
    c['out'] = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}
    c['eng'] = eng
    return c


def test_forward_edge_sets_match_reference_counts(run):
    ec, got = run['meta']['edge_counts'], run['eng'].edge_counts
    assert got['t'] == ec['t'][0]
    assert got['a'] + got['a2sa'] == ec['a'][0] and got['a2sa'] == ec['a2sa'][0]
    assert got['m'] + got['m2sa'] == ec['m'][0] and got['m2sa'] == ec['m2sa'][0]
    assert got['a2sa_refine'] == ec['a2sa'][1] and got['m2sa_refine'] == ec['m2sa'][1]


@pytest.mark.parametrize('key', EXACT)
def test_forward_bookkeeping_exact(run, key):
    ref, got = run['z']['out_' + key], run['out'][key]
    assert ref.shape == got.shape, (ref.shape, got.shape)
    if ref.dtype.kind == 'f':
        assert np.abs(ref - got).max() <= 1e-6
    else:
        assert np.array_equal(ref, got)


@pytest.mark.parametrize('key', sorted(TOL))
def test_forward_activations(run, key):
    ref, got = run['z']['out_' + key], run['out'][key]
    assert ref.shape == got.shape
    err = float(np.abs(ref - got).max())
    assert err <= TOL[key], err
    # the kernels' own error level (three-term fp16 split = fp32 accuracy): far below the reference tolerance
    assert err <= 1e-4 * max(1.0, float(np.abs(ref).max())), err


def test_forward_heads_and_occupancy(run):
    z, o = run['z'], run['out']
    lg = z['out_next_token_prob']
    part = np.partition(lg, -2, axis=-1)
    sure = (part[..., -1] - part[..., -2]) > 1e-3
    assert np.array_equal(z['out_next_token_idx'][..., 0][sure], o['next_token_idx'][..., 0][sure])
    for k in ('next_state_idx', 'next_state_idx_seed', 'next_type_idx_seed'):
        assert (z['out_' + k] != o[k]).mean() <= 0.01, k
    assert np.array_equal(z['out_grid_agent_occ_gt_seed'], o['grid_agent_occ_gt_seed'].astype(np.int8))
    assert np.array_equal(z['out_grid_pt_occ_gt_seed'], o['grid_pt_occ_gt_seed'].astype(np.int8))
    for k in ('grid_agent_occ_seed', 'grid_pt_occ_seed'):
        assert np.abs(z['out_' + k] - o[k][[0, 1, 10, 11]]).max() <= 1e-3
        assert np.abs(z['out_' + k + '_rowsum'] - o[k].astype(np.float64).sum(-1)).max() <= 5e-2
    for k in ('neighbor_agent_grid_idx', 'neighbor_pt_grid_idx'):
        assert np.abs(z['out_' + k] - o[k][:128]).max() <= 1e-3
        assert np.abs(z['out_' + k + '_max'] - o[k].max(-1)).max() <= 1e-3


def test_forward_matches_cpu_oracle_on_a_second_batch(run):
    """different map tokens and another generator seed (other candidate rows in the refine stage): HIP vs the CPU oracle"""
    from infgen_amd import engine, forward_engine
    from oracle import forward_oracle as fo
    from forward_case import build_batch
    c = run
    batch = build_batch(c['cfg'], c['vocab'], (96, 200), 9001)
    # cells of the new map tokens in the egos' frames (what _fetch_enterings provides), through the oracle's encoder
    from oracle import rollout_oracle as ro
    ag, pt = batch['agent'], batch['pt_token']
    T, M = ag['state_idx'].shape[1], pt['num_nodes']
    cells = np.full((T, M), -1, np.int64)
    grid_t = torch.from_numpy(c['grid']).float()
    for b in range(2):
        sel = np.nonzero(pt['batch'] == b)[0]
        pp = torch.from_numpy(pt['position'][sel, :2])
        a0 = int(ag['av_index'][b])
        for t in range(T):
            ep = torch.from_numpy(ag['token_pos'][a0, t])[None]
            eh = torch.from_numpy(ag['token_heading'][a0:a0 + 1, t])
            near = ((pp - ep) ** 2).sum(-1).sqrt() <= c['cfg'].pl2seed_radius
            if near.any():
                cells[t, sel[near.numpy()]] = ro.encode_pos(grid_t, pp[near], ep.expand(int(near.sum()), -1), eh).numpy()
    batch['agent']['pt_grid_token_idx'] = cells
    tsd = {k: torch.from_numpy(v) for k, v in c['sd'].items()}
    torch.manual_seed(77)
    ref = fo.run_forward(tsd, batch, c['cfg'], c['vocab'], c['map_vocab'], c['grid'])
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], c['cfg'], dev)
    eng = forward_engine.ForwardEngine(w, batch, c['vocab'], c['map_vocab'], c['grid'])
    torch.manual_seed(77)
    out = eng.run()
    assert eng.edge_counts['t'] == ref['_edges_t']
    for k in ('a', 'a2sa', 'm', 'm2sa', 'a2sa_refine', 'm2sa_refine'):
        assert eng.edge_counts[k] == ref['_edges'][k], k
    for k in EXACT:
        r, g = ref[k].numpy(), out[k].cpu().numpy()
        assert r.shape == g.shape and (np.abs(r.astype(np.float64) - g).max() <= 1e-6 if r.size else True), k
    for k in sorted(TOL):
        if k == 'x_pt':
            continue
        err = float((ref[k] - out[k].cpu()).abs().max())
        assert err <= 1e-4 * max(1.0, float(ref[k].abs().max())), (k, err)


def test_infgen_decoder_forward_drop_in(run):
    """the module entry: InfGenDecoder.forward(data) with the reference's dict keys (infgen_decoder.py:114-121)"""
    from test_boundary_cpu import _decoder
    from test_modules_gpu import _load
    c = run
    dev = torch.device('cuda:0')
    dec = _decoder(c['cfg'])
    _load(dec, c['sd'])
    dec = dec.to(dev).eval()
    data = {}
    for k, v in c['batch'].items():
        if isinstance(v, dict):
            data[k] = {kk: (torch.from_numpy(np.ascontiguousarray(vv)).to(dev) if isinstance(vv, np.ndarray) else vv) for kk, vv in v.items()}
        else:
            data[k] = torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v
    data[('pt_token', 'to', 'map_polygon')] = data.pop('pt_token__to__map_polygon')
    for k in ('agent_valid_mask', 'category', 'valid_mask', 'av_index', 'shape'):
        data[k] = data['agent'][k]
    torch.manual_seed(c['meta']['rng_seed'])
    out = dec(data)
    z = c['z']
    for k in ('x_a', 'next_token_prob', 'next_token_idx_gt', 'next_token_eval_mask', 'next_state_prob', 'next_state_idx_gt',
              'next_state_eval_mask', 'next_state_idx_seed', 'next_state_idx_gt_seed', 'grid_agent_occ_seed', 'grid_pt_occ_seed',
              'grid_agent_occ_gt_seed', 'grid_pt_occ_gt_seed', 'next_head_eval_mask_seed', 'target_indices', 'x_pt',
              'map_next_token_prob', 'scenario_id', 'av_index'):
        assert k in out, k
    assert np.abs(out['next_token_prob'].cpu().numpy() - z['out_next_token_prob']).max() <= 1e-4
    assert np.array_equal(out['next_token_eval_mask'].cpu().numpy(), z['out_next_token_eval_mask'])
    assert np.array_equal(out['next_head_eval_mask_seed'].cpu().numpy(), z['out_next_head_eval_mask_seed'])
    # the open-loop validation loss of the reference (infgen/model/infgen.py:627-653) from these outputs
    m = torch.from_numpy(z['out_next_token_eval_mask'])
    ref = torch.nn.functional.cross_entropy(torch.from_numpy(z['out_next_token_prob'])[m], torch.from_numpy(z['out_next_token_idx_gt'])[m],
                                            label_smoothing=0.1)
    got = torch.nn.functional.cross_entropy(out['next_token_prob'][out['next_token_eval_mask']],
                                            out['next_token_idx_gt'][out['next_token_eval_mask']], label_smoothing=0.1)
    assert abs(float(ref) - float(got)) <= 1e-5
