"""The two-scene batch of the teacher-forced ``forward`` fixture (tests/golden/forward_a40.npz): inputs rebuilt from the
committed fixtures of the tokeniser and of ``_fetch_enterings`` plus seeded map tokens - the same function feeds the
reference (tests/golden/make_golden_forward.py, build container) and the tests (CPU oracle, GPU)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def build_batch(cfg, vocab, m_per_scene=(160, 140), map_seed=7711):
    """numpy batch dict (the layout infgen_amd.synth scenes use, two graphs) from the two committed fixtures"""
    tk = np.load(os.path.join(GOLDEN, 'tokenize_a40.npz'))
    en = np.load(os.path.join(GOLDEN, 'enterings_a40.npz'))
    A = tk['out_state_idx'].shape[0]
    batch = en['batch'].astype(np.int64)
    ptr = np.array([0, int((batch == 0).sum()), A], np.int64)
    av_local = en['av_index'].astype(np.int64)
    av = av_local + ptr[:-1]
    rng = np.random.default_rng(map_seed)
    M = int(sum(m_per_scene))
    pt_batch = np.repeat(np.arange(2), m_per_scene).astype(np.int64)
    ego0 = en['token_pos'][av, 2]                                     # ego position at the current step, per scene
    pos = (ego0[pt_batch] + rng.uniform(-70, 70, (M, 2))).astype(np.float32)
    npoly = M // 8
    agent = {
        'num_nodes': A,
        'av_index': av,
        'batch': batch, 'ptr': ptr,
        'type': tk['in_type'].astype(np.uint8),
        'id': np.arange(A, dtype=np.int64),
        'state_idx': en['state_idx'].astype(np.int64),
        'token_idx': tk['out_token_idx'].astype(np.int64),
        'token_pos': en['token_pos'].astype(np.float32),
        'token_heading': en['token_heading'].astype(np.float32),
        'raw_agent_valid_mask': tk['out_raw_agent_valid_mask'].astype(bool),
        'agent_valid_mask': tk['out_agent_valid_mask'].astype(bool),
        'valid_mask': tk['out_valid_mask'].astype(bool),
        'shape': tk['out_shape'].astype(np.float32),
        'category': tk['in_category'].astype(np.uint8),
        'grid_token_idx': en['out_grid_token_idx'], 'grid_offset_xy': en['out_grid_offset_xy'],
        'heading_token_idx': en['out_heading_token_idx'], 'sort_indices': en['out_sort_indices'],
        'inrange_mask': en['out_inrange_mask'], 'bos_mask': en['out_bos_mask'],
        'pos_xy': en['out_pos_xy'], 'heading_theta': en['out_heading_theta'],
        'trajectory_token_veh': vocab['veh'], 'trajectory_token_ped': vocab['ped'], 'trajectory_token_cyc': vocab['cyc'],
    }
    pt = {
        'num_nodes': M,
        'position': np.concatenate([pos, np.zeros((M, 1), np.float32)], -1),
        'orientation': rng.uniform(-np.pi, np.pi, size=(M,)).astype(np.float32),
        'type': rng.integers(0, 17, size=(M,)).astype(np.uint8),
        'pl_type': rng.integers(0, 4, size=(M,)).astype(np.uint8),
        'token_idx': rng.integers(0, 1024, size=(M,)).astype(np.int64),
        'pt_valid_mask': np.ones((M,), bool), 'pt_pred_mask': np.zeros((M,), bool), 'pt_target_mask': np.zeros((M,), bool),
        'batch': pt_batch, 'ptr': np.array([0, m_per_scene[0], M], np.int64),
    }
    poly = {'num_nodes': npoly, 'light_type': rng.integers(0, 4, size=(npoly,)).astype(np.uint8)}
    tok2pl = np.stack([np.arange(M), rng.integers(0, npoly, size=(M,))]).astype(np.int64)
    return {'agent': agent, 'pt_token': pt, 'map_polygon': poly, 'pt_token__to__map_polygon': {'edge_index': tok2pl},
            'batch_size_a': np.diff(ptr), 'batch_size_pl': np.asarray(m_per_scene, np.int64),
            'ego_pos': en['token_pos'][av].astype(np.float32),
            'scenario_id': ['fwd_0', 'fwd_1'], 'num_graphs': 2}


def load_forward_case():
    """-> dict(batch, cfg, vocab, map_vocab, grid, sd, z, meta) of tests/golden/forward_a40.npz"""
    from infgen_amd import synth
    z = np.load(os.path.join(GOLDEN, 'forward_a40.npz'))
    meta = json.loads(str(z['meta']))
    cfg = synth.standard_config()
    vocab, map_vocab = synth.make_agent_vocab(cfg.token_size), synth.make_map_vocab()
    grid = synth.build_grid(cfg.grid_range, cfg.grid_interval, cfg.pl2seed_radius)
    batch = build_batch(cfg, vocab, tuple(meta['m_per_scene']), meta['map_seed'])
    batch['agent']['pt_grid_token_idx'] = z['pt_grid_token_idx']
    with open(os.path.join(GOLDEN, 'state_dict_shapes.json')) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    sd = synth.fill_state_dict(shapes, seed=meta['weight_seed'], rich=True, head_gain=meta['head_gain'])
    return dict(batch=batch, cfg=cfg, vocab=vocab, map_vocab=map_vocab, grid=grid, sd=sd, z=z, meta=meta)
