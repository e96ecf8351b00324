import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_shapes():
    with open(os.path.join(GOLDEN, 'state_dict_shapes.json')) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


def make_weights(seed=1, head_gain=1.0, rich=True):
    from infgen_amd import synth
    return synth.fill_state_dict(load_shapes(), seed=seed, rich=rich, head_gain=head_gain)


def load_case(name):
    """golden fixture + the regenerated inputs it was produced from"""
    from infgen_amd import synth
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    meta = json.loads(str(z['meta']))
    cfg = synth.smart_config() if meta['cfg'] == 'smart' else synth.standard_config()
    if meta.get('R'):
        cfg = synth.standard_config(num_recurrent_steps_val=meta['R'])
    meta.setdefault('insertion', '')
    vocab = synth.make_agent_vocab(cfg.token_size)
    map_vocab = synth.make_map_vocab()
    grid = synth.build_grid(cfg.grid_range, cfg.grid_interval, cfg.pl2seed_radius)
    scene = synth.make_scene(meta['seed'], meta['A'], meta['M'], cfg, ego_last=meta['ego_last'],
                             edge_cases=meta['edge_cases'], vocab=vocab, grid=grid, slip=float(meta.get('slip', 0.0)))
    sd = make_weights(seed=meta['weight_seed'], head_gain=meta['head_gain'])
    return dict(z=z, meta=meta, cfg=cfg, vocab=vocab, map_vocab=map_vocab, grid=grid, scene=scene, sd=sd)


GOLDEN_CASES = ['c1_a8_m128', 'a24_m256_edge', 'a16_m128_egofirst_state', 'c2_a32_m512', 'c3_a64_m1024']


@pytest.fixture(scope='session')
def torch_sd():
    import torch

    def conv(sd):
        return {k: torch.from_numpy(v) for k, v in sd.items()}
    return conv
