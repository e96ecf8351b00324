"""CPU: the drop-in boundary — C-ABI library exports, loud failure without it, state_dict
compatibility with the reference checkpoint layout, packing layouts, host-side scene setup."""
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import REPO, load_case, load_shapes


def _declared_symbols():
    txt = open(os.path.join(REPO, 'include', 'infgen_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(infgen_[a-z_0-9]+)\s*\(', txt)))


def test_library_loads_and_exports_every_declared_symbol():
    from infgen_amd import _lib
    lib = _lib.load()                     # dlopen works without a GPU
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/infgen_hip.h but not exported'
    assert set(_lib.SYMBOLS) == set(declared), 'python binding and header disagree'
    assert lib.infgen_layout_query(_lib.Q_TILE_ROWS) == 32
    assert lib.infgen_layout_query(_lib.Q_ABI_VERSION) == 1


def test_product_path_fails_loudly_without_the_library(monkeypatch):
    from infgen_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libinfgen_hip.so')
    with pytest.raises(_lib.InfgenHipError):
        _lib.load()


def test_modules_refuse_cpu_tensors():
    from infgen_amd import _lib
    from infgen_amd.modules import MLPEmbedding
    m = MLPEmbedding(8, 128)
    with pytest.raises(_lib.InfgenHipError):
        m(torch.zeros(4, 8))


def _decoder(cfg):
    from infgen_amd import synth
    from infgen_amd.modules import Attr_Tokenizer, InfGenDecoder
    tok = Attr_Tokenizer(cfg.grid_range, cfg.grid_interval, cfg.pl2seed_radius, cfg.angle_interval)
    return InfGenDecoder(
        decoder_type='agent_decoder', dataset='waymo', input_dim=2, hidden_dim=128, num_historical_steps=11,
        pl2pl_radius=cfg.pl2pl_radius, time_span=cfg.time_span, pl2a_radius=cfg.pl2a_radius,
        pl2seed_radius=cfg.pl2seed_radius, a2a_radius=cfg.a2a_radius, a2sa_radius=cfg.a2sa_radius,
        pl2sa_radius=cfg.pl2sa_radius, num_freq_bands=64, num_map_layers=3, num_agent_layers=6, num_heads=8,
        head_dim=16, dropout=0.1, map_token={'traj_src': torch.from_numpy(synth.make_map_vocab())}, token_size=2048,
        attr_tokenizer=tok, predict_motion=True, predict_state=True, predict_map=False, predict_occ=True,
        disable_insertion=True, state_token=cfg.state_token, seed_size=1, buffer_size=128,
        num_recurrent_steps_val=cfg.num_recurrent_steps_val)


def test_state_dict_keys_match_the_reference_checkpoint_layout():
    """fixture = state_dict shapes dumped from the reference's InfGenDecoder (make_golden.py)"""
    from infgen_amd import synth
    dec = _decoder(synth.standard_config())
    mine = {k: tuple(v.shape) for k, v in dec.state_dict().items()}
    assert mine == load_shapes()
    # strict load of a reference-layout state dict
    sd = synth.fill_state_dict(load_shapes(), seed=1)
    full = {k: torch.from_numpy(sd[k]) if k in sd else v for k, v in dec.state_dict().items()}
    dec.load_state_dict(full, strict=True)
    # non-bipartite layers share ONE prenorm under two names (reference layers.py:52-53)
    l0 = dec.agent_encoder.t_attn_layers[0]
    assert l0.attn_prenorm_x_dst is l0.attn_prenorm_x_src


def test_attr_grid_replica():
    from infgen_amd import synth
    g = synth.build_grid()
    assert g.shape == (1961, 2) and (g[1961 // 2] == 0).all()
    d = np.sqrt((g ** 2).sum(-1))
    assert d.max() <= 75.0 and set(np.unique(g % 3)) == {0.0}


def test_pack_matrix_layout():
    from infgen_amd import packing
    w = np.arange(40 * 13, dtype=np.float32).reshape(40, 13)       # [N][K]
    p = packing.pack_matrix(w)
    kp, npad = 16, 64
    assert p.size == kp * npad
    for (k, n) in [(0, 0), (5, 7), (12, 39), (8, 1)]:
        assert p[((k // 8) * npad + n) * 8 + (k % 8)] == w[n, k]
    assert p[((15 // 8) * npad + 3) * 8 + 7] == 0.0               # K padding


def test_attention_pack_folds_prenorm_r():
    from infgen_amd import packing, synth, _lib
    sd = synth.fill_state_dict(load_shapes(), seed=2)
    p = 'agent_encoder.a2a_attn_layers.0'
    pack = packing.pack_attention_layer(sd, p)
    lib = _lib.load()
    assert pack.size == lib.infgen_layout_query(_lib.Q_ATTN_PACK_SIZE)
    o = lib.infgen_attn_pack_offset(b'bvr')
    ref = sd[p + '.to_v_r.weight'] @ sd[p + '.attn_prenorm_r.bias'] + sd[p + '.to_v_r.bias']
    assert np.allclose(pack[o:o + 128], ref, atol=1e-6)
    o = lib.infgen_attn_pack_offset(b'bq')
    assert np.allclose(pack[o:o + 128], sd[p + '.to_q.bias'] * 0.25)


def test_host_scene_setup_matches_oracle_masks():
    """engine._setup_scene (reference agent_decoder.py:1609-1719) against the oracle's setup"""
    from infgen_amd.engine import RolloutEngine
    from oracle import rollout_oracle as ro
    c = load_case('a24_m256_edge')
    dummy = RolloutEngine.__new__(RolloutEngine)
    dummy.cfg = c['cfg']
    h = RolloutEngine._setup_scene(dummy, c['scene'])
    sd = {k: torch.from_numpy(v) for k, v in c['sd'].items()}
    # run only the first decode step of the oracle to get its masks
    cfg1 = type(c['cfg'])(**{**c['cfg'].__dict__})
    out = ro.run_scene(sd, c['scene'], cfg1, c['vocab'], c['map_vocab'], c['grid'])
    assert h['A'] == out['pos_a'].shape[0] == 23 and h['av'] == out['ego_index']
    assert np.array_equal(h['tmask'][:, :2], out['tmask'].numpy()[:, :2])
    assert np.array_equal(h['imask'][:, :2], out['imask'].numpy()[:, :2])
    assert np.array_equal(h['grid'][:, :2], out['gridtok'].numpy()[:, :2])


def test_synth_is_deterministic():
    from infgen_amd import synth
    cfg = synth.standard_config()
    a = synth.make_scene(11, 12, 64, cfg)
    b = synth.make_scene(11, 12, 64, cfg)
    assert np.array_equal(a['agent']['token_pos'], b['agent']['token_pos'])
    assert np.array_equal(a['pt_token']['position'], b['pt_token']['position'])
    w1 = synth.fill_state_dict({'x.weight': (4, 4)}, seed=3)
    w2 = synth.fill_state_dict({'x.weight': (4, 4)}, seed=3)
    assert np.array_equal(w1['x.weight'], w2['x.weight'])


def test_infgen_caller_mirror_host_side(tmp_path):
    """infgen_amd.model.InfGen (mirror of infgen/model/infgen.py): constructor from a model_config with the reference's
    attribute names, checkpoint layout `encoder.*`, map-token sample points (:201-211), mode switches; the data path
    itself refuses to run without a GPU"""
    from types import SimpleNamespace
    from infgen_amd import synth
    from infgen_amd.model import InfGen
    cfg = synth.standard_config()
    dec = SimpleNamespace(num_future_steps=80, pl2seed_radius=cfg.pl2seed_radius, token_size=cfg.token_size, seed_size=1,
                          num_map_layers=3, num_agent_layers=6, pl2pl_radius=cfg.pl2pl_radius, pl2a_radius=cfg.pl2a_radius,
                          a2a_radius=cfg.a2a_radius, a2sa_radius=cfg.a2sa_radius, pl2sa_radius=cfg.pl2sa_radius,
                          time_span=cfg.time_span, buffer_size=128)
    mc = SimpleNamespace(dataset='waymo', input_dim=2, hidden_dim=128, num_historical_steps=11, num_freq_bands=64, num_heads=8,
                         head_dim=16, dropout=0.1, decoder_type='agent_decoder', predict_motion=True, predict_state=True,
                         predict_map=False, predict_occ=True, state_token=cfg.state_token, grid_range=cfg.grid_range,
                         grid_interval=cfg.grid_interval, angle_interval=cfg.angle_interval,
                         num_recurrent_steps_val=80, decoder=dec)
    with pytest.raises(ValueError):
        InfGen(mc)                                        # no map token table given
    mv = synth.make_map_vocab()
    m = InfGen(mc, save_path=str(tmp_path), map_token_traj=mv, agent_tokens=synth.make_agent_vocab(2048))
    assert {k[len('encoder.'):] for k in m.state_dict() if k.startswith('encoder.')} == set(load_shapes())
    # besides the encoder only the tokenizer's buffers, like the reference's checkpoints
    assert {k for k in m.state_dict() if not k.startswith('encoder.')} == {'attr_tokenizer.grid', 'attr_tokenizer.dist', 'attr_tokenizer.dir'}
    assert m.map_token['sample_pt'].shape == (mv.shape[0], 3, 2)
    assert np.array_equal(m.map_token['sample_pt'].numpy(), mv[:, [0, 5, 10]])
    assert m.noise and not m._online_metric
    m.set('validation')
    assert m._online_metric and m._save_validate_reuslts
    from infgen_amd import _lib as _l
    with pytest.raises(_l.InfgenHipError):        # the teacher-forced forward exists, but only on a GPU: no CPU fallback
        m(None)
    data = {'agent': {'av_idx': 0, 'valid_mask': torch.ones(2, 91, dtype=torch.bool), 'heading': torch.zeros(2, 91),
                      'position': torch.zeros(2, 91, 3), 'velocity': torch.zeros(2, 91, 2), 'type': torch.zeros(2),
                      'shape': torch.ones(2, 91, 3)}}
    with pytest.raises(RuntimeError, match='GPU only'):
        m.validation_step(data, 0)


def test_attr_tokenizer_matches_reference_fixture():
    """Attr_Tokenizer's host-side methods against the reference's own class on seeded inputs
    (tests/golden/make_golden_tokenizer.py): buffers and indices exact, positions to float32 rounding"""
    import torch
    from infgen_amd.modules import Attr_Tokenizer
    z = np.load(os.path.join(REPO, 'tests', 'golden', 'attr_tokenizer.npz'))
    tok = Attr_Tokenizer(grid_range=150., grid_interval=3., radius=75., angle_interval=3.)
    assert np.array_equal(tok.grid.numpy(), z['grid']) and np.array_equal(tok.square_mask, z['square_mask'])
    assert np.allclose(tok.dist.numpy(), z['dist'], atol=1e-5) and np.allclose(tok.dir.numpy(), z['dir'], atol=1e-6)
    x, y, th = (torch.from_numpy(z[k]) for k in ('x', 'y', 'theta'))
    idx_r, off_r = tok.encode_pos(x, y, th)
    idx_n, off_n = tok.encode_pos(x, y)
    # a rotated point that sits within rounding of a cell border may fall to either side: compare through the distances
    same = idx_r.numpy() == z['idx_r']
    assert same.mean() > 0.97 and np.abs(off_r.numpy()[same] - z['off_r'][same]).max() <= 1e-4
    assert np.array_equal(idx_n.numpy(), z['idx_n']) and np.abs(off_n.numpy() - z['off_n']).max() <= 1e-5
    gi = torch.from_numpy(z['idx_r'])
    assert np.abs(tok.decode_pos(gi, y, th).numpy() - z['dec_r']).max() <= 1e-4
    assert np.array_equal(tok.decode_pos(gi, y).numpy(), z['dec_n']) and np.array_equal(tok.decode_pos(gi).numpy(), z['dec_0'])
    head = torch.from_numpy(z['head'])
    assert np.array_equal(tok.encode_heading(head).numpy(), z['hbin'])
    assert np.array_equal(tok.decode_heading(torch.from_numpy(z['hbin'])).numpy(), z['hdec'])
    assert np.abs(tok.get_grid(y[:1], th).numpy() - z['grid_w']).max() <= 1e-4
    pad, pidx = tok.pad_square(z['prob'], np.array([0, 5, tok.grid_size - 1, -1]))
    assert np.array_equal(pad, z['pad']) and np.array_equal(pidx, z['pidx'])


def test_torch_library_ops_are_registered_with_shape_functions():
    """the operators are torch.library ops (namespace infgen_hip); on meta tensors only the registered shape
    functions run (no GPU needed)"""
    import torch
    from infgen_amd import torch_ops  # noqa: F401
    for name in ('fourier_embed', 'radius_firstk', 'attn_layer', 'token_state_head', 'mlp_layer', 'mlp_embedding'):
        assert hasattr(torch.ops.infgen_hip, name), name
    m = lambda *s, dt=torch.float32: torch.empty(*s, device='meta', dtype=dt)
    assert torch.ops.infgen_hip.fourier_embed(m(7, 3), m(10), True).shape == (7, 128)
    idx, cnt = torch.ops.infgen_hip.radius_firstk(m(5, 2), m(9, 2), m(2, dt=torch.int64), m(2, dt=torch.int64), 3.0, 4)
    assert idx.shape == (5, 4) and cnt.shape == (5,) and idx.dtype == torch.int32
    assert torch.ops.infgen_hip.attn_layer(m(6, 128), m(10), m(6, dt=torch.int32), m(6, dt=torch.int32), m(3, dt=torch.int32),
                                           m(3, 128), None).shape == (6, 128)
    tok, st, lg = torch.ops.infgen_hip.token_state_head(m(6, 128), m(10), m(10), 2048, True)
    assert tok.shape == (6,) and lg.shape == (6, 2048)
    assert torch.ops.infgen_hip.mlp_layer(m(6, 128), m(10), 120).shape == (6, 120)


def test_library_has_no_packed_fp32_op_sel_broadcast_of_a_vgpr():
    """packed fp32 instructions that broadcast one half of a VGPR pair through op_sel / op_sel_hi compute with
    the wrong register now and then while other waves of the CU execute MFMAs (stand-alone reproducer tools/hazard_repro2.hip).
    The library is built without a single such instruction - checked here on the disassembly of the built code objects."""
    import importlib.util
    from infgen_amd import _lib
    spec = importlib.util.spec_from_file_location('pk_opsel_scan', os.path.join(REPO, 'tools', 'pk_opsel_scan.py'))
    scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scan)
    if not os.path.exists(scan.OBJDUMP):
        pytest.skip('llvm-objdump not found')
    tot, risky, per_func, n = scan.scan_library(_lib.LIB_PATH)
    assert n >= 10 and tot > 3000, 'the scan must see the kernels (packed fp32 arithmetic is used on purpose)'
    assert risky == 0, f'op_sel broadcasts of a VGPR half in: {sorted(per_func.items(), key=lambda kv: -kv[1])[:5]}'


def test_library_writes_m0_only_for_its_own_lds_dma():
    """split.cuh: lds_dma16 issues the LDS-DMA loads as inline assembly (through the builtin hipcc waits for every piece before the
    next LDS access, which defeats the weight ring) and sets M0 itself.  M0 is not a legal clobber, so the built code is checked
    instead: every M0 write sits in front of a global_load_lds, every LDS-DMA load has one, nothing else touches M0."""
    import importlib.util
    from infgen_amd import _lib
    spec = importlib.util.spec_from_file_location('pk_opsel_scan', os.path.join(REPO, 'tools', 'pk_opsel_scan.py'))
    scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scan)
    if not os.path.exists(scan.OBJDUMP):
        pytest.skip('llvm-objdump not found')
    writes, dma, bad = scan.scan_library_m0(_lib.LIB_PATH)
    assert dma > 50 and writes == dma, (writes, dma)
    assert not bad, bad[:5]


def test_infgen_import_surface_is_the_mi355x_package():
    """`import infgen.model.infgen` with <repo>/compat and <repo> on the path (instead of the reference) resolves to infgen_amd -
    the module objects are the same (reference run.py:103-105); the real modules keep their own __spec__ / __package__ (no
    ImportWarning from their relative imports, reload works)"""
    import subprocess
    import sys
    code = ('import sys, importlib; sys.path.insert(0, %r); sys.path.insert(0, %r); '
            'import infgen.model.infgen as a, infgen_amd.model.infgen as b; '
            'from infgen.modules.infgen_decoder import InfGenDecoder as D1; from infgen_amd.modules import InfGenDecoder as D2; '
            'from infgen.utils.func import wrap_angle; from infgen.metrics.compute_metrics import LongMetric; '
            'import infgen_amd.metrics.compute_metrics as cm; importlib.reload(cm); '
            'print(a is b, D1 is D2, a.InfGen.__module__, cm.__spec__.name, cm.__package__)'
            % (REPO, os.path.join(REPO, 'compat')))
    out = subprocess.run([sys.executable, '-W', 'error::ImportWarning', '-W', 'error::DeprecationWarning', '-c', code],
                         capture_output=True, text=True, timeout=120, cwd='/tmp')
    assert out.returncode == 0, out.stderr[-1500:]
    assert out.stdout.split() == ['True', 'True', 'infgen_amd.model.infgen', 'infgen_amd.metrics.compute_metrics',
                                  'infgen_amd.metrics']


def test_repository_root_has_no_package_called_infgen():
    """the reference's `infgen/` is a namespace package: a regular package of that name in the import root would shadow it for
    tests/golden/make_golden*.py - the alias lives under compat/"""
    assert not os.path.exists(os.path.join(REPO, 'infgen'))
    assert os.path.isfile(os.path.join(REPO, 'compat', 'infgen', '__init__.py'))


def test_operator_level_options_are_per_thread():
    """the operator-level entries read a per-thread option block when one is installed
    (infgen_thread_options), so engines driven from different host threads never see each other's switches"""
    import ctypes as C
    import threading
    from infgen_amd import _lib
    lib = _lib.load()
    base = _lib.Options()
    _lib.check(lib.infgen_get_options(C.byref(base)))
    mine = _lib.Options.from_buffer_copy(bytes(base))
    mine.attn_mode, mine.gemm_terms, mine.edge_fuse = 1, 1, 2
    seen = {}

    def eff():
        e = _lib.Options()
        _lib.check(lib.infgen_get_effective_options(C.byref(e)))
        return (e.attn_mode, e.gemm_terms, e.edge_fuse)

    with _lib.thread_options(mine):
        seen['inside'] = eff()
        inner = _lib.Options.from_buffer_copy(bytes(mine))
        inner.gemm_terms = 3
        with _lib.thread_options(inner):
            seen['nested'] = eff()
        seen['restored'] = eff()
        t = threading.Thread(target=lambda: seen.__setitem__('other', eff()))
        t.start(); t.join()
    seen['after'] = eff()
    default = (base.attn_mode, base.gemm_terms, base.edge_fuse)
    assert seen['inside'] == (1, 1, 2) and seen['nested'] == (1, 3, 2) and seen['restored'] == (1, 1, 2)
    assert seen['other'] == default and seen['after'] == default


def test_stacked_host_setup_equals_the_per_scene_path():
    """RolloutEngine._setup_scenes: a one-shape batch is set up with stacked numpy statements (the drop-in entry's host time);
    every array must equal the per-scene loop's - history edge cases included - and ragged / filtered batches must fall back"""
    from infgen_amd import engine, synth
    cfg = synth.standard_config()
    vocab = synth.make_agent_vocab(cfg.token_size)
    grid = synth.build_grid(cfg.grid_range, cfg.grid_interval, cfg.pl2seed_radius)
    e = engine.RolloutEngine.__new__(engine.RolloutEngine)
    e.cfg, e.T, e.hc = cfg, cfg.num_columns, cfg.hist_columns

    def check(scenes, expect_stacked):
        e.S, e.A_cap = len(scenes), 64
        e.M_cap = max(int(np.asarray(sc['pt_token']['position']).shape[0]) for sc in scenes)
        e.M_cap = (e.M_cap + 31) // 32 * 32
        e._stacked = None
        fast = e._setup_scenes(scenes)
        assert (e._stacked is not None) == expect_stacked
        arr_fast = e._scene_arrays(fast)
        e._stacked = None
        slow = [e._setup_scene(sc) for sc in scenes]
        arr_slow = e._scene_arrays(slow)
        for f, s_ in zip(fast, slow):
            assert f.keys() == s_.keys()
            for k in f:
                assert np.array_equal(np.asarray(f[k]), np.asarray(s_[k])), k
        for k in arr_slow:
            assert arr_fast[k].dtype == arr_slow[k].dtype and np.array_equal(arr_fast[k], arr_slow[k]), k

    uniform = [synth.make_scene(900 + i, 40, 200, cfg, ego_last=(i % 2 == 0), edge_cases=(i % 3 == 0), vocab=vocab, grid=grid)
               for i in range(12)]
    uniform = [sc for sc in uniform if (np.asarray(sc['agent']['state_idx'])[:, 1] != 0).all()]
    assert len(uniform) >= 8
    check(uniform, True)
    ragged = uniform[:7] + [synth.make_scene(77, 24, 100, cfg, vocab=vocab, grid=grid)] + uniform[7:]
    check(ragged, False)


def test_device_side_setup_equals_the_host_setup():
    """RolloutEngine._setup_device (the drop-in entry's reload of a batch that arrives as device tensors: the setup statements
    and the epilogue's inputs as torch ops) writes exactly the arrays the host path uploads - run here on CPU tensors against
    _setup_scenes / _scene_arrays / _epi_from_hosts, history edge cases included; a batch with a filtered row is refused"""
    from infgen_amd import engine, synth
    from infgen_amd.modules.infgen_decoder import stack_datas, _LazyScenes
    from test_modules_gpu import _to_data
    cfg = synth.standard_config()
    vocab = synth.make_agent_vocab(cfg.token_size)
    grid = synth.build_grid(cfg.grid_range, cfg.grid_interval, cfg.pl2seed_radius)
    scenes = [synth.make_scene(900 + i, 40, 200, cfg, ego_last=(i % 2 == 0), edge_cases=(i % 3 == 0), vocab=vocab, grid=grid)
              for i in range(12)]
    scenes = [sc for sc in scenes if (np.asarray(sc['agent']['state_idx'])[:, 1] != 0).all()]
    assert len(scenes) >= 8
    dev = torch.device('cpu')
    datas = [_to_data(sc, dev) for sc in scenes]
    assert stack_datas(datas) is None                      # CPU tensors: the product path keeps its host setup
    k = stack_datas(datas, any_device=True)
    assert k is not None and k['agent']['state_idx'].shape[0] == len(scenes)
    assert stack_datas(datas[:4], any_device=True) is None
    assert stack_datas(datas + [_to_data(synth.make_scene(77, 24, 100, cfg, vocab=vocab, grid=grid), dev)], any_device=True) is None

    def blank(S):
        e = engine.RolloutEngine.__new__(engine.RolloutEngine)
        e.cfg, e.T, e.hc, e.device = cfg, cfg.num_columns, cfg.hist_columns, dev
        e.S, e.A_cap, e.M_cap, e.R = S, 64, 224, cfg.num_recurrent_steps_val
        e.insertion, e.teacher_token, e._amax0, e._stacked = False, None, 40, None
        return e
    S = len(scenes)
    host = blank(S)
    host.scenes = scenes
    host.hosts = host._setup_scenes(scenes)
    arr = host._scene_arrays(host.hosts)
    epi_h = host._epi_from_hosts()
    d = blank(S)
    T, A_cap, M_cap = d.T, d.A_cap, d.M_cap
    junk = lambda *shape, dt: torch.full(shape, 7, dtype=dt)         # stale contents of a reused engine's buffers
    d.pos, d.head = junk(S, T, A_cap, 2, dt=torch.float32), junk(S, T, A_cap, dt=torch.float32)
    d.state, d.token, d.gridtok = (junk(S, T, A_cap, dt=torch.int32) for _ in range(3))
    d.tmask, d.imask, d.catflag = (junk(S, T, A_cap, dt=torch.uint8) for _ in range(3))
    d.atype, d.bos, d._shape10 = junk(S, A_cap, dt=torch.int32), junk(S, A_cap, dt=torch.int32), junk(S, A_cap, 3, dt=torch.float32)
    d.n_agents, d.n_map, d.av = (junk(S, dt=torch.int32) for _ in range(3))
    d.map_pos, d.map_orient = junk(S, M_cap, 2, dt=torch.float32), junk(S, M_cap, dt=torch.float32)
    d._map_cat = tuple(junk(S, M_cap, dt=torch.int64) for _ in range(4))
    assert d.fits_device(k)
    assert d._setup_device(k) is True
    for name in engine.RolloutEngine._SCENE_ARRAYS:
        got = getattr(d, name).numpy()
        assert got.dtype == arr[name].dtype and np.array_equal(got, arr[name]), name
    for got, name in zip(d._map_cat, ('map_tok', 'map_type', 'map_pl', 'map_light')):
        assert np.array_equal(got.numpy(), arr[name]), name
    assert d._epi.keys() == epi_h.keys()
    for name, v in epi_h.items():
        got = d._epi[name]
        got, v = (np.asarray(got), np.asarray(v)) if not isinstance(v, torch.Tensor) else (got.numpy(), v.numpy())
        assert got.dtype == v.dtype and np.array_equal(got, v), name
    assert d._gt_len == host._gt_len
    assert [(h['A'], h['M'], h['av']) for h in d.hosts] == [(h['A'], h['M'], h['av']) for h in host.hosts]
    # the host-side outputs() of such an engine rebuilds the full per-scene dicts from the lazily made host scenes
    d.scenes = _LazyScenes(datas)
    full = d._full_hosts()
    assert set(full[0]) == set(host.hosts[0]) and np.array_equal(full[3]['tmask'], host.hosts[3]['tmask'])
    # a filtered row (invalid at the last history column): refused before anything is written
    bad = [dict(x) for x in datas]
    bad[2] = dict(bad[2]); bad[2]['agent'] = dict(bad[2]['agent'])
    st = bad[2]['agent']['state_idx'].clone(); st[5, cfg.hist_columns - 1] = 0
    bad[2]['agent']['state_idx'] = st
    before = d.pos.clone()
    assert d._setup_device(stack_datas(bad, any_device=True)) is False and torch.equal(d.pos, before)


def test_lazy_out_behaves_like_the_dict_it_replaces():
    """engine.LazyOut (the per-scene result of outputs_device / inference_batch): values are cut on first access, every dict
    operation that could observe one resolves it - same keys, same values as the plain dict of rounds 1 - 3"""
    from infgen_amd.engine import LazyOut
    calls = []

    def thunk(v):
        return lambda: (calls.append(v), v)[1]
    o = LazyOut({'a': 1}, {'b': thunk(2), 'c': thunk(3)})
    assert set(o) == {'a', 'b', 'c'} and len(o) == 3 and 'b' in o and 'z' not in o and not calls
    assert o['b'] == 2 and calls == [2] and o['b'] == 2 and calls == [2]          # resolved once
    assert o.get('c') == 3 and o.get('z', 9) == 9
    m = LazyOut({'a': 1}, {'b': thunk(20)}).merged(first={'x': 0, 'a': 5}, last={'y': 7})
    assert 20 not in calls and list(m)[0] == 'x' and m['a'] == 1 and m['y'] == 7 and m['b'] == 20
    assert dict(LazyOut({'a': 1}, {'b': thunk(30)})) == {'a': 1, 'b': 30}
    assert {**LazyOut({'a': 1}, {'b': thunk(40)})} == {'a': 1, 'b': 40}
    p = LazyOut({}, {'k': thunk(50)})
    assert p.pop('k') == 50 and 'k' not in p and p.pop('k', None) is None
    q = LazyOut({}, {'k': thunk(60)})
    q['k'] = 61                                                                    # overwriting a pending value drops its thunk
    assert q['k'] == 61 and 60 not in calls
    assert q.setdefault('n', 5) == 5 and q.setdefault('n', 6) == 5
    assert sorted(LazyOut({'a': 1}, {'b': thunk(70)}).items()) == [('a', 1), ('b', 70)]
    # the bulk operations go through the same bookkeeping (update over a pending key, clear, popitem, |=, key views)
    u = LazyOut({'a': 1}, {'b': thunk(80), 'c': thunk(81)})
    u.update(b=5)
    assert len(u) == 3 and sorted(u) == ['a', 'b', 'c'] and u['b'] == 5 and 80 not in calls
    u |= {'c': 6, 'd': 7}
    assert u['c'] == 6 and 81 not in calls and len(u) == 4
    assert (u | {'e': 1})['e'] == 1 and 'e' not in u
    assert u.keys() & {'a', 'z'} == {'a'}
    k = LazyOut({}, {'p': thunk(90)})
    assert k.popitem() == ('p', 90) and len(k) == 0
    c = LazyOut({'a': 1}, {'b': thunk(91)})
    c.clear()
    assert len(c) == 0 and 'b' not in c and list(c) == [] and 91 not in calls
