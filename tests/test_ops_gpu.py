"""GPU: operator-level parity of the HIP kernels (through the C ABI) against the CPU oracle's
restatement of the reference operators (oracle/rollout_oracle.py).  fp32 tolerances are
written next to each check."""
import numpy as np
import pytest
import torch

from conftest import make_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    from infgen_amd import _lib, packing, engine
    assert torch.cuda.is_available(), 'these tests need the GPU box'
    dev = torch.device('cuda:0')
    sd = make_weights(seed=3)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    return dict(lib=_lib.load(), packing=packing, ops=engine.Ops(dev), dev=dev, sd=sd, tsd=tsd)


def _dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


@pytest.mark.parametrize('rows,k,n', [(1, 8, 128), (70, 3, 128), (33, 22, 128), (64, 128, 128), (40, 128, 2048),
                                      (37, 512, 128), (5, 1961, 128), (96, 128, 3), (31, 128, 120)])
def test_linear_matches_fp32(env, rows, k, n):
    """MFMA tile GEMM incl. A/B fragment and C/D layouts: asymmetric random operands."""
    rng = np.random.default_rng(rows * 1000 + k)
    x = rng.standard_normal((rows, k)).astype(np.float32)
    w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    npad = (n + 31) // 32 * 32
    pack = np.concatenate([env['packing'].pack_matrix(w), b, np.zeros(npad - n, np.float32)])
    wp = _dev(pack, env['dev'])
    y = env['ops'].linear(_dev(x, env['dev']), wp, 0, n, k, bias_off=pack.size - npad)
    ref = x.astype(np.float64) @ w.T.astype(np.float64) + b
    err = np.abs(y.cpu().numpy() - ref).max()
    assert err <= 2e-5 * max(1.0, np.abs(ref).max()), err      # fp32 fmaf chain vs fp64


def test_linear_layernorm_relu_epilogue(env):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((50, 128)).astype(np.float32)
    w = (rng.standard_normal((128, 128)) / 11).astype(np.float32)
    b, g, be = (rng.standard_normal(128).astype(np.float32) for _ in range(3))
    pack = np.concatenate([env['packing'].pack_matrix(w), b, g, be])
    y = env['ops'].linear(_dev(x, env['dev']), _dev(pack, env['dev']), 0, 128, 128, bias_off=16384,
                          post_ln_off=16384 + 128, relu=True)
    ref = torch.relu(torch.nn.functional.layer_norm(torch.from_numpy(x) @ torch.from_numpy(w).T + torch.from_numpy(b),
                                                    (128,), torch.from_numpy(g), torch.from_numpy(be)))
    assert np.abs(y.cpu().numpy() - ref.numpy()).max() <= 2e-5


@pytest.mark.parametrize('mode', [1, 0])
@pytest.mark.parametrize('n,prefix,E', [(2, 'agent_encoder.x_a_emb', 77), (3, 'agent_encoder.r_a2a_emb', 77),
                                        (4, 'agent_encoder.r_t_emb', 77), (3, 'map_encoder.r_pt2pt_emb', 1),
                                        (3, 'agent_encoder.r_pt2a_emb', 70001)])
def test_fourier_embedding(env, n, prefix, E, mode):
    """mode 1: fp16 MFMA with the three-term hi/lo split (k_fourier_h); mode 0: fp32-input MFMA (k_fourier).
    Same tolerances for both: the split keeps 21 bits per product."""
    from oracle import rollout_oracle as ro
    from infgen_amd import _lib
    _lib.check(env['lib'].infgen_set_fourier_mode(mode))
    try:
        _fourier_case(env, n, prefix, E)
    finally:
        _lib.check(env['lib'].infgen_set_fourier_mode(1))


def _fourier_case(env, n, prefix, E):
    from oracle import rollout_oracle as ro
    rng = np.random.default_rng(n)
    raw = np.zeros((E, 4), np.float32)
    raw[:, 0] = rng.uniform(0, 60, E)
    raw[:, 1:n] = rng.uniform(-np.pi, np.pi, (E, n - 1))
    if n == 4:
        raw[:, 3] = -rng.integers(1, 13, E)
    if E > 1000:
        raw[7, 0] = 4.0e5            # |z| >= 1e5 rad: the fp64 range-reduction path
        raw[8, 0] = 0.0
    cat = rng.standard_normal((E, 128)).astype(np.float32) * 0.1 if n == 2 else None
    pack = _dev(env['packing'].pack_fourier(env['sd'], prefix, n), env['dev'])
    out = torch.empty(E, 128, device=env['dev'])
    env['ops'].fourier(_dev(raw, env['dev']), n, pack, out, cat=_dev(cat, env['dev']) if cat is not None else None)
    with torch.no_grad():
        ref = ro.fourier_embedding(env['tsd'], prefix, torch.from_numpy(raw[:, :n]),
                                   [torch.from_numpy(cat), torch.zeros(E, 128)] if cat is not None else None)
    err = np.abs(out.cpu().numpy() - ref.numpy()).max()
    assert err <= 5e-5, err       # sin/cos arguments up to ~25 rad: ocml vs sleef differ by an ulp or two
    # normalised variant == affine-free LayerNorm of the same
    out2 = torch.empty(E, 128, device=env['dev'])
    env['ops'].fourier(_dev(raw, env['dev']), n, pack, out2, cat=_dev(cat, env['dev']) if cat is not None else None,
                       normalize=True)
    ref2 = torch.nn.functional.layer_norm(ref, (128,))
    assert np.abs(out2.cpu().numpy() - ref2.numpy()).max() <= 2e-4


def test_embedding_sum4_is_torch_indexing(env):
    """infgen_embedding_sum4 (the map-token embedding + its three nn.Embedding rows, map_decoder.py:87-89) bitwise against the
    torch expression it replaces; out-of-range indices are clamped"""
    from infgen_amd import _lib
    dev = env['dev']
    g = torch.Generator().manual_seed(5)
    tabs = [torch.randn(n, 128, generator=g).to(dev) for n in (1024, 17, 4, 4)]
    rows = 70001
    idx = [torch.randint(0, t.shape[0], (rows,), generator=g).to(dev) for t in tabs]
    out = torch.empty(rows, 128, device=dev)
    args = []
    for t, i in zip(tabs, idx):
        args += [_lib.ptr(t), _lib.ptr(i), t.shape[0]]
    _lib.check(env['lib'].infgen_embedding_sum4(*args, rows, _lib.ptr(out), env['ops'].stream))
    ref = tabs[0][idx[0]] + ((tabs[1][idx[1]] + tabs[2][idx[2]]) + tabs[3][idx[3]])
    assert torch.equal(out, ref)
    bad = idx[1].clone()
    bad[:5] = torch.tensor([-3, 17, 99, 16, 0], device=dev)
    args[4] = _lib.ptr(bad)
    _lib.check(env['lib'].infgen_embedding_sum4(*args, rows, _lib.ptr(out), env['ops'].stream))
    ref = tabs[0][idx[0]] + ((tabs[1][bad.clamp(0, 16)] + tabs[2][idx[2]]) + tabs[3][idx[3]])
    assert torch.equal(out, ref)


@pytest.mark.parametrize('terms', [3, 1])
def test_fourier_time_gap_table(env, terms):
    """the temporal edges' fourth input (time gap -1 .. -16) as a lookup of its branch (infgen_fourier_last_dim_table /
    infgen_fourier_embed_tab) against the oracle (layers.py:142-160) and against the per-edge evaluation of the same kernel"""
    from oracle import rollout_oracle as ro
    from infgen_amd import _lib
    lib, dev = env['lib'], env['dev']
    E, prefix = 40003, 'agent_encoder.r_t_emb'
    rng = np.random.default_rng(44)
    raw = np.zeros((E, 4), np.float32)
    raw[:, 0] = rng.uniform(0, 60, E)
    raw[:, 1:3] = rng.uniform(-np.pi, np.pi, (E, 2))
    raw[:, 3] = -rng.integers(1, 17, E)
    pack = _dev(env['packing'].pack_fourier(env['sd'], prefix, 4), dev)
    rawd = _dev(raw, dev)
    _lib.check(lib.infgen_set_gemm_terms(terms))
    try:
        tab = torch.zeros(32, 128, device=dev)
        _lib.check(lib.infgen_fourier_last_dim_table(_lib.ptr(pack), 4, _lib.ptr(tab), env['ops'].stream))
        outs = []
        for normalize in (0, 1):
            a, b = torch.empty(E, 128, device=dev), torch.empty(E, 128, device=dev)
            _lib.check(lib.infgen_fourier_embed_tab(_lib.ptr(rawd), 4, None, E, _lib.ptr(pack), _lib.ptr(tab), a.data_ptr(), 128,
                                                    normalize, env['ops'].stream))
            env['ops'].fourier(rawd, 4, pack, b, normalize=bool(normalize))
            outs.append((a.cpu().numpy(), b.cpu().numpy()))
    finally:
        _lib.check(lib.infgen_set_gemm_terms(3))
    with torch.no_grad():
        ref = ro.fourier_embedding(env['tsd'], prefix, torch.from_numpy(raw), None)
    ref2 = torch.nn.functional.layer_norm(ref, (128,)).numpy()
    (a0, b0), (a1, b1) = outs
    # three-term split: only the fp32 summation order of the branches differs; plain fp16 operands: a last-bit change of the sum
    # can move the fp16 rounding of an activation (2^-11), the mode's own error level
    assert np.abs(a0 - b0).max() <= (2e-6 if terms == 3 else 2e-3) * max(1.0, np.abs(b0).max())
    assert np.abs(a1 - b1).max() <= (1e-5 if terms == 3 else 5e-3)
    if terms == 3:
        assert np.abs(a0 - ref.numpy()).max() <= 5e-5
        assert np.abs(a1 - ref2).max() <= 2e-4


def test_fourier_three_wave_group_variant_gives_the_same_rows(env):
    """large edge sets take k_fourier_h12 (three wave groups, 192-edge tiles; csrc/fourier_h12.hip), chosen by the set's
    CAPACITY: the same 90,001 counted rows through a small-capacity launch (k_fourier_h) and a large-capacity one must be
    bitwise equal - fp32 rows and packed 24-bit rows, n = 3 and 4"""
    from infgen_amd import _lib
    lib, dev = env['lib'], env['dev']
    E, cap_small, cap_large = 90001, 100000, 400000
    for n, prefix in ((3, 'agent_encoder.r_a2a_emb'), (4, 'agent_encoder.r_t_emb')):
        rng = np.random.default_rng(n)
        raw = np.zeros((cap_large, 4), np.float32)
        raw[:, 0] = rng.uniform(0, 60, cap_large)
        raw[:, 1:n] = rng.uniform(-np.pi, np.pi, (cap_large, n - 1))
        if n == 4:
            raw[:, 3] = -rng.integers(1, 13, cap_large)
        rawd = _dev(raw, dev)
        pack = _dev(env['packing'].pack_fourier(env['sd'], prefix, n), dev)
        count = torch.tensor([E], device=dev, dtype=torch.int32)
        outs = []
        for cap in (cap_small, cap_large):
            o32 = torch.zeros(cap_large, 128, device=dev)
            _lib.check(lib.infgen_fourier_embed(_lib.ptr(rawd), n, _lib.ptr(count), cap, _lib.ptr(pack), None, 0, _lib.ptr(o32), 128, 1,
                                                env['ops'].stream))
            o24 = torch.zeros(cap_large, 96, device=dev)            # 384 bytes per row
            _lib.check(lib.infgen_fourier_embed_r24(_lib.ptr(rawd), n, _lib.ptr(count), cap, _lib.ptr(pack), o24.data_ptr(),
                                                    env['ops'].stream))
            outs.append((o32, o24))
        torch.cuda.synchronize()
        assert float(outs[0][0][:E].abs().max()) > 0 and float(outs[0][0][E:].abs().max()) == 0
        for a_, b_ in zip(outs[0], outs[1]):
            assert torch.equal(a_.view(torch.int32), b_.view(torch.int32)), n


def test_fourier_split_is_deterministic(env):
    """350k rows (every CU busy for ~10 tiles), four launches: bitwise identical, and equal to the fp32-MFMA kernel
    within the split's accuracy.  Guards the one-workgroup-per-CU placement of k_fourier_h (csrc/fourier_h.hip)."""
    from infgen_amd import _lib
    E, n, prefix = 350000, 3, 'agent_encoder.r_a2a_emb'
    rng = np.random.default_rng(0)
    raw = np.zeros((E, 4), np.float32)
    raw[:, 0] = rng.uniform(0, 60, E)
    raw[:, 1:n] = rng.uniform(-np.pi, np.pi, (E, n - 1))
    rawd = _dev(raw, env['dev'])
    pack = _dev(env['packing'].pack_fourier(env['sd'], prefix, n), env['dev'])
    outs = []
    for _ in range(4):
        out = torch.empty(E, 128, device=env['dev'])
        env['ops'].fourier(rawd, n, pack, out, normalize=True)
        outs.append(out)
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o.view(torch.int32), outs[0].view(torch.int32))
    _lib.check(env['lib'].infgen_set_fourier_mode(0))
    try:
        ref = torch.empty(E, 128, device=env['dev'])
        env['ops'].fourier(rawd, n, pack, ref, normalize=True)
        torch.cuda.synchronize()
    finally:
        _lib.check(env['lib'].infgen_set_fourier_mode(1))
    assert float((outs[0] - ref).abs().max()) <= 5e-5


def test_attn_split_is_deterministic(env):
    """k_attn_h runs two workgroups per CU; 64 k rows, six launches, bitwise identical (the Fourier kernel of the same
    family is NOT reproducible with several workgroups per CU - csrc/fourier_h.hip - so this one is watched)"""
    from infgen_amd import _lib
    dev, lib = env['dev'], env['lib']
    rows = 65536
    p1 = _dev(env['packing'].pack_attention_layer(env['sd'], 'agent_encoder.t_attn_layers.0'), dev)
    p2 = _dev(env['packing'].pack_attention_layer(env['sd'], 'agent_encoder.pt2a_attn_layers.0'), dev)
    g = torch.Generator(device='cpu').manual_seed(0)
    X0 = torch.randn(rows, 128, generator=g).to(dev)
    AGG = (torch.randn(rows, 128, generator=g) * 0.5).to(dev)
    Z = (torch.randn(rows, 8, 128, generator=g) * 0.3).to(dev)
    SIG = torch.rand(rows, 8, generator=g).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.infgen_set_attn_mode(1))
    try:
        outs = []
        for _ in range(6):
            X = X0.clone()
            Q, K, V = (torch.empty(rows, 128, device=dev) for _ in range(3))
            U = torch.empty(rows, 8, 128, device=dev)
            _lib.check(lib.infgen_attn_post_pre(X.data_ptr(), rows, p1.data_ptr(), AGG.data_ptr(), Z.data_ptr(), SIG.data_ptr(), 1,
                                                p2.data_ptr(), Q.data_ptr(), U.data_ptr(), K.data_ptr(), V.data_ptr(), st))
            outs.append((X, Q, U, K, V))
        torch.cuda.synchronize()
    finally:
        _lib.check(lib.infgen_set_attn_mode(2))
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.parametrize('rows', [16, 50, 512, 4099])
@pytest.mark.parametrize('has_pos', [1, 0])
def test_attn_small_row_kernel_equals_tile_kernel(env, rows, has_pos):
    """k_attn_hs (one 16-row group per workgroup, feature tiles dealt to the waves) issues k_attn_h's products in k_attn_h's
    order and runs its LayerNorm / split code on the same register layout: x, q, k, v must be bit-identical; the absorbed
    query u differs by its per-head (instead of per-row) operand scale only"""
    from infgen_amd import _lib
    dev, lib = env['dev'], env['lib']
    p1 = _dev(env['packing'].pack_attention_layer(env['sd'], 'agent_encoder.a2a_attn_layers.1'), dev)
    p2 = _dev(env['packing'].pack_attention_layer(env['sd'], 'agent_encoder.t_attn_layers.2'), dev)
    g = torch.Generator(device='cpu').manual_seed(rows)
    X0 = torch.randn(rows, 128, generator=g).to(dev)
    AGG = (torch.randn(rows, 128, generator=g) * 0.5).to(dev)
    Z = (torch.randn(rows, 8, 128, generator=g) * 0.3).to(dev)
    SIG = torch.rand(rows, 8, generator=g).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    try:
        for mode in (1, 3):
            _lib.check(lib.infgen_set_attn_mode(mode))
            X = X0.clone()
            Q, K, V = (torch.zeros(rows, 128, device=dev) for _ in range(3))
            U = torch.zeros(rows, 8, 128, device=dev)
            _lib.check(lib.infgen_attn_post_pre(X.data_ptr(), rows, p1.data_ptr(), AGG.data_ptr(), Z.data_ptr(), SIG.data_ptr(),
                                                has_pos, p2.data_ptr(), Q.data_ptr(), U.data_ptr(), K.data_ptr(), V.data_ptr(), st))
            # the pre part alone, LayerNorm with the source parameters (K / V of a bipartite source)
            K2, V2 = torch.zeros(rows, 128, device=dev), torch.zeros(rows, 128, device=dev)
            _lib.check(lib.infgen_attn_pre(X0.data_ptr(), rows, p2.data_ptr(), 1, None, None, K2.data_ptr(), V2.data_ptr(), st))
            torch.cuda.synchronize()
            res[mode] = (X, Q, K, V, K2, V2, U)
    finally:
        _lib.check(lib.infgen_set_attn_mode(2))
    for a, b in zip(res[1][:6], res[3][:6]):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert float((res[1][6] - res[3][6]).abs().max()) <= 2e-6 * max(1.0, float(res[1][6].abs().max()))


def test_attn_small_row_kernel_is_deterministic(env):
    from infgen_amd import _lib
    dev, lib = env['dev'], env['lib']
    rows = 8192
    p1 = _dev(env['packing'].pack_attention_layer(env['sd'], 'agent_encoder.t_attn_layers.0'), dev)
    p2 = _dev(env['packing'].pack_attention_layer(env['sd'], 'agent_encoder.pt2a_attn_layers.0'), dev)
    g = torch.Generator(device='cpu').manual_seed(5)
    X0 = torch.randn(rows, 128, generator=g).to(dev)
    AGG = (torch.randn(rows, 128, generator=g) * 0.5).to(dev)
    Z = (torch.randn(rows, 8, 128, generator=g) * 0.3).to(dev)
    SIG = torch.rand(rows, 8, generator=g).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.infgen_set_attn_mode(3))
    try:
        outs = []
        for _ in range(8):
            X = X0.clone()
            Q, K, V = (torch.empty(rows, 128, device=dev) for _ in range(3))
            U = torch.empty(rows, 8, 128, device=dev)
            _lib.check(lib.infgen_attn_post_pre(X.data_ptr(), rows, p1.data_ptr(), AGG.data_ptr(), Z.data_ptr(), SIG.data_ptr(), 1,
                                                p2.data_ptr(), Q.data_ptr(), U.data_ptr(), K.data_ptr(), V.data_ptr(), st))
            outs.append((X, Q, U, K, V))
        torch.cuda.synchronize()
    finally:
        _lib.check(lib.infgen_set_attn_mode(2))
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def _random_graph(rng, n_dst, n_src, max_deg, empty_rows=()):
    off, cnt, src, dst = [], [], [], []
    e = 0
    for i in range(n_dst):
        d = 0 if i in empty_rows else int(rng.integers(1, max_deg + 1))
        s = rng.choice(n_src, size=min(d, n_src), replace=False)
        off.append(e); cnt.append(len(s))
        src += list(s); dst += [i] * len(s)
        e += len(s)
    return np.array(off, np.int32), np.array(cnt, np.int32), np.array(src, np.int32), np.array(dst, np.int64)


@pytest.fixture(params=[1, 3, 0], ids=['split16', 'split16-16row', 'fp32mfma'])
def attn_mode(request, env):
    """1: node-side GEMMs on the fp16 matrix pipe with the three-term split (k_attn_h); 3: the same arithmetic, one 16-row group
    per workgroup (k_attn_hs, the kernel small launches take by default); 0: fp32-input MFMA kernels"""
    from infgen_amd import _lib
    _lib.check(env['lib'].infgen_set_attn_mode(request.param))
    yield request.param
    _lib.check(env['lib'].infgen_set_attn_mode(2))


@pytest.mark.parametrize('wide', [False, True, 'fused'])
@pytest.mark.parametrize('prefix,bip', [('agent_encoder.a2a_attn_layers.2', False),
                                        ('agent_encoder.pt2a_attn_layers.1', True),
                                        ('agent_encoder.t_attn_layers.0', False)])
def test_attention_layer(env, prefix, bip, wide, attn_mode):
    """pre + edge attention + post == AttentionLayer.forward (layers.py:61-113), incl. rows without
    incoming edges (exact-zero aggregate) and ragged degrees up to 70."""
    from oracle import rollout_oracle as ro
    rng = np.random.default_rng(11)
    n_dst, n_src = 45, (60 if bip else 45)
    x = rng.standard_normal((n_dst, 128)).astype(np.float32)
    xs = rng.standard_normal((n_src, 128)).astype(np.float32) if bip else None
    off, cnt, src, dst = _random_graph(rng, n_dst, n_src, 44 if not bip else 59, empty_rows=(0, 7, 44))
    E = len(src)
    r = rng.standard_normal((E, 128)).astype(np.float32)
    with torch.no_grad():
        ref = ro.attention_layer(env['tsd'], prefix, torch.from_numpy(x), torch.from_numpy(r),
                                 torch.from_numpy(src).long(), torch.from_numpy(dst),
                                 x_src_raw=torch.from_numpy(xs) if bip else None).numpy()
    dev = env['dev']
    pack = _dev(env['packing'].pack_attention_layer(env['sd'], prefix), dev)
    rhat = torch.nn.functional.layer_norm(torch.from_numpy(r), (128,)).to(dev).contiguous()
    xd = _dev(x, dev)
    env['ops'].attention_layer(xd, pack, torch.from_numpy(off).to(dev), torch.from_numpy(cnt).to(dev),
                               torch.from_numpy(src).to(dev), rhat, x_src=_dev(xs, dev) if bip else None, wide=wide)
    err = np.abs(xd.cpu().numpy() - ref).max()
    assert err <= 1e-4, err


def test_attention_layer_edgeless(env, attn_mode):
    from oracle import rollout_oracle as ro
    rng = np.random.default_rng(12)
    x = rng.standard_normal((20, 128)).astype(np.float32)
    z = torch.zeros(0, dtype=torch.long)
    prefix = 'agent_encoder.t_attn_layers.3'
    with torch.no_grad():
        ref = ro.attention_layer(env['tsd'], prefix, torch.from_numpy(x), None, z, z).numpy()
    dev = env['dev']
    pack = _dev(env['packing'].pack_attention_layer(env['sd'], prefix), dev)
    xd = _dev(x, dev)
    zi = torch.zeros(20, dtype=torch.int32, device=dev)
    env['ops'].attention_layer(xd, pack, zi, zi, torch.zeros(1, dtype=torch.int32, device=dev),
                               torch.zeros(1, 128, device=dev))
    assert np.abs(xd.cpu().numpy() - ref).max() <= 5e-5


def test_heads_argmax_and_logits(env, attn_mode):
    from oracle import rollout_oracle as ro
    rng = np.random.default_rng(13)
    rows = 50
    x = rng.standard_normal((rows, 128)).astype(np.float32)
    dev = env['dev']
    tokp = _dev(env['packing'].pack_mlp_layer(env['sd'], 'agent_encoder.token_predict_head'), dev)
    stp = _dev(env['packing'].pack_mlp_layer(env['sd'], 'agent_encoder.state_predict_head', row_major_out=True), dev)
    logits = torch.empty(rows, 2048, device=dev)
    nt = torch.zeros(rows, dtype=torch.int32, device=dev)
    ns = torch.zeros(rows, dtype=torch.int32, device=dev)
    from infgen_amd import _lib
    _lib.check(env['lib'].infgen_heads(_dev(x, dev).data_ptr(), rows, tokp.data_ptr(), stp.data_ptr(), 2048,
                                       logits.data_ptr(), nt.data_ptr(), ns.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream))
    with torch.no_grad():
        ref = ro.mlp_layer(env['tsd'], 'agent_encoder.token_predict_head', torch.from_numpy(x))
        refs = ro.mlp_layer(env['tsd'], 'agent_encoder.state_predict_head', torch.from_numpy(x))
    lg = logits.cpu().numpy()
    assert np.abs(lg - ref.numpy()).max() <= 2e-5
    assert np.array_equal(nt.cpu().numpy(), lg.argmax(-1))
    assert np.array_equal(ns.cpu().numpy(), refs.numpy().argmax(-1))


@pytest.mark.parametrize('n_dst,n_src,max_deg', [(45, 45, 44), (200, 260, 255), (16, 1500, 1200), (1000, 64, 63), (5000, 300, 70)])
def test_edge_fused3_equals_the_vector_loop(env, n_dst, n_src, max_deg):
    """k_edge_fused3 (lane = (head, 16-column slice), rhat rows staged through LDS-DMA into the row's own U / Z slot; InfgenOptions.
    edge_kernel) against k_edge_fused on the same fp32 rows (reference layers.py:78-92,109): the same agg' up to fp32 summation
    order, over ragged degrees from 0 to beyond the 300-neighbour cap (lists of more than 64 edges: several index chunks), rows
    without edges (exactly b' * 0), odd list lengths (the last LDS-DMA pair repeats the last row), score ranges that move the
    online-softmax reference (x8 queries), and twice the same bits"""
    import ctypes as C
    from infgen_amd import _lib
    rng = np.random.default_rng(n_dst + max_deg)
    dev, lib = env['dev'], env['lib']
    prefix = 'agent_encoder.a2a_attn_layers.1'
    pack = _dev(env['packing'].pack_attention_layer(env['sd'], prefix), dev)
    off, cnt, src, dst = _random_graph(rng, n_dst, n_src, max_deg, empty_rows=(0, 7, n_dst - 1))
    E = len(src)
    r = torch.nn.functional.layer_norm(torch.from_numpy(rng.standard_normal((E, 128)).astype(np.float32) *
                                                        rng.uniform(0.2, 5.0, (E, 1)).astype(np.float32)), (128,)).to(dev).contiguous()
    o = _lib.Options()
    _lib.check(lib.infgen_get_options(C.byref(o)))
    for qscale in (1.0, 8.0):
        q = _dev(rng.standard_normal((n_dst, 128)) * qscale, dev)
        k = _dev(rng.standard_normal((n_src, 128)), dev)
        v = _dev(rng.standard_normal((n_src, 128)) * 3.0, dev)
        offd, cntd, srcd = (torch.from_numpy(a).to(dev) for a in (off, cnt, src))
        outs = []
        for kern in (0, 2, 2):
            o.edge_kernel, o.use = kern, 0
            agg = torch.full((n_dst, 128), float('nan'), device=dev)
            with _lib.thread_options(o):
                env['ops'].edge_attn(n_dst, q, pack, k, v, offd, cntd, srcd, r, agg, None, None, wide='fused')
            outs.append(agg)
        torch.cuda.synchronize()
        ref, a1, a2 = outs
        assert torch.equal(a1.view(torch.int32), a2.view(torch.int32))
        assert not torch.equal(a1, ref)                                  # (another kernel ran)
        scale = float(ref.abs().max())
        err = float((a1 - ref).abs().max())
        assert err <= 2e-5 * scale, (qscale, err, scale)
        assert torch.equal(a1[0], ref[0]) and torch.equal(a1[7], ref[7])       # rows without edges
