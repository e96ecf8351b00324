"""GPU: the torch.library ops of namespace ``infgen_hip`` (infgen_amd/torch_ops.py, SURVEY 8b last row) against the CPU oracle's
operators and against torch_cluster.radius semantics."""
import numpy as np
import pytest
import torch

from conftest import make_weights
from test_ops_gpu import _dev, _random_graph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    from infgen_amd import packing, torch_ops  # noqa: F401  (registers the ops)
    dev = torch.device('cuda:0')
    sd = make_weights(seed=3)
    return dict(packing=packing, dev=dev, sd=sd, tsd={k: torch.from_numpy(v) for k, v in sd.items()})


def test_fourier_embed_op(env):
    from oracle import rollout_oracle as ro
    rng = np.random.default_rng(1)
    x = np.stack([rng.uniform(0, 60, 300), rng.uniform(-np.pi, np.pi, 300), rng.uniform(-np.pi, np.pi, 300)], -1).astype(np.float32)
    pack = _dev(env['packing'].pack_fourier(env['sd'], 'agent_encoder.r_a2a_emb', 3), env['dev'])
    out = torch.ops.infgen_hip.fourier_embed(_dev(x, env['dev']), pack, False)
    ref = ro.fourier_embedding(env['tsd'], 'agent_encoder.r_a2a_emb', torch.from_numpy(x))
    assert float((out.cpu() - ref).abs().max()) <= 5e-5
    outn = torch.ops.infgen_hip.fourier_embed(_dev(x, env['dev']), pack, True)
    assert float((outn.cpu() - torch.nn.functional.layer_norm(ref, (128,))).abs().max()) <= 2e-4


def test_radius_firstk_op(env):
    """first K in ascending index with strict d^2 < r^2 inside the query's batch; ragged batches, an empty one"""
    from oracle import rollout_oracle as ro
    rng = np.random.default_rng(2)
    nq, nx = [5, 0, 9], [40, 7, 130]
    pq = rng.uniform(-20, 20, (sum(nq), 2)).astype(np.float32)
    px = rng.uniform(-20, 20, (sum(nx), 2)).astype(np.float32)
    ptr_q = torch.tensor(np.concatenate([[0], np.cumsum(nq)]))
    ptr_x = torch.tensor(np.concatenate([[0], np.cumsum(nx)]))
    K, r = 6, 9.0
    idx, cnt = torch.ops.infgen_hip.radius_firstk(_dev(pq, env['dev']), _dev(px, env['dev']), ptr_q, ptr_x, r, K)
    idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
    for b in range(3):
        q0, q1, x0, x1 = int(ptr_q[b]), int(ptr_q[b + 1]), int(ptr_x[b]), int(ptr_x[b + 1])
        yi, xi = ro.radius_first_k(torch.from_numpy(px[x0:x1]), torch.from_numpy(pq[q0:q1]), r, K)
        for q in range(q1 - q0):
            want = (xi[yi == q] + x0).numpy()
            assert cnt[q0 + q] == len(want)
            assert np.array_equal(idx[q0 + q, :len(want)], want) and (idx[q0 + q, len(want):] == -1).all()


@pytest.mark.parametrize('bip', [False, True])
def test_attn_layer_op(env, bip):
    from oracle import rollout_oracle as ro
    prefix = 'agent_encoder.pt2a_attn_layers.1' if bip else 'agent_encoder.a2a_attn_layers.1'
    rng = np.random.default_rng(3)
    n_dst, n_src = 70, 90 if bip else 70
    off, cnt, src, dst = _random_graph(rng, n_dst, n_src, 9, empty_rows=(3, 11))
    x = rng.standard_normal((n_dst, 128)).astype(np.float32)
    xs = rng.standard_normal((n_src, 128)).astype(np.float32) if bip else None
    r = rng.standard_normal((len(src), 128)).astype(np.float32)
    pack = _dev(env['packing'].pack_attention_layer(env['sd'], prefix), env['dev'])
    dev = env['dev']
    rhat = torch.nn.functional.layer_norm(torch.from_numpy(r), (128,))
    i32 = lambda a: torch.from_numpy(a).to(dev)
    out = torch.ops.infgen_hip.attn_layer(_dev(x, dev), pack, i32(off), i32(cnt), i32(src), rhat.to(dev),
                                          _dev(xs, dev) if bip else None)
    ref = ro.attention_layer(env['tsd'], prefix, torch.from_numpy(x), torch.from_numpy(r), torch.from_numpy(src).long(),
                             torch.from_numpy(dst), x_src_raw=torch.from_numpy(xs) if bip else None)
    assert float((out.cpu() - ref).abs().max()) <= 1e-4


def test_heads_and_mlp_ops(env):
    from oracle import rollout_oracle as ro
    rng = np.random.default_rng(4)
    x = rng.standard_normal((50, 128)).astype(np.float32)
    dev = env['dev']
    tokp = _dev(env['packing'].pack_mlp_layer(env['sd'], 'agent_encoder.token_predict_head'), dev)
    stp = _dev(env['packing'].pack_mlp_layer(env['sd'], 'agent_encoder.state_predict_head', row_major_out=True), dev)
    tok, st, lg = torch.ops.infgen_hip.token_state_head(_dev(x, dev), tokp, stp, 2048, True)
    ref = ro.mlp_layer(env['tsd'], 'agent_encoder.token_predict_head', torch.from_numpy(x))
    assert float((lg.cpu() - ref).abs().max()) <= 1e-4
    part = np.partition(ref.numpy(), -2, axis=-1)
    sure = (part[:, -1] - part[:, -2]) > 1e-3
    assert np.array_equal(tok.cpu().numpy()[sure], ref.argmax(-1).numpy()[sure])
    rs = ro.mlp_layer(env['tsd'], 'agent_encoder.state_predict_head', torch.from_numpy(x))
    assert np.array_equal(st.cpu().numpy(), rs.argmax(-1).numpy())
    tok2, _, lg2 = torch.ops.infgen_hip.token_state_head(_dev(x, dev), tokp, stp, 2048, False)
    assert lg2.numel() == 0 and torch.equal(tok2, tok)
    hp = _dev(env['packing'].pack_mlp_layer(env['sd'], 'agent_encoder.seed_heading_rel_token_predict_head'), dev)
    h = torch.ops.infgen_hip.mlp_layer(_dev(x, dev), hp, 120)
    assert float((h.cpu() - ro.mlp_layer(env['tsd'], 'agent_encoder.seed_heading_rel_token_predict_head', torch.from_numpy(x))).abs().max()) <= 1e-4
    ep = _dev(env['packing'].pack_mlp_embedding(env['sd'], 'agent_encoder.token_emb_veh'), dev)
    x8 = rng.standard_normal((33, 8)).astype(np.float32)
    e = torch.ops.infgen_hip.mlp_embedding(_dev(x8, dev), ep)
    assert float((e.cpu() - ro.mlp_embedding(env['tsd'], 'agent_encoder.token_emb_veh', torch.from_numpy(x8))).abs().max()) <= 1e-4


def test_ops_refuse_cpu_tensors(env):
    from infgen_amd import _lib
    with pytest.raises(_lib.InfgenHipError):
        torch.ops.infgen_hip.mlp_embedding(torch.zeros(4, 8), torch.zeros(10))


def test_integrate_tokenise_op(env):
    """token -> contour -> next pose -> grid cell (agent_decoder.py:2175-2239, attr_tokenizer.py:77-89) against the oracle's
    restatement of the same lines; ragged scenes, an invalid and an exit state, ego first / last"""
    from infgen_amd import synth
    from oracle import rollout_oracle as ro
    rng = np.random.default_rng(5)
    cfg = synth.standard_config()
    vocab = synth.make_agent_vocab(cfg.token_size)
    grid = synth.build_grid(cfg.grid_range, cfg.grid_interval, cfg.pl2seed_radius)
    tabs = np.stack([vocab[k] for k in ('veh', 'ped', 'cyc')]).astype(np.float32)
    S, A = 3, 40
    n = np.array([40, 7, 33], np.int32)
    ego = np.array([39, 0, 5], np.int32)
    tok = rng.integers(0, cfg.token_size, (S, A)).astype(np.int32)
    st = np.ones((S, A), np.int32)
    st[0, 3], st[2, 9] = 0, 2                    # one invalid row, one 'exit' (class 2 -> state 3)
    ty = rng.integers(0, 3, (S, A)).astype(np.int32)
    pos = rng.uniform(-60, 60, (S, A, 2)).astype(np.float32)
    head = rng.uniform(-np.pi, np.pi, (S, A)).astype(np.float32)
    dev = env['dev']
    d = lambda a: torch.from_numpy(a).to(dev)
    npos, nhead, traj, phead, cell, nst = torch.ops.infgen_hip.integrate_tokenise(
        d(tok), d(st), d(ty), d(pos), d(head), d(n), d(ego), d(tabs), d(grid.astype(np.float32)))
    for s in range(S):
        a = int(n[s])
        c = torch.from_numpy(tabs)[torch.from_numpy(ty[s, :a]).long(), torch.from_numpy(tok[s, :a]).long()]     # (a, 6, 4, 2)
        c = ro.rot_right(c.view(a, 24, 2), torch.from_numpy(head[s, :a])).view(a, 6, 4, 2) + torch.from_numpy(pos[s, :a])[:, None, None]
        want_traj = c[:, 1:].mean(dim=2)
        dxy = c[:, 1:, 0] - c[:, 1:, 3]
        want_head = torch.atan2(dxy[..., 1], dxy[..., 0])
        assert float((traj[s, :a].cpu() - want_traj).abs().max()) <= 1e-4
        assert float((phead[s, :a].cpu() - want_head).abs().max()) <= 1e-5
        want_state = st[s, :a].copy()
        want_state[want_state == 2] = 3
        want_state[ego[s]] = 1
        assert np.array_equal(nst[s, :a].cpu().numpy(), want_state)
        ok = torch.from_numpy(want_state != 0)
        p_n, h_n = want_traj[:, -1], want_head[:, -1]
        assert float((npos[s, :a].cpu() - p_n)[ok].abs().max()) <= 1e-4
        e = int(ego[s])
        want_cell = ro.encode_pos(torch.from_numpy(grid).float(), p_n, p_n[e][None].expand(a, 2), h_n[e]).numpy()
        got = cell[s, :a].cpu().numpy()
        okn = ok.numpy()
        # a cell border within float noise of the position can go either way: require the same distance there
        diff = (got != want_cell) & okn
        if diff.any():
            g = torch.from_numpy(grid).float()
            cx = ro.rot_right((p_n - p_n[e])[:, None], (-(h_n[e] - np.pi / 2)).expand(a))[:, 0]
            d_got = (cx - g[torch.from_numpy(got).long().clamp(min=0)]).norm(dim=-1)
            d_want = (cx - g[torch.from_numpy(want_cell).long()]).norm(dim=-1)
            assert float((d_got - d_want)[torch.from_numpy(diff)].abs().max()) <= 1e-4
            assert diff.sum() <= 1
        assert (got[~okn] == -1).all() and float(npos[s, :a].cpu()[~ok].abs().sum()) == 0.0


def test_decode_step_op_equals_the_engine(env):
    """torch.ops.infgen_hip.decode_step over the engine's state block reproduces RolloutEngine.step bit for bit"""
    from conftest import load_case
    from infgen_amd import engine
    c = load_case('a24_m256_edge')
    dev = env['dev']
    w = engine.PackedWeights(c['sd'], c['cfg'], dev)
    mk = lambda: engine.RolloutEngine(w, [c['scene']], c['vocab'], c['map_vocab'], c['grid'], use_graph=False)
    a, b = mk(), mk()
    a.prologue(); b.prologue()
    ctx = b.ctx_tensor()
    for t in range(3):
        a.step(t)
        tok, st = torch.ops.infgen_hip.decode_step(ctx, t, b.pos, b.head, b.state, b.token, b.gridtok, b.X, b.next_token, b.next_state)
        assert torch.equal(tok, a.next_token) and torch.equal(st, a.next_state)
    for k in ('pos', 'head', 'state', 'token', 'gridtok', 'X'):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    z = c['z']
    assert np.array_equal(b.outputs()[0]['next_token_idx'][:, :5], z['next_token_idx'][:, :5])
    with pytest.raises(Exception):
        torch.ops.infgen_hip.decode_step(ctx, 3, b.pos.clone(), b.head, b.state, b.token, b.gridtok, b.X, b.next_token, b.next_state)
