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
