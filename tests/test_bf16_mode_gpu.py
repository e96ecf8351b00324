"""GPU: the optional bf16-operand mode (InfgenOptions.gemm_terms = 2; BASELINE config C5 quotes the reference at "bf16").

The reference's trainer can run its Linear layers under bf16 autocast: both operands of every matrix product are rounded to 8
significant bits, products accumulate in fp32.  The k_*_b16 kernels (csrc/*_b16.hip: the split kernels' source with
split.cuh's IG_BF16_OPERANDS) do exactly that to their operands - activations in the kernel (v_cvt_pk_bf16_f32, round to nearest
even), weights at pack time (packing.operand_bits(8)) - and carry the rounded values on the f16 matrix pipe, where they are exact.

Each operator test evaluates ONE operator on the same inputs
    full   gemm_terms = 3 (fp32-accurate split)            f16   gemm_terms = 1 (fp16 operands)
    b16    gemm_terms = 2 with bf16 packs                  emu   the fp64 oracle with both operands of every Linear rounded to bf16
against the fp64 oracle, and asserts
    rms |b16 - fp64|  within [0.8, 1.25] x rms |emu - fp64|    (measured 0.98 - 1.00: the mode is what it says)
    rms |b16 - fp64|  >=  6 x rms |f16 - fp64|                 (three bits fewer than the fp16 mode: measured 7.5 - 8.1 x)
(the kernels round folded / pre-scaled weights, the emulation the checkpoint's: the two agree statistically, not bit for bit).
"""
import numpy as np
import pytest
import torch

from conftest import load_case
from test_precision_gpu import _dev, _f64, _graph, _modes, env  # noqa: F401  (env: module fixture)

pytestmark = pytest.mark.gpu


def _rb(t: torch.Tensor) -> torch.Tensor:
    return t.float().bfloat16().to(t.dtype)


class _bf16_linears:
    """the oracle's Linear (rollout_oracle._lin) with both operands rounded to bf16"""

    def __enter__(self):
        from oracle import rollout_oracle as ro
        self.ro, self.old = ro, ro._lin

        def lin(sd, p, x, bias=True):
            w = sd[p + '.weight']
            if x.shape[-1] == 129:          # FourierEmbedding's [cos 64 | sin 64 | x]: the kernels keep the rank-1 term of the raw x in fp32
                xr, wr = torch.cat([_rb(x[..., :128]), x[..., 128:]], -1), torch.cat([_rb(w[:, :128]), w[:, 128:]], -1)
            else:
                xr, wr = _rb(x), _rb(w)
            return torch.nn.functional.linear(xr, wr, sd[p + '.bias'] if bias else None)
        ro._lin = lin

    def __exit__(self, *exc):
        self.ro._lin = self.old
        return False


def _rms(a, ref):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(ref, np.float64)) ** 2)))


def _judge(name, outs, ref, emu):
    e = {k: _rms(v, ref) for k, v in outs.items()}
    ee = _rms(emu, ref)
    print(f'{name}: rms error vs fp64 - full {e["full"]:.2e}, fp16 {e["f16"]:.2e}, bf16 kernels {e["b16"]:.2e}, bf16 emulation {ee:.2e}')
    assert 0.8 * ee <= e['b16'] <= 1.25 * ee, (name, e, ee)
    assert e['b16'] >= 6.0 * e['f16'] > 6.0 * e['full'], (name, e)


def _three(env, run, pack_fn, **modes):
    """run(pack) under gemm_terms 3 / 1 with default packs and under 2 with bf16 packs"""
    packing = env['packing']
    outs = {}
    for tag, terms in (('full', 3), ('f16', 1), ('b16', 2)):
        if terms == 2:
            with packing.operand_bits(8):
                pack = pack_fn()
        else:
            pack = pack_fn()
        with _modes(env, gemm_terms=terms, **modes):
            outs[tag] = run(pack)
    return outs


def test_weight_planes_of_a_bf16_pack(env):
    """hi plane = bf16-rounded weights (exact in fp16), lo plane = 0; the default split is unchanged"""
    packing = env['packing']
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(4096) * np.exp(rng.uniform(-6, 6, 4096))).astype(np.float32)
    hi, lo = packing.split_f16(x)
    assert np.abs(hi.view(np.float16).astype(np.float64) + lo.view(np.float16).astype(np.float64) - x).max() <= 2.0 ** -21 * np.abs(x).max()
    with packing.operand_bits(8):
        h8, l8 = packing.split_f16(x)
    assert not l8.any()
    want = torch.from_numpy(x).bfloat16().float().numpy()
    ok = np.abs(want) >= 2.0 ** -14                                    # (fp16 normal range)
    assert np.array_equal(h8.view(np.float16).astype(np.float32)[ok], want[ok])
    assert packing.current_operand_bits() == 11


@pytest.mark.parametrize('n,prefix,E', [(3, 'agent_encoder.r_a2a_emb', 70001), (4, 'agent_encoder.r_t_emb', 33000)])
def test_fourier_embedding_bf16_operands(env, n, prefix, E):
    from oracle import rollout_oracle as ro
    rng = np.random.default_rng(n)
    raw = np.zeros((E, 4), np.float32)                       # (rows of the raw edge attributes are four floats wide)
    raw[:, 0] = rng.uniform(0, 60, E)
    raw[:, 1:3] = rng.uniform(-np.pi, np.pi, (E, 2))
    if n == 4:
        raw[:, 3] = -rng.integers(1, 17, E)
    dev = env['dev']
    rawd = _dev(raw, dev)

    def run(pack):
        out = torch.empty(E, 128, device=dev)
        env['ops'].fourier(rawd, n, pack, out, normalize=False)
        return out.cpu().numpy()
    outs = _three(env, run, lambda: _dev(env['packing'].pack_fourier(env['sd'], prefix, n), dev), fourier_mode=1)
    with _f64(), torch.no_grad():
        ref = ro.fourier_embedding(env['tsd64'], prefix, torch.from_numpy(raw[:, :n]).double(), None).numpy()
        with _bf16_linears():
            emu = ro.fourier_embedding(env['tsd64'], prefix, torch.from_numpy(raw[:, :n]).double(), None).numpy()
    _judge(f'fourier[{prefix}]', outs, ref, emu)


@pytest.mark.parametrize('rows,mode', [(12000, 1), (512, 3)], ids=['k_attn_h_b16', 'k_attn_hs_b16'])
def test_attention_layer_bf16_operands(env, rows, mode):
    from oracle import rollout_oracle as ro
    prefix = 'agent_encoder.a2a_attn_layers.1'
    rng = np.random.default_rng(rows)
    x = rng.standard_normal((rows, 128)).astype(np.float32)
    off, cnt, src, dst = _graph(rng, rows, rows, 40)
    r = (rng.standard_normal((len(src), 128)) * rng.uniform(0.3, 3.0, (len(src), 1))).astype(np.float32)
    dev = env['dev']
    rhat = torch.nn.functional.layer_norm(torch.from_numpy(r).double(), (128,)).float()
    offd, cntd, srcd = (torch.from_numpy(a).to(dev) for a in (off, cnt, src))
    rhd = rhat.to(dev).contiguous()

    def run(pack):
        xd = _dev(x, dev)
        env['ops'].attention_layer(xd, pack, offd, cntd, srcd, rhd, wide='fused')
        return xd.cpu().numpy()
    outs = _three(env, run, lambda: _dev(env['packing'].pack_attention_layer(env['sd'], prefix), dev), attn_mode=mode)
    with _f64(), torch.no_grad():
        args = (env['tsd64'], prefix, torch.from_numpy(x).double(), torch.from_numpy(r).double(), torch.from_numpy(src).long(),
                torch.from_numpy(dst))
        ref = ro.attention_layer(*args).numpy()
        with _bf16_linears():
            emu = ro.attention_layer(*args).numpy()
    _judge(f'attention[{prefix},rows={rows}]', outs, ref, emu)


def test_heads_and_mlp_embedding_bf16_operands(env):
    from oracle import rollout_oracle as ro
    from infgen_amd import _lib
    rng = np.random.default_rng(21)
    rows = 12000
    dev = env['dev']
    x = rng.standard_normal((rows, 128)).astype(np.float32)

    def run_heads(packs):
        logits = torch.empty(rows, 2048, device=dev)
        nt = torch.zeros(rows, dtype=torch.int32, device=dev)
        ns = torch.zeros(rows, dtype=torch.int32, device=dev)
        _lib.check(env['lib'].infgen_heads(_dev(x, dev).data_ptr(), rows, packs[0].data_ptr(), packs[1].data_ptr(), 2048,
                                           logits.data_ptr(), nt.data_ptr(), ns.data_ptr(), env['ops'].stream))
        return logits.cpu().numpy()
    P = env['packing']
    outs = _three(env, run_heads, lambda: (_dev(P.pack_mlp_layer(env['sd'], 'agent_encoder.token_predict_head'), dev),
                                           _dev(P.pack_mlp_layer(env['sd'], 'agent_encoder.state_predict_head', row_major_out=True), dev)),
                  attn_mode=1)
    with _f64(), torch.no_grad():
        ref = ro.mlp_layer(env['tsd64'], 'agent_encoder.token_predict_head', torch.from_numpy(x).double()).numpy()
        with _bf16_linears():
            emu = ro.mlp_layer(env['tsd64'], 'agent_encoder.token_predict_head', torch.from_numpy(x).double()).numpy()
    _judge('heads[token_predict_head]', outs, ref, emu)

    x5 = rng.standard_normal((rows, 512)).astype(np.float32)

    def run_mlp(pack):
        y = torch.empty(rows, 128, device=dev)
        t1, t2 = torch.empty(rows, 128, device=dev), torch.empty(rows, 128, device=dev)
        _lib.check(env['lib'].infgen_mlp_embedding(_dev(x5, dev).data_ptr(), 512, rows, 512, pack.data_ptr(), t1.data_ptr(),
                                                   t2.data_ptr(), y.data_ptr(), 128, env['ops'].stream))
        return y.cpu().numpy()
    outs = _three(env, run_mlp, lambda: _dev(P.pack_mlp_embedding(env['sd'], 'agent_encoder.fusion_emb'), dev), attn_mode=1)
    with _f64(), torch.no_grad():
        ref = ro.mlp_embedding(env['tsd64'], 'agent_encoder.fusion_emb', torch.from_numpy(x5).double()).numpy()
        with _bf16_linears():
            emu = ro.mlp_embedding(env['tsd64'], 'agent_encoder.fusion_emb', torch.from_numpy(x5).double()).numpy()
    _judge('mlp_embedding[fusion_emb]', outs, ref, emu)


def test_bf16_rollout_bar_and_pack_mismatch_is_refused():
    """Teacher-forced rollout of fixture c2 (32 agents, 512 map tokens; unsharpened head) with every split kernel on bf16 operands:
    stated bar against the reference's fp32 logits - error <= 2e-2 at every (step, agent), mean <= 3e-3, arg-max agreement >= 95 %
    (measured 1.3e-2 / 1.7e-3 / 98.0 %; the fp16 mode: 1.4e-3 / 2.0e-4);
    the error is above the fp16 mode's (the mode is on).  An engine refuses packs of the wrong operand width, both ways."""
    from infgen_amd import engine
    c = load_case('c2_a32_m512')
    z = c['z']
    dev = torch.device('cuda:0')
    w11 = engine.PackedWeights(c['sd'], c['cfg'], dev)
    w8 = engine.PackedWeights(c['sd'], c['cfg'], dev, operand_bits=8)
    teacher = [(z['next_token_idx'], z['next_state_idx'])]
    split = dict(attn_mode=1, fourier_mode=1, layers_p=0)

    def run(w, terms):
        eng = engine.RolloutEngine(w, [c['scene']], c['vocab'], c['map_vocab'], c['grid'], store_logits=True, teacher=teacher,
                                   options=dict(split, gemm_terms=terms))
        eng.rollout()
        return eng.outputs()[0]['logits']
    ref = z['logits']
    d16 = np.abs(run(w11, 1) - ref)
    d8 = np.abs(run(w8, 2) - ref)
    lg = run(w8, 2)
    agree = float((lg.argmax(-1) == ref.argmax(-1)).mean())
    print(f'bf16 mode vs reference logits: max {d8.max():.2e} mean {d8.mean():.2e} (fp16 mode: max {d16.max():.2e} mean {d16.mean():.2e}), '
          f'arg-max agreement {agree:.4f}')
    assert d8.max() <= 2e-2 and d8.mean() <= 3e-3 and agree >= 0.95
    assert d8.mean() >= 6.0 * d16.mean()
    for w, terms in ((w11, 2), (w8, 3), (w8, 1)):
        with pytest.raises(ValueError, match='operand'):
            run(w, terms)
