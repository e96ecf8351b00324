"""CPU: the weight side of the bf16-operand mode (packing.operand_bits(8)): hi plane = the weights rounded to bf16 (nearest even,
exact in fp16 above its subnormal range), lo plane = 0; per thread; the default split is untouched."""
import threading

import numpy as np
import torch

from infgen_amd import packing


def test_round_bf16_is_torch_bfloat16():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(20000) * np.exp(rng.uniform(-20, 20, 20000))).astype(np.float32)
    x[:4] = [0.0, -0.0, 1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -9]           # (two ties: to even)
    assert np.array_equal(packing.round_bf16(x).view(np.uint32), torch.from_numpy(x).bfloat16().float().numpy().view(np.uint32))


def test_bf16_planes_and_default_planes():
    rng = np.random.default_rng(2)
    x = (rng.standard_normal(4096) * np.exp(rng.uniform(-6, 6, 4096))).astype(np.float32)
    hi, lo = packing.split_f16(x)
    back = hi.view(np.float16).astype(np.float64) + lo.view(np.float16).astype(np.float64)
    assert np.abs(back - x).max() <= 2.0 ** -21 * np.abs(x).max() and lo.any()
    with packing.operand_bits(8):
        h8, l8 = packing.split_f16(x)
        seen = {}
        t = threading.Thread(target=lambda: seen.setdefault('bits', packing.current_operand_bits()))
        t.start()
        t.join()
        assert seen['bits'] == 11 and packing.current_operand_bits() == 8          # (another thread packs at the default width)
    assert packing.current_operand_bits() == 11 and not l8.any()
    want = torch.from_numpy(x).bfloat16().float().numpy()
    ok = np.abs(want) >= 2.0 ** -14
    assert np.array_equal(h8.view(np.float16).astype(np.float32)[ok], want[ok])
    h2, l2 = packing.split_f16(x)
    assert np.array_equal(h2, hi) and np.array_equal(l2, lo)


def test_a_bf16_attention_pack_differs_only_in_its_fp16_planes():
    from conftest import make_weights
    sd = make_weights(seed=3)
    a = packing.pack_attention_layer(sd, 'agent_encoder.a2a_attn_layers.0')
    with packing.operand_bits(8):
        b = packing.pack_attention_layer(sd, 'agent_encoder.a2a_attn_layers.0')
    assert a.shape == b.shape and not np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # the header (version, scales) and the fp32 vectors are the same: the pack validates like a default one
    n_hdr = 16
    assert np.array_equal(a[:n_hdr].view(np.uint32), b[:n_hdr].view(np.uint32))
