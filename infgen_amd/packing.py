"""Host-side weight packing: reference ``state_dict`` tensors -> the kernel-ready flat fp32
layouts of infgen_amd/csrc/layout.h (offsets are queried from the shared library, so the C
header is the single source of truth).

GEMM operands are stored as P(K, N) = Wp[K/8][N][8] (see tile.cuh); the relative-position
projections ``to_k_r`` / ``to_v_r`` are stored with the layer's ``attn_prenorm_r`` affine folded in
(DESIGN.md "absorbed relative-position attention").
"""
from __future__ import annotations

import contextlib
import threading
from typing import Dict, Mapping

import numpy as np

from . import _lib

HEADS, HEAD_DIM, D = 8, 16, 128


def pack_matrix(w: np.ndarray) -> np.ndarray:
    """torch Linear.weight [N][K] -> P(Kp, Np) flat float32 (K padded to 8, N to 32)."""
    w = np.asarray(w, dtype=np.float32)
    n, k = w.shape
    kp, npad = (k + 7) // 8 * 8, (n + 31) // 32 * 32
    wt = np.zeros((kp, npad), dtype=np.float32)
    wt[:k, :n] = w.T
    return np.ascontiguousarray(wt.reshape(kp // 8, 8, npad).transpose(0, 2, 1)).reshape(-1)


def _get(sd: Mapping[str, np.ndarray], key: str) -> np.ndarray:
    v = sd[key]
    if hasattr(v, 'detach'):
        v = v.detach().cpu().numpy()
    return np.asarray(v, dtype=np.float32)


def pack_attention_layer(sd, prefix: str, has_pos_emb: bool = True) -> np.ndarray:
    """AttentionLayer (reference infgen/modules/layers.py:16-59) -> AttnLayout."""
    lib = _lib.load()
    out = np.zeros(lib.infgen_layout_query(_lib.Q_ATTN_PACK_SIZE), dtype=np.float32)

    def put(field, arr):
        o = lib.infgen_attn_pack_offset(field.encode())
        assert o >= 0, field
        arr = np.asarray(arr, dtype=np.float32).reshape(-1)
        out[o:o + arr.size] = arr

    g = lambda k: _get(sd, f'{prefix}.{k}')
    scale = np.float32(HEAD_DIM ** -0.5)
    put('ln_src_g', g('attn_prenorm_x_src.weight')); put('ln_src_b', g('attn_prenorm_x_src.bias'))
    put('ln_dst_g', g('attn_prenorm_x_dst.weight')); put('ln_dst_b', g('attn_prenorm_x_dst.bias'))
    put('wq', pack_matrix(g('to_q.weight') * scale)); put('bq', g('to_q.bias') * scale)
    put('wk', pack_matrix(g('to_k.weight')))
    put('wv', pack_matrix(g('to_v.weight'))); put('bv', g('to_v.bias'))
    if has_pos_emb:
        gam, bet = g('attn_prenorm_r.weight'), g('attn_prenorm_r.bias')
        wkr = g('to_k_r.weight') * gam[None, :]            # [c'][d]
        wvr = g('to_v_r.weight') * gam[None, :]
        kr = [pack_matrix(wkr[HEAD_DIM * h:HEAD_DIM * (h + 1), :].T) for h in range(HEADS)]   # B_h[c][d]: N=128, K=16
        put('wkr', np.concatenate(kr))
        vr = []
        for h in range(HEADS):
            wh = wvr[HEAD_DIM * h:HEAD_DIM * (h + 1), :]                    # [c][d]
            vr.append(np.ascontiguousarray(wh.reshape(HEAD_DIM, 8, 4, 4).transpose(1, 0, 2, 3)).reshape(-1))
        put('wvr', np.concatenate(vr))
        put('bvr', g('to_v_r.weight') @ bet + g('to_v_r.bias'))
    put('ws', pack_matrix(g('to_s.weight'))); put('bs', g('to_s.bias'))
    put('wg', pack_matrix(g('to_g.weight'))); put('bg', g('to_g.bias'))
    put('wo', pack_matrix(g('to_out.weight'))); put('bo', g('to_out.bias'))
    put('ln_post_g', g('attn_postnorm.weight')); put('ln_post_b', g('attn_postnorm.bias'))
    put('ln_ffpre_g', g('ff_prenorm.weight')); put('ln_ffpre_b', g('ff_prenorm.bias'))
    put('w1', pack_matrix(g('ff_mlp.0.weight'))); put('b1', g('ff_mlp.0.bias'))
    put('w2', pack_matrix(g('ff_mlp.3.weight'))); put('b2', g('ff_mlp.3.bias'))
    put('ln_ffpost_g', g('ff_postnorm.weight')); put('ln_ffpost_b', g('ff_postnorm.bias'))

    # ---- fp16-split section for k_attn_h (AH_* in csrc/layout.h): every matrix prescaled by a power of two ----
    pow2 = lambda w: float(min(2.0 ** np.floor(np.log2(H_TARGET / max(float(np.abs(w).max()), 1e-30))), 2.0 ** 14))
    hdr = np.zeros(16, np.float32)
    wq = g('to_q.weight') * scale
    wk, wv = g('to_k.weight'), g('to_v.weight')
    ws, wg, wo = g('to_s.weight'), g('to_g.weight'), g('to_out.weight')
    w1, w2 = g('ff_mlp.0.weight'), g('ff_mlp.3.weight')
    zero = np.zeros((128, 128), np.float32)
    wkr_ = wkr if has_pos_emb else zero
    wvr_ = wvr if has_pos_emb else zero
    sc = [pow2(wq), pow2(wkr_) if has_pos_emb else 1.0, pow2(wk), pow2(wv), pow2(wvr_) if has_pos_emb else 1.0,
          pow2(wg), pow2(ws), pow2(wo), pow2(w1), pow2(w2)]
    hdr[:10] = [1.0 / v for v in sc]
    # [10..13]: largest |gamma|, |beta| of ff_prenorm and of attn_prenorm_x_dst - k_layers_p bounds a row's LayerNorm output with them
    # (one power-of-two scale per row for the fp16 split of a GEMM operand, csrc/layers_p.hip: scale_bits)
    hdr[10], hdr[11] = np.abs(g('ff_prenorm.weight')).max(), np.abs(g('ff_prenorm.bias')).max()
    hdr[12], hdr[13] = np.abs(g('attn_prenorm_x_dst.weight')).max(), np.abs(g('attn_prenorm_x_dst.bias')).max()
    # [14]: version of this header (csrc/layout.h: AH_HDR_VERSION).  The library checks it before it lets k_layers_p use [10..13]: a
    # pack from an older packer has zeros there, the bound would be 0 and the operand scale 2^126
    hdr[14] = ATTN_HDR_VERSION
    pre = [pack_matrix_h(wq * sc[0], natural_k=False), pack_wkr_h(wkr_ * sc[1]),
           pack_matrix_h(wk * sc[2], natural_k=False), pack_matrix_h(wv * sc[3], natural_k=False)]
    post = [pack_wvr_h(wvr_ * sc[4]), pack_matrix_h(wg[:, :128] * sc[5], natural_k=False),
            pack_matrix_h(wg[:, 128:] * sc[5], natural_k=False), pack_matrix_h(ws * sc[6], natural_k=False),
            pack_matrix_h(wo * sc[7], natural_k=False)]
    for c in range(4):
        post.append(pack_matrix_h(w1[128 * c:128 * (c + 1), :] * sc[8], natural_k=False))
        post.append(pack_matrix_h(w2[:, 128 * c:128 * (c + 1)] * sc[9], natural_k=False))
    o = lib.infgen_attn_pack_offset(b'h_hdr')
    out[o:o + 16] = hdr
    halfs = np.concatenate(pre + post)
    assert halfs.size == 68 * 8192
    o = lib.infgen_attn_pack_offset(b'h_pre')
    out[o:o + halfs.size // 2] = halfs.view(np.float32)
    return out


def pack_fourier(sd, prefix: str, n: int) -> np.ndarray:
    """FourierEmbedding (layers.py:116-141) -> FourierLayout."""
    lib = _lib.load()
    size = lib.infgen_layout_query({2: _lib.Q_FOURIER_N2, 3: _lib.Q_FOURIER_N3, 4: _lib.Q_FOURIER_N4}[n])
    out = np.zeros(size, dtype=np.float32)

    def put(field, dim, arr):
        o = lib.infgen_fourier_pack_offset(field.encode(), n, dim)
        assert o >= 0, field
        arr = np.asarray(arr, dtype=np.float32).reshape(-1)
        out[o:o + arr.size] = arr

    g = lambda k: _get(sd, f'{prefix}.{k}')
    freqs = g('freqs.weight')
    b2sum = np.zeros(D, dtype=np.float32)
    for i in range(n):
        w1 = g(f'mlps.{i}.0.weight')                    # [128][129]
        put('freq', i, freqs[i])
        put('w1', i, pack_matrix(w1[:, :128]))
        put('w1x', i, w1[:, 128])
        put('b1', i, g(f'mlps.{i}.0.bias'))
        put('ln_g', i, g(f'mlps.{i}.1.weight')); put('ln_b', i, g(f'mlps.{i}.1.bias'))
        put('w2', i, pack_matrix(g(f'mlps.{i}.3.weight')))
        b2sum = b2sum + g(f'mlps.{i}.3.bias')
    put('b2sum', 0, b2sum)
    put('lno_g', 0, g('to_out.0.weight')); put('lno_b', 0, g('to_out.0.bias'))
    put('w3', 0, pack_matrix(g('to_out.2.weight'))); put('b3', 0, g('to_out.2.bias'))

    # ---- fp16-split section for k_fourier_h (FourierHLayout in csrc/layout.h) ----
    pow2 = lambda bound: float(2.0 ** np.floor(np.log2(H_TARGET / max(float(bound), 1e-30))))
    ln_bound = lambda gam, bet: LN_MAX * float(np.abs(gam).max()) + float(np.abs(bet).max())
    sf = 1024.0                                                     # |cos|, |sin| <= 1
    sa1 = min(min(pow2(ln_bound(g(f'mlps.{i}.1.weight'), g(f'mlps.{i}.1.bias'))) for i in range(n)), 4096.0)
    sa2 = min(pow2(ln_bound(g('to_out.0.weight'), g('to_out.0.bias'))), 4096.0)
    sw1 = [min(pow2(np.abs(g(f'mlps.{i}.0.weight')[:, :128]).max()), 2.0 ** 14) for i in range(n)]
    sw2 = min(min(pow2(np.abs(g(f'mlps.{i}.3.weight')).max()) for i in range(n)), 2.0 ** 14)
    sw3 = min(pow2(np.abs(g('to_out.2.weight')).max()), 2.0 ** 14)
    hdr = np.zeros(16, np.float32)
    for i in range(n):
        hdr[i] = 1.0 / (sw1[i] * sf)
    hdr[4], hdr[5], hdr[6] = 1.0 / (sw2 * sa1), 1.0 / (sw3 * sa2), sf
    put('h_hdr', 0, hdr)
    mats = []
    for i in range(n):
        w1 = g(f'mlps.{i}.0.weight')
        put('h_freq', i, freqs[i])
        put('h_wx', i, w1[:, 128]); put('h_b1', i, g(f'mlps.{i}.0.bias'))
        put('h_g1', i, g(f'mlps.{i}.1.weight') * sa1); put('h_be1', i, g(f'mlps.{i}.1.bias') * sa1)
        mats.append(pack_matrix_h(w1[:, :128] * sw1[i], natural_k=True))
        mats.append(pack_matrix_h(g(f'mlps.{i}.3.weight') * sw2, natural_k=False))
    put('h_b2sum', 0, b2sum)
    put('h_g2', 0, g('to_out.0.weight') * sa2); put('h_be2', 0, g('to_out.0.bias') * sa2)
    put('h_b3', 0, g('to_out.2.bias'))
    mats.append(pack_matrix_h(g('to_out.2.weight') * sw3, natural_k=False))
    halfs = np.concatenate(mats)                                   # uint16
    o = lib.infgen_fourier_pack_offset(b'h_mat', n, 0)
    out[o:o + halfs.size // 2] = halfs.view(np.float32)            # raw bits; never touched as floats
    return out


ATTN_HDR_VERSION = 2.0      # csrc/layout.h: AH_HDR_VERSION
H_TARGET = 32000.0      # |scaled operand| bound: fp16 max is 65504
LN_MAX = 11.3           # max |(x - mean) / std| over 128 values is sqrt(127)


class _PackState(threading.local):
    """significand bits of the hi term of the packs being built by THIS thread: 11 = fp16 (default), 8 = bf16 (``operand_bits``)"""
    bits = 11


_STATE = _PackState()


def current_operand_bits() -> int:
    return _STATE.bits


@contextlib.contextmanager
def operand_bits(bits: int):
    """Packs built inside (by the calling thread) carry bf16-precision weights (bits = 8) in their fp16 hi plane and an all-zero lo
    plane - the weights of the ``*_b16`` kernels (InfgenOptions.gemm_terms = 2; csrc/split.cuh: IG_BF16_OPERANDS).  The fp32 planes of
    a pack are unchanged."""
    assert bits in (8, 11)
    prev, _STATE.bits = _STATE.bits, bits
    try:
        yield
    finally:
        _STATE.bits = prev


def round_bf16(x: np.ndarray) -> np.ndarray:
    """fp32 -> the nearest bf16 value (round to nearest even on the upper 16 bits, as v_cvt_pk_bf16_f32), returned as fp32"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    u = (u + np.uint32(0x7fff) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xffff0000)
    return u.view(np.float32)


def split_f16(x: np.ndarray):
    """x -> (hi, lo) fp16 bit patterns with x ~= hi + lo (error <= 2^-23 |x|): hi = x rounded to nearest even at 11
    significand bits, lo = the remainder x - hi (exact in fp32) rounded to nearest even - the same split the kernels
    apply to activations (csrc/split.cuh: split_pair, v_cvt_pk_f16_f32).  Under ``operand_bits(8)``: hi = x rounded to bf16
    (exact in fp16 above the subnormal range: every weight is pre-scaled into it), lo = 0."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if _STATE.bits == 8:
        hi = round_bf16(x).astype(np.float16)
        assert np.all(np.isfinite(hi)), 'fp16 overflow in the weight split'
        return hi.view(np.uint16), np.zeros(hi.shape, np.uint16)
    hi = x.astype(np.float16)
    assert np.all(np.isfinite(hi)), 'fp16 overflow in the weight split'
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.view(np.uint16), lo.view(np.uint16)


def pack_wvr_h(wvr: np.ndarray) -> np.ndarray:
    """W'_vr [128 = 16 h + c][128 d] -> 4 quarters of two heads: [head 2][k-step 4][hi, lo][lane 64][slot 8];
    lane (i, kg) slot p of head h, k-step s = wvr[16 h + i][32 s + 8 kg + p]  (B fragments come from Z in natural order)"""
    h_, s_, l_, p_ = np.meshgrid(np.arange(8), np.arange(4), np.arange(64), np.arange(8), indexing='ij')
    vals = np.asarray(wvr, np.float32)[16 * h_ + (l_ & 15), 32 * s_ + 8 * (l_ >> 4) + p_]       # [8][4][64][8]
    hi, lo = split_f16(vals)
    return np.stack([hi, lo], axis=2).reshape(-1)                                                # [8][4][2][64][8]


def pack_wkr_h(wkr: np.ndarray) -> np.ndarray:
    """W'_kr [128 = 16 h + d][128 c] -> 4 quarters of two heads for the K = 16 MFMA (16x16x16):
    [head 2][column tile 8][hi, lo][lane 64][slot 4]; lane (i, kg) slot p = wkr[16 h + 4 kg + p][16 ct + i]"""
    h_, c_, l_, p_ = np.meshgrid(np.arange(8), np.arange(8), np.arange(64), np.arange(4), indexing='ij')
    vals = np.asarray(wkr, np.float32)[16 * h_ + 4 * (l_ >> 4) + p_, 16 * c_ + (l_ & 15)]       # [8][8][64][4]
    hi, lo = split_f16(vals)
    return np.stack([hi, lo], axis=2).reshape(-1)                                                # [8][8][2][64][4]


def pack_matrix_h(w: np.ndarray, natural_k: bool) -> np.ndarray:
    """nn.Linear weight [128 out][128 in] (already prescaled) -> 4 quarter-matrices (one k-step of 32 each) of
    16x16x32 MFMA A fragments: [k-step 4][feature tile 8][hi, lo][lane 64][slot 8] fp16.
    lane = (i = lane & 15, kg = lane >> 4) holds W[16 t + i][k(s, kg, p)] with
      natural_k:  k = 32 s + 8 kg + p                      (input = [cos 64 | sin 64] features)
      otherwise:  k = 32 s + 16 (p >> 2) + 4 kg + (p & 3)  (input = C registers of the previous GEMM)"""
    w = np.asarray(w, dtype=np.float32)
    assert w.shape == (128, 128)
    s_, t_, l_, p_ = np.meshgrid(np.arange(4), np.arange(8), np.arange(64), np.arange(8), indexing='ij')
    i_, kg_ = l_ & 15, l_ >> 4
    k_ = 32 * s_ + 8 * kg_ + p_ if natural_k else 32 * s_ + 16 * (p_ >> 2) + 4 * kg_ + (p_ & 3)
    vals = w[16 * t_ + i_, k_]                                    # [4][8][64][8]
    hi, lo = split_f16(vals)
    return np.stack([hi, lo], axis=2).reshape(-1)                 # [4][8][2][64][8]


def pack_mlp_embedding(sd, prefix: str) -> np.ndarray:
    """MLPEmbedding (layers.py:163-179): P(K0p,128) b ln_g ln_b | P(128,128) b ln_g ln_b | P(128,128) b"""
    g = lambda k: _get(sd, f'{prefix}.{k}')
    out = np.concatenate([
        pack_matrix(g('mlp.0.weight')), g('mlp.0.bias'), g('mlp.1.weight'), g('mlp.1.bias'),
        pack_matrix(g('mlp.3.weight')), g('mlp.3.bias'), g('mlp.4.weight'), g('mlp.4.bias'),
        pack_matrix(g('mlp.6.weight')), g('mlp.6.bias')]).astype(np.float32)
    w0 = g('mlp.0.weight')
    if w0.shape[1] % 128 == 0:
        # split section for k_mlpemb_h: [16] inverse weight scales, then quarter-matrices W0 (one unit per 128 inputs), W1, W2
        ws = [w0, g('mlp.3.weight'), g('mlp.6.weight')]
        sc = [_pow2_scale(w) for w in ws]
        hdr = np.zeros(16, np.float32)
        hdr[:3] = [1.0 / v for v in sc]
        mats = [pack_matrix_h(w0[:, 128 * j:128 * (j + 1)] * sc[0], natural_k=False) for j in range(w0.shape[1] // 128)]
        mats += [pack_matrix_h(ws[1] * sc[1], natural_k=False), pack_matrix_h(ws[2] * sc[2], natural_k=False)]
        assert out.size % 4 == 0
        out = np.concatenate([out, hdr, np.concatenate(mats).view(np.float32)])
    return out


def _pow2_scale(w) -> float:
    return float(min(2.0 ** np.floor(np.log2(H_TARGET / max(float(np.abs(w).max()), 1e-30))), 2.0 ** 14))


def mlp_embedding_offsets(k0: int):
    """float offsets of the three stages inside a pack_mlp_embedding() buffer"""
    k0p = (k0 + 7) // 8 * 8
    o1 = 0
    o2 = k0p * 128 + 3 * 128
    o3 = o2 + 16384 + 3 * 128
    return k0p, o1, o2, o3


def pack_mlp_layer(sd, prefix: str, row_major_out: bool = False) -> np.ndarray:
    """MLPLayer (layers.py:195-212): P(128,128) W0 | b0 | ln_g | ln_b | W3 | b3
    W3 is P(128, Np) unless ``row_major_out`` (tiny heads read by scalar code)."""
    g = lambda k: _get(sd, f'{prefix}.{k}')
    w3 = g('mlp.3.weight')
    w3p = w3.reshape(-1) if row_major_out else pack_matrix(w3)
    b3 = g('mlp.3.bias')
    if not row_major_out:
        npad = (w3.shape[0] + 31) // 32 * 32
        b3 = np.concatenate([b3, np.zeros(npad - b3.size, dtype=np.float32)])
    out = np.concatenate([pack_matrix(g('mlp.0.weight')), g('mlp.0.bias'), g('mlp.1.weight'), g('mlp.1.bias'),
                          w3p, b3]).astype(np.float32)
    # split section for k_heads_h (16-byte aligned): [16] inverse weight scales, W0 quarters, and for the wide head the
    # quarters of every 128-output chunk of W3
    w0 = g('mlp.0.weight')
    if w0.shape == (128, 128) and (row_major_out or w3.shape[0] % 128 == 0):
        out = np.concatenate([out, np.zeros((-out.size) % 4, np.float32)])
        s0 = _pow2_scale(w0)
        hdr = np.zeros(16, np.float32)
        hdr[0] = 1.0 / s0
        mats = [pack_matrix_h(w0 * s0, natural_k=False)]
        if not row_major_out:
            s3 = _pow2_scale(w3)
            hdr[1] = 1.0 / s3
            mats += [pack_matrix_h(w3[128 * c:128 * (c + 1), :] * s3, natural_k=False) for c in range(w3.shape[0] // 128)]
        out = np.concatenate([out, hdr, np.concatenate(mats).view(np.float32)])
    return out


MLP_LAYER_W3_OFFSET = 16384 + 3 * 128
