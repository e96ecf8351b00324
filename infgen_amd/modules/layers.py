"""Operator modules with the reference's names, constructor signatures, parameter names
(``state_dict`` compatible) and ``forward`` argument meaning — reference
infgen/modules/layers.py:16-215 — whose arithmetic runs in libinfgen_hip.so.

These are inference-only (no autograd).  Each module packs its own parameters into the
kernel layout lazily and re-packs when a parameter's version counter changes.
"""
from __future__ import annotations

from typing import List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from .. import _lib, packing
from ..engine import Ops
from ..utils.func import weight_init

__all__ = ['AttentionLayer', 'FourierEmbedding', 'MLPEmbedding', 'MLPLayer']


class _Packed(nn.Module):
    """mixin: lazily packed device copy of the module's own parameters"""

    def _pack_fn(self, sd):       # -> np.ndarray
        raise NotImplementedError

    def packed(self) -> torch.Tensor:
        ps = list(self.parameters())
        dev = ps[0].device
        if dev.type != 'cuda':
            raise _lib.InfgenHipError('the HIP path needs the module on a cuda device (no CPU fallback)')
        ver = (dev, tuple(p._version for p in ps), tuple(p.data_ptr() for p in ps))
        if getattr(self, '_pk_ver', None) != ver:
            sd = {'m.' + k: v.detach().cpu().numpy() for k, v in self.state_dict().items()}
            self._pk = torch.from_numpy(self._pack_fn(sd)).to(dev)
            self._pk_ver = ver
        return self._pk

    def _ops(self) -> Ops:
        dev = next(self.parameters()).device
        if getattr(self, '_ops_obj', None) is None or self._ops_obj.device != dev:
            self._ops_obj = Ops(dev)
        return self._ops_obj


class AttentionLayer(_Packed):
    """reference infgen/modules/layers.py:16-113"""

    def __init__(self, hidden_dim: int, num_heads: int, head_dim: int, dropout: float, bipartite: bool,
                 has_pos_emb: bool, **kwargs) -> None:
        super().__init__()
        assert hidden_dim == 128 and num_heads == 8 and head_dim == 16, 'kernels are specialised for 128 = 8 x 16'
        self.num_heads, self.head_dim, self.has_pos_emb = num_heads, head_dim, has_pos_emb
        self.bipartite = bipartite
        self.scale = head_dim ** -0.5
        self.to_q = nn.Linear(hidden_dim, head_dim * num_heads)
        self.to_k = nn.Linear(hidden_dim, head_dim * num_heads, bias=False)
        self.to_v = nn.Linear(hidden_dim, head_dim * num_heads)
        if has_pos_emb:
            self.to_k_r = nn.Linear(hidden_dim, head_dim * num_heads, bias=False)
            self.to_v_r = nn.Linear(hidden_dim, head_dim * num_heads)
        self.to_s = nn.Linear(hidden_dim, head_dim * num_heads)
        self.to_g = nn.Linear(head_dim * num_heads + hidden_dim, head_dim * num_heads)
        self.to_out = nn.Linear(head_dim * num_heads, hidden_dim)
        self.attn_drop = nn.Dropout(dropout)
        self.ff_mlp = nn.Sequential(nn.Linear(hidden_dim, hidden_dim * 4), nn.ReLU(inplace=True), nn.Dropout(dropout),
                                    nn.Linear(hidden_dim * 4, hidden_dim))
        if bipartite:
            self.attn_prenorm_x_src = nn.LayerNorm(hidden_dim)
            self.attn_prenorm_x_dst = nn.LayerNorm(hidden_dim)
        else:
            self.attn_prenorm_x_src = nn.LayerNorm(hidden_dim)
            self.attn_prenorm_x_dst = self.attn_prenorm_x_src
        if has_pos_emb:
            self.attn_prenorm_r = nn.LayerNorm(hidden_dim)
        self.attn_postnorm = nn.LayerNorm(hidden_dim)
        self.ff_prenorm = nn.LayerNorm(hidden_dim)
        self.ff_postnorm = nn.LayerNorm(hidden_dim)
        self.apply(weight_init)

    def _pack_fn(self, sd):
        return packing.pack_attention_layer(sd, 'm', has_pos_emb=self.has_pos_emb)

    @torch.no_grad()
    def forward(self, x: Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]], r: Optional[torch.Tensor],
                edge_index: torch.Tensor) -> torch.Tensor:
        """x: (N,128) or (x_src, x_dst); r: (E,128) relative-position embedding or None;
        edge_index: (2,E) long, [source; destination] (PyG flow source_to_target)."""
        ops = self._ops()
        if isinstance(x, torch.Tensor):
            x_src, x_dst = None, x
        else:
            x_src, x_dst = x
        out = x_dst.detach().to(torch.float32).contiguous().clone()
        n = out.shape[0]
        dev = out.device
        src, dst = edge_index[0].long(), edge_index[1].long()
        order = torch.argsort(dst, stable=True)
        cnt = torch.bincount(dst, minlength=n).to(torch.int32)
        off = (torch.cumsum(cnt, 0) - cnt).to(torch.int32)
        src_s = src[order].to(torch.int32).contiguous()
        rhat = None
        if self.has_pos_emb and r is not None and r.shape[0] > 0:
            rs = r.detach().to(torch.float32)[order].contiguous()
            rhat = torch.empty_like(rs)
            _lib.check(ops.lib.infgen_layernorm(rs.data_ptr(), rs.shape[0], None, None, rhat.data_ptr(), ops.stream),
                       'infgen_layernorm')
        if src_s.numel() == 0:
            src_s = torch.zeros(1, dtype=torch.int32, device=dev)
        ops.attention_layer(out, self.packed(), off, cnt, src_s, rhat,
                            x_src=x_src.detach().to(torch.float32).contiguous() if x_src is not None else None)
        return out


class FourierEmbedding(_Packed):
    """reference infgen/modules/layers.py:116-160"""

    def __init__(self, input_dim: int, hidden_dim: int, num_freq_bands: int) -> None:
        super().__init__()
        assert hidden_dim == 128 and num_freq_bands == 64 and 1 <= input_dim <= 4
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        self.freqs = nn.Embedding(input_dim, num_freq_bands)
        self.mlps = nn.ModuleList([nn.Sequential(nn.Linear(num_freq_bands * 2 + 1, hidden_dim), nn.LayerNorm(hidden_dim),
                                                 nn.ReLU(inplace=True), nn.Linear(hidden_dim, hidden_dim))
                                   for _ in range(input_dim)])
        self.to_out = nn.Sequential(nn.LayerNorm(hidden_dim), nn.ReLU(inplace=True), nn.Linear(hidden_dim, hidden_dim))
        self.apply(weight_init)

    def _pack_fn(self, sd):
        return packing.pack_fourier(sd, 'm', self.input_dim)

    @torch.no_grad()
    def forward(self, continuous_inputs: Optional[torch.Tensor] = None,
                categorical_embs: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        if continuous_inputs is None:
            raise ValueError('the HIP FourierEmbedding needs continuous_inputs (the reference path always passes them)')
        ops = self._ops()
        x = continuous_inputs.detach().to(torch.float32)
        e = x.shape[0]
        raw = torch.zeros(e, 4, device=x.device)
        raw[:, :self.input_dim] = x
        cat = None
        if categorical_embs is not None:
            cat = torch.stack(categorical_embs).sum(dim=0).to(torch.float32).contiguous()
        out = torch.empty(e, self.hidden_dim, device=x.device)
        if e:
            ops.fourier(raw, self.input_dim, self.packed(), out, cat=cat)
        return out


class MLPEmbedding(_Packed):
    """reference infgen/modules/layers.py:163-192"""

    def __init__(self, input_dim: int, hidden_dim: int) -> None:
        super().__init__()
        assert hidden_dim == 128
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        self.mlp = nn.Sequential(nn.Linear(input_dim, 128), nn.LayerNorm(128), nn.ReLU(inplace=True),
                                 nn.Linear(128, hidden_dim), nn.LayerNorm(hidden_dim), nn.ReLU(inplace=True),
                                 nn.Linear(hidden_dim, hidden_dim))
        self.apply(weight_init)

    def _pack_fn(self, sd):
        return packing.pack_mlp_embedding(sd, 'm')

    @torch.no_grad()
    def forward(self, continuous_inputs: Optional[torch.Tensor] = None,
                categorical_embs: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        if continuous_inputs is None:
            if categorical_embs is None:
                raise ValueError('Both continuous_inputs and categorical_embs are None')
            return torch.stack(categorical_embs).sum(dim=0)
        x = continuous_inputs.detach().to(torch.float32).contiguous()
        y = self._ops().mlp_embedding(x, self.packed(), self.input_dim)
        if categorical_embs is not None:
            y = y + torch.stack(categorical_embs).sum(dim=0)
        return y


class MLPLayer(_Packed):
    """reference infgen/modules/layers.py:195-215"""

    def __init__(self, input_dim: int, hidden_dim: int = None, output_dim: int = None) -> None:
        super().__init__()
        if hidden_dim is None:
            hidden_dim = output_dim
        self.input_dim, self.hidden_dim, self.output_dim = input_dim, hidden_dim, output_dim
        self.mlp = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.LayerNorm(hidden_dim), nn.ReLU(inplace=True),
                                 nn.Linear(hidden_dim, output_dim))
        self.apply(weight_init)

    def _pack_fn(self, sd):
        g = lambda k: packing._get(sd, 'm.' + k)
        w3 = g('mlp.3.weight')
        npad = (w3.shape[0] + 31) // 32 * 32
        b3 = np.concatenate([g('mlp.3.bias'), np.zeros(npad - w3.shape[0], np.float32)])
        return np.concatenate([packing.pack_matrix(g('mlp.0.weight')), g('mlp.0.bias'), g('mlp.1.weight'),
                               g('mlp.1.bias'), packing.pack_matrix(w3), b3]).astype(np.float32)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        assert self.hidden_dim == 128, 'kernels are specialised for a 128-wide hidden layer'
        ops = self._ops()
        x = x.detach().to(torch.float32).contiguous()
        k0p = (self.input_dim + 7) // 8 * 8
        pk = self.packed()
        o_b0 = k0p * 128
        h = ops.linear(x, pk, 0, 128, self.input_dim, bias_off=o_b0, post_ln_off=o_b0 + 128, relu=True)
        o_w3 = o_b0 + 3 * 128
        npad = (self.output_dim + 31) // 32 * 32
        return ops.linear(h, pk, o_w3, self.output_dim, 128, bias_off=o_w3 + 128 * npad)
