"""Batched closed-loop rollout engine on one MI355X.

S independent scenes are decoded in lockstep; every per-step kernel works on the
``S * A_cap`` agent rows of the current token column only (windowed K/V ring instead of the
reference's all-columns recompute).  Host code here only allocates device
memory (torch), packs weights and sequences C-ABI calls; all arithmetic of the path runs in
libinfgen_hip.so.

Reference path: ``InfGenDecoder.inference`` -> ``InfGenMapDecoder.forward`` +
``InfGenAgentDecoder.inference`` (infgen/modules/infgen_decoder.py:123-130,
map_decoder.py:70-130, agent_decoder.py:1605-2389), greedy decoding, insertion disabled.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
from typing import Dict, List, Mapping, Optional, Sequence

import numpy as np
import torch

from . import _lib, packing
from .synth import INVALID, VALID, ENTER, EXIT, AGENT_SHAPE, RolloutConfig

D = 128
SEED_TYPE = 3
INVALID_SHAPE = 0.1


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class LazyOut(dict):
    """One scene's slice of a batch result: a ``dict`` whose values are created on first access.  ``outputs_device`` of a
    512-scene batch used to spend ~30 ms cutting 17 views per scene (and ``InfGenDecoder._run`` another ~60 ms copying and
    completing the dicts) before anything was read; a consumer such as ``validation_step`` reads a handful of keys of one
    scene.  Pending values are thunks in ``_lazy``; every dict operation that could observe a value resolves it first, so
    the object behaves like the plain dict it replaces (same keys, same tensors - views of the batch arrays)."""
    __slots__ = ('_lazy',)

    def __init__(self, eager=None, lazy=None):
        super().__init__(eager or {})
        self._lazy = dict(lazy or {})

    def _force(self, k):
        f = self._lazy.pop(k, None)
        if f is not None:
            super().__setitem__(k, f())

    def _force_all(self):
        for k in list(self._lazy):
            self._force(k)

    def __missing__(self, k):
        if k in self._lazy:
            self._force(k)
            return super().__getitem__(k)
        raise KeyError(k)

    def __contains__(self, k):
        return super().__contains__(k) or k in self._lazy

    def __setitem__(self, k, v):
        self._lazy.pop(k, None)
        super().__setitem__(k, v)

    def __delitem__(self, k):
        if self._lazy.pop(k, None) is None:
            super().__delitem__(k)

    def __len__(self):
        return super().__len__() + len(self._lazy)

    def __iter__(self):
        yield from super().__iter__()
        yield from list(self._lazy)

    def keys(self):
        # a real view (set operations such as ``d.keys() & other`` work); the pending keys are resolved first - every caller that
        # only wants the names uses ``in`` / iteration, which stay lazy
        self._force_all()
        return super().keys()

    def update(self, *args, **kw):
        for k, v in dict(*args, **kw).items():
            self[k] = v

    def __ior__(self, other):
        self.update(other)
        return self

    def __or__(self, other):
        out = self.copy()
        out.update(other)
        return out

    def clear(self):
        self._lazy.clear()
        super().clear()

    def popitem(self):
        self._force_all()
        return super().popitem()

    def values(self):
        self._force_all()
        return super().values()

    def items(self):
        self._force_all()
        return super().items()

    def get(self, k, default=None):
        return self[k] if k in self else default

    def pop(self, k, *default):
        if k in self._lazy:
            self._force(k)
        return super().pop(k, *default)

    def setdefault(self, k, default=None):
        if k in self:
            return self[k]
        self[k] = default
        return default

    def set_lazy(self, k, thunk):
        super().pop(k, None)
        self._lazy[k] = thunk

    def merged(self, first=None, last=None) -> 'LazyOut':
        """{**first, **self, **last} without resolving this dict's pending values"""
        eager = dict(first or {})
        eager.update({k: super(LazyOut, self).__getitem__(k) for k in super().__iter__()})
        out = LazyOut(eager, self._lazy)
        for k, v in (last or {}).items():
            out[k] = v
        return out

    def copy(self):
        return LazyOut({k: super(LazyOut, self).__getitem__(k) for k in super().__iter__()}, self._lazy)

    def __eq__(self, other):
        self._force_all()
        return super().__eq__(other)

    def __repr__(self):
        return f'LazyOut(resolved={list(super().__iter__())}, pending={list(self._lazy)})'


class InsertionHeadroomError(RuntimeError):
    """scenario insertion ran out of agent rows in some scene: results would differ from the reference's, so the rollout
    stops instead of dropping the insertion; the caller re-runs with more ``insert_headroom``"""

    def __init__(self, msg, needed=0):
        super().__init__(msg)
        self.needed = needed


class PackedWeights:
    """Device-resident packed weights of one checkpoint (shared by every engine on the GPU)."""

    def __init__(self, sd: Mapping[str, np.ndarray], cfg: RolloutConfig, device: torch.device,
                 agent_prefix: str = 'agent_encoder', map_prefix: str = 'map_encoder', operand_bits: int = 11):
        """``operand_bits`` = 8: the fp16 planes of every pack hold bf16-precision weights (packing.operand_bits) - the packs
        of the reduced bf16 mode, which only engines running ``gemm_terms = 2`` accept (and which accept no other)"""
        self.operand_bits = int(operand_bits)
        with packing.operand_bits(self.operand_bits):
            self._pack(sd, cfg, device, agent_prefix, map_prefix)

    def _pack(self, sd, cfg, device, agent_prefix, map_prefix):
        self.cfg = cfg
        self.device = device
        ap, mp = agent_prefix, map_prefix
        self.sd = sd
        self.ap, self.mp = ap, mp
        L = cfg.num_agent_layers

        def dev(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)

        self.attn_t = [dev(packing.pack_attention_layer(sd, f'{ap}.t_attn_layers.{i}')) for i in range(L)]
        self.attn_m = [dev(packing.pack_attention_layer(sd, f'{ap}.pt2a_attn_layers.{i}')) for i in range(L)]
        self.attn_a = [dev(packing.pack_attention_layer(sd, f'{ap}.a2a_attn_layers.{i}')) for i in range(L)]
        self.attn_pt = [dev(packing.pack_attention_layer(sd, f'{mp}.pt2pt_layers.{i}'))
                        for i in range(cfg.num_map_layers)]
        self.four_t = dev(packing.pack_fourier(sd, f'{ap}.r_t_emb', 4))
        self._time_gap_tables = {}
        self.four_m = dev(packing.pack_fourier(sd, f'{ap}.r_pt2a_emb', 3))
        self.four_a = dev(packing.pack_fourier(sd, f'{ap}.r_a2a_emb', 3))
        self.four_xa = dev(packing.pack_fourier(sd, f'{ap}.x_a_emb', 2))
        self.four_pt = dev(packing.pack_fourier(sd, f'{mp}.r_pt2pt_emb', 3))
        self.fusion = dev(packing.pack_mlp_embedding(sd, f'{ap}.fusion_emb'))
        self.shape_emb = dev(packing.pack_mlp_embedding(sd, f'{ap}.shape_emb'))
        self.tok_emb = [dev(packing.pack_mlp_embedding(sd, f'{ap}.token_emb_{k}')) for k in ('veh', 'ped', 'cyc')]
        self.grid_emb = dev(packing.pack_mlp_embedding(sd, f'{ap}.token_emb_grid'))
        self.map_tok_emb = dev(packing.pack_mlp_embedding(sd, f'{mp}.token_emb'))
        self.tok_head = dev(packing.pack_mlp_layer(sd, f'{ap}.token_predict_head'))
        self.st_head = dev(packing.pack_mlp_layer(sd, f'{ap}.state_predict_head', row_major_out=True))
        g = lambda k: dev(packing._get(sd, k))
        self.type_a_emb = g(f'{ap}.type_a_emb.weight')
        self.state_a_emb = g(f'{ap}.state_a_emb.weight')
        self.no_token = g(f'{ap}.no_token_emb.weight')
        self.bos_token = g(f'{ap}.bos_token_emb.weight')
        self.invalid_offset = g(f'{ap}.invalid_offset_token_emb.weight')
        self.type_pt_emb = g(f'{mp}.type_pt_emb.weight')
        self.polygon_type_emb = g(f'{mp}.polygon_type_emb.weight')
        self.light_pl_emb = g(f'{mp}.light_pl_emb.weight')
        # scenario insertion (agent_decoder.py:236-290)
        self.attn_pt2sa = [dev(packing.pack_attention_layer(sd, f'{ap}.pt2sa_attn_layers.{i}')) for i in range(3)]
        self.attn_a2sa = [dev(packing.pack_attention_layer(sd, f'{ap}.a2sa_attn_layers.{i}')) for i in range(3)]
        self.attn_occ2sa = [dev(packing.pack_attention_layer(sd, f'{ap}.occ2sa_attn_layers.{i}', has_pos_emb=False))
                            for i in range(3)]
        self.four_pt2sa = dev(packing.pack_fourier(sd, f'{ap}.r_pt2sa_emb', 3))
        self.four_a2sa = dev(packing.pack_fourier(sd, f'{ap}.r_a2sa_emb', 3))
        self.heads = {k: dev(packing.pack_mlp_layer(sd, f'{ap}.{k}')) for k in
                      ('seed_state_predict_head', 'seed_type_predict_head', 'seed_shape_predict_head',
                       'seed_pos_rel_token_predict_head', 'seed_heading_rel_token_predict_head',
                       'seed_offset_xy_predict_head', 'seed_agent_occ_embed')}
        self._tables = None
        self._tables_key = None
        self._tables_by_key = {}

    def time_gap_table(self, lib, terms: int, stream):
        """[32][128] table of r_t_emb's fourth branch (the time gap of a temporal edge is one of -1 .. -16, agent_decoder.py:586-600)
        under the arithmetic ``terms`` (infgen_fourier_last_dim_table), cached per pack"""
        tab = self._time_gap_tables.get(terms)
        if tab is None:
            # (the arithmetic is chosen through the calling thread's option block, not by editing the process-wide default)
            o = _lib.Options()
            _lib.check(lib.infgen_get_effective_options(C.byref(o)), 'infgen_get_effective_options')
            tab = torch.zeros(32, D, device=self.four_t.device)
            o.gemm_terms, o.use = terms, 0
            with _lib.thread_options(o):
                _lib.check(lib.infgen_fourier_last_dim_table(_lib.ptr(self.four_t), 4, _lib.ptr(tab), stream),
                           'infgen_fourier_last_dim_table')
            torch.cuda.synchronize(tab.device)
            self._time_gap_tables[terms] = tab
        return tab

    def tables(self, ops: 'Ops', vocab_dev: torch.Tensor, grid_dev: torch.Tensor, map_vocab_dev: torch.Tensor, key: bytes):
        """Per-checkpoint constants (the reference recomputes them in every inference call,
        agent_decoder.py:347-373, map_decoder.py:77-78): token-embedding tables with the bos /
        no_token rows appended, the grid-embedding table with the invalid row, the seed
        categorical embedding and the map-token embedding table.  ``key``: a content hash of the vocabularies / grid the
        caller computed on the host before the upload (``tables_key``) - device addresses say nothing about contents, the
        caching allocator hands a freed engine's addresses to the next one."""
        # one entry PER key, never evicted: engines of different vocabularies share a pack and keep raw pointers to their
        # tables in their contexts / captured graphs (a single-entry cache freed tables a live engine still addressed)
        hit = self._tables_by_key.get(key)
        if hit is not None:
            self._tables, self._tables_key = hit, key
            return hit
        dev, ts, G = self.device, self.cfg.token_size, grid_dev.shape[0]
        tok_tab = torch.empty(3, ts + 2, D, device=dev)
        for k in range(3):
            ops.mlp_embedding(vocab_dev[k][:, -1].reshape(ts, 8).contiguous(), self.tok_emb[k], 8, out=tok_tab[k, :ts])
            tok_tab[k, ts] = self.bos_token[0]
            tok_tab[k, ts + 1] = self.no_token[0]
        grid_tab = torch.empty(G + 1, D, device=dev)
        ops.mlp_embedding(grid_dev, self.grid_emb, 2, out=grid_tab[:G])
        grid_tab[G] = self.invalid_offset[0]
        seed_shape = ops.mlp_embedding(torch.full((1, 3), INVALID_SHAPE, device=dev), self.shape_emb, 3)
        cat_seed = (self.type_a_emb[SEED_TYPE] + seed_shape[0]).contiguous()
        map_tab = ops.mlp_embedding(map_vocab_dev, self.map_tok_emb, map_vocab_dev.shape[1])
        # the all-invalid seed query row (agent_decoder.py:1814-1818): a constant of the weights
        raw = torch.tensor([[2.0 * 2.0 ** 0.5, -2.356194490192345, 0.0, 0.0]], device=dev)   # |(-2,-2)|, atan2(-2,-2)
        fus = torch.zeros(1, 4 * D, device=dev)
        fus[0, :D] = self.no_token[0]
        fus[0, 2 * D:3 * D] = self.state_a_emb[0]
        fus[0, 3 * D:] = grid_tab[G // 2]
        ops.fourier(raw, 2, self.four_xa, fus[:, D:2 * D], cat=cat_seed[None].contiguous())
        f_seed = ops.mlp_embedding(fus, self.fusion, 4 * D)
        self._tables = dict(tok_tab=tok_tab, grid_tab=grid_tab, cat_seed=cat_seed, map_tab=map_tab, f_seed=f_seed)
        self._tables_key = key
        self._tables_by_key[key] = self._tables
        return self._tables

    @staticmethod
    def tables_key(*arrays: np.ndarray) -> bytes:
        h = hashlib.blake2b(digest_size=16)
        for a in arrays:
            a = np.ascontiguousarray(a)
            h.update(str((a.shape, a.dtype.str)).encode())
            h.update(a.tobytes())
        return h.digest()


class Ops:
    """Thin typed wrappers over the C ABI (torch tensors in, device pointers out)."""

    def __init__(self, device: torch.device):
        self.lib = _lib.load()
        self.device = device

    @property
    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def linear(self, x, w_pack, w_off, n, k, bias_off=None, post_ln_off=None, relu=False, pre_ln=None,
               out=None, ldx=None, gather=None, rows=None):
        """y = [LN relu]( x @ W^T + b ); weight/bias/LN live inside ``w_pack`` at float offsets."""
        rows = x.shape[0] if rows is None else rows
        npad = _round_up(n, 32)
        if out is None:
            out = torch.empty(rows, n, device=self.device, dtype=torch.float32)
        base = w_pack.data_ptr()
        bias = base + 4 * bias_off if bias_off is not None else None
        pg = base + 4 * post_ln_off if post_ln_off is not None else None
        pb = base + 4 * (post_ln_off + 128) if post_ln_off is not None else None
        prg, prb = (pre_ln if pre_ln is not None else (None, None))
        _lib.check(self.lib.infgen_linear(_lib.ptr(x), ldx or x.stride(0), _lib.ptr(gather), rows, k,
                                          base + 4 * w_off, npad, bias, n, prg, prb, pg, pb, int(relu),
                                          _lib.ptr(out), out.stride(0), self.stream), 'infgen_linear')
        return out

    def mlp_embedding(self, x, pack, k0, out=None):
        """MLPEmbedding.forward (reference infgen/modules/layers.py:180-192)"""
        k0p, o1, o2, o3 = packing.mlp_embedding_offsets(k0)
        h = self.linear(x, pack, o1, 128, k0, bias_off=o1 + k0p * 128, post_ln_off=o1 + k0p * 128 + 128, relu=True)
        h = self.linear(h, pack, o2, 128, 128, bias_off=o2 + 16384, post_ln_off=o2 + 16384 + 128, relu=True)
        return self.linear(h, pack, o3, 128, 128, bias_off=o3 + 16384, out=out)

    def mlp_layer(self, x, pack, k0, n_out, out=None):
        """MLPLayer.forward (reference infgen/modules/layers.py:214): pack = packing.pack_mlp_layer"""
        k0p = _round_up(k0, 8)
        o_b0 = k0p * 128
        h = self.linear(x, pack, 0, 128, k0, bias_off=o_b0, post_ln_off=o_b0 + 128, relu=True)
        o_w3 = o_b0 + 3 * 128
        return self.linear(h, pack, o_w3, n_out, 128, bias_off=o_w3 + 128 * _round_up(n_out, 32), out=out)

    def mlp_layers(self, x, packs_nout):
        """several MLPLayers (``mlp_layer``) on the same input in two launches: all first Linears (+ LayerNorm, ReLU), then
        all output Linears.  packs_nout: [(pack, n_out), ...] -> list of outputs"""
        rows, n = x.shape[0], len(packs_nout)
        hid = torch.empty(n, rows, 128, device=self.device, dtype=torch.float32)
        outs = [torch.empty(rows, no, device=self.device, dtype=torch.float32) for _, no in packs_nout]
        d1, d2 = (_lib.LinearDesc * n)(), (_lib.LinearDesc * n)()
        o_b0, o_w3 = 128 * 128, 128 * 128 + 3 * 128
        for i, (pack, no) in enumerate(packs_nout):
            base = pack.data_ptr()
            a = d1[i]
            a.X, a.ldx, a.gather, a.rows, a.K = x.data_ptr(), x.stride(0), None, rows, 128
            a.Wp, a.Np, a.bias, a.N = base, 128, base + 4 * o_b0, 128
            a.pre_g = a.pre_b = None
            a.post_g, a.post_b, a.relu = base + 4 * (o_b0 + 128), base + 4 * (o_b0 + 256), 1
            a.Y, a.ldy = hid[i].data_ptr(), 128
            b = d2[i]
            npad = _round_up(no, 32)
            b.X, b.ldx, b.gather, b.rows, b.K = hid[i].data_ptr(), 128, None, rows, 128
            b.Wp, b.Np, b.bias, b.N = base + 4 * o_w3, npad, base + 4 * (o_w3 + 128 * npad), no
            b.pre_g = b.pre_b = b.post_g = b.post_b = None
            b.relu = 0
            b.Y, b.ldy = outs[i].data_ptr(), outs[i].stride(0)
        _lib.check(self.lib.infgen_linear_multi(d1, n, self.stream), 'infgen_linear_multi')
        _lib.check(self.lib.infgen_linear_multi(d2, n, self.stream), 'infgen_linear_multi')
        return outs

    def fourier(self, raw, n, pack, out, count_dev=None, rows=None, cat=None, normalize=False):
        rows = raw.shape[0] if rows is None else rows
        _lib.check(self.lib.infgen_fourier_embed(_lib.ptr(raw), n, _lib.ptr(count_dev), rows, _lib.ptr(pack),
                                                 _lib.ptr(cat), cat.stride(0) if cat is not None else 0,
                                                 out.data_ptr(), out.stride(0), int(normalize), self.stream),
                   'infgen_fourier_embed')
        return out

    def attn_pre(self, x, pack, use_src_ln=False, q=None, u=None, k=None, v=None, rows=None):
        rows = x.shape[0] if rows is None else rows
        _lib.check(self.lib.infgen_attn_pre(_lib.ptr(x), rows, _lib.ptr(pack), int(use_src_ln), _lib.ptr(q),
                                            _lib.ptr(u), _lib.ptr(k), _lib.ptr(v), self.stream), 'infgen_attn_pre')

    def edge_attn(self, rows, q, u, ksrc, vsrc, off, cnt, src, rhat, agg, z, sig, wide=None):
        args = (rows, _lib.ptr(q), _lib.ptr(u), _lib.ptr(ksrc), _lib.ptr(vsrc), _lib.ptr(off), _lib.ptr(cnt),
                _lib.ptr(src), _lib.ptr(rhat), _lib.ptr(agg), _lib.ptr(z), _lib.ptr(sig))
        if wide == 'fused':         # u is the layer pack: U / Z / SIG stay on chip, agg already holds the positional part
            _lib.check(self.lib.infgen_edge_attn_fused(rows, _lib.ptr(q), _lib.ptr(u), *args[3:10], self.stream),
                       'infgen_edge_attn_fused')
        elif wide is None:
            _lib.check(self.lib.infgen_edge_attn(*args, self.stream), 'infgen_edge_attn')
        else:
            _lib.check(self.lib.infgen_edge_attn_mode(*args, int(wide), self.stream), 'infgen_edge_attn_mode')

    def attn_post(self, x, pack, agg, z, sig, has_pos=True, rows=None):
        rows = x.shape[0] if rows is None else rows
        _lib.check(self.lib.infgen_attn_post(_lib.ptr(x), rows, _lib.ptr(pack), _lib.ptr(agg), _lib.ptr(z),
                                             _lib.ptr(sig), int(has_pos), self.stream), 'infgen_attn_post')

    def attn_post_pre(self, x, pack, agg, z, sig, next_pack, has_pos=True, q=None, u=None, k=None, v=None):
        """the post part of one layer and the pre part (q / absorbed query / K / V) of the next in one launch"""
        _lib.check(self.lib.infgen_attn_post_pre(_lib.ptr(x), x.shape[0], _lib.ptr(pack), _lib.ptr(agg), _lib.ptr(z),
                                                 _lib.ptr(sig), int(has_pos), _lib.ptr(next_pack), _lib.ptr(q), _lib.ptr(u),
                                                 _lib.ptr(k), _lib.ptr(v), self.stream), 'infgen_attn_post_pre')

    def attention_layer(self, x, pack, off, cnt, src, rhat, x_src=None, scratch=None, wide=None):
        """AttentionLayer.forward (layers.py:61-76) on CSR edges; in place on ``x``."""
        rows = x.shape[0]
        dev = self.device
        sc = scratch if scratch is not None else {}
        q = sc.get('Q', torch.empty(rows, D, device=dev))
        u = sc.get('U', torch.empty(rows, 8 * D, device=dev))
        agg = sc.get('AGG', torch.empty(rows, D, device=dev))
        z = sc.get('Z', torch.empty(rows, 8 * D, device=dev))
        sig = sc.get('SIG', torch.empty(rows, 8, device=dev))
        fused = wide == 'fused'
        if x_src is None:
            k = torch.empty(rows, D, device=dev)
            v = torch.empty(rows, D, device=dev)
            self.attn_pre(x, pack, q=q, u=None if fused else u, k=k, v=v)
        else:
            k = torch.empty(x_src.shape[0], D, device=dev)
            v = torch.empty(x_src.shape[0], D, device=dev)
            self.attn_pre(x_src, pack, use_src_ln=True, k=k, v=v)
            self.attn_pre(x, pack, q=q, u=None if fused else u)
        self.edge_attn(rows, q, pack if fused else u, k, v, off, cnt, src, rhat, agg, z, sig, wide=wide)
        self.attn_post(x, pack, agg, z, sig, has_pos=rhat is not None and not fused)
        return x


class RolloutEngine:
    """Device state + launch sequence for a batch of scenes."""
    copies = 1          # rollouts per scene over one map encoding (instance attribute when given; see __init__)

    def __init__(self, weights: PackedWeights, scenes: Sequence[Mapping], vocab: Mapping[str, np.ndarray],
                 map_vocab: np.ndarray, grid: np.ndarray, a_cap: Optional[int] = None, m_cap: Optional[int] = None,
                 store_logits: bool = False, live_state: bool = False,
                 teacher: Optional[Sequence] = None, x_pt_override: Optional[Sequence] = None,
                 force_enter: bool = False, insert_headroom: Optional[int] = None,
                 sample_k: int = 1, sample_uniforms: Optional[np.ndarray] = None, options: Optional[Mapping[str, int]] = None,
                 insert_k: int = 1, insert_uniforms: Optional[np.ndarray] = None, seed_outputs: bool = False,
                 use_graph: Optional[bool] = None, copies: int = 1, flags: Optional[Mapping[str, bool]] = None,
                 tap_layers: bool = False):
        self.w = weights
        self.options = dict(options) if options else None      # per-engine kernel switches (fields of InfgenOptions)
        # per-engine launch-sequence switches (none changes what is computed beyond fp32 summation order): read from the environment
        # ONCE, here, as defaults (diagnostic A/B runs), overridden by ``flags`` - two engines of one process may differ
        env = os.environ.get
        self.flags = dict(dt_table=env('INFGEN_NO_DT_TAB', '0') != '1',        # the temporal edges' time-gap branch as a lookup
                          map_fuse=env('INFGEN_MAP_FUSE', '1') != '0',        # map encoder's edge side through k_edge_fused
                          row_groups=env('INFGEN_ROW_GROUPS', '1') != '0',    # insertion: visit only the 16-row groups that hold agents
                          row_groups_tight=env('INFGEN_ROW_GROUPS_TIGHT', '1') != '0',
                          graph=env('INFGEN_GRAPH', '0'))                     # '1': replay decode steps from a HIP graph, '2': whole rollout
        self.flags.update(flags or {})
        # test hook (InfgenRollout.tap_x): the residual stream after every (temporal, map, agent) triple of the last decode step
        self._tap_layers = bool(tap_layers)
        self.tap_x = None
        self.cfg = cfg = weights.cfg
        self.device = dev = weights.device
        self.ops = Ops(dev)
        self.lib = self.ops.lib
        # copies = n: every scene is decoded n times in lockstep (the reference's n_rollout_close_val loop, infgen/model/infgen.py:
        # 704-706, whose inference_no_map(data, map_enc), infgen_decoder.py:132-134, exists so that the map is encoded once): the
        # batch has S = len(scenes) * n agent-side scenes (scene i's copies are rows i n .. i n + n - 1: own state, own uniforms)
        # over S0 = len(scenes) map-side scenes - ONE map-token graph, map encoding and set of map K / V rows per scene, found
        # through InfgenRollout.map_scene
        self.copies = int(copies)
        assert self.copies >= 1
        self.scenes = scenes
        self.S0 = len(scenes)
        self.S = S = len(scenes) * self.copies
        self.T = T = cfg.num_columns
        self.R = R = cfg.num_recurrent_steps_val
        self.hc = hc = cfg.hist_columns
        assert hc == 2, 'the kernels assume num_historical_steps=11, shift=5'
        self.W = cfg.window
        self.ring = self.W + 1
        self.store_logits = store_logits
        self.sample_k = int(sample_k)
        self._sample_uniforms = sample_uniforms      # [steps][S][A] float32 in [0,1) (top-k inverse-CDF sampling)
        # scenario insertion: the cell of a new agent from the insert_k most probable ones (reference insert_beam_size = 10,
        # agent_decoder.py:1900-1904) with insert_uniforms [steps][10][S]; 1: arg-max
        self.insert_k = int(insert_k)
        self._insert_u = None
        # record the seed node's per-insertion outputs of the reference's return dict (agent_decoder.py:2099-2113, :2364-2386:
        # next_state_prob_seed, next_pos_rel_prob_seed, grid_*_occ_seed - plot inputs of the reference; two more heads per iteration)
        self.seed_outputs = bool(seed_outputs)
        self.seed_out = None
        # capture the decode steps of a rollout (no insertion: ~47 launches per step, all shapes static) in a HIP graph at the
        # first run and replay it afterwards
        # use_graph = 'all': the WHOLE rollout (reset, map encoder, column-0 chain, every decode step) as one graph, replayed on a
        # stream of the engine's own - what lets one host thread keep several engines on several streams busy (rollout_many): a
        # rollout is ~1,000 launches, and issuing four engines' launches one after the other takes longer than the GPU needs
        self._graph_all = use_graph == 'all' or (use_graph is None and str(self.flags['graph']) == '2')
        if self._graph_all:
            use_graph = False
        self._wgraph = None
        self._wgraph_opts = None
        self._use_graph_arg = use_graph              # None: by size (set below, once the row count is known)
        self.use_graph = bool(use_graph)
        self._graph = None
        self._graph_opts = None
        if self.insert_k > 1:
            assert insert_uniforms is not None, 'cell sampling needs caller-supplied uniforms [steps][10][S]'
            self._insert_u = torch.from_numpy(np.ascontiguousarray(insert_uniforms, dtype=np.float32)).to(weights.device)
        self._x_pt_override = x_pt_override
        self.force_valid = bool(cfg.disable_insertion) and not live_state
        self.insertion = not cfg.disable_insertion
        self.force_enter = force_enter

        # ------------------------------------------------ host-side scene setup
        self._stacked = None
        self._hosts_light = False
        hosts = self._replicate(self._setup_scenes(scenes))
        self.hosts = hosts
        amax = max(h['A'] for h in hosts)
        mmax = max(h['M'] for h in hosts)
        self._amax0 = amax
        head = 0
        if self.insertion:
            head = insert_headroom if insert_headroom is not None else min(10 * cfg.num_decode_steps, 96)
        self.A_cap = A_cap = a_cap or min(_round_up(max(amax + head, 1), 32), self.lib.infgen_layout_query(_lib.Q_MAX_AGENTS))
        self.M_cap = M_cap = m_cap or _round_up(max(mmax, 1), 32)
        assert amax <= A_cap <= self.lib.infgen_layout_query(_lib.Q_MAX_AGENTS) and A_cap % 32 == 0
        assert mmax <= M_cap
        self.rows = rows = S * A_cap
        if self._use_graph_arg is None:
            # replaying the decode steps from a captured HIP graph: opt-in (use_graph=True or INFGEN_GRAPH=1).  Measured in round 3:
            # 64 scenes 25.9 ms with and without it (the step is its kernels' dependency chains, not launch overhead)
            self.use_graph = str(self.flags['graph']) == '1' and not self.insertion

        arr = self._map_side(self._scene_arrays(hosts))
        t = lambda a: torch.from_numpy(a).to(dev)
        for k in self._SCENE_ARRAYS:
            setattr(self, k, t(arr[k]))
        self._map_cat = tuple(t(arr[k]) for k in ('map_tok', 'map_type', 'map_pl', 'map_light'))
        self.map_scene = (torch.arange(S, device=dev, dtype=torch.int32) // self.copies).contiguous() if self.copies > 1 else None
        vocab_np = np.stack([vocab[k] for k in ('veh', 'ped', 'cyc')]).astype(np.float32)
        map_vocab_np = np.asarray(map_vocab, dtype=np.float32).reshape(map_vocab.shape[0], -1)
        grid_np = np.asarray(grid, dtype=np.float32)
        self._tables_key = PackedWeights.tables_key(vocab_np, grid_np, map_vocab_np)
        self.vocab, self._map_vocab, self.grid_xy = t(vocab_np), t(map_vocab_np), t(grid_np)
        self.G = int(grid.shape[0])
        self.teacher_token = self.teacher_state = None
        self.teacher_grid = self.teacher_pos = self.teacher_head = None
        if teacher is not None:
            tt = np.full((S, T, A_cap), -1, np.int32); ts = np.zeros((S, T, A_cap), np.int32)
            tg = np.full((S, T, A_cap), -2, np.int32)
            tp, th = None, None
            for s, tch in enumerate(teacher):
                tok_s, st_s = tch[0], tch[1]
                A = min(np.asarray(tok_s).shape[0], A_cap)        # rows beyond the initial agents: inserted ones (insertion on)
                tt[s, :, :A] = np.asarray(tok_s)[:A].T; ts[s, :, :A] = np.asarray(st_s)[:A].T
                if len(tch) > 2 and tch[2] is not None:           # optional third entry: grid cells (A, T) of the teacher state
                    tg[s, :, :A] = np.asarray(tch[2])[:A].T
                if len(tch) > 4 and tch[3] is not None:           # optional fourth / fifth: poses (A, T, 2) / (A, T) of the teacher state
                    if tp is None:
                        tp, th = np.zeros((S, T, A_cap, 2), np.float32), np.zeros((S, T, A_cap), np.float32)
                    tp[s, :, :A] = np.asarray(tch[3], np.float32)[:A].transpose(1, 0, 2); th[s, :, :A] = np.asarray(tch[4], np.float32)[:A].T
            self.teacher_token, self.teacher_state = t(tt), t(ts)
            if (tg > -2).any():
                self.teacher_grid = t(tg)
            if tp is not None:
                self.teacher_pos, self.teacher_head = t(tp), t(th)

        # ------------------------------------------------ scratch / caches
        f = lambda *shape: torch.zeros(*shape, device=dev, dtype=torch.float32)
        i32 = lambda *shape: torch.zeros(*shape, device=dev, dtype=torch.int32)
        L = cfg.num_agent_layers
        self.X, self.Q, self.U = f(rows, D), f(rows, D), f(rows, 8 * D)
        if self._tap_layers:
            self.tap_x = f(cfg.num_agent_layers, rows, D)
        self.Ka, self.Va, self.AGG, self.Z, self.SIG = f(rows, D), f(rows, D), f(rows, D), f(rows, 8 * D), f(rows, 8)
        self.ringK = [f(self.ring, rows, D) for _ in range(L)]
        self.ringV = [f(self.ring, rows, D) for _ in range(L)]
        self.mapK = [f(self.S0 * M_cap, D) for _ in range(L)]
        self.mapV = [f(self.S0 * M_cap, D) for _ in range(L)]
        self.edges = {}
        totals = i32(3)        # the three edge totals back to back: infgen_build_edges clears them with one memset
        for k, (name, cap) in enumerate((('t', rows * self.W), ('m', rows * 5), ('a', rows * (A_cap - 1)))):
            cap = max(cap, 32)
            self.edges[name] = dict(off=i32(rows), cnt=i32(rows), src=i32(cap), raw=f(cap, 4), rhat=f(cap, D),
                                    total=totals[k:k + 1], cap=cap)
        self.raw2, self.cat, self.fus_in = f(rows, 4), f(rows, D), f(rows, 4 * D)
        self.tmp1, self.tmp2 = f(rows, D), f(rows, D)
        self.next_token, self.next_state = i32(rows), i32(rows)
        steps = cfg.num_decode_steps
        self.logits = f(steps, rows, cfg.token_size) if store_logits else None
        self.pred_traj, self.pred_head, self.pred_state = f(S, A_cap, R, 2), f(S, A_cap, R), f(S, A_cap, R)
        self.sample_u = self.logits_scratch = None
        if self.sample_k > 1:
            assert sample_uniforms is not None, 'top-k sampling needs caller-supplied uniforms'
            self.sample_u = torch.from_numpy(self._uniform_rows(sample_uniforms, amax)).to(dev)
            if self.logits is None:
                self.logits_scratch = f(rows, cfg.token_size)
        self.tok_tab = self.grid_tab = self.cat_agent = self.cat_seed = None
        self.x_pt = None
        self._ctx = None
        self._init = None
        self.ins = None
        self._mg = None
        self._mg_checked = False
        self._map_nbr_cap = 40          # compacted pt<->pt edges per map token (grown on overflow)
        self._prologue_done = False
        self._decoded_rows = torch.zeros((), device=dev, dtype=torch.int64)

    # ------------------------------------------------------------------ scene arrays (host -> device)
    _SCENE_ARRAYS = ('pos', 'head', 'state', 'token', 'gridtok', 'tmask', 'imask', 'catflag', 'atype', 'bos', 'n_agents', 'n_map',
                     'av', 'map_pos', 'map_orient', '_shape10')

    _MAP_SIDE = ('n_map', 'map_pos', 'map_orient', 'map_tok', 'map_type', 'map_pl', 'map_light')

    def _replicate(self, hosts0):
        """the per-scene host dicts of the agent-side batch: scene i's copies are adjacent (the dicts are shared, read-only)"""
        if self.copies == 1:
            return hosts0
        self._stacked = None                   # (the stacked one-shape arrays describe the S0 distinct scenes)
        return [h for h in hosts0 for _ in range(self.copies)]

    def _map_side(self, arr):
        """the map-side arrays of a batch with copies: one entry per DISTINCT scene (every copies-th of the replicated batch)"""
        if self.copies > 1:
            for k in self._MAP_SIDE:
                arr[k] = np.ascontiguousarray(arr[k][::self.copies])
        return arr

    def _scene_arrays(self, hosts) -> Dict[str, np.ndarray]:
        """the padded [S][T][A_cap] / [S][M_cap] arrays of a batch (section 4 of DESIGN.md) from the per-scene host dicts"""
        S, T, A_cap, M_cap = self.S, self.T, self.A_cap, self.M_cap

        def zeros(shape, dtype):
            return np.zeros(shape, dtype=dtype)
        a = dict(pos=zeros((S, T, A_cap, 2), np.float32), head=zeros((S, T, A_cap), np.float32),
                 state=zeros((S, T, A_cap), np.int32), token=np.full((S, T, A_cap), -1, np.int32),
                 gridtok=np.full((S, T, A_cap), -1, np.int32), tmask=zeros((S, T, A_cap), np.uint8),
                 imask=zeros((S, T, A_cap), np.uint8), catflag=zeros((S, T, A_cap), np.uint8),
                 atype=zeros((S, A_cap), np.int32), bos=zeros((S, A_cap), np.int32),
                 _shape10=np.full((S, A_cap, 3), INVALID_SHAPE, np.float32),
                 n_agents=zeros((S,), np.int32), n_map=zeros((S,), np.int32), av=zeros((S,), np.int32),
                 map_pos=zeros((S, M_cap, 2), np.float32), map_orient=zeros((S, M_cap), np.float32),
                 map_tok=zeros((S, M_cap), np.int64), map_type=zeros((S, M_cap), np.int64),
                 map_pl=zeros((S, M_cap), np.int64), map_light=zeros((S, M_cap), np.int64))
        k = getattr(self, '_stacked', None)
        if k is not None and len(hosts) == S:                      # one-shape batch: the stacked arrays of _setup_scenes_stacked
            A, M = k['A'], k['M']
            a['n_agents'][:], a['n_map'][:], a['av'][:] = A, M, k['av']
            a['pos'][:, :, :A] = k['pos'].transpose(0, 2, 1, 3); a['head'][:, :, :A] = k['head'].transpose(0, 2, 1)
            for dst, src in (('state', 'state'), ('token', 'token'), ('gridtok', 'grid'), ('tmask', 'tmask'), ('imask', 'imask'),
                             ('catflag', 'catflag')):
                a[dst][:, :, :A] = k[src].transpose(0, 2, 1)
            a['atype'][:, :A] = k['type']; a['bos'][:, :A] = k['bos']; a['_shape10'][:, :A] = k['shape10']
            a['map_pos'][:, :M] = k['map_pos']; a['map_orient'][:, :M] = k['map_orient']
            a['map_tok'][:, :M] = k['map_tok']; a['map_type'][:, :M] = k['map_type']
            a['map_pl'][:, :M] = k['map_pl']; a['map_light'][:, :M] = k['map_light']
            return a
        for s, h in enumerate(hosts):
            A, M = h['A'], h['M']
            a['n_agents'][s], a['n_map'][s], a['av'][s] = A, M, h['av']
            a['pos'][s, :, :A] = h['pos'].transpose(1, 0, 2); a['head'][s, :, :A] = h['head'].T
            a['state'][s, :, :A] = h['state'].T; a['token'][s, :, :A] = h['token'].T; a['gridtok'][s, :, :A] = h['grid'].T
            a['tmask'][s, :, :A] = h['tmask'].T; a['imask'][s, :, :A] = h['imask'].T; a['catflag'][s, :, :A] = h['catflag'].T
            a['atype'][s, :A] = h['type']; a['bos'][s, :A] = h['bos']; a['_shape10'][s, :A] = h['shape10']
            a['map_pos'][s, :M] = h['map_pos']; a['map_orient'][s, :M] = h['map_orient']
            a['map_tok'][s, :M] = h['map_tok']; a['map_type'][s, :M] = h['map_type']
            a['map_pl'][s, :M] = h['map_pl']; a['map_light'][s, :M] = h['map_light']
        return a

    def fits(self, scenes: Sequence[Mapping]) -> bool:
        """can ``reload`` take this batch? (same scene count, agents + insertion head-room and map tokens inside the rows this
        engine allocated)"""
        if len(scenes) != self.S0 or self.teacher_token is not None:
            return False
        amax = max(int((np.asarray(sc['agent']['state_idx'])[:, self.hc - 1] != INVALID).sum()) for sc in scenes)
        mmax = max(int(np.asarray(sc['pt_token']['position']).shape[0]) for sc in scenes)
        head = (self.A_cap - self._amax0) if self.insertion else 0
        return amax + head <= self.A_cap and mmax <= self.M_cap

    def reload(self, scenes: Sequence[Mapping], sample_uniforms: Optional[np.ndarray] = None,
               insert_uniforms: Optional[np.ndarray] = None, x_pt_override: Optional[Sequence] = None):
        """a new batch of scenes of the same layout into this engine's device buffers: one upload per array, no allocation, the
        context block / captured graph / scratch stay (the drop-in entry keeps one engine per layout across calls)"""
        assert self.fits(scenes), 'batch does not fit this engine (RolloutEngine.fits)'
        self.scenes = scenes
        self._stacked = None
        self._hosts_light = False
        self.hosts = hosts = self._replicate(self._setup_scenes(scenes))
        arr = self._map_side(self._scene_arrays(hosts))
        for k in self._SCENE_ARRAYS:
            getattr(self, k).copy_(torch.from_numpy(arr[k]), non_blocking=False)
        for dst, k in zip(self._map_cat, ('map_tok', 'map_type', 'map_pl', 'map_light')):
            dst.copy_(torch.from_numpy(arr[k]))
        if self.sample_k > 1:
            assert sample_uniforms is not None, 'top-k sampling needs caller-supplied uniforms'
            self.sample_u.copy_(torch.from_numpy(self._uniform_rows(sample_uniforms, max(h['A'] for h in hosts))))
        if self.insert_k > 1:
            assert insert_uniforms is not None, 'cell sampling needs caller-supplied uniforms [steps][10][S]'
            self._insert_u.copy_(torch.from_numpy(np.ascontiguousarray(insert_uniforms, dtype=np.float32)))
        self._x_pt_override = x_pt_override
        self._init = None                  # reset() snapshots the new initial state
        self._wgraph = None                # (the whole-rollout graph restores the state from the old snapshot's buffers)
        self._epi = None
        self._mg_checked = False           # the new map may hold more pt <-> pt edges than the buffers
        self._prologue_done = False

    # ------------------------------------------------------------------ a batch that is already on the device
    def fits_device(self, k: Mapping) -> bool:
        """``fits`` for a stacked device batch (``_setup_device``): shapes only, no host copy"""
        ag, pt = k['agent'], k['pt_token']
        if self.copies > 1 or int(ag['state_idx'].shape[0]) != self.S or self.teacher_token is not None:
            return False                        # (batches with copies take the host path: reload)
        A, T0, M = int(ag['state_idx'].shape[1]), int(ag['state_idx'].shape[2]), int(pt['position'].shape[1])
        head = (self.A_cap - self._amax0) if self.insertion else 0
        return A + head <= self.A_cap and M <= self.M_cap and T0 <= self.T and A >= 1

    def reload_device(self, k: Mapping, scenes, sample_uniforms: Optional[np.ndarray] = None,
                      insert_uniforms: Optional[np.ndarray] = None, x_pt_override: Optional[Sequence] = None) -> bool:
        """``reload`` for a batch whose scenes arrive as DEVICE tensors of one shape, stacked per key (``k``: what
        ``modules.infgen_decoder.stack_datas`` returns): the setup statements of ``_setup_scenes_stacked`` / ``_scene_arrays`` and
        the epilogue's input arrays run as torch ops on the device and write this engine's buffers - no device -> host -> device
        round trip of the scene arrays (~70 ms of a 512-scene call before the first launch).  ``scenes`` is the (lazy) host form
        of the same batch, only read if somebody asks for the host-side ``outputs()``.  Returns False, having changed nothing,
        when a row would be filtered (the per-scene host path handles that: ``reload``)."""
        assert self.fits_device(k), 'batch does not fit this engine (RolloutEngine.fits_device)'
        if not self._setup_device(k):
            return False
        self.scenes = scenes
        self._stacked = None
        if self.sample_k > 1:
            assert sample_uniforms is not None, 'top-k sampling needs caller-supplied uniforms'
            self.sample_u.copy_(torch.from_numpy(self._uniform_rows(sample_uniforms, self.hosts[0]['A'])))
        if self.insert_k > 1:
            assert insert_uniforms is not None, 'cell sampling needs caller-supplied uniforms [steps][10][S]'
            self._insert_u.copy_(torch.from_numpy(np.ascontiguousarray(insert_uniforms, dtype=np.float32)))
        self._x_pt_override = x_pt_override
        self._init = None
        self._wgraph = None
        self._mg_checked = False
        self._prologue_done = False
        return True

    def _setup_device(self, k: Mapping) -> bool:
        """reference agent_decoder.py:1609-1719 (pad, zero the future, masks) for a one-shape batch on the device; the same
        statements as ``_setup_scenes_stacked`` (tests/test_boundary_cpu.py compares the two array by array on CPU tensors)"""
        cfg = self.cfg
        T, hc, H, S, A_cap, M_cap = cfg.num_columns, cfg.hist_columns, cfg.num_historical_steps, self.S, self.A_cap, self.M_cap
        ag, pt = k['agent'], k['pt_token']
        state0 = ag['state_idx'].long()                                                   # [S, A, T0]
        A, T0, M = int(state0.shape[1]), int(state0.shape[2]), int(pt['position'].shape[1])
        dev = state0.device
        av = ag['av_index'].reshape(S, -1)[:, 0].long()
        # the one host copy of the way in: the ego rows + "is any row filtered?" (then the host path takes the batch)
        chk = torch.cat([av, (state0[:, :, hc - 1] == INVALID).any().long()[None]]).cpu().numpy()
        if chk[-1]:
            return False
        av_host = chk[:-1]

        def pad(x, val):
            if x.shape[2] == T:
                return x.clone()
            shp = tuple(x.shape[:2]) + (T - x.shape[2],) + tuple(x.shape[3:])
            return torch.cat([x, torch.full(shp, val, dtype=x.dtype, device=dev)], dim=2)
        pos = pad(ag['token_pos'].float(), 0.0)
        head = pad(ag['token_heading'].float(), 0.0)
        token = pad(ag['token_idx'].long(), -1)
        state = pad(state0, INVALID)
        grid = pad(ag['grid_token_idx'].long(), -1)
        valid = pad(ag['raw_agent_valid_mask'].bool(), True)
        pos[:, :, hc:] = 0; head[:, :, hc:] = 0; token[:, :, hc:] = -1; state[:, :, hc:] = INVALID; grid[:, :, hc:] = -1
        valid[:, :, hc:] = True
        valid &= ag['valid_mask'][:, :, H - 1].bool()[..., None]
        is_bos, is_eos = state == ENTER, state == EXIT
        bos = torch.where(is_bos.any(2), is_bos.int().argmax(2), 0)
        eos = torch.where(is_eos.any(2), is_eos.int().argmax(2), T - 1)
        cols = torch.arange(T, device=dev)[None, None, :]
        motion = (cols > bos[..., None]) & (cols <= eos[..., None])
        motion[:, :, H // cfg.shift:] = False
        tmask = torch.where(motion, valid, True)
        nonmotion = ~motion
        nonmotion[:, :, H // cfg.shift:] = False
        imask = ~nonmotion
        imask |= state == ENTER
        imask[torch.arange(S, device=dev), av] = True
        tmask[:, :, hc:] = True
        imask[:, :, hc:] = True
        catflag = state != INVALID

        def put(dst, src, fill=0):          # [S, A, T, ...] -> the engine's [S, T, A_cap, ...]
            dst.fill_(fill)
            dst[:, :, :A] = src.transpose(1, 2)
        put(self.pos, pos); put(self.head, head); put(self.state, state); put(self.token, token, -1); put(self.gridtok, grid, -1)
        put(self.tmask, tmask); put(self.imask, imask); put(self.catflag, catflag)
        self.atype.zero_(); self.atype[:, :A] = ag['type']
        self.bos.zero_(); self.bos[:, :A] = bos
        self._shape10.fill_(INVALID_SHAPE); self._shape10[:, :A] = ag['shape'][:, :, H - 1]
        self.n_agents.fill_(A); self.n_map.fill_(M); self.av.copy_(av)
        self.map_pos.zero_(); self.map_pos[:, :M] = pt['position'][:, :, :2]
        self.map_orient.zero_(); self.map_orient[:, :M] = pt['orientation']
        light = torch.gather(k['light_type'].long(), 1, k['edge_index'][:, 1].long())
        for dst, src in zip(self._map_cat, (pt['token_idx'], pt['type'], pt['pl_type'], light)):
            dst.zero_()
            dst[:, :M] = src
        # the epilogue's inputs (outputs_device: E)
        P = int(ag['position'].shape[2])
        Rg = P - H
        z = lambda *shape, dt=torch.float32: torch.zeros(*shape, dtype=dt, device=dev)
        htok, hst = z(S, A_cap, hc, dt=torch.int64), z(S, A_cap, hc, dt=torch.int64)
        htok[:, :A] = ag['token_idx'][:, :, :hc]; hst[:, :A] = state0[:, :, :hc]
        p0, h0, shp = z(S, A_cap, 2), z(S, A_cap), z(S, A_cap, 3)
        p0[:, :A] = ag['position'][:, :, 0, :2]; h0[:, :A] = ag['heading'][:, :, 0]; shp[:, :A] = ag['shape'][:, :, hc - 1]
        gt = z(S, A_cap, Rg, 2); gt[:, :A] = ag['position'][:, :, H:, :2]
        val = z(S, A_cap, T, dt=torch.bool); val[:, :A] = valid
        ids = z(S, A_cap, dt=torch.int64)
        i0 = ag['id'].long()
        ids[:, :A] = i0
        ids[:, A:] = i0.max(dim=1).values[:, None] + 1 + torch.arange(A_cap - A, device=dev)[None, :]
        n0_host = np.full(S, A, np.int64)
        self._gt_len = [Rg] * S
        self._epi = dict(htok=htok, hst=hst, p0=p0, h0=h0, ids=ids, shp=shp, gt=gt, val=val, n0=torch.full((S,), A, device=dev),
                         n0_host=n0_host, eval_shape=torch.tensor([[4.3, 1.8, 1.0], [0.5, 0.5, 1.0], [1.9, 0.5, 1.0]], device=dev))
        filt = np.ones(A, bool)
        self.hosts = [dict(A=A, M=M, av=int(av_host[s]), filt=filt) for s in range(S)]
        self._hosts_light = True             # (the host-side ``outputs()`` rebuilds the full per-scene dicts when asked)
        return True

    def _full_hosts(self):
        """the per-scene host dicts with every array of ``_setup_scene`` (a device-side reload keeps only A / M / av / filt)"""
        if getattr(self, '_hosts_light', False):
            self.scenes = list(self.scenes)
            epi = self._epi
            self.hosts = self._replicate(self._setup_scenes(self.scenes))
            self._stacked = None
            self._epi = epi
            self._hosts_light = False
        return self.hosts

    def _uniform_rows(self, sample_uniforms, amax) -> np.ndarray:
        steps, S, A_cap = self.cfg.num_decode_steps, self.S, self.A_cap
        su = np.asarray(sample_uniforms, dtype=np.float32)
        # a missing column would sample with u = 0, i.e. greedily - never fill in silently
        need = A_cap if self.insertion else amax
        assert su.ndim == 3 and su.shape[0] >= steps and su.shape[1] >= S and su.shape[2] >= need, \
            f'sample_uniforms {su.shape} does not cover [steps={steps}][S={S}][rows={need}]'
        u = np.zeros((steps, S, A_cap), np.float32)
        u[:, :, :min(A_cap, su.shape[2])] = su[:steps, :S, :A_cap]
        return u.reshape(steps, S * A_cap)

    # ------------------------------------------------------------------ host setup of a batch
    def _setup_scenes(self, scenes) -> List[Dict[str, np.ndarray]]:
        """``_setup_scene`` for every scene of a batch.  A 512-scene batch spends ~100 ms in 512 x ~40 small numpy calls; when the
        scenes have one shape (same agent / column / map-token counts) and no row is filtered - what a batch of one dataset
        looks like - the same statements run once on stacked arrays, and the per-scene dicts are views of them."""
        try:
            out = self._setup_scenes_stacked(scenes)
        except (ValueError, KeyError, TypeError):
            out = None
        return out if out is not None else [self._setup_scene(sc) for sc in scenes]

    def _setup_scenes_stacked(self, scenes):
        cfg = self.cfg
        T, hc, H = cfg.num_columns, cfg.hist_columns, cfg.num_historical_steps
        S = len(scenes)
        if S < 8:
            return None
        st = lambda grp, k: np.stack([np.asarray(sc[grp][k]) for sc in scenes])          # raises ValueError on ragged shapes
        state0 = st('agent', 'state_idx').astype(np.int64)                                # [S, A, T0]
        if (state0[:, :, hc - 1] == INVALID).any():
            return None                                                                   # a filtered row: per-scene path
        A, T0 = state0.shape[1], state0.shape[2]
        if T0 > T:
            return None
        av = np.stack([np.asarray(sc['agent']['av_index']).reshape(-1)[0] for sc in scenes]).astype(np.int64)

        def pad(x, val):
            if x.shape[2] == T:
                return x.copy()
            shp = x.shape[:2] + (T - x.shape[2],) + tuple(x.shape[3:])
            return np.concatenate([x, np.full(shp, val, dtype=x.dtype)], axis=2)
        pos = pad(st('agent', 'token_pos').astype(np.float32), 0.0)
        head = pad(st('agent', 'token_heading').astype(np.float32), 0.0)
        token = pad(st('agent', 'token_idx').astype(np.int64), -1)
        state = pad(state0, INVALID)
        grid = pad(st('agent', 'grid_token_idx').astype(np.int64), -1)
        valid = pad(st('agent', 'raw_agent_valid_mask').astype(bool), True)
        pos[:, :, hc:] = 0; head[:, :, hc:] = 0; token[:, :, hc:] = -1; state[:, :, hc:] = INVALID; grid[:, :, hc:] = -1
        valid[:, :, hc:] = True
        eval_mask = np.stack([np.asarray(sc['agent']['valid_mask'])[:, H - 1] for sc in scenes]).astype(bool)
        valid[~eval_mask] = False
        is_bos, is_eos = state == ENTER, state == EXIT
        bos = np.where(is_bos.any(2), is_bos.argmax(2), 0)
        eos = np.where(is_eos.any(2), is_eos.argmax(2), T - 1)
        cols = np.arange(T)[None, None, :]
        motion = (cols > bos[..., None]) & (cols <= eos[..., None])
        motion[:, :, H // cfg.shift:] = False
        tmask = np.ones((S, A, T), bool)
        tmask[motion] = valid[motion]
        imask = np.ones((S, A, T), bool)
        nonmotion = ~motion
        nonmotion[:, :, H // cfg.shift:] = False
        imask[nonmotion] = False
        imask[state == ENTER] = True
        imask[np.arange(S), av] = True
        tmask[:, :, hc:] = True
        imask[:, :, hc:] = True
        catflag = state != INVALID
        atype = st('agent', 'type').astype(np.int64)
        shape10 = np.stack([np.asarray(sc['agent']['shape'])[:, H - 1] for sc in scenes]).astype(np.float32)
        mpos = np.stack([np.asarray(sc['pt_token']['position'])[:, :2] for sc in scenes]).astype(np.float32)
        morient = st('pt_token', 'orientation').astype(np.float32)
        mtok, mtype, mpl = (st('pt_token', k).astype(np.int64) for k in ('token_idx', 'type', 'pl_type'))
        e1 = np.stack([np.asarray(sc['pt_token__to__map_polygon']['edge_index'])[1] for sc in scenes]).astype(np.int64)
        lt = [np.asarray(sc['map_polygon']['light_type']).astype(np.int64) for sc in scenes]
        light = np.stack([l[e] for l, e in zip(lt, e1)])
        M = mpos.shape[1]
        filt = np.ones(A, bool)
        self._stacked = dict(pos=pos, head=head, state=state, token=token, grid=grid, tmask=tmask, imask=imask, catflag=catflag,
                             type=atype, bos=bos, shape10=shape10, av=av, map_pos=mpos, map_orient=morient, map_tok=mtok,
                             map_type=mtype, map_pl=mpl, map_light=light, A=A, M=M)
        return [dict(A=A, M=M, av=int(av[s]), filt=filt, pos=pos[s], head=head[s], token=token[s], state=state[s], grid=grid[s],
                     valid=valid[s], tmask=tmask[s], imask=imask[s], catflag=catflag[s], bos=bos[s], type=atype[s],
                     shape10=shape10[s], map_pos=mpos[s], map_orient=morient[s], map_tok=mtok[s], map_type=mtype[s],
                     map_pl=mpl[s], map_light=light[s]) for s in range(S)]

    # ------------------------------------------------------------------ host setup of one scene
    def _setup_scene(self, scene) -> Dict[str, np.ndarray]:
        """reference agent_decoder.py:1609-1719 (filter, pad, zero the future, masks)"""
        cfg = self.cfg
        ag = scene['agent']
        T, hc = cfg.num_columns, cfg.hist_columns
        state0 = np.asarray(ag['state_idx']).astype(np.int64)
        filt = state0[:, hc - 1] != INVALID
        av0 = int(np.asarray(ag['av_index']).reshape(-1)[0])
        av = av0 - int((~filt[:av0]).sum())

        def take(k):
            return np.asarray(ag[k])[filt]

        def pad(x, val):
            if x.shape[1] >= T:
                return x[:, :T].copy() if x.shape[1] > T else x.copy()
            shp = (x.shape[0], T - x.shape[1]) + tuple(x.shape[2:])
            return np.concatenate([x, np.full(shp, val, dtype=x.dtype)], axis=1)
        pos = pad(take('token_pos').astype(np.float32), 0.0)
        head = pad(take('token_heading').astype(np.float32), 0.0)
        token = pad(take('token_idx').astype(np.int64), -1)
        state = pad(state0[filt], INVALID)
        grid = pad(take('grid_token_idx').astype(np.int64), -1)
        valid = pad(take('raw_agent_valid_mask').astype(bool), True)
        assert pos.shape[1] == T, 'token arrays longer than the rollout are not supported (SURVEY a-Q15)'
        A = pos.shape[0]
        pos[:, hc:] = 0; head[:, hc:] = 0; token[:, hc:] = -1; state[:, hc:] = INVALID; grid[:, hc:] = -1
        valid[:, hc:] = True
        eval_mask = np.asarray(ag['valid_mask'])[filt][:, cfg.num_historical_steps - 1].astype(bool)
        valid[~eval_mask] = False
        is_bos, is_eos = state == ENTER, state == EXIT
        bos = np.where(is_bos.any(1), is_bos.argmax(1), 0)
        eos = np.where(is_eos.any(1), is_eos.argmax(1), T - 1)
        cols = np.arange(T)[None, :]
        motion = (cols > bos[:, None]) & (cols <= eos[:, None])
        motion[:, cfg.num_historical_steps // cfg.shift:] = False
        tmask = np.ones((A, T), bool)
        tmask[motion] = valid[motion]
        imask = np.ones((A, T), bool)
        nonmotion = ~motion
        nonmotion[:, cfg.num_historical_steps // cfg.shift:] = False
        imask[nonmotion] = False
        imask[state == ENTER] = True
        imask[av] = True
        tmask[:, hc:] = True
        imask[:, hc:] = True
        catflag = (state != INVALID)
        pt = scene['pt_token']
        e = np.asarray(scene['pt_token__to__map_polygon']['edge_index'])
        light = np.asarray(scene['map_polygon']['light_type']).astype(np.int64)[e[1].astype(np.int64)]
        return dict(
            A=A, M=int(np.asarray(pt['position']).shape[0]), av=av, filt=filt, pos=pos, head=head, token=token,
            state=state, grid=grid, valid=valid, tmask=tmask, imask=imask, catflag=catflag, bos=bos,
            type=take('type').astype(np.int64), shape10=np.asarray(ag['shape'])[filt][:, cfg.num_historical_steps - 1].astype(np.float32),
            map_pos=np.asarray(pt['position'])[:, :2].astype(np.float32), map_orient=np.asarray(pt['orientation']).astype(np.float32),
            map_tok=np.asarray(pt['token_idx']).astype(np.int64), map_type=np.asarray(pt['type']).astype(np.int64),
            map_pl=np.asarray(pt['pl_type']).astype(np.int64), map_light=light)

    def _fourier_split(self) -> bool:
        """is the Fourier embedding of the operator-level entries the split kernel (infgen_set_fourier_mode != 0)?  (they read
        the process-wide switches, like every call of the map encoder)"""
        o = _lib.Options()
        _lib.check(self.lib.infgen_get_effective_options(C.byref(o)), 'infgen_get_effective_options')
        return o.fourier_mode != 0

    # ------------------------------------------------------------------ prologue (once per rollout)
    def reset(self):
        """restore the scene state to the rollout's initial condition (device-to-device copies)"""
        if self._init is None:
            self._init = {k: getattr(self, k).clone() for k in self._STATE}
        else:
            for k in self._STATE:
                getattr(self, k).copy_(self._init[k])
        for buf in (self.pred_traj, self.pred_head, self.pred_state):
            buf.zero_()

    _STATE = ('pos', 'head', 'state', 'token', 'gridtok', 'imask', 'catflag', 'tmask', 'atype', 'bos', 'n_agents')

    def prologue(self, map_only: bool = False):
        """per-scene constants and the first columns: agent categorical embeddings, map encoder
        (map_decoder.py:70-130), map K/V of the six map->agent layers, the edgeless column-0 chain
        (SURVEY a-Q3) and column 1's raw feature."""
        # the operator-level calls below (map encoder, tables) run under THIS engine's switches: installed as the calling thread's
        # option block for the duration of the prologue (re-entrant across host threads; include/infgen_hip.h)
        with _lib.thread_options(self._effective_options()):
            return self._prologue(map_only)

    def _effective_options(self):
        o = _lib.Options()
        _lib.check(self.lib.infgen_get_options(C.byref(o)), 'infgen_get_options')
        for k, v in (self.options or {}).items():
            setattr(o, k, int(v))
        o.use, o.row_groups, o.n_row_groups = 0, None, None
        return o

    def _prologue(self, map_only: bool = False):
        ops, w, cfg, dev = self.ops, self.w, self.cfg, self.device
        S, A_cap, M_cap, rows = self.S, self.A_cap, self.M_cap, self.rows
        tabs = w.tables(ops, self.vocab, self.grid_xy, self._map_vocab, self._tables_key)
        self.tok_tab, self.grid_tab, self.cat_seed = tabs['tok_tab'], tabs['grid_tab'], tabs['cat_seed']
        self.f_seed = tabs['f_seed']                 # (this engine's entry, not whatever the pack built last)
        self.reset()
        # categorical embedding rows (agent_decoder.py:376-380,492)
        if self.cat_agent is None:
            self.cat_agent = torch.empty(rows, D, device=dev)
        shp = ops.mlp_embedding(self._shape10.reshape(rows, 3), w.shape_emb, 3)
        torch.add(w.type_a_emb[self.atype.reshape(-1).long()], shp, out=self.cat_agent)

        if self._x_pt_override is not None:
            return self._prologue_with_given_map()
        # ---- map encoder (once per DISTINCT scene)
        S0 = self.S0
        mrows = S0 * M_cap
        mtok, mtype, mpl, mlight = self._map_cat
        if self.x_pt is None:
            self.x_pt = torch.empty(mrows, D, device=dev)
        # token-table row + (type + polygon type) + light embedding (map_decoder.py:87-89) in one pass over the rows
        mt = tabs['map_tab']
        _lib.check(self.lib.infgen_embedding_sum4(_lib.ptr(mt), _lib.ptr(mtok), mt.shape[0], _lib.ptr(w.type_pt_emb), _lib.ptr(mtype),
                                                  w.type_pt_emb.shape[0], _lib.ptr(w.polygon_type_emb), _lib.ptr(mpl),
                                                  w.polygon_type_emb.shape[0], _lib.ptr(w.light_pl_emb), _lib.ptr(mlight),
                                                  w.light_pl_emb.shape[0], mrows, _lib.ptr(self.x_pt), ops.stream),
                   'infgen_embedding_sum4')
        x_pt = self.x_pt
        K = 100
        if self._mg is None:
            cap = mrows * self._map_nbr_cap
            self._mg = dict(off=torch.zeros(mrows, device=dev, dtype=torch.int32),
                            cnt=torch.zeros(mrows, device=dev, dtype=torch.int32),
                            src=torch.zeros(cap, device=dev, dtype=torch.int32), raw=torch.zeros(cap, 4, device=dev),
                            rhat=torch.empty(cap, D, device=dev), total=torch.zeros(1, device=dev, dtype=torch.int32),
                            cap=cap, K=torch.empty(mrows, D, device=dev), V=torch.empty(mrows, D, device=dev),
                            Q=torch.empty(mrows, D, device=dev), AGG=torch.empty(mrows, D, device=dev))
        g = self._mg
        _lib.check(self.lib.infgen_map_graph(S0, M_cap, _lib.ptr(self.n_map), _lib.ptr(self.map_pos),
                                             _lib.ptr(self.map_orient), float(cfg.pl2pl_radius), K, _lib.ptr(g['off']),
                                             _lib.ptr(g['cnt']), _lib.ptr(g['src']), _lib.ptr(g['raw']),
                                             _lib.ptr(g['total']), g['cap'], ops.stream), 'infgen_map_graph')
        if not self._mg_checked:
            tot = int(g['total'].item())           # one host sync, first prologue only
            if tot > g['cap']:
                self._map_nbr_cap = (tot + mrows - 1) // mrows + 4
                self._mg = None
                return self._prologue(map_only=map_only)
            self._mg_checked = True
        fused = bool(self.flags['map_fuse'])
        # rhat rows of the pt <-> pt edges: fp32 by default; options['rhat_format'] = 1: the packed 24-bit form when both ends know
        # it (split Fourier kernel -> k_edge_fused), like the rollout's own edge sets (include/infgen_hip.h: InfgenOptions.rhat_format)
        eo = self._effective_options()
        r24 = fused and eo.rhat_format == 1 and eo.fourier_mode != 0
        if r24:
            _lib.check(self.lib.infgen_fourier_embed_r24(_lib.ptr(g['raw']), 3, _lib.ptr(g['total']), g['cap'], _lib.ptr(w.four_pt),
                                                         g['rhat'].data_ptr(), ops.stream), 'infgen_fourier_embed_r24')
        else:
            ops.fourier(g['raw'], 3, w.four_pt, g['rhat'], count_dev=g['total'], rows=g['cap'], normalize=True)
        for i in range(cfg.num_map_layers):
            if not fused:                                           # the unfused sequence (comparison)
                if 'U' not in g:
                    g.update(U=torch.empty(mrows, 8 * D, device=dev), Z=torch.empty(mrows, 8 * D, device=dev),
                             SIG=torch.empty(mrows, 8, device=dev))
                ops.attn_pre(x_pt, w.attn_pt[i], q=g['Q'], u=g['U'], k=g['K'], v=g['V'])
                ops.edge_attn(mrows, g['Q'], g['U'], g['K'], g['V'], g['off'], g['cnt'], g['src'], g['rhat'],
                              g['AGG'], g['Z'], g['SIG'])
                ops.attn_post(x_pt, w.attn_pt[i], g['AGG'], g['Z'], g['SIG'])
                continue
            # edge side with k_edge_fused: the absorbed query and the positional aggregate stay on chip (no U / Z arrays); the pre
            # part of layer i + 1 rides in the post launch of layer i (as in the agent layers)
            if i == 0:
                ops.attn_pre(x_pt, w.attn_pt[0], q=g['Q'], k=g['K'], v=g['V'])
            if r24:
                _lib.check(self.lib.infgen_edge_attn_fused_r24(mrows, _lib.ptr(g['Q']), _lib.ptr(w.attn_pt[i]), _lib.ptr(g['K']),
                                                               _lib.ptr(g['V']), _lib.ptr(g['off']), _lib.ptr(g['cnt']),
                                                               _lib.ptr(g['src']), g['rhat'].data_ptr(), _lib.ptr(g['AGG']),
                                                               ops.stream), 'infgen_edge_attn_fused_r24')
            else:
                ops.edge_attn(mrows, g['Q'], w.attn_pt[i], g['K'], g['V'], g['off'], g['cnt'], g['src'], g['rhat'],
                              g['AGG'], None, None, wide='fused')
            if i + 1 < cfg.num_map_layers:
                ops.attn_post_pre(x_pt, w.attn_pt[i], g['AGG'], None, None, w.attn_pt[i + 1], has_pos=False,
                                  q=g['Q'], k=g['K'], v=g['V'])
            else:
                ops.attn_post(x_pt, w.attn_pt[i], g['AGG'], None, None, has_pos=False)
        if map_only:
            return
        self._finish_prologue()

    def _prologue_with_given_map(self):
        """inference_no_map: x_pt comes from the caller (reference infgen_decoder.py:132-134)"""
        S0, M_cap, dev = self.S0, self.M_cap, self.device
        if self.x_pt is None:
            self.x_pt = torch.zeros(S0 * M_cap, D, device=dev)
        assert len(self._x_pt_override) == S0, 'one x_pt per distinct scene'
        for s, xp in enumerate(self._x_pt_override):
            M = self.hosts[s * self.copies]['M']
            self.x_pt[s * M_cap:s * M_cap + M] = xp.detach().to(dev, torch.float32)
        self._finish_prologue()

    def _finish_prologue(self):
        ops, w, cfg = self.ops, self.w, self.cfg
        x_pt = self.x_pt
        # map K/V of the six pt2a layers (bipartite source LayerNorm)
        for i in range(cfg.num_agent_layers):
            ops.attn_pre(x_pt, w.attn_m[i], use_src_ln=True, k=self.mapK[i], v=self.mapV[i])

        if self.insertion:
            self._alloc_insertion()
            for i in range(3):
                ops.attn_pre(x_pt, w.attn_pt2sa[i], use_src_ln=True, k=self.ins['mapK'][i], v=self.ins['mapV'][i])
        if self._ctx is None:
            self._build_ctx()
        self._refresh_opts()
        self._decoded_rows.zero_()
        st = ops.stream
        # column 0: edgeless chain, its K/V land in ring slot 0 (SURVEY a-Q3); then column 1's raw feature
        _lib.check(self.lib.infgen_raw_feature(C.byref(self._ctx), 0, st), 'raw_feature(0)')
        _lib.check(self.lib.infgen_decode_layers(C.byref(self._ctx), 0, 1, st), 'decode_layers(0)')
        _lib.check(self.lib.infgen_raw_feature(C.byref(self._ctx), 1, st), 'raw_feature(1)')
        self._prologue_done = True

    def rollout(self):
        """one full pass of the hot path over the batch: prologue + every decode step"""
        if (self._graph_all and not self.insertion and self._x_pt_override is None and self._mg_checked and self._init is not None
                and self._ctx is not None and not _lib.prof_active()):
            self._rollout_graph()                # (the first rollout of a batch runs eagerly: buffers, tables, edge capacities)
            return
        self.prologue()
        self.run()

    def _rollout_graph(self):
        """prologue + decode steps as ONE HIP-graph replay on the engine's own stream, fenced against the caller's stream on both
        sides (see _run_graph for why not on the caller's stream).  Nothing in a rollout without insertion depends on the host
        after the first one: the scene state is restored from device copies, every launch has static shapes."""
        cur = torch.cuda.current_stream(self.device)
        if getattr(self, '_gstream', None) is None:
            self._gstream = torch.cuda.Stream(device=self.device)
        gs = self._gstream
        gs.wait_stream(cur)
        with torch.cuda.stream(gs):
            self._refresh_opts(groups=True)
            snap = bytes(self._ctx.opts)[:_lib.OPTIONS_VALUE_BYTES] + bytes([self._ctx.four_t_dt is not None])
            if self._wgraph is not None and snap != self._wgraph_opts:
                self._wgraph = None
            if self._wgraph is None:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=gs):
                    self.prologue()
                    self.run()
                self._wgraph, self._wgraph_opts = g, snap
            self._wgraph.replay()
            self._prologue_done = True
        cur.wait_stream(gs)

    # ------------------------------------------------------------------ scenario insertion (host-sequenced)
    def _alloc_insertion(self):
        if self.ins is not None:
            for k in ('n_new', 'active', 'inserted'):
                self.ins[k].zero_()
            self.ins['shape_all'].fill_(INVALID_SHAPE)
            self.ins['first_new'].fill_(self.A_cap)
            self.ins['inserted_rows'] = [[] for _ in range(self.S)]
            if self.seed_out is not None:
                for v in self.seed_out.values():
                    v.zero_()
            return
        dev, S, rows, A_cap, M_cap, G = self.device, self.S, self.rows, self.A_cap, self.M_cap, self.G
        f = lambda *shape: torch.zeros(*shape, device=dev, dtype=torch.float32)
        i32 = lambda *shape: torch.zeros(*shape, device=dev, dtype=torch.int32)

        def ebuf(cap):
            cap = max(cap, 32)
            return dict(off=i32(S), cnt=i32(S), src=i32(cap), raw=f(cap, 4), rhat=f(cap, D), total=i32(1), cap=cap)
        ar = torch.arange(S, device=dev, dtype=torch.int32)
        self.ins = dict(
            occ=f(S, G), occ_emb=f(S, D), Kocc=[f(S, D) for _ in range(3)], Vocc=[f(S, D) for _ in range(3)],
            mapK=[f(self.S0 * M_cap, D) for _ in range(3)], mapV=[f(self.S0 * M_cap, D) for _ in range(3)],
            Ksa=[f(rows, D) for _ in range(3)], Vsa=[f(rows, D) for _ in range(3)],
            Kh=[f(rows, D) for _ in range(3)], Vh=[f(rows, D) for _ in range(3)],
            Xc=f(rows, D), AGG0=f(rows, D), Z0=f(rows, 8 * D), SIG0=f(rows, 8),
            XS=f(2 * S, D), QS=f(2 * S, D), US=f(2 * S, 8 * D), AGGS=f(2 * S, D), ZS=f(2 * S, 8 * D), SIGS=f(2 * S, 8),
            KN=f(2 * S, D), VN=f(2 * S, D),
            ea_s=ebuf(S * A_cap), em_s=ebuf(S * min(M_cap, 2048)), ea_h=ebuf(S * 24), em_h=ebuf(S * 128),
            occ_off=ar.clone(), occ_cnt=torch.ones(S, device=dev, dtype=torch.int32), occ_src=ar.clone(),
            dec3=i32(3 * S), n_new=i32(S), new_cell=i32(S), new_shape=f(S, 3),
            first_new=torch.full((S,), A_cap, device=dev, dtype=torch.int32), hv_ovr=f(S, 2), shape_all=torch.full((rows, 3), INVALID_SHAPE, device=dev),
            scene_base=(ar * A_cap).contiguous(), inserted_rows=[[] for _ in range(S)],
            groups=i32((rows + 15) // 16), n_groups=i32(1),
            host_dec=torch.zeros(3, S, dtype=torch.int32).pin_memory(),
            new_local=i32(S), prev_row=i32(S), prev_mask=i32(S), pend_row=i32(S), pend_mask=i32(S),
            hid=f(6, S, D), lg_state=f(S, 2), lg_type=f(S, 3), shape=f(S, 3), lg_pos=f(S, G), lg_heading=f(S, int(360.0 / self.cfg.angle_interval)),
            offset=f(S, 2), t1=f(S, D), t2=f(S, D), shp=f(S, D), struct=None,
            host_ev=torch.cuda.Event())
        # inserted | new_row | active back to back: one device-to-host copy per iteration hands all three over
        d3 = self.ins['dec3']
        self.ins['inserted'], self.ins['new_row'], self.ins['active'] = d3[:S], d3[S:2 * S], d3[2 * S:]
        if self.seed_outputs:
            steps = self.cfg.num_decode_steps
            self.seed_out = dict(state=f(S, 11, steps), pos=f(S, 11, steps, G), occ_a=f(S, 11, steps, G), occ_p=f(S, 11, steps, G),
                                 occ_gt=f(S, 11, steps, G))
            sd, ap, dv = self.w.sd, self.w.ap, self.device
            if not hasattr(self.w, 'fwd_heads'):
                self.w.fwd_heads = {k: torch.from_numpy(np.ascontiguousarray(packing.pack_mlp_layer(sd, f'{ap}.{k}'), dtype=np.float32)).to(dv)
                                    for k in ('grid_index_head', 'grid_agent_occ_head', 'grid_pt_occ_head')}

    def _ebuf_struct(self, e):
        b = _lib.EdgeBuf()
        P = _lib.ptr
        b.off, b.cnt, b.src, b.raw, b.rhat, b.total, b.cap = P(e['off']), P(e['cnt']), P(e['src']), P(e['raw']), P(e['rhat']), P(e['total']), e['cap']
        return b

    def _insertion_struct(self):
        """InfgenInsertion: the device arrays / weights of the sub-loop for the library's sequencing (include/infgen_hip.h)"""
        I, w, cfg, P = self.ins, self.w, self.cfg, _lib.ptr
        b = _lib.Insertion()
        for i in range(3):
            b.attn_occ2sa[i], b.attn_pt2sa[i], b.attn_a2sa[i] = P(w.attn_occ2sa[i]), P(w.attn_pt2sa[i]), P(w.attn_a2sa[i])
            b.Kocc[i], b.Vocc[i], b.mapK[i], b.mapV[i] = P(I['Kocc'][i]), P(I['Vocc'][i]), P(I['mapK'][i]), P(I['mapV'][i])
            b.Ksa[i], b.Vsa[i], b.Kh[i], b.Vh[i] = P(I['Ksa'][i]), P(I['Vsa'][i]), P(I['Kh'][i]), P(I['Vh'][i])
        H = w.heads
        b.four_a2sa, b.four_pt2sa = P(w.four_a2sa), P(w.four_pt2sa)
        b.head_state, b.head_type, b.head_shape = P(H['seed_state_predict_head']), P(H['seed_type_predict_head']), P(H['seed_shape_predict_head'])
        b.head_pos, b.head_heading = P(H['seed_pos_rel_token_predict_head']), P(H['seed_heading_rel_token_predict_head'])
        b.head_offset, b.occ_embed = P(H['seed_offset_xy_predict_head']), P(H['seed_agent_occ_embed'])
        b.shape_emb, b.type_a_emb, b.f_seed = P(w.shape_emb), P(w.type_a_emb), P(self.f_seed)
        b.occ, b.occ_emb, b.Xc = P(I['occ']), P(I['occ_emb']), P(I['Xc'])
        b.zero_agg, b.zero_z, b.zero_sig = P(I['AGG0']), P(I['Z0']), P(I['SIG0'])
        for k in ('XS', 'QS', 'US', 'AGGS', 'ZS', 'SIGS', 'KN', 'VN'):
            setattr(b, k, P(I[k]))
        b.ea_s, b.em_s, b.ea_h, b.em_h = (self._ebuf_struct(I[k]) for k in ('ea_s', 'em_s', 'ea_h', 'em_h'))
        b.occ_off, b.occ_cnt, b.occ_src = P(I['occ_off']), P(I['occ_cnt']), P(I['occ_src'])
        for k in ('active', 'n_new', 'inserted', 'new_row', 'new_cell', 'new_local', 'new_shape', 'prev_row', 'prev_mask', 'pend_row',
                  'pend_mask', 'hv_ovr', 'shape_all', 'hid', 'lg_state', 'lg_type', 'shape', 'lg_pos', 'lg_heading', 'offset', 't1', 't2', 'shp'):
            setattr(b, k, P(I[k]))
        b.host_dec = I['host_dec'].data_ptr()
        b.r_seed, b.r_a2sa, b.r_pl2sa, b.angle_interval = float(cfg.pl2seed_radius), float(cfg.a2sa_radius), float(cfg.pl2sa_radius), float(cfg.angle_interval)
        b.n_heading, b.force_enter, b.insert_k, b.max_new = int(360.0 / cfg.angle_interval), int(self.force_enter), self.insert_k, 10
        return b

    def _insert_step(self, t: int):
        """the insertion sub-loop of decode step t (reference agent_decoder.py:1773-2105).  The launches of an
        iteration are sequenced by the library (infgen_insert_seed -> decisions -> infgen_insert_heading: two C calls instead of
        ~60 Python-level launches); the data-dependent loop stays here.  A generator: once per iteration the per-scene decisions
        (inserted?, which row, still active?) arrive in pinned host memory asynchronously and the event of that copy is yielded -
        ``run`` just waits for it, ``rollout_many`` sequences other engines' streams meanwhile."""
        lib, I, S = self.lib, self.ins, self.S
        ctx = C.byref(self._ctx)
        if I.get('struct') is None:
            I['struct'] = self._insertion_struct()
        blk = C.byref(I['struct'])
        I['first_new'].copy_(self.n_agents)
        I['active'].fill_(1)
        I['n_new'].zero_()
        riders = riders_h = h_ready = False
        hd = I['host_dec'].numpy()
        for it in range(10):
            st = self.ops.stream
            u = _lib.ptr(self._insert_u[t, it]) if self._insert_u is not None else None
            _lib.check(lib.infgen_insert_seed(ctx, blk, t, it, int(riders), u, st), 'infgen_insert_seed')
            ev = I['host_ev']
            ev.record(torch.cuda.current_stream(self.device))
            yield ev
            ev.synchronize()
            ins_host = hd[0].copy()
            if (ins_host < 0).any():
                full = np.nonzero(ins_host < 0)[0]
                raise InsertionHeadroomError(
                    f'decode step {t}: scene(s) {full[:8].tolist()} have used all {self.A_cap} agent rows '
                    f'({int(self.n_agents[int(full[0])].item())} agents) and the seed head asks for another insertion; '
                    f're-run with a larger insert_headroom (rows per scene <= {self.lib.infgen_layout_query(_lib.Q_MAX_AGENTS)})',
                    needed=self.A_cap)
            ins_host = ins_host > 0
            if not ins_host.any():
                if self.insert_k > 1 and hd[2].any():
                    riders = False          # sampled cells were occupied everywhere: the iteration is spent, active scenes draw again
                    continue
                break
            ins_idx_host = np.nonzero(ins_host)[0]
            nr_host = hd[1][ins_idx_host].astype(np.int64)
            for s_i, r_i in zip(ins_idx_host, nr_host):
                I['inserted_rows'][int(s_i)].append((int(r_i), t))
            if self.seed_out is not None:
                # slot = the scene's insertion count of this step after the append (agent_decoder.py:2099-2105)
                ops, w = self.ops, self.w
                ins = torch.from_numpy(ins_idx_host.astype(np.int64)).to(self.device)
                so, slot = self.seed_out, I['n_new'][ins].long()
                XS_in = I['XS'][:S][ins].contiguous()
                so['state'][ins, slot, t] = torch.softmax(I['lg_state'][ins], dim=-1)[:, -1]
                so['pos'][ins, slot, t] = torch.softmax(I['lg_pos'][ins], dim=-1)
                so['occ_a'][ins, slot, t] = ops.mlp_layer(XS_in, w.fwd_heads['grid_agent_occ_head'], D, self.G)
                so['occ_p'][ins, slot, t] = ops.mlp_layer(XS_in, w.fwd_heads['grid_pt_occ_head'], D, self.G)
                so['occ_gt'][ins, slot, t] = I['occ'][ins]
            # heading stage of the rows just appended; the rows of the previous heading stage ride along (their K / V of the motion
            # layers 0..2 are refreshed) - not necessarily the previous iteration's: an occupied sampled cell spends iterations
            _lib.check(lib.infgen_insert_heading(ctx, blk, t, int(h_ready), int(riders_h and h_ready), self.ops.stream), 'infgen_insert_heading')
            h_ready = True
            riders = riders_h = True

    def _build_ctx(self):
        cfg, w = self.cfg, self.w
        c = _lib.Rollout()
        c.S, c.A_cap, c.T, c.M_cap, c.W, c.ring, c.R = self.S, self.A_cap, self.T, self.M_cap, self.W, self.ring, self.R
        c.token_size, c.grid_size, c.num_layers = cfg.token_size, self.G, cfg.num_agent_layers
        c.force_valid, c.store_logits = int(self.force_valid), int(self.store_logits)
        c.r_map, c.r_agent = float(cfg.pl2a_radius), float(cfg.a2a_radius)
        P = _lib.ptr
        c.n_agents, c.n_map, c.av_index = P(self.n_agents), P(self.n_map), P(self.av)
        c.pos, c.head, c.state, c.token, c.grid = P(self.pos), P(self.head), P(self.state), P(self.token), P(self.gridtok)
        c.tmask, c.imask, c.catflag, c.type, c.bos = P(self.tmask), P(self.imask), P(self.catflag), P(self.atype), P(self.bos)
        c.map_pos, c.map_orient = P(self.map_pos), P(self.map_orient)
        c.map_scene = P(self.map_scene)
        c.tap_x = P(self.tap_x)
        for i in range(cfg.num_agent_layers):
            c.attn_t[i], c.attn_m[i], c.attn_a[i] = P(w.attn_t[i]), P(w.attn_m[i]), P(w.attn_a[i])
            c.ringK[i], c.ringV[i], c.mapK[i], c.mapV[i] = P(self.ringK[i]), P(self.ringV[i]), P(self.mapK[i]), P(self.mapV[i])
        c.four_t, c.four_m, c.four_a, c.four_xa = P(w.four_t), P(w.four_m), P(w.four_a), P(w.four_xa)
        c.fusion_pack, c.tok_head_pack, c.st_head_pack = P(w.fusion), P(w.tok_head), P(w.st_head)
        c.tok_tab, c.grid_tab, c.state_emb = P(self.tok_tab), P(self.grid_tab), P(w.state_a_emb)
        c.cat_agent, c.cat_seed, c.vocab, c.grid_xy = P(self.cat_agent), P(self.cat_seed), P(self.vocab), P(self.grid_xy)
        c.X, c.Q, c.U, c.Ka, c.Va, c.AGG, c.Z, c.SIG = (P(self.X), P(self.Q), P(self.U), P(self.Ka), P(self.Va),
                                                          P(self.AGG), P(self.Z), P(self.SIG))
        for name, field in (('t', 'et'), ('m', 'em'), ('a', 'ea')):
            e, b = self.edges[name], getattr(c, field)
            b.off, b.cnt, b.src, b.raw, b.rhat, b.total, b.cap = (P(e['off']), P(e['cnt']), P(e['src']), P(e['raw']),
                                                                   P(e['rhat']), P(e['total']), e['cap'])
        c.raw2, c.cat, c.fus_in, c.tmp1, c.tmp2 = P(self.raw2), P(self.cat), P(self.fus_in), P(self.tmp1), P(self.tmp2)
        c.next_token, c.next_state, c.logits = P(self.next_token), P(self.next_state), P(self.logits)
        c.teacher_token, c.teacher_state = P(self.teacher_token), P(self.teacher_state)
        c.teacher_grid = P(self.teacher_grid)
        c.teacher_pos, c.teacher_head = P(self.teacher_pos), P(self.teacher_head)
        c.pred_traj, c.pred_head, c.pred_state = P(self.pred_traj), P(self.pred_head), P(self.pred_state)
        if self.insertion:
            c.first_new, c.hv_ovr = P(self.ins['first_new']), P(self.ins['hv_ovr'])
        c.sample_k, c.sample_u, c.logits_scratch = self.sample_k, P(self.sample_u), P(self.logits_scratch)
        self._ctx = c
        self._refresh_opts()
        _lib.check(self.lib.infgen_rollout_validate(C.byref(c)), 'infgen_rollout_validate')      # (the packs' headers, looked at afresh)

    def _refresh_opts(self, groups: bool = False):
        """the context carries its own kernel switches (InfgenRollout.opts, re-entrant): explicit ``options`` of this engine,
        otherwise a snapshot of the library's process-wide defaults (infgen_set_*) taken at every prologue / run"""
        o = self._ctx.opts
        _lib.check(self.lib.infgen_get_options(C.byref(o)), 'infgen_get_options')
        for k, v in (self.options or {}).items():
            setattr(o, k, int(v))
        o.use = 1
        if (int(o.gemm_terms) == 2) != (self.w.operand_bits == 8):
            raise ValueError(f'gemm_terms = {int(o.gemm_terms)} with packs of {self.w.operand_bits}-bit operands: the bf16 mode '
                             '(gemm_terms = 2) takes PackedWeights(..., operand_bits=8), every other mode operand_bits=11')
        # the temporal edges' time-gap input as a lookup of its r_t_emb branch (include/infgen_hip.h: four_t_dt), built once per
        # weight pack and arithmetic; INFGEN_NO_DT_TAB=1: evaluated per edge as before
        self._ctx.four_t_dt = None
        if o.fourier_mode != 0 and self.flags['dt_table']:
            self._ctx.four_t_dt = _lib.ptr(self.w.time_gap_table(self.lib, int(o.gemm_terms), self.ops.stream))
        o.row_groups = o.n_row_groups = None
        o.row_group_margin = 0
        if groups and self.insertion and self.ins is not None and self.flags['row_groups']:
            # rows are padded to A_cap per scene: the split node kernels and the edge kernels visit only the 16-row groups
            # that hold agents (or may receive one of the <= 10 rows a step appends)
            o.row_groups, o.n_row_groups, o.row_group_margin = _lib.ptr(self.ins['groups']), _lib.ptr(self.ins['n_groups']), 10

    # ------------------------------------------------------------------ rollout
    def run(self, t0: int = 0, t1: Optional[int] = None):
        for _ in self.run_gen(t0, t1):      # (the generator waits for each event itself when it is resumed)
            pass

    def run_gen(self, t0: int = 0, t1: Optional[int] = None):
        """``run`` as a generator: yields a ``torch.cuda.Event`` wherever the host needs a result of the device (the insertion
        sub-loop's per-iteration decisions) - everything up to the event is already enqueued on the current stream.  Lets one
        host thread keep several engines on several streams busy (``rollout_many``)."""
        if not self._prologue_done:
            self.prologue()
        t1 = self.cfg.num_decode_steps if t1 is None else t1
        self._refresh_opts(groups=True)        # (the group list is rebuilt on the device at every decode step below)
        if not self.insertion:
            # a replay runs what was captured: fall back to the eager sequence while per-kernel profiling is on (events are not
            # part of the graph) and re-capture when the context's kernel switches changed since the capture
            if self.use_graph and (t0, t1) == (0, self.cfg.num_decode_steps) and not _lib.prof_active():
                self._run_graph(t0, t1)
                return
            _lib.check(self.lib.infgen_rollout_run(C.byref(self._ctx), t0, t1, self.ops.stream), 'infgen_rollout_run')
            return
        lib, I = self.lib, self.ins
        use_groups = bool(self._ctx.opts.row_groups)
        for t in range(t0, t1):
            if use_groups:
                _lib.check(lib.infgen_active_row_groups(_lib.ptr(self.n_agents), self.S, self.A_cap, 10,
                                                        _lib.ptr(I['groups']), _lib.ptr(I['n_groups']), self.ops.stream),
                           'infgen_active_row_groups')
            if t > 0:
                yield from self._insert_step(t)
            self._decoded_rows.add_(self.n_agents.sum())       # A_t: rows decoded at this step, incl. the inserted ones
            tight = use_groups and self.flags['row_groups_tight']
            if tight:
                # the step's insertions are done: the motion stage runs on exactly the rows that hold agents now - the ten rows of
                # head-room the sub-loop's lists carry put one more 16-row group per scene into 60 % of the node / edge launches
                _lib.check(lib.infgen_active_row_groups(_lib.ptr(self.n_agents), self.S, self.A_cap, 0,
                                                        _lib.ptr(I['groups']), _lib.ptr(I['n_groups']), self.ops.stream),
                           'infgen_active_row_groups')
                self._ctx.opts.row_group_margin = 0
            self.step(t)
            if tight:
                self._ctx.opts.row_group_margin = 10

    def _run_graph(self, t0, t1):
        """the decode steps as a HIP-graph replay.  Replays launched into the LEGACY DEFAULT stream fault now and then on this
        ROCm 7.2 stack (memory access fault inside the second or a later replay, 2 of 4 runs; never with AMD_SERIALIZE_KERNEL=3, never
        on a created stream: profiles/r03_graph_replay_fault.log), so capture and replay always run on a stream of the engine's
        own, fenced against the caller's stream on both sides."""
        cur = torch.cuda.current_stream(self.device)
        if getattr(self, '_gstream', None) is None:
            self._gstream = torch.cuda.Stream(device=self.device)
        gs = self._gstream
        gs.wait_stream(cur)
        with torch.cuda.stream(gs):
            snap = bytes(self._ctx.opts)[:_lib.OPTIONS_VALUE_BYTES] + bytes([self._ctx.four_t_dt is not None])
            if self._graph is not None and snap != self._graph_opts:
                self._graph = None
            if self._graph is None and not getattr(self, '_graph_warm', False):
                # the first rollout runs eagerly: kernels are loaded lazily at their first launch, which must not happen
                # inside a capture
                self._graph_warm = True
                _lib.check(self.lib.infgen_rollout_run(C.byref(self._ctx), t0, t1, self.ops.stream), 'infgen_rollout_run')
            else:
                if self._graph is None:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=gs):
                        _lib.check(self.lib.infgen_rollout_run(C.byref(self._ctx), t0, t1, self.ops.stream), 'infgen_rollout_run')
                    self._graph = g
                    self._graph_opts = snap
                self._graph.replay()
        cur.wait_stream(gs)

    def edge_totals(self):
        """(temporal, map, agent) edge counts of the last decode step's edge sets (one host sync)"""
        return tuple(int(self.edges[k]['total'].item()) for k in ('t', 'm', 'a'))

    def scenes_at_row_cap(self) -> int:
        """scenes whose insertion head-room ran out (n_agents == A_cap): their later insertions were dropped - give the engine a
        larger ``insert_headroom`` (A_cap <= 1024) to avoid that"""
        return int((self.n_agents >= self.A_cap).sum().item()) if self.insertion else 0

    def step(self, t: int):
        _lib.check(self.lib.infgen_decode_step(C.byref(self._ctx), t, self.ops.stream), 'infgen_decode_step')

    def ctx_tensor(self) -> torch.Tensor:
        """the bytes of this engine's InfgenRollout block (after ``prologue``) - the state handle of
        ``torch.ops.infgen_hip.decode_step``"""
        assert self._ctx is not None, 'run prologue() first'
        self._refresh_opts(groups=True)
        return torch.frombuffer(bytearray(bytes(self._ctx)), dtype=torch.uint8)

    # ------------------------------------------------------------------ outputs (reference :2303-2389)
    def outputs(self) -> List[Dict[str, np.ndarray]]:
        """the reference's return dict per scene (agent_decoder.py:2303-2389); rows appended by the
        insertion loop follow the initial ones like in the reference"""
        torch.cuda.synchronize(self.device)
        self._full_hosts()
        cfg, hc, H = self.cfg, self.hc, self.cfg.num_historical_steps
        pos, head = self.pos.cpu().numpy(), self.head.cpu().numpy()
        state, token = self.state.cpu().numpy(), self.token.cpu().numpy()
        ptraj, phead, pstate = self.pred_traj.cpu().numpy(), self.pred_head.cpu().numpy(), self.pred_state.cpu().numpy()
        logits = self.logits.cpu().numpy() if self.logits is not None else None
        x_pt = self.x_pt.cpu().numpy() if self.x_pt is not None else None
        tabs = self.vocab.cpu().numpy()
        n_fin = self.n_agents.cpu().numpy()
        atype_dev = self.atype.cpu().numpy()
        bos_dev = self.bos.cpu().numpy()
        shape_all = self.ins['shape_all'].cpu().numpy().reshape(self.S, self.A_cap, 3) if self.ins is not None else None
        outs = []
        for s, h in enumerate(self.hosts):
            A0, M = h['A'], h['M']
            A = int(n_fin[s])
            sc = self.scenes[s // self.copies]['agent']
            filt = h['filt']
            pos_a = pos[s, :, :A].transpose(1, 0, 2).copy()
            head_a = head[s, :, :A].T.copy()
            nstate = state[s, :, :A].T.astype(np.int64)
            ntok = token[s, :, :A].T.astype(np.int64)
            # history columns of next_token_idx / next_state_idx are the *input* tokens (:1733-1735)
            ntok[:A0, :hc] = np.asarray(sc['token_idx'])[filt][:, :hc]
            nstate[:A0, :hc] = np.asarray(sc['state_idx'])[filt][:, :hc]
            for a in range(A0, A):                      # inserted rows: no token up to and including bos (:2303-2305)
                ntok[a, :bos_dev[s, a] + 1] = -1
            pt = np.concatenate([np.zeros((A, H, 2), np.float32), ptraj[s, :A]], axis=1)
            ph = np.concatenate([np.zeros((A, H), np.float32), phead[s, :A]], axis=1)
            ps = np.concatenate([np.zeros((A, H), np.float32), pstate[s, :A]], axis=1)
            pt[:A0, 0] = np.asarray(sc['position'])[filt][:, 0, :2]
            ph[:A0, 0] = np.asarray(sc['heading'])[filt][:, 0]
            ps[:A0, 1:H] = np.repeat(np.asarray(sc['state_idx'])[filt][:, :hc], cfg.shift, axis=1)
            htok = np.asarray(sc['token_idx'])[filt][:, :hc].astype(np.int64).copy()
            htok[htok < 0] = 0
            atype0 = h['type']
            hcont = tabs[atype0[:, None], htok]                      # (A0, hc, 6, 4, 2)
            th = head_a[:A0, 0].astype(np.float32)
            cs, sn = np.cos(th)[:, None, None, None], np.sin(th)[:, None, None, None]
            x, y = hcont[..., 0], hcont[..., 1]
            hx = x * cs - y * sn + pos_a[:A0, 0, 0][:, None, None, None]
            hy = x * sn + y * cs + pos_a[:A0, 0, 1][:, None, None, None]
            pt[:A0, 1:H, 0] = hx[:, :, 1:].mean(axis=3).reshape(A0, -1)
            pt[:A0, 1:H, 1] = hy[:, :, 1:].mean(axis=3).reshape(A0, -1)
            ph[:A0, 1:H] = np.arctan2(hy[:, :, 1:, 0] - hy[:, :, 1:, 3], hx[:, :, 1:, 0] - hx[:, :, 1:, 3]).reshape(A0, -1)
            atype = atype_dev[s, :A].astype(np.int64)
            eval_shape = np.asarray([[4.3, 1.8, 1.0], [0.5, 0.5, 1.0], [1.9, 0.5, 1.0]], np.float32)[atype]
            ids0 = np.asarray(sc['id'])[filt]
            ids = np.concatenate([ids0, ids0.max() + 1 + np.arange(A - A0, dtype=ids0.dtype)])
            pshape = np.asarray(sc['shape'])[filt][:, hc - 1].astype(np.float32)
            if A > A0:
                pshape = np.concatenate([pshape, shape_all[s, A0:A]])
            o = dict(ego_index=h['av'], agent_id=ids, valid_mask=h['valid'],
                     pos_a=pos_a, head_a=head_a, pred_traj=pt, pred_head=ph, pred_state=ps,
                     pred_valid=(ps != INVALID) & (ps != ENTER), pred_type=atype,
                     pred_shape=pshape, eval_shape=eval_shape,
                     pred_z=np.zeros_like(ph), next_token_idx=ntok, next_state_idx=nstate,
                     gt_traj=np.asarray(sc['position'])[filt][:, H:, :2].copy(), num_inserted=A - A0)
            if self.ins is not None:
                # label 'A<k>' on the first column after the bos column of the k-th agent a step inserted (:1996-1999)
                labels = [[None] * self.T for _ in range(A)]
                per_step = {}
                for row, t_ in self.ins['inserted_rows'][s]:
                    k_ = per_step[t_] = per_step.get(t_, 0) + 1
                    a_ = row - s * self.A_cap
                    if a_ < A and hc + t_ < self.T:
                        labels[a_][hc + t_] = f'A{k_}'
                o['agent_labels'] = labels
            if self.seed_out is not None:
                so = self.seed_out
                o.update(next_state_prob_seed=so['state'][s].cpu().numpy(), next_pos_rel_prob_seed=so['pos'][s].cpu().numpy(),
                         grid_agent_occ_seed=so['occ_a'][s].cpu().numpy(), grid_pt_occ_seed=so['occ_p'][s].cpu().numpy(),
                         grid_agent_occ_gt_seed=so['occ_gt'][s].cpu().numpy())
            if logits is not None:
                o['logits'] = logits[:, s * self.A_cap:s * self.A_cap + A].copy()
            if x_pt is not None:
                ms = s // self.copies
                o['x_pt'] = x_pt[ms * self.M_cap:ms * self.M_cap + M].copy()
            outs.append(o)
        return outs

    def _epi_from_hosts(self):
        """padded copies of the inputs the device epilogue reads (``outputs_device``), from the per-scene host dicts: one upload
        per array (``_setup_device`` builds the same arrays on the device)"""
        cfg, hc, H, dev = self.cfg, self.hc, self.cfg.num_historical_steps, self.device
        S, A_cap, T, R = self.S, self.A_cap, self.T, self.R
        # padded copies of the inputs the epilogue reads (one upload per array)
        z = lambda *shape, dt=np.float32: np.zeros(shape, dt)
        Rg = max(int(np.asarray(sc['agent']['position']).shape[1]) - H for sc in self.scenes)
        row_scenes = self.scenes if self.copies == 1 else [sc for sc in self.scenes for _ in range(self.copies)]
        htok, hst = z(S, A_cap, hc, dt=np.int64), z(S, A_cap, hc, dt=np.int64)
        p0, h0, ids, shp = z(S, A_cap, 2), z(S, A_cap), z(S, A_cap, dt=np.int64), z(S, A_cap, 3)
        gt, val, n0 = z(S, A_cap, Rg, 2), z(S, A_cap, T, dt=bool), z(S, dt=np.int64)
        for s, (h, sc_) in enumerate(zip(self.hosts, row_scenes)):
            sc, f, A0 = sc_['agent'], h['filt'], h['A']
            n0[s] = A0
            htok[s, :A0] = np.asarray(sc['token_idx'])[f][:, :hc]
            hst[s, :A0] = np.asarray(sc['state_idx'])[f][:, :hc]
            pos = np.asarray(sc['position'])[f]
            p0[s, :A0] = pos[:, 0, :2]
            g = pos[:, H:, :2]
            gt[s, :A0, :g.shape[1]] = g
            h0[s, :A0] = np.asarray(sc['heading'])[f][:, 0]
            i0 = np.asarray(sc['id'])[f]
            ids[s, :A0] = i0
            ids[s, A0:] = (i0.max() if len(i0) else -1) + 1 + np.arange(A_cap - A0)
            shp[s, :A0] = np.asarray(sc['shape'])[f][:, hc - 1]
            val[s, :A0] = h['valid']
        t = lambda a: torch.from_numpy(a).to(dev)
        self._gt_len = [int(np.asarray(sc['agent']['position']).shape[1]) - H for sc in row_scenes]
        return dict(htok=t(htok), hst=t(hst), p0=t(p0), h0=t(h0), ids=t(ids), shp=t(shp), gt=t(gt), val=t(val), n0=t(n0),
                         n0_host=n0, eval_shape=t(np.asarray([[4.3, 1.8, 1.0], [0.5, 0.5, 1.0], [1.9, 0.5, 1.0]], np.float32)))

    def outputs_device(self, detach: bool = False) -> List[Dict[str, torch.Tensor]]:
        """``outputs`` without the host round trip: the same per-scene dicts as device tensors (views of the batch arrays where
        the layout allows), the epilogue of agent_decoder.py:2303-2389 evaluated for all scenes at once on the device.
        VIEWS: the per-scene tensors are slices of batch-wide results of this call (an in-place edit of one scene's tensor edits
        that slice only); ``pos_a`` / ``head_a`` / ``logits`` / ``x_pt`` additionally alias this engine's own buffers, which the next
        rollout overwrites - ``detach=True`` copies those four batch arrays once (what ``InfGenDecoder`` does, whose engines are
        reused across calls)."""
        cfg, hc, H, dev = self.cfg, self.hc, self.cfg.num_historical_steps, self.device
        S, A_cap, T, R = self.S, self.A_cap, self.T, self.R
        if getattr(self, '_epi', None) is None:
            self._epi = self._epi_from_hosts()
        E = self._epi
        n_fin = self.n_agents.long()
        row = torch.arange(A_cap, device=dev)[None, :]
        init = row < E['n0'][:, None]                                                 # rows of the initial agents
        pos_a = self.pos.permute(0, 2, 1, 3)                                          # [S, A_cap, T, 2]
        head_a = self.head.permute(0, 2, 1)
        lg_all, x_pt_all = self.logits, self.x_pt
        if detach:
            pos_a, head_a = pos_a.contiguous(), head_a.contiguous()
            lg_all = lg_all.clone() if lg_all is not None else None
            x_pt_all = x_pt_all.clone() if x_pt_all is not None else None
        nstate = self.state.permute(0, 2, 1).long().clone()
        ntok = self.token.permute(0, 2, 1).long().clone()
        # history columns of next_token_idx / next_state_idx are the input tokens (:1733-1735); inserted rows: no token up to and
        # including their bos column (:2303-2305)
        ntok[:, :, :hc] = torch.where(init[..., None], E['htok'], ntok[:, :, :hc])
        nstate[:, :, :hc] = torch.where(init[..., None], E['hst'], nstate[:, :, :hc])
        cols = torch.arange(T, device=dev)[None, None, :]
        ntok.masked_fill_((~init)[..., None] & (cols <= self.bos.long()[..., None]), -1)      # (a masked assignment would wait for the device)
        zf = lambda *shape: torch.zeros(*shape, device=dev)
        pt = torch.cat([zf(S, A_cap, H, 2), self.pred_traj], dim=2)
        ph = torch.cat([zf(S, A_cap, H), self.pred_head], dim=2)
        ps = torch.cat([zf(S, A_cap, H), self.pred_state], dim=2)
        # history prefill of the initial agents: step 0 = the logged pose, steps 1..H-1 from the history tokens' contours
        atype = self.atype.long()
        hcont = self.vocab[atype[..., None].expand(S, A_cap, hc), E['htok'].clamp(min=0)]        # [S, A_cap, hc, 6, 4, 2]
        th = head_a[:, :, 0]
        cs, sn = torch.cos(th)[..., None, None, None], torch.sin(th)[..., None, None, None]
        x, y = hcont[..., 0], hcont[..., 1]
        hx = x * cs - y * sn + pos_a[:, :, 0, 0][..., None, None, None]
        hy = x * sn + y * cs + pos_a[:, :, 0, 1][..., None, None, None]
        i3 = init[..., None, None]
        pt[:, :, 0] = torch.where(init[..., None], E['p0'], pt[:, :, 0])
        ph[:, :, 0] = torch.where(init, E['h0'], ph[:, :, 0])
        hist_xy = torch.stack([hx[:, :, :, 1:].mean(dim=4), hy[:, :, :, 1:].mean(dim=4)], dim=-1).reshape(S, A_cap, H - 1, 2)
        pt[:, :, 1:H] = torch.where(i3, hist_xy, pt[:, :, 1:H])
        hist_h = torch.atan2(hy[:, :, :, 1:, 0] - hy[:, :, :, 1:, 3], hx[:, :, :, 1:, 0] - hx[:, :, :, 1:, 3]).reshape(S, A_cap, H - 1)
        ph[:, :, 1:H] = torch.where(init[..., None], hist_h, ph[:, :, 1:H])
        ps[:, :, 1:H] = torch.where(init[..., None], E['hst'].float().repeat_interleave(cfg.shift, dim=2), ps[:, :, 1:H])
        pvalid = (ps != INVALID) & (ps != ENTER)
        pshape = E['shp']
        if self.ins is not None:
            pshape = torch.where(init[..., None], E['shp'], self.ins['shape_all'].view(S, A_cap, 3))
        eval_shape = E['eval_shape'][atype]
        # the only host copy: the final agent counts - known without asking when nothing can be inserted (then the call returns
        # with the epilogue enqueued behind the rollout and nothing waited for)
        n_host = n_fin.cpu().numpy() if self.insertion else E['n0_host']
        outs = []
        batch = dict(agent_id=E['ids'], pos_a=pos_a, head_a=head_a, pred_traj=pt, pred_head=ph, pred_state=ps, pred_valid=pvalid,
                     pred_type=atype, pred_shape=pshape, eval_shape=eval_shape, next_token_idx=ntok, next_state_idx=nstate)
        gt_all, val_all, gt_len = E['gt'], E['val'], self._gt_len
        seed_out, ins = self.seed_out, self.ins
        A_capl, M_capl = A_cap, self.M_cap

        def cut(t, s, A):          # (default arguments bind the loop variables: the thunk runs later)
            return lambda: t[s, :A]
        for s, h in enumerate(self.hosts):
            A, A0, M = int(n_host[s]), h['A'], h['M']
            lazy = {k: cut(t, s, A) for k, t in batch.items()}
            lazy['valid_mask'] = cut(val_all, s, A0)
            lazy['pred_z'] = (lambda s=s, A=A: torch.zeros_like(ph[s, :A]))
            lazy['gt_traj'] = (lambda s=s, A0=A0: gt_all[s, :A0, :gt_len[s]])
            o = LazyOut(dict(ego_index=h['av'], num_inserted=A - A0), lazy)
            if ins is not None:
                def labels(s=s, A=A):
                    lab = [[None] * T for _ in range(A)]
                    per_step = {}
                    for r_, t_ in ins['inserted_rows'][s]:
                        k_ = per_step[t_] = per_step.get(t_, 0) + 1
                        a_ = r_ - s * A_capl
                        if a_ < A and hc + t_ < T:
                            lab[a_][hc + t_] = f'A{k_}'
                    return lab
                o.set_lazy('agent_labels', labels)
            if seed_out is not None:
                # (detach: the seed arrays are engine-owned and zeroed / rewritten by the next rollout of a reused engine - cloned NOW)
                for k_out, k_in in (('next_state_prob_seed', 'state'), ('next_pos_rel_prob_seed', 'pos'), ('grid_agent_occ_seed', 'occ_a'),
                                    ('grid_pt_occ_seed', 'occ_p'), ('grid_agent_occ_gt_seed', 'occ_gt')):
                    o[k_out] = seed_out[k_in][s].clone() if detach else seed_out[k_in][s]
            if lg_all is not None:
                o.set_lazy('logits', (lambda s=s, A=A: lg_all[:, s * A_capl:s * A_capl + A]))
            if x_pt_all is not None:
                o.set_lazy('x_pt', (lambda ms=s // self.copies, M=M: x_pt_all[ms * M_capl:ms * M_capl + M]))
            outs.append(o)
        return outs

    def agent_steps(self) -> int:
        """agent-steps (10 Hz) decoded by the last full rollout of this batch (the rows decoded at every step, incl. the
        ones scenario insertion appended, x 5 simulated steps per decode step)"""
        if self.insertion and self._prologue_done:
            return int(self._decoded_rows.item()) * self.cfg.shift
        return int(sum(h['A'] for h in self.hosts)) * self.R


def rollout_many(engines: Sequence[RolloutEngine], streams: Optional[Sequence[torch.cuda.Stream]] = None):
    """Full rollouts of several engines (disjoint scene sets of one GPU), each on its own HIP stream, sequenced by ONE host
    thread: an engine runs until it needs a result of the device (``RolloutEngine.run_gen``), then the next engine's launches
    are enqueued, round robin.  With scenario insertion on, the sub-loop of a decode step is a chain of small dependent launches
    whose length is set by the slowest scene of the batch; several smaller batches in different phases fill the chip where one
    large batch leaves it idle.  Without insertion the engines' launch sequences simply interleave."""
    if not engines:
        return
    dev = engines[0].device
    if streams is None or len(engines) == 1:
        for e in engines:
            e.rollout()
        return
    # (k_layers_p launches of several streams: the library keeps every launch within the device's resident capacity and orders such
    # launches of different streams behind each other - csrc/api.hip: layers_p_launch - so engines of any size may share the GPU;
    # the launch itself is a plain one by default, cooperative with options['layers_p'] = 2)
    return _rollout_many_streams(engines, streams, dev)


def _rollout_many_streams(engines, streams, dev):
    cur = torch.cuda.current_stream(dev)
    if all(e._graph_all and not e.insertion for e in engines):
        # whole-rollout graphs: one replay per engine, nothing for the host to sequence
        try:
            for e, st in zip(engines, streams):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    e.rollout()
        finally:
            for st in streams:
                cur.wait_stream(st)
        return
    live = []
    try:
        for e, st in zip(engines, streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                e.prologue()
                live.append((e.run_gen(), st))
        while live:
            nxt = []
            for g, st in live:
                with torch.cuda.stream(st):
                    try:
                        next(g)
                        nxt.append((g, st))
                    except StopIteration:
                        pass
            live = nxt
    finally:
        for st in streams:          # also when a generator raised (InsertionHeadroomError): the caller's stream waits for all
            cur.wait_stream(st)
