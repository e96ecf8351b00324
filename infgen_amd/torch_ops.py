"""``torch.library`` registration of the hot-path operators (last row): namespace ``infgen_hip``.

Every op is a thin binding of an ``extern "C"`` entry of libinfgen_hip.so (include/infgen_hip.h) - tensors are borrowed, outputs
are allocated here, launches go to torch's current HIP stream, errors surface as ``RuntimeError`` (``InfgenHipError``).  The ops
take the PACKED parameter blocks the library reads (``infgen_amd.packing``; the ``infgen_amd.modules`` layers pack their own
``state_dict`` parameters lazily), so a checkpoint is packed once and the ops are pure functions of tensors:

    torch.ops.infgen_hip.fourier_embed(x (E, n), pack, normalize)                          -> (E, 128)
    torch.ops.infgen_hip.radius_firstk(pos_q (Nq, 2), pos_x (Nx, 2), ptr_q, ptr_x, r, K)    -> idx (Nq, K) int32 (-1 padded), cnt (Nq,)
    torch.ops.infgen_hip.attn_layer(x_dst (N, 128), pack, off, cnt, src, rhat?, x_src?)     -> (N, 128)
    torch.ops.infgen_hip.token_state_head(x (N, 128), tok_pack, st_pack, token_size, want_logits) -> token, state, logits
    torch.ops.infgen_hip.mlp_layer(x (N, K), pack, n_out)                                   -> (N, n_out)
    torch.ops.infgen_hip.mlp_embedding(x (N, K), pack)                                      -> (N, 128)

    torch.ops.infgen_hip.integrate_tokenise(token, state, type, pos, head, n_agents, ego, vocab, grid) -> pos', head', pred_traj, pred_head, grid, state'
    torch.ops.infgen_hip.decode_step(ctx_bytes, t, pos, head, state, token, grid, x, next_token, next_state) -> next_token, next_state

(``decode_step`` works on the persistent state block ``InfgenRollout`` of include/infgen_hip.h, which ``RolloutEngine.ctx_tensor()``
hands out as bytes; whole rollouts stay a C-ABI call, ``infgen_rollout_run``, driven by infgen_amd/engine.py).  ``register_fake`` gives every op a shape function, so they trace under
``torch.compile`` / ``make_fx`` as opaque calls.  Inference only: no autograd formula is registered.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from .engine import Ops, D

_OPS = {}


def _ops(dev: torch.device) -> Ops:
    if dev.type != 'cuda':
        raise _lib.InfgenHipError('infgen_hip ops need cuda tensors: the HIP path has no CPU fallback')
    key = (dev.type, dev.index)
    if key not in _OPS:
        _OPS[key] = Ops(dev)
    return _OPS[key]


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous().float()


@torch.library.custom_op('infgen_hip::fourier_embed', mutates_args=())
def fourier_embed(x: torch.Tensor, pack: torch.Tensor, normalize: bool = False) -> torch.Tensor:
    """FourierEmbedding.forward (reference infgen/modules/layers.py:142-160) of (E, n) continuous inputs, n <= 4; ``normalize``
    appends the affine-free LayerNorm the attention layers share (``attn_prenorm_r`` without gamma / beta)"""
    ops = _ops(x.device)
    n = x.shape[1]
    raw = torch.zeros(x.shape[0], 4, device=x.device)
    raw[:, :n] = x
    out = torch.empty(x.shape[0], D, device=x.device)
    if x.shape[0]:
        ops.fourier(raw, n, _f32(pack), out, normalize=normalize)
    return out


@fourier_embed.register_fake
def _(x, pack, normalize=False):
    return x.new_empty(x.shape[0], D, dtype=torch.float32)


@torch.library.custom_op('infgen_hip::radius_firstk', mutates_args=())
def radius_firstk(pos_q: torch.Tensor, pos_x: torch.Tensor, ptr_q: torch.Tensor, ptr_x: torch.Tensor, r: float,
                  K: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """torch_cluster.radius(x=pos_x, y=pos_q, r, batch_x, batch_y, max_num_neighbors=K) for batches given as CSR pointers
    (agent_decoder.py:710-711 and the other call sites): per query the first K points of its batch in ascending index with
    d^2 < r^2, as a (Nq, K) index table padded with -1 and the per-query counts"""
    dev = pos_q.device
    ops = _ops(dev)
    nq, nx = pos_q.shape[0], pos_x.shape[0]
    idx = torch.full((nq, K), -1, device=dev, dtype=torch.int32)
    cnt = torch.zeros(nq, device=dev, dtype=torch.int32)
    if nq == 0 or nx == 0:
        return idx, cnt
    batch_q = torch.repeat_interleave(torch.arange(ptr_q.numel() - 1, device=dev), (ptr_q[1:] - ptr_q[:-1]).to(dev))
    i32 = lambda t: t.to(device=dev, dtype=torch.int32).contiguous()
    ar = torch.arange(nq, device=dev, dtype=torch.int32)
    c0, c1 = i32(ptr_x.to(dev)[batch_q]), i32(ptr_x.to(dev)[batch_q + 1])
    cap = nq * K
    off = torch.zeros(nq, device=dev, dtype=torch.int32)
    src = torch.zeros(cap, device=dev, dtype=torch.int32)
    raw = torch.empty(cap, 4, device=dev)
    total = torch.zeros(1, device=dev, dtype=torch.int32)
    zq, zx = torch.zeros(nq, device=dev), torch.zeros(nx, device=dev)
    a = _lib.RadiusEdges()
    P = _lib.ptr
    pq, px = _f32(pos_q), _f32(pos_x)
    a.n_q, a.q_node, a.q_pt, a.q_c0, a.q_c1 = nq, P(ar), P(ar), P(c0), P(c1)
    a.p_pos, a.p_head, a.c_pos, a.c_head = P(pq), P(zq), P(px), P(zx)
    a.radius, a.K = float(r), int(K)
    e = _lib.EdgeBuf()
    e.off, e.cnt, e.src, e.raw, e.total, e.cap = P(off), P(cnt), P(src), P(raw), P(total), cap
    _lib.check(ops.lib.infgen_radius_edges(C.byref(a), C.byref(e), ops.stream), 'infgen_radius_edges')
    pos = torch.arange(K, device=dev, dtype=torch.int32)[None, :]
    take = pos < cnt[:, None]
    gather = (off[:, None] + pos).clamp_(0, cap - 1).long()
    idx = torch.where(take, src[gather], idx)
    return idx, cnt


@radius_firstk.register_fake
def _(pos_q, pos_x, ptr_q, ptr_x, r, K):
    return pos_q.new_empty(pos_q.shape[0], K, dtype=torch.int32), pos_q.new_empty(pos_q.shape[0], dtype=torch.int32)


@torch.library.custom_op('infgen_hip::attn_layer', mutates_args=())
def attn_layer(x_dst: torch.Tensor, pack: torch.Tensor, off: torch.Tensor, cnt: torch.Tensor, src: torch.Tensor,
               rhat: Optional[torch.Tensor] = None, x_src: Optional[torch.Tensor] = None) -> torch.Tensor:
    """AttentionLayer.forward (reference infgen/modules/layers.py:61-113) on edges in CSR form by destination (``off`` / ``cnt``
    per destination row, ``src`` per edge indexes the source rows: of ``x_src`` for a bipartite layer, else of ``x_dst``);
    ``rhat`` (E, 128): normalised relative-position embedding of every edge (``fourier_embed(..., normalize=True)``)"""
    ops = _ops(x_dst.device)
    x = _f32(x_dst).clone()
    if x.shape[0]:
        ops.attention_layer(x, _f32(pack), off.int().contiguous(), cnt.int().contiguous(), src.int().contiguous(),
                            None if rhat is None else _f32(rhat), x_src=None if x_src is None else _f32(x_src))
    return x


@attn_layer.register_fake
def _(x_dst, pack, off, cnt, src, rhat=None, x_src=None):
    return torch.empty_like(x_dst, dtype=torch.float32)


@torch.library.custom_op('infgen_hip::token_state_head', mutates_args=())
def token_state_head(x: torch.Tensor, tok_pack: torch.Tensor, st_pack: torch.Tensor, token_size: int,
                     want_logits: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """token_predict_head / state_predict_head with the greedy arg-max (agent_decoder.py:2161-2167): next token (N,), next state
    (N,) and, on request, the (N, token_size) logits (else an empty tensor)"""
    ops = _ops(x.device)
    n = x.shape[0]
    tok = torch.zeros(n, device=x.device, dtype=torch.int32)
    st = torch.zeros(n, device=x.device, dtype=torch.int32)
    lg = torch.empty(n if want_logits else 0, token_size, device=x.device)
    if n:
        _lib.check(ops.lib.infgen_heads(_lib.ptr(_f32(x)), n, _lib.ptr(_f32(tok_pack)), _lib.ptr(_f32(st_pack)), int(token_size),
                                        _lib.ptr(lg) if want_logits else None, _lib.ptr(tok), _lib.ptr(st), ops.stream),
                   'infgen_heads')
    return tok, st, lg


@token_state_head.register_fake
def _(x, tok_pack, st_pack, token_size, want_logits=False):
    n = x.shape[0]
    return (x.new_empty(n, dtype=torch.int32), x.new_empty(n, dtype=torch.int32),
            x.new_empty(n if want_logits else 0, token_size, dtype=torch.float32))


@torch.library.custom_op('infgen_hip::mlp_layer', mutates_args=())
def mlp_layer(x: torch.Tensor, pack: torch.Tensor, n_out: int) -> torch.Tensor:
    """MLPLayer.forward (layers.py:195-215): Linear - LayerNorm - ReLU - Linear"""
    ops = _ops(x.device)
    if x.shape[0] == 0:
        return x.new_empty(0, n_out, dtype=torch.float32)
    return ops.mlp_layer(_f32(x), _f32(pack), x.shape[1], int(n_out))


@mlp_layer.register_fake
def _(x, pack, n_out):
    return x.new_empty(x.shape[0], n_out, dtype=torch.float32)


@torch.library.custom_op('infgen_hip::mlp_embedding', mutates_args=())
def mlp_embedding(x: torch.Tensor, pack: torch.Tensor) -> torch.Tensor:
    """MLPEmbedding.forward (layers.py:163-192)"""
    ops = _ops(x.device)
    if x.shape[0] == 0:
        return x.new_empty(0, D, dtype=torch.float32)
    return ops.mlp_embedding(_f32(x), _f32(pack), x.shape[1])


@mlp_embedding.register_fake
def _(x, pack):
    return x.new_empty(x.shape[0], D, dtype=torch.float32)


@torch.library.custom_op('infgen_hip::integrate_tokenise', mutates_args=())
def integrate_tokenise(token: torch.Tensor, state: torch.Tensor, agent_type: torch.Tensor, pos: torch.Tensor, head: torch.Tensor,
                       n_agents: torch.Tensor, ego: torch.Tensor, vocab: torch.Tensor, grid_xy: torch.Tensor
                       ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """token -> trajectory -> next pose -> grid cell for dense batches (reference agent_decoder.py:2175-2239,
    attr_tokenizer.py:77-89): ``token`` / ``state`` / ``agent_type`` (S, A) int, ``pos`` (S, A, 2), ``head`` (S, A) the current
    pose, ``n_agents`` / ``ego`` (S,), ``vocab`` (3, token_size, 6, 4, 2), ``grid_xy`` (G, 2).  ``state`` is the state head's
    class index (2 -> exit; the ego is forced valid).  Returns the next pose ``pos'`` (S, A, 2), ``head'`` (S, A) (zeros for
    invalid rows), the five intermediate poses ``pred_traj`` (S, A, 5, 2) / ``pred_head`` (S, A, 5), the cell of the new position in
    the ego's new frame ``grid`` (S, A) (-1 invalid) and the stored ``state'`` (S, A).  One launch of ``infgen_integrate``."""
    dev = pos.device
    ops = _ops(dev)
    S, A = token.shape
    A_cap = max(32, (A + 31) // 32 * 32)
    if A_cap > ops.lib.infgen_layout_query(_lib.Q_MAX_AGENTS):
        raise _lib.InfgenHipError('integrate_tokenise: more rows per scene than the layout holds')
    i32 = lambda *shape: torch.zeros(*shape, device=dev, dtype=torch.int32)
    T = 3
    P = torch.zeros(S, T, A_cap, 2, device=dev); H = torch.zeros(S, T, A_cap, device=dev)
    ST, TK, GR = i32(S, T, A_cap), i32(S, T, A_cap), i32(S, T, A_cap)
    IM = torch.ones(S, T, A_cap, device=dev, dtype=torch.uint8); CF = torch.ones_like(IM); TM = torch.ones_like(IM)
    P[:, 1, :A] = pos.float(); H[:, 1, :A] = head.float()
    ty, nt, ns = i32(S, A_cap), i32(S, A_cap), i32(S, A_cap)
    ty[:, :A] = agent_type.int(); nt[:, :A] = token.int(); ns[:, :A] = state.int()
    bos = i32(S, A_cap)
    na, av = n_agents.to(dev).int().contiguous(), ego.to(dev).int().contiguous()
    traj, phead, pstate = torch.zeros(S, A_cap, 5, 2, device=dev), torch.zeros(S, A_cap, 5, device=dev), torch.zeros(S, A_cap, 5, device=dev)
    voc, gxy = _f32(vocab), _f32(grid_xy)
    c = _lib.Rollout()
    Pp = _lib.ptr
    c.S, c.A_cap, c.T, c.M_cap, c.W, c.ring, c.R = S, A_cap, T, 32, 1, 2, 5
    c.token_size, c.grid_size, c.num_layers = int(voc.shape[1]), int(gxy.shape[0]), 1
    c.n_agents, c.n_map, c.av_index = Pp(na), Pp(i32(S)), Pp(av)
    c.pos, c.head, c.state, c.token, c.grid = Pp(P), Pp(H), Pp(ST), Pp(TK), Pp(GR)
    c.tmask, c.imask, c.catflag, c.type, c.bos = Pp(TM), Pp(IM), Pp(CF), Pp(ty), Pp(bos)
    c.next_token, c.next_state = Pp(nt.view(-1)), Pp(ns.view(-1))
    c.vocab, c.grid_xy = Pp(voc), Pp(gxy)
    c.pred_traj, c.pred_head, c.pred_state = Pp(traj), Pp(phead), Pp(pstate)
    if S and A:
        _lib.check(ops.lib.infgen_integrate(C.byref(c), 0, ops.stream), 'infgen_integrate')
    return (P[:, 2, :A].contiguous(), H[:, 2, :A].contiguous(), traj[:, :A].contiguous(), phead[:, :A].contiguous(),
            GR[:, 2, :A].contiguous(), ST[:, 2, :A].contiguous())


@integrate_tokenise.register_fake
def _(token, state, agent_type, pos, head, n_agents, ego, vocab, grid_xy):
    S, A = token.shape
    f = lambda *shape: pos.new_empty(*shape, dtype=torch.float32)
    i = lambda *shape: pos.new_empty(*shape, dtype=torch.int32)
    return f(S, A, 2), f(S, A), f(S, A, 5, 2), f(S, A, 5), i(S, A), i(S, A)


@torch.library.custom_op('infgen_hip::decode_step',
                         mutates_args=('pos', 'head', 'state', 'token', 'grid', 'x', 'next_token', 'next_state'))
def decode_step(ctx: torch.Tensor, t: int, pos: torch.Tensor, head: torch.Tensor, state: torch.Tensor, token: torch.Tensor,
                grid: torch.Tensor, x: torch.Tensor, next_token: torch.Tensor, next_state: torch.Tensor
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    """one decode step over a persistent state block (reference agent_decoder.py:1740-2301 without the
    insertion sub-loop): edge sets of column 1 + t, the 18 sublayers, heads, token -> pose -> grid cell, raw feature of the new
    column - ``infgen_decode_step``.  ``ctx``: the bytes of the ``InfgenRollout`` block (include/infgen_hip.h) as a uint8 CPU
    tensor (``RolloutEngine.ctx_tensor()``); the arrays the block points to that a step writes are passed (and declared
    mutated) so that a tracer sees the data flow: ``pos`` / ``head`` / ``state`` / ``token`` / ``grid`` [S][T][A_cap], the residual
    stream ``x`` [rows][128] and the heads' outputs ``next_token`` / ``next_state`` [rows], copies of which are returned."""
    ops = _ops(pos.device)
    blk = _lib.Rollout.from_buffer_copy(ctx.numpy().tobytes())
    for name, ten in (('pos', pos), ('head', head), ('state', state), ('token', token), ('grid', grid), ('X', x),
                      ('next_token', next_token), ('next_state', next_state)):
        if int(getattr(blk, name) or 0) != ten.data_ptr():
            raise _lib.InfgenHipError(f'decode_step: `{name}` is not the array the state block points to')
    _lib.check(ops.lib.infgen_decode_step(C.byref(blk), int(t), ops.stream), 'infgen_decode_step')
    return next_token.clone(), next_state.clone()


@decode_step.register_fake
def _(ctx, t, pos, head, state, token, grid, x, next_token, next_state):
    return torch.empty_like(next_token), torch.empty_like(next_state)
