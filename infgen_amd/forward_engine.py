"""Teacher-forced ``forward`` on one MI355X: every token column of a batch of scenes at once.

Reference: ``InfGenDecoder.forward`` = ``InfGenMapDecoder.forward`` + ``InfGenAgentDecoder.forward``
(infgen/modules/infgen_decoder.py:114-121, map_decoder.py:70-130, agent_decoder.py:1104-1603) - the open-loop validation /
training forward: motion stage (6 x temporal / map / agent sublayers over all (agent + seed) x column nodes, token and state
heads), coarse stage of the seed rows (3 x occupancy / map / agent sublayers from the raw features, seed heads), refine stage
(candidate rows re-featured as "entering with the ego's heading", motion layers 0..2 on their 10 m neighbourhood).

All arithmetic runs in libinfgen_hip.so: ``infgen_radius_edges`` builds the seven edge sets and the map-token graph on the
device, the Fourier embeddings / attention sublayers / MLP heads are the rollout's operators (``engine.Ops``).  Host code
here marshals inputs (padding with the seed rows, query lists of the edge builder), draws the reference's ``randperm``
selections from torch's CPU generator, and does the index bookkeeping of the evaluation masks / ground-truth gathers
(agent_decoder.py:1387-1540: python loops over integer arrays in the reference as well).

Node layout: scene-contiguous rows - scene b owns rows [o_b, o_b + A_b + 10), its agents first, then its ten seed rows (which
sit at the scene's ego) - and step-major nodes ``t * N + row``: every (scene, column) group is one contiguous candidate range of
the edge builder.  ``perm`` maps these rows to the reference's order (all agents, then all seed rows).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Mapping

import numpy as np
import torch

from . import _lib, packing
from .engine import Ops, PackedWeights, D, SEED_TYPE, INVALID_SHAPE
from .synth import INVALID, ENTER, EXIT

NS = 10                 # num_seed_feature (agent_decoder.py:292)


class _Edges:
    """one CSR edge buffer over all nodes (device) + the zone bookkeeping of sets that share it"""

    def __init__(self, dev, n_nodes: int, cap: int):
        cap = max(int(cap), 32)
        i32 = lambda n: torch.zeros(n, device=dev, dtype=torch.int32)
        self.off, self.cnt, self.src, self.total = i32(n_nodes), i32(n_nodes), i32(cap), i32(1)
        self.raw = torch.zeros(cap, 4, device=dev)
        self.rhat = torch.zeros(cap, D, device=dev)
        self.cap = cap

    def struct(self, base: int = 0, cap: int = None, total=None, off=None, cnt=None):
        b = _lib.EdgeBuf()
        cap = self.cap - base if cap is None else cap
        b.off, b.cnt = _lib.ptr(self.off if off is None else off), _lib.ptr(self.cnt if cnt is None else cnt)
        b.src, b.raw, b.rhat = self.src.data_ptr() + 4 * base, self.raw.data_ptr() + 16 * base, self.rhat.data_ptr() + 4 * D * base
        b.total, b.cap = _lib.ptr(self.total if total is None else total), cap
        return b


class ForwardEngine:
    def __init__(self, weights: PackedWeights, batch: Mapping, vocab: Mapping[str, np.ndarray], map_vocab: np.ndarray,
                 grid: np.ndarray):
        self.w, self.cfg, self.device = weights, weights.cfg, weights.device
        if getattr(weights, 'operand_bits', 11) != 11:
            raise ValueError('the teacher-forced forward runs in fp32 arithmetic: it takes default packs (operand_bits=11), not the '
                             'bf16-operand packs of the rollout\'s reduced mode')
        self.ops = Ops(self.device)
        self.lib = self.ops.lib
        self.batch = batch
        dev = self.device
        t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a if dt is None else np.asarray(a).astype(dt))).to(dev)
        self.vocab = t(np.stack([vocab[k] for k in ('veh', 'ped', 'cyc')]), np.float32)
        self._map_vocab = t(np.asarray(map_vocab, np.float32).reshape(map_vocab.shape[0], -1))
        self.grid_xy = t(grid, np.float32)
        self._tables_key = PackedWeights.tables_key(np.stack([vocab[k] for k in ('veh', 'ped', 'cyc')]).astype(np.float32),
                                                    np.asarray(grid, dtype=np.float32),
                                                    np.asarray(map_vocab, dtype=np.float32).reshape(map_vocab.shape[0], -1))
        self.G = int(grid.shape[0])
        sd, ap = weights.sd, weights.ap
        if not hasattr(weights, 'fwd_heads'):
            dv = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
            weights.fwd_heads = {k: dv(packing.pack_mlp_layer(sd, f'{ap}.{k}'))
                                 for k in ('grid_index_head', 'grid_agent_occ_head', 'grid_pt_occ_head')}
        self._setup()

    # ------------------------------------------------------------------ host marshalling
    def _setup(self):
        ag, pt, cfg = self.batch['agent'], self.batch['pt_token'], self.cfg
        f32, i64 = np.float32, np.int64
        pos = np.asarray(ag['token_pos'], f32)
        head = np.asarray(ag['token_heading'], f32)
        state = np.asarray(ag['state_idx']).astype(i64)
        self.A, self.T = A, T = pos.shape[0], pos.shape[1]
        ptr = np.asarray(ag['ptr']).astype(i64)
        self.B = B = len(ptr) - 1
        av = np.asarray(ag['av_index']).astype(i64).reshape(-1)
        self.S = S = B * NS
        self.N = N = A + S
        bsz = np.diff(ptr)
        o = ptr[:-1] + NS * np.arange(B)                                  # first row of every scene block
        # perm[row'] = reference row;  rowp[reference row] = row'
        perm = np.concatenate([np.concatenate([np.arange(ptr[b], ptr[b + 1]), A + b * NS + np.arange(NS)]) for b in range(B)])
        rowp = np.empty(N, i64)
        rowp[perm] = np.arange(N)
        graph_of = np.repeat(np.arange(B), bsz)
        seed_of = np.repeat(av, NS)
        pad = lambda x: np.concatenate([x, x[seed_of]])[perm]              # (N, T, ...) in scene-contiguous order
        self.h = h = dict(pos=pos, head=head, state=state, ptr=ptr, av=av, bsz=bsz, o=o, perm=perm, rowp=rowp,
                          graph_of=graph_of, seed_of=seed_of,
                          token=np.asarray(ag['token_idx']).astype(i64), atype=np.asarray(ag['type']).astype(i64),
                          gidx=np.asarray(ag['grid_token_idx']).astype(i64), sort_idx=np.asarray(ag['sort_indices']).astype(i64),
                          shape=np.asarray(ag['shape'], f32)[:, cfg.num_historical_steps - 1].copy(),
                          mask=np.asarray(ag['raw_agent_valid_mask']).astype(bool),
                          pt_ptr=np.asarray(pt['ptr']).astype(i64))
        inv = state == INVALID
        is_bos, is_eos = state == ENTER, state == EXIT
        bos = np.where(is_bos.any(1), is_bos.argmax(1), 0)
        eos = np.where(is_eos.any(1), is_eos.argmax(1), T - 1)
        cols = np.arange(T)[None, :]
        motion = (cols > bos[:, None]) & (cols <= eos[:, None])
        tmask = np.ones((A, T), bool)
        tmask[motion] = h['mask'][motion]
        imask = h['mask'].copy()
        imask[is_bos] = True
        win = int(cfg.time_span // cfg.shift)
        start = np.clip(bos - win + 1, 0, None)
        hist = tmask & (cols >= bos[:, None]) & (cols >= start[:, None])
        h.update(inv=inv, is_bos=is_bos, is_eos=is_eos, imask=imask, hist=hist, win=win)

        dev = self.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        sm = lambda x: np.ascontiguousarray(np.swapaxes(x, 0, 1))        # (rows, T, ...) -> step-major (T, rows, ...)
        is_agent = np.concatenate([np.ones(A, bool), np.zeros(S, bool)])[perm]
        im_p = np.concatenate([imask, np.ones((S, T), bool)])[perm]
        # node-indexed (step-major) point arrays
        self.n_pos, self.n_head = t(sm(pad(pos))), t(sm(pad(head)))
        self.n_inv = t(sm(pad(inv)).astype(np.uint8))
        self.n_ok = t(sm(im_p & is_agent[:, None]).astype(np.uint8))      # candidate filter: unmasked agent
        # agent-major arrays of the temporal builder and of the raw features
        self.a_pos, self.a_head = t(pos), t(head)
        self.a_state = t(state.astype(np.int32))
        self.a_inv, self.a_hist = t(inv.astype(np.uint8)), t(hist.astype(np.uint8))
        node_of = (np.arange(T)[None, :] * N + rowp[:A, None]).astype(np.int32)          # (A, T): node of agent a at column t
        self.a_node = t(node_of)
        self.map_pos = t(np.asarray(pt['position'], f32)[:, :2])
        self.map_orient = t(np.asarray(pt['orientation'], f32))
        self.M = int(self.map_pos.shape[0])
        h['node_of'] = node_of
        h['is_agent'] = is_agent

    def _queries(self, **cols):
        """int32 device arrays of one query list"""
        return {k: torch.from_numpy(np.ascontiguousarray(np.asarray(v).astype(np.int32))).to(self.device) for k, v in cols.items()}

    def _build(self, e: _Edges, q, p, c, radius, K, gap_rule=0, index_diff=0, self_drop=False, pair=None, base=0, cap=None,
               total=None, off=None, cnt=None):
        """one infgen_radius_edges launch; p / c = (pos, head, inv[, ok, src]) device arrays of queries / candidates"""
        if q['node'].numel() == 0:
            return
        r = _lib.RadiusEdges()
        P = _lib.ptr
        r.n_q = int(q['node'].numel())
        r.q_node, r.q_pt, r.q_c0, r.q_c1 = P(q['node']), P(q['pt']), P(q['c0']), P(q['c1'])
        r.q_self = P(q['pt']) if self_drop else None
        r.q_pair_off = P(q['pair']) if pair is not None else None
        r.pair_ok = P(pair)
        r.p_pos, r.p_head, r.p_inv = P(p[0]), P(p[1]), P(p[2])
        r.c_pos, r.c_head, r.c_inv = P(c[0]), P(c[1]), P(c[2])
        r.c_ok = P(c[3]) if len(c) > 3 else None
        r.c_src = P(c[4]) if len(c) > 4 else None
        r.radius, r.K, r.gap_rule, r.index_diff, r.e_base = float(radius), int(K), int(gap_rule), int(index_diff), int(base)
        eb = e.struct(base, cap, total, off, cnt)
        _lib.check(self.lib.infgen_radius_edges(C.byref(r), C.byref(eb), self.ops.stream), 'infgen_radius_edges')

    # ------------------------------------------------------------------ map encoder (map_decoder.py:70-130)
    def map_encoder(self, tabs):
        w, cfg, ops, dev = self.w, self.cfg, self.ops, self.device
        pt, h, M = self.batch['pt_token'], self.h, self.M
        lt = lambda a: torch.from_numpy(np.asarray(a).astype(np.int64)).to(dev)
        e2 = np.asarray(self.batch['pt_token__to__map_polygon']['edge_index']).astype(np.int64)
        light = np.asarray(self.batch['map_polygon']['light_type']).astype(np.int64)[e2[1]]
        x = tabs['map_tab'][lt(pt['token_idx'])]
        cat = (w.type_pt_emb[lt(pt['type'])] + w.polygon_type_emb[lt(pt['pl_type'])]) + w.light_pl_emb[lt(light)]
        x_pt = (x + cat).contiguous()
        scene_of = np.repeat(np.arange(self.B), np.diff(h['pt_ptr']))
        q = self._queries(node=np.arange(M), pt=np.arange(M), c0=h['pt_ptr'][scene_of], c1=h['pt_ptr'][scene_of + 1])
        zero_inv = torch.zeros(M, device=dev, dtype=torch.uint8)
        cap = int(sum(int(m) * min(int(m) - 1, 100) for m in np.diff(h['pt_ptr'])))
        e = _Edges(dev, M, cap)
        arr = (self.map_pos, self.map_orient, zero_inv)
        self._build(e, q, arr, arr, cfg.pl2pl_radius, 100 + 1, self_drop=True)
        ops.fourier(e.raw, 3, w.four_pt, e.rhat, count_dev=e.total, rows=e.cap, normalize=True)
        for i in range(cfg.num_map_layers):
            ops.attention_layer(x_pt, w.attn_pt[i], e.off, e.cnt, e.src, e.rhat)
        return x_pt

    # ------------------------------------------------------------------ raw features (agent_decoder.py:332-509)
    def _features(self, tabs, state, tok_emb, types_cat, gap_mask=None, head=None):
        """fusion_emb([token | x_a | state | grid]) for the (A, T) agent-major rows -> (A * T, D)"""
        w, ops, dev, h = self.w, self.ops, self.device, self.h
        A, T = self.A, self.T
        raw2 = torch.empty(A * T, 4, device=dev)
        st = torch.from_numpy(state.astype(np.int32)).to(dev)
        gm = torch.from_numpy(gap_mask.astype(np.uint8)).to(dev) if gap_mask is not None else None
        hd = self.a_head if head is None else head
        _lib.check(self.lib.infgen_motion_features(_lib.ptr(self.a_pos), _lib.ptr(hd), _lib.ptr(st), _lib.ptr(gm), A, T,
                                                   _lib.ptr(raw2), ops.stream), 'infgen_motion_features')
        fus = torch.empty(A * T, 4 * D, device=dev)
        fus[:, :D] = tok_emb
        fus[:, 2 * D:3 * D] = w.state_a_emb[st.reshape(-1).long()]
        fus[:, 3 * D:] = tabs['grid_tab'][torch.from_numpy(h['gidx'].reshape(-1)).to(dev)]
        ops.fourier(raw2, 2, w.four_xa, fus[:, D:2 * D], cat=types_cat)
        return ops.mlp_embedding(fus, w.fusion, 4 * D)

    # ------------------------------------------------------------------ the forward
    @torch.no_grad()
    def run(self) -> Dict[str, torch.Tensor]:
        w, cfg, ops, dev, h = self.w, self.cfg, self.ops, self.device, self.h
        A, T, B, S, N, G = self.A, self.T, self.B, self.S, self.N, self.G
        nn = T * N
        lt = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a).astype(np.int64))).to(dev)
        tabs = w.tables(ops, self.vocab, self.grid_xy, self._map_vocab, self._tables_key)
        x_pt = self.map_encoder(tabs)
        out = {'x_pt': x_pt, 'ego_pos': torch.from_numpy(h['pos'][h['av']]).to(dev)}

        # ---- raw features: agents (state-dependent type / shape embedding, :376-380), seed rows = the constant seed feature
        shp = ops.mlp_embedding(torch.from_numpy(h['shape']).to(dev), w.shape_emb, 3)
        cat_agent = w.type_a_emb[lt(h['atype'])] + shp                                     # (A, D)
        inv_d = lt(h['inv'].reshape(-1)).bool()
        cat = torch.where(inv_d[:, None], tabs['cat_seed'][None, :], cat_agent.repeat_interleave(T, dim=0)).contiguous()
        tok = tabs['tok_tab'][lt(np.repeat(h['atype'], T)), lt(h['token'].reshape(-1))]
        raw_a = self._features(tabs, h['state'], tok, cat)                                 # (A * T, D) agent-major
        a_node = self.a_node.reshape(-1).long()
        seed_nodes_np = (np.arange(T)[:, None] * N + h['rowp'][A:][None, :]).reshape(-1)   # (t, seed k) order
        seed_nodes = lt(seed_nodes_np)
        X0 = torch.empty(nn, D, device=dev)
        X0[a_node] = raw_a
        X0[seed_nodes] = tabs['f_seed'][0]

        # ---- edge sets
        cols = np.arange(T)
        node_of = h['node_of']
        # temporal: queries = (a, j) with hist; candidates (a, i), i in [j - win, j)
        qa, qj = np.nonzero(h['hist'])
        q_t = self._queries(node=node_of[qa, qj], pt=qa * T + qj, c0=qa * T + np.maximum(qj - h['win'], 0), c1=qa * T + qj)
        e_t = _Edges(dev, nn, len(qa) * h['win'])
        a_arr = (self.a_pos.reshape(-1, 2), self.a_head.reshape(-1), self.a_inv.reshape(-1), self.a_hist.reshape(-1),
                 self.a_node.reshape(-1))
        self._build(e_t, q_t, a_arr, a_arr, 1e30, h['win'], gap_rule=1, index_diff=1)
        ops.fourier(e_t.raw, 4, w.four_t, e_t.rhat, count_dev=e_t.total, rows=e_t.cap, normalize=True)

        # group ranges of every node
        row_scene = np.concatenate([np.full(int(h['bsz'][b]) + NS, b) for b in range(B)])   # scene of row'
        g0 = (cols[:, None] * N + h['o'][row_scene][None, :]).reshape(-1)                    # first node of the node's group
        g1 = g0 + (h['bsz'][row_scene] + NS)[None, :].repeat(T, 0).reshape(-1)
        ga1 = g0 + h['bsz'][row_scene][None, :].repeat(T, 0).reshape(-1)                     # ... agents only
        m0 = h['pt_ptr'][row_scene][None, :].repeat(T, 0).reshape(-1)
        m1 = h['pt_ptr'][row_scene + 1][None, :].repeat(T, 0).reshape(-1)
        is_agent_n = np.tile(h['is_agent'], T)
        ok_n = self.n_ok.cpu().numpy().reshape(-1).astype(bool)
        nodes = np.arange(nn)
        qd = nodes[ok_n]                                                                     # unmasked agent nodes
        qs = seed_nodes_np                                                                   # seed nodes in (t, k) order
        n_arr = (self.n_pos.reshape(-1, 2), self.n_head.reshape(-1), self.n_inv.reshape(-1), self.n_ok.reshape(-1))
        map_arr = (self.map_pos, self.map_orient, torch.zeros(self.M, device=dev, dtype=torch.uint8))
        # seq_mask of _build_seq (:994-1054) restricted to the scene's own agents: pair[b][t][s][local agent]
        amax = int(h['bsz'].max())
        pair = np.ones((B, T, NS, amax), np.uint8)
        for b in range(B):
            bs = h['sort_idx'][h['ptr'][b]:h['ptr'][b + 1]]
            lo = int(h['ptr'][b])
            for s in range(min(NS, bs.shape[0])):
                for t_ in range(T):
                    cleared = bs[s:, t_] - lo                       # the reference clears scene-LOCAL indices as global columns
                    cleared = cleared[(cleared >= 0) & (cleared < h['bsz'][b])]
                    pair[b, t_, s, cleared] = 0
            pair[b, :, :, int(h['av'][b]) - lo] = 1
        pair_d = torch.from_numpy(pair).to(dev)
        k_seed = np.tile(np.arange(S), T)                           # seed index of qs entries
        t_seed = np.repeat(cols, S)
        pair_off = ((k_seed // NS * T + t_seed) * NS + k_seed % NS) * amax

        cap_a = int(sum(T * int(a_) * max(int(a_) - 1, 0) for a_ in h['bsz']))
        cap_s = int(sum(T * NS * min(int(a_), 300) for a_ in h['bsz']))
        e_a = _Edges(dev, nn, cap_a + cap_s)                        # zones: agent <-> agent | agent -> seed
        self._build(e_a, self._queries(node=qd, pt=qd, c0=g0[qd], c1=g1[qd]), n_arr, n_arr, cfg.a2a_radius, 300 + 1,
                    gap_rule=1, self_drop=True, cap=cap_a)
        s_off, s_cnt, s_tot = (torch.zeros(nn, device=dev, dtype=torch.int32), torch.zeros(nn, device=dev, dtype=torch.int32),
                               torch.zeros(1, device=dev, dtype=torch.int32))
        q_s = self._queries(node=qs, pt=qs, c0=g0[qs], c1=g1[qs], pair=pair_off)
        self._build(e_a, q_s, n_arr, n_arr, cfg.pl2seed_radius, 300, pair=pair_d, base=cap_a, cap=cap_s, total=s_tot,
                    off=s_off, cnt=s_cnt)
        ops.fourier(e_a.raw, 3, w.four_a, e_a.rhat, count_dev=e_a.total, rows=cap_a, normalize=True)
        ops.fourier(e_a.raw[cap_a:], 3, w.four_a2sa, e_a.rhat[cap_a:], count_dev=s_tot, rows=cap_s, normalize=True)

        mb = np.diff(h['pt_ptr'])
        cap_m = 5 * len(qd)
        cap_q = int(sum(T * NS * min(int(m_), 2048) for m_ in mb))
        e_m = _Edges(dev, nn, cap_m + cap_q)                        # zones: map -> agent | map -> seed
        self._build(e_m, self._queries(node=qd, pt=qd, c0=m0[qd], c1=m1[qd]), n_arr, map_arr, cfg.pl2a_radius, 5, gap_rule=2,
                    cap=cap_m)
        q_off, q_cnt, q_tot = (torch.zeros(nn, device=dev, dtype=torch.int32), torch.zeros(nn, device=dev, dtype=torch.int32),
                               torch.zeros(1, device=dev, dtype=torch.int32))
        self._build(e_m, self._queries(node=qs, pt=qs, c0=m0[qs], c1=m1[qs]), n_arr, map_arr, cfg.pl2seed_radius, 2048,
                    base=cap_m, cap=cap_q, total=q_tot, off=q_off, cnt=q_cnt)
        ops.fourier(e_m.raw, 3, w.four_m, e_m.rhat, count_dev=e_m.total, rows=cap_m, normalize=True)
        ops.fourier(e_m.raw[cap_m:], 3, w.four_pt2sa, e_m.rhat[cap_m:], count_dev=q_tot, rows=cap_q, normalize=True)
        totals = [int(x.item()) for x in (e_t.total, e_a.total, s_tot, e_m.total, q_tot)]     # one host sync: overflow check
        assert totals[0] <= e_t.cap and totals[1] <= cap_a and totals[2] <= cap_s and totals[3] <= cap_m and totals[4] <= cap_q
        self.edge_counts = dict(t=totals[0], a=totals[1], a2sa=totals[2], m=totals[3], m2sa=totals[4])
        # the motion layers run on agent and seed edges alike (agent_decoder.py:679 / :756 return the total, :1209-1211)
        aa_off, aa_cnt = e_a.off + s_off, e_a.cnt + s_cnt
        mm_off, mm_cnt = e_m.off + q_off, e_m.cnt + q_cnt

        # ---- motion stage (:1201-1227)
        X = X0.clone()
        for i in range(cfg.num_agent_layers):
            ops.attention_layer(X, w.attn_t[i], e_t.off, e_t.cnt, e_t.src, e_t.rhat)
            ops.attention_layer(X, w.attn_m[i], mm_off, mm_cnt, e_m.src, e_m.rhat, x_src=x_pt)
            ops.attention_layer(X, w.attn_a[i], aa_off, aa_cnt, e_a.src, e_a.rhat)
        ref_nodes = lt((np.arange(T)[None, :] * N + h['rowp'][:, None]))                  # (N, T): node of reference row r
        out['x_a'] = X[ref_nodes.reshape(-1)].view(N, T, D)
        fe = X[a_node]                                                                      # (A * T, D)
        out['next_token_prob'] = ops.mlp_layer(fe, w.tok_head, D, cfg.token_size).view(A, T, -1)
        out['next_token_idx'] = torch.topk(torch.softmax(out['next_token_prob'], dim=-1), k=10, dim=-1)[1]
        out['next_state_prob'] = ops.mlp_layer(fe, self._state_head(), D, 3).view(A, T, 3)
        out['next_state_idx'] = out['next_state_prob'].softmax(-1).argmax(-1, keepdim=True)

        # ---- occupancy ground truth (:1056-1102) from the agent -> seed edges
        # edge list in the reference's order: seed nodes in (t, scene, s) order, sources ascending
        eo = torch.cat([torch.arange(int(o_), int(o_) + int(c_), device=dev) for o_, c_ in
                        zip((s_off[seed_nodes] - cap_a).tolist(), s_cnt[seed_nodes].tolist())]) if totals[2] else torch.zeros(0, dtype=torch.long, device=dev)
        es_src = e_a.src[cap_a:][eo].long()
        es_dst = torch.repeat_interleave(seed_nodes, s_cnt[seed_nodes].long())
        gidx_n = torch.full((nn,), 0, device=dev, dtype=torch.long)
        gidx_n[a_node] = lt(h['gidx'].reshape(-1))
        node_row = torch.arange(nn, device=dev) % N
        seed_k = lt(h['perm'])[node_row] - A                                                # reference seed index of a seed node
        occ_a = torch.zeros(S, T, G, device=dev, dtype=torch.long)
        occ_a[seed_k[es_dst], es_dst // N, gidx_n[es_src]] = 1
        pt_cells = lt(self.batch['agent']['pt_grid_token_idx'])
        occ_m = torch.zeros(B, T, G, device=dev, dtype=torch.long)
        for b in range(B):
            c = pt_cells[:, int(h['pt_ptr'][b]):int(h['pt_ptr'][b + 1])]
            tt = torch.arange(T, device=dev)[:, None].expand_as(c)
            keep = c != -1
            occ_m[b, tt[keep], c[keep]] = 1
        out['grid_agent_occ_gt_seed'] = occ_a
        out['grid_pt_occ_gt_seed'] = occ_m.repeat_interleave(NS, dim=0)

        # ---- coarse stage (:1236-1302)
        occ_emb = ops.mlp_layer(occ_a.transpose(0, 1).reshape(-1, G).float().contiguous(), w.heads['seed_agent_occ_embed'], G, D)
        o_off, o_cnt = torch.zeros(nn, device=dev, dtype=torch.int32), torch.zeros(nn, device=dev, dtype=torch.int32)
        o_off[seed_nodes] = torch.arange(T * S, device=dev, dtype=torch.int32)              # occ row t * S + k -> seed node
        o_cnt[seed_nodes] = 1
        o_src = torch.arange(T * S, device=dev, dtype=torch.int32)
        Xs = X0.clone()
        for i in range(3):
            ops.attention_layer(Xs, w.attn_occ2sa[i], o_off, o_cnt, o_src, None, x_src=occ_emb)
            ops.attention_layer(Xs, w.attn_pt2sa[i], q_off, q_cnt, e_m.src, e_m.rhat, x_src=x_pt)
            ops.attention_layer(Xs, w.attn_a2sa[i], s_off, s_cnt, e_a.src, e_a.rhat)
        f_seed = Xs[seed_nodes].view(T, S, D).transpose(0, 1).contiguous()                  # (S, T, D)
        fs2 = f_seed.view(S * T, D)
        H = w.heads
        st_prob = ops.mlp_layer(fs2, H['seed_state_predict_head'], D, 2).view(S, T, 2)
        type_prob = ops.mlp_layer(fs2, H['seed_type_predict_head'], D, 3).view(S, T, 3)
        shape_seed = ops.mlp_layer(fs2, H['seed_shape_predict_head'], D, 3).view(S, T, 3)
        pos_prob = ops.mlp_layer(fs2, H['seed_pos_rel_token_predict_head'], D, G).view(S, T, G)
        out['raw_next_state_prob_seed'] = st_prob.clone()
        out['grid_agent_occ_seed'] = ops.mlp_layer(fs2, w.fwd_heads['grid_agent_occ_head'], D, G).view(S, T, G)
        out['grid_pt_occ_seed'] = ops.mlp_layer(fs2, w.fwd_heads['grid_pt_occ_head'], D, G).view(S, T, G)
        # grid_index_head over the seed edges' embeddings, in the reference's edge order (:1288-1295)
        eq = torch.cat([torch.arange(int(o_), int(o_) + int(c_), device=dev) for o_, c_ in
                        zip((q_off[seed_nodes] - cap_m).tolist(), q_cnt[seed_nodes].tolist())]) if totals[4] else torch.zeros(0, dtype=torch.long, device=dev)
        r_s = self._unnormalised(e_a.raw[cap_a:][eo], w.four_a2sa)
        r_q = self._unnormalised(e_m.raw[cap_m:][eq], w.four_pt2sa)
        out['neighbor_agent_grid_idx'] = ops.mlp_layer(r_s, w.fwd_heads['grid_index_head'], D, G)
        out['neighbor_pt_grid_idx'] = ops.mlp_layer(r_q, w.fwd_heads['grid_index_head'], D, G)
        eq_src = e_m.src[cap_m:][eq].long()
        eq_dst = torch.repeat_interleave(seed_nodes, q_cnt[seed_nodes].long())
        gp_n = gidx_n.clone()
        ego_nodes = lt((np.arange(T)[:, None] * N + h['rowp'][h['seed_of']][None, :]).reshape(-1))     # ego node behind every seed node
        gp_n[seed_nodes] = gidx_n[ego_nodes]
        out['neighbor_agent_grid_index_gt'] = gp_n[es_src]
        out['neighbor_pt_grid_index_gt'] = pt_cells.reshape(-1)[(eq_dst // N) * self.M + eq_src]
        ma = torch.zeros(es_src.numel(), dtype=torch.bool)
        mp = torch.zeros(eq_src.numel(), dtype=torch.bool)
        ma[torch.randperm(ma.shape[0])[:180]] = True
        mp[torch.randperm(mp.shape[0])[:600]] = True
        out['neighbor_agent_grid_index_eval_mask'], out['neighbor_pt_grid_index_eval_mask'] = ma.to(dev), mp.to(dev)

        # ---- refine stage (:1304-1385); candidate rows from torch's CPU generator like the reference on the CPU
        state, gidx, mask = h['state'], h['gidx'], h['mask']
        mask_sa = np.zeros((A, T), bool)
        for t_ in range(T):
            avail = np.nonzero((state[:, t_] != INVALID) & (gidx[:, t_] != -1))[0]
            mask_sa[avail[torch.randperm(avail.shape[0])[:B * 10].numpy()], t_] = True
        mask_sa[h['is_bos']] = True
        mask_sa[:, 0] = False
        mask_sa[h['av']] = False
        state_sa = np.full_like(state, INVALID)
        state_sa[mask_sa] = ENTER
        head_sa = h['head'].copy()
        ego_of = h['av'][h['graph_of']]
        head_sa[mask_sa] = h['head'][ego_of][mask_sa]
        head_sa_d = torch.from_numpy(head_sa).to(dev)
        tok_sa = w.no_token[0].expand(A * T, D).clone()
        tok_sa[lt(mask_sa.reshape(-1)).bool()] = w.bos_token[0]
        f_sa = self._features(tabs, state_sa, tok_sa, cat_agent.repeat_interleave(T, dim=0).contiguous(), gap_mask=mask_sa,
                              head=head_sa_d)
        keep_raw = lt(~mask_sa.reshape(-1)).bool()
        f_sa[keep_raw] = raw_a[keep_raw]
        Xr = torch.zeros(nn, D, device=dev)
        Xr[a_node] = f_sa
        # node-indexed arrays with the candidates' headings
        n_head_sa = self.n_head.clone().reshape(-1)
        n_head_sa[a_node] = head_sa_d.reshape(-1)
        sa_n = np.zeros(nn, bool)
        sa_n[node_of.reshape(-1)] = mask_sa.reshape(-1)
        ok_sa = torch.from_numpy((ok_n & ~sa_n).astype(np.uint8)).to(dev)
        qx = nodes[sa_n]
        zero_inv = torch.zeros(nn, device=dev, dtype=torch.uint8)
        p_sa = (self.n_pos.reshape(-1, 2), n_head_sa, zero_inv, ok_sa)
        e_x = _Edges(dev, nn, 8 * len(qx))
        e_y = _Edges(dev, nn, 32 * len(qx))
        self._build(e_x, self._queries(node=qx, pt=qx, c0=g0[qx], c1=ga1[qx]), p_sa, p_sa, cfg.a2sa_radius, 8)
        self._build(e_y, self._queries(node=qx, pt=qx, c0=m0[qx], c1=m1[qx]), p_sa, map_arr, cfg.pl2sa_radius, 32)
        ops.fourier(e_x.raw, 3, w.four_a, e_x.rhat, count_dev=e_x.total, rows=e_x.cap, normalize=True)
        ops.fourier(e_y.raw, 3, w.four_m, e_y.rhat, count_dev=e_y.total, rows=e_y.cap, normalize=True)
        for i in range(3):
            ops.attention_layer(Xr, w.attn_m[i], e_y.off, e_y.cnt, e_y.src, e_y.rhat, x_src=x_pt)
            ops.attention_layer(Xr, w.attn_a[i], e_x.off, e_x.cnt, e_x.src, e_x.rhat)
        fr = Xr[a_node]
        n_head = int(360.0 / cfg.angle_interval)
        out['next_head_rel_prob_seed'] = ops.mlp_layer(fr, H['seed_heading_rel_token_predict_head'], D, n_head).view(A, T, n_head)
        out['next_offset_xy_seed'] = (torch.tanh(ops.mlp_layer(fr, H['seed_offset_xy_predict_head'], D, 2)) * 2).view(A, T, 2)
        self.edge_counts.update(a2sa_refine=int(e_x.total.item()), m2sa_refine=int(e_y.total.item()))
        has_edge = ((e_x.cnt + e_y.cnt) > 0).cpu().numpy()
        mask_sa &= has_edge[node_of.reshape(-1)].reshape(A, T)                                # :1353-1356

        self._bookkeeping(out, mask_sa, st_prob, type_prob, shape_seed, pos_prob)
        return out

    def _state_head(self):
        w = self.w
        if not hasattr(w, 'st_head_packed'):
            w.st_head_packed = torch.from_numpy(np.ascontiguousarray(
                packing.pack_mlp_layer(w.sd, f'{w.ap}.state_predict_head'), dtype=np.float32)).to(self.device)
        return w.st_head_packed

    def _unnormalised(self, raw, pack):
        """Fourier embedding WITHOUT the shared affine-free LayerNorm of the attention layers (grid_index_head reads r itself)"""
        out = torch.empty(raw.shape[0], D, device=self.device)
        if raw.shape[0]:
            self.ops.fourier(raw.contiguous(), 3, pack, out, normalize=False)
        return out

    # ------------------------------------------------------------------ masks / ground truth (agent_decoder.py:1387-1540)
    def _bookkeeping(self, out, mask_sa, st_prob, type_prob, shape_seed, pos_prob):
        h, cfg, dev = self.h, self.cfg, self.device
        A, T, B, S, G = self.A, self.T, self.B, self.S, self.G
        state, mask, ptr, av = h['state'], h['mask'], h['ptr'], h['av']
        roll = lambda x, s: np.roll(x, s, axis=1)
        tok_eval = mask & roll(mask, -1) & roll(mask, 1)
        st_eval = tok_eval.copy()
        for a_, c_ in np.argwhere(h['is_bos']):
            tok_eval[a_, c_:c_ + 1] = True
            tok_eval[a_, c_ + 1:c_ + 2] = mask[a_, c_ + 2:c_ + 3]
            st_eval[a_, :c_] = False
            st_eval[a_, c_:c_ + 1] = True
            st_eval[a_, c_ + 1:c_ + 2] = mask[a_, c_ + 2:c_ + 3]
        eos_idx = np.argwhere(h['is_eos'])
        tok_eval[eos_idx[:, 0], eos_idx[:, 1]] = False
        for a_, c_ in eos_idx:
            st_eval[a_, c_ + 1:] = True
            st_eval[a_, c_:c_ + 1] = mask[a_, c_ - 1:c_]
        tok_eval[:, 0] = mask[:, 0] & mask[:, 1]
        st_eval[:, 0] = mask[:, 0] & mask[:, 1]
        tok_eval[:, -1] = False
        st_eval[:, -1] = False
        token_gt = roll(h['token'], -1)
        if (token_gt[tok_eval] < 0).any():
            raise RuntimeError('Found invalid motion index.')                               # :1423-1424
        seed_st_eval = np.ones((S, T), bool)
        seed_st_eval[:, 0] = False
        pred, gt = [], []
        for b in range(B):
            bs = h['sort_idx'][ptr[b]:ptr[b + 1]]
            n_b = min(NS, bs.shape[0])
            pred.append(np.repeat((np.arange(n_b) + b * NS)[:, None], T, axis=1))
            gt.append(bs[:n_b] + ptr[b])
        pred, gt = np.concatenate(pred), np.concatenate(gt)
        n = pred.shape[0]
        rest = []
        for t_ in range(T):
            used = np.zeros(S, bool)
            used[pred[:, t_]] = True
            rest.append(np.arange(S)[~used])
        padded = np.concatenate([pred, np.stack(rest, axis=1)])
        lt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        pred_d, gt_d, padded_d = lt(pred), lt(gt), lt(padded)
        st_idx = st_prob.softmax(-1).argmax(-1, keepdim=True)
        type_idx = type_prob.softmax(-1).argmax(-1, keepdim=True)
        out['next_state_idx_seed'] = torch.gather(st_idx, 0, padded_d[..., None])
        out['next_state_prob_seed'] = torch.gather(st_prob, 0, padded_d[..., None].expand(-1, -1, 2))
        st_gt_seed = np.concatenate([np.take_along_axis(state, gt, 0), np.zeros((S - n, T), np.int64)])
        enter = st_gt_seed == ENTER
        type_gt = np.repeat(h['atype'][:, None], T, axis=1)
        shape_gt = np.repeat(h['shape'][:, None], T, axis=1)
        pos_gt = np.take_along_axis(h['gidx'], gt, 0)
        attr_eval = enter[:n].copy()
        attr_eval[:, 0] = False
        attr_eval[pos_gt == G // 2] = False
        st_eval[av] = False
        state_gt = roll(state, -1).copy()
        state_gt[state_gt == EXIT] = 2
        ag = self.batch['agent']
        f = lambda k: np.asarray(ag[k], np.float32)
        occ_eval = torch.ones(S, T, G, dtype=torch.bool, device=dev)
        occ_eval[:, 0] = False
        occ_eval[..., G // 2] = False
        tgt = pred.copy()
        tgt[~attr_eval] = -1
        shape_gt_seed = np.take_along_axis(shape_gt, gt[..., None], 0)
        if n > 0 and ((np.take_along_axis(type_gt, gt, 0)[attr_eval] == SEED_TYPE).any()
                      or (shape_gt_seed[attr_eval] == INVALID_SHAPE).all(-1).any() or (pos_gt[attr_eval] < 0).any()):
            raise ValueError('Found invalid gt values.')                                    # :1513-1516
        out.update(
            next_token_idx_gt=lt(token_gt), next_token_eval_mask=lt(tok_eval), next_state_idx_gt=lt(state_gt),
            next_state_eval_mask=lt(st_eval), next_state_idx_gt_seed=lt(enter.astype(np.int64)),
            next_type_idx_seed=torch.gather(type_idx, 0, pred_d[..., None]),
            next_type_prob_seed=torch.gather(type_prob, 0, pred_d[..., None].expand(-1, -1, 3)),
            next_type_idx_gt_seed=lt(np.take_along_axis(type_gt, gt, 0)),
            next_pos_rel_prob_seed=torch.gather(pos_prob, 0, pred_d[..., None].expand(-1, -1, G)),
            next_pos_rel_index_gt_seed=lt(pos_gt), next_pos_rel_xy_seed=None,
            next_pos_rel_xy_gt_seed=lt(np.take_along_axis(f('pos_xy') / np.float32(cfg.pl2seed_radius), gt[..., None], 0)),
            next_head_rel_index_gt_seed=lt(np.asarray(ag['heading_token_idx']).astype(np.int64)), next_head_rel_theta_seed=None,
            next_head_rel_theta_gt_seed=lt(f('heading_theta') / np.float32(math.pi)),
            next_offset_xy_gt_seed=lt(f('grid_offset_xy')),
            next_shape_seed=torch.gather(shape_seed, 0, pred_d[..., None].expand(-1, -1, 3)), next_shape_gt_seed=lt(shape_gt_seed),
            target_indices=lt(tgt), next_state_eval_mask_seed=lt(seed_st_eval), next_attr_eval_mask_seed=lt(attr_eval),
            next_head_eval_mask_seed=lt(mask_sa), grid_agent_occ_eval_mask_seed=occ_eval, grid_pt_occ_eval_mask_seed=occ_eval)
