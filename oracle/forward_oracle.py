"""CPU ORACLE — test infrastructure, not product code.

Restatement, in plain fp32 torch CPU ops, of the reference's teacher-forced ``forward``
(``InfGenDecoder.forward`` = map encoder + ``InfGenAgentDecoder.forward``, reference
infgen/modules/infgen_decoder.py:114-121, agent_decoder.py:1104-1603): every token column of a
batch of scenes at once, ten seed rows per scene appended to the agent rows, three stages -
  motion   : 6 x (temporal, map -> agent, agent <-> agent) over all (agent + seed) x column nodes, token / state heads;
  coarse   : 3 x (occupancy -> seed, map -> seed, agent -> seed) from the raw features, the seed heads;
  refine   : candidate rows re-featured as "entering with the ego's heading", motion layers 0..2 on their 10 m
             neighbourhood, heading / offset heads.
It keeps the reference's arithmetic (oracle/rollout_oracle.py's operators) but not its data flow: edges are built per
(scene, column) group instead of through flat radius calls over step-major copies of everything.

Reference quirks reproduced on purpose (each visible in the fixture):
  * ``na2a`` / ``npl2a`` are the TOTAL edge counts (agent_decoder.py:679, :756 return the length after the seed edges were
    appended), so the motion layers run on the seed edges as well (:1209-1211) - only the seed rows of ``x_a`` see it;
  * ``_build_seq`` clears ``seq_mask`` columns with scene-LOCAL ``sort_indices`` (no ``ptr`` offset, :1009): for every
    scene but the first the cleared columns belong to scene 0's agents;
  * the seed rows carry the ego's state, so ``is_bos`` of a seed row is the ego's (:547) before :554 masks them anyway;
  * grid cell -1 indexes the last row of the grid-embedding table (python negative indexing, :373, :1330).

Only ``tests/`` may import this module.  PINNED by tests/test_forward_oracle.py on tests/golden/forward_a40.npz, the
reference's own output (tests/golden/make_golden_forward.py).  The candidate rows of the refine stage and the two
neighbour-grid evaluation masks come from ``torch.randperm`` on torch's global CPU generator (:1294-1295, :1312); the oracle
draws from the same generator in the same order, so seeding it like the fixture script reproduces the selection.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from .rollout_oracle import (INVALID, ENTER, EXIT, SEED_TYPE, NUM_SEED_FEATURE, INVALID_SHAPE, MOTION_GAP, HEADING_GAP,
                             INVALID_MOTION, INVALID_HEAD, wrap_angle, angle_between, fourier_embedding, mlp_embedding,
                             mlp_layer, attention_layer, radius_first_k, map_encoder, _t)

NS = NUM_SEED_FEATURE


def build_vector(pos, head, state):
    """agent_decoder.py:426-447 (the ``==`` at :444 is a comparison: the heading is left alone)"""
    A = pos.shape[0]
    mv = torch.cat([pos.new_zeros(A, 1, 2), pos[:, 1:] - pos[:, :-1]], dim=1)
    inv = state == INVALID
    mv[inv] = INVALID_MOTION
    last_inv = (state.roll(1, 1) == INVALID) & ~inv
    last_inv[:, 0] = state[:, 0] == ENTER
    mv[last_inv] = MOTION_GAP
    last_val = (state.roll(1, 1) != INVALID) & inv
    last_val[:, 0] = False
    mv[last_val] = -MOTION_GAP
    return mv, torch.stack([head.cos(), head.sin()], dim=-1)


def _gap_rules(dp, dth, s_inv, d_inv):
    """agent_decoder.py:595-601 / :647-653 (the :598 / :650 line is an always-false mask)"""
    dp[s_inv & ~d_inv] = -MOTION_GAP
    dp[~s_inv & d_inv] = MOTION_GAP
    dth[s_inv & ~d_inv] = -HEADING_GAP
    dp[s_inv & d_inv] = INVALID_MOTION
    dth[s_inv & d_inv] = INVALID_HEAD


class ForwardOracle:
    def __init__(self, sd: Dict[str, torch.Tensor], cfg, grid: np.ndarray, prefix: str = 'agent_encoder'):
        self.sd, self.cfg, self.p = sd, cfg, prefix
        self.grid = _t(grid).float()
        self.G = self.grid.shape[0]

    # ------------------------------------------------------------------ features (agent_decoder.py:332-509)
    def tables(self, vocab):
        sd, p = self.sd, self.p
        bos, no = sd[p + '.bos_token_emb.weight'], sd[p + '.no_token_emb.weight']
        tabs = [torch.cat([mlp_embedding(sd, f'{p}.token_emb_{n}', _t(vocab[n]).float()[:, -1].flatten(1, 2)), bos, no])
                for n in ('veh', 'ped', 'cyc')]
        grid_tab = torch.cat([mlp_embedding(sd, p + '.token_emb_grid', self.grid), sd[p + '.invalid_offset_token_emb.weight']])
        return torch.stack(tabs), grid_tab

    def agent_feature(self, mv, hv, tok_emb, grid_emb, type_idx, shape, state):
        """_build_agent_feature (:449-509) on (n, T) rows"""
        sd, p = self.sd, self.p
        n, T = state.shape
        feat = torch.stack([torch.norm(mv, p=2, dim=-1), angle_between(hv, mv)], dim=-1)
        cat = [sd[p + '.type_a_emb.weight'][type_idx.reshape(-1)], mlp_embedding(sd, p + '.shape_emb', shape.reshape(-1, 3))]
        x_a = fourier_embedding(sd, p + '.x_a_emb', feat.view(-1, 2), cat).view(n, T, -1)
        s_a = sd[p + '.state_a_emb.weight'][state.reshape(-1)].view(n, T, -1)
        return mlp_embedding(sd, p + '.fusion_emb', torch.cat([tok_emb, x_a, s_a, grid_emb], dim=-1))

    # ------------------------------------------------------------------ the forward
    @torch.no_grad()
    def forward(self, batch, x_pt: torch.Tensor, vocab) -> Dict[str, torch.Tensor]:
        sd, p, cfg = self.sd, self.p, self.cfg
        ag, pt = batch['agent'], batch['pt_token']
        pos = _t(ag['token_pos']).float().clone()
        head = _t(ag['token_heading']).float().clone()
        A, T, _ = pos.shape
        H = cfg.num_historical_steps
        shape = _t(ag['shape']).float()[:, H - 1].clone()
        token = _t(ag['token_idx']).long()
        state = _t(ag['state_idx']).long()
        atype = _t(ag['type']).long()
        av = _t(ag['av_index']).long()
        ptr = _t(ag['ptr']).long()
        B = av.shape[0]
        S = B * NS
        N = A + S
        gidx = _t(ag['grid_token_idx']).long()
        sort_idx = _t(ag['sort_indices']).long()
        bsz = ptr[1:] - ptr[:-1]
        graph_of = torch.repeat_interleave(torch.arange(B), bsz)                      # scene of every agent row
        pt_ptr = _t(pt['ptr']).long()
        map_pos = _t(pt['position']).float()[:, :2].contiguous()
        map_orient = _t(pt['orientation']).float()
        M = map_pos.shape[0]
        out = {'ego_pos': pos[av]}

        # ---- raw features of the agents and the (constant) seed rows  (:1131-1140, :332-424)
        tok_tab, grid_tab = self.tables(vocab)
        mv, hv = build_vector(pos, head, state)
        tok_emb = tok_tab[atype[:, None].expand(A, T), token]
        inv = state == INVALID
        types = atype[:, None].expand(A, T).clone()
        types[inv] = SEED_TYPE
        shapes = shape[:, None, :].expand(A, T, 3).clone()
        shapes[inv] = INVALID_SHAPE
        raw_a = self.agent_feature(mv, hv, tok_emb, grid_tab[gidx], types, shapes, state)
        st_seed = torch.full((S, T), INVALID)
        mv_s, hv_s = build_vector(torch.zeros(S, T, 2), torch.zeros(S, T), st_seed)
        no_tok = sd[p + '.no_token_emb.weight'][0].expand(S, T, -1)
        raw_seed = self.agent_feature(mv_s, hv_s, no_tok, grid_tab[self.G // 2].expand(S, T, -1),
                                      torch.full((1,), SEED_TYPE).expand(S * T).reshape(S, T), torch.full((S, T, 3), INVALID_SHAPE), st_seed)
        feat = torch.cat([raw_a, raw_seed])                                            # (N, T, D)

        # ---- masks (:1143-1159)
        mask = _t(ag['raw_agent_valid_mask']).bool().clone()
        is_bos, is_eos = state == ENTER, state == EXIT
        bos = torch.where(is_bos.any(1), torch.argmax(is_bos.long(), 1), torch.tensor(0))
        eos = torch.where(is_eos.any(1), torch.argmax(is_eos.long(), 1), torch.tensor(T - 1))
        cols = torch.arange(T)[None, :]
        motion_mask = (cols > bos[:, None]) & (cols <= eos[:, None])
        tmask = torch.ones(A, T, dtype=torch.bool)
        tmask[motion_mask] = mask[motion_mask]
        imask = mask.clone()
        imask[is_bos] = True

        # ---- padded arrays: seed rows sit at their scene's ego (:511-526)
        seed_of = av.repeat_interleave(NS)                                              # ego row of every seed row
        pos_p, head_p, state_p, hv_p = (torch.cat([x, x[seed_of]]) for x in (pos, head, state, hv))
        graph_p = torch.cat([graph_of, torch.arange(B).repeat_interleave(NS)])
        inv_p = state_p == INVALID

        # ---- temporal edges, agent-major node ids a * T + t (:540-610)
        hist = tmask & (cols >= bos[:, None])
        start = torch.clamp(bos - cfg.time_span / cfg.shift + 1, min=0)
        hist = hist & (cols >= start[:, None])
        pair = hist[:, :, None] & hist[:, None, :]
        dj = cols[0][None, :] - cols[0][:, None]                                        # dst col - src col
        pair = pair & (dj > 0)[None] & (dj <= cfg.time_span / cfg.shift)[None]
        nz = torch.nonzero(pair)
        ta, ti, tj = nz[:, 0], nz[:, 1], nz[:, 2]
        dp = pos[ta, ti] - pos[ta, tj]
        dth = wrap_angle(head[ta, ti] - head[ta, tj])
        _gap_rules(dp, dth, inv[ta, ti], inv[ta, tj])
        r_t = torch.stack([torch.norm(dp, p=2, dim=-1), angle_between(hv[ta, tj], dp), dth, (ti - tj).float()], dim=-1)
        r_t = fourier_embedding(sd, p + '.r_t_emb', r_t)
        e_t = (ta * T + ti, ta * T + tj)
        out['_edges_t'] = int(ta.numel())

        # ---- seq_mask of _build_seq (:994-1054): seed (b, s) at column t may attend agent column r?
        seq = torch.ones(S, T, N, dtype=torch.bool)
        seq[..., A:] = False
        for b in range(B):
            bs = sort_idx[ptr[b]:ptr[b + 1]]
            for t in range(T):
                for s in range(NS):
                    seq[b * NS + s, t, bs[s:, t]] = False                               # scene-local indices, as in the reference
        seq[..., av] = True

        # ---- interaction / map edges per (scene, column) group, step-major node ids t * N + row (:612-758, :760-904)
        rows_of = [torch.cat([torch.arange(int(ptr[b]), int(ptr[b + 1])), A + b * NS + torch.arange(NS)]) for b in range(B)]
        ea_s, ea_d, ra = [], [], []               # agent <-> agent
        es_s, es_d, rs = [], [], []               # agent -> seed
        em_s, em_d, rm = [], [], []               # map -> agent
        eq_s, eq_d, rq = [], [], []               # map -> seed
        for t in range(T):
            for b in range(B):
                rows = rows_of[b]
                P, Hd = pos_p[rows, t], head_p[rows, t]
                is_seed = rows >= A
                im = torch.cat([imask[rows[~is_seed], t], torch.ones(NS, dtype=torch.bool)])
                iv = inv_p[rows, t]
                hvr = hv_p[rows, t]
                # agent <-> agent: first 300 + 1 in range over ALL rows of the group, then both ends unmasked agents
                yi, xi = radius_first_k(P, P, cfg.a2a_radius, 300 + 1)
                k = (yi != xi) & im[yi] & im[xi] & ~is_seed[yi] & ~is_seed[xi]
                d, s = yi[k], xi[k]
                dp = P[s] - P[d]
                dth = wrap_angle(Hd[s] - Hd[d])
                _gap_rules(dp, dth, iv[s], iv[d])
                ra.append(torch.stack([torch.norm(dp, p=2, dim=-1), angle_between(hvr[d], dp), dth], dim=-1))
                ea_s.append(t * N + rows[s]); ea_d.append(t * N + rows[d])
                # agent -> seed (mode 'insert', :760-849): first 300 of all rows within pl2seed_radius, sources = unmasked agents
                sr = torch.nonzero(is_seed)[:, 0]
                yi, xi = radius_first_k(P, P[sr], cfg.pl2seed_radius, 300)
                k = ~is_seed[xi] & im[xi]
                k = k & seq[rows[sr[yi]] - A, t, rows[xi]]
                d, s = sr[yi[k]], xi[k]
                dp = P[s] - P[d]
                dth = wrap_angle(Hd[s] - Hd[d])
                rs.append(torch.stack([torch.norm(dp, p=2, dim=-1), angle_between(hvr[d], dp), dth], dim=-1))
                es_s.append(t * N + rows[s]); es_d.append(t * N + rows[d])
                # map -> agent: first 5 map tokens of the scene (:683-758); map -> seed: first 2048 within pl2seed_radius
                m0, m1 = int(pt_ptr[b]), int(pt_ptr[b + 1])
                MP, MO = map_pos[m0:m1], map_orient[m0:m1]
                yi, xi = radius_first_k(MP, P, cfg.pl2a_radius, 5)
                k = im[yi] & ~is_seed[yi]
                d, s = yi[k], xi[k]
                dp = MP[s] - P[d]
                dth = wrap_angle(MO[s] - Hd[d])
                dp[iv[d]] = MOTION_GAP
                dth[iv[d]] = HEADING_GAP
                rm.append(torch.stack([torch.norm(dp, p=2, dim=-1), angle_between(hvr[d], dp), dth], dim=-1))
                em_s.append(t * M + m0 + s); em_d.append(t * N + rows[d])
                yi, xi = radius_first_k(MP, P[sr], cfg.pl2seed_radius, 2048)
                d, s = sr[yi], xi
                dp = MP[s] - P[d]
                dth = wrap_angle(MO[s] - Hd[d])
                rq.append(torch.stack([torch.norm(dp, p=2, dim=-1), angle_between(hvr[d], dp), dth], dim=-1))
                eq_s.append(t * M + m0 + s); eq_d.append(t * N + rows[d])
        cat = torch.cat
        r_a = fourier_embedding(sd, p + '.r_a2a_emb', cat(ra))
        r_s = fourier_embedding(sd, p + '.r_a2sa_emb', cat(rs))
        r_m = fourier_embedding(sd, p + '.r_pt2a_emb', cat(rm))
        r_q = fourier_embedding(sd, p + '.r_pt2sa_emb', cat(rq))
        e_a, e_s, e_m, e_q = (cat(ea_s), cat(ea_d)), (cat(es_s), cat(es_d)), (cat(em_s), cat(em_d)), (cat(eq_s), cat(eq_d))
        out['_edges'] = dict(a=int(e_a[0].numel()), a2sa=int(e_s[0].numel()), m=int(e_m[0].numel()), m2sa=int(e_q[0].numel()))
        e_aa = (cat([e_a[0], e_s[0]]), cat([e_a[1], e_s[1]]))                           # what the motion layers run on (quirk 1)
        e_mm = (cat([e_m[0], e_q[0]]), cat([e_m[1], e_q[1]]))
        r_aa, r_mm = cat([r_a, r_s]), cat([r_m, r_q])
        x_pt_s = x_pt.repeat(T, 1)                                                      # step-major copies: row t * M + m

        # ---- motion stage (:1201-1227)
        f = feat
        for i in range(cfg.num_agent_layers):
            f = attention_layer(sd, f'{p}.t_attn_layers.{i}', f.reshape(-1, f.shape[-1]), r_t, e_t[0], e_t[1]).view(N, T, -1)
            fs = f.transpose(0, 1).reshape(T * N, -1)
            fs = attention_layer(sd, f'{p}.pt2a_attn_layers.{i}', fs, r_mm, e_mm[0], e_mm[1], x_src_raw=x_pt_s)
            fs = attention_layer(sd, f'{p}.a2a_attn_layers.{i}', fs, r_aa, e_aa[0], e_aa[1])
            f = fs.view(T, N, -1).transpose(0, 1)
        out['x_a'] = f
        fe = f[:A]
        out['next_token_prob'] = mlp_layer(sd, p + '.token_predict_head', fe)
        out['next_token_idx'] = torch.topk(torch.softmax(out['next_token_prob'], dim=-1), k=10, dim=-1)[1]
        out['next_token_idx_gt'] = token.roll(-1, 1)
        out['next_state_prob'] = mlp_layer(sd, p + '.state_predict_head', fe)
        out['next_state_idx'] = out['next_state_prob'].softmax(-1).argmax(-1, keepdim=True)
        state_gt = state.roll(-1, 1)

        # ---- occupancy ground truth (:1056-1102): cells of the agents every seed row attends, cells of the scene's map tokens
        occ_a = torch.zeros(S, T, self.G, dtype=torch.long)
        src_row, src_col = e_s[0] % N, e_s[0] // N
        occ_a[e_s[1] % N - A, e_s[1] // N, gidx[src_row, src_col]] = 1
        pt_cells = _t(ag['pt_grid_token_idx']).long()
        occ_m = torch.zeros(B, T, self.G, dtype=torch.long)
        for b in range(B):
            for t in range(T):
                c = pt_cells[t, pt_ptr[b]:pt_ptr[b + 1]]
                occ_m[b, t, c[c != -1]] = 1
        out['grid_agent_occ_gt_seed'] = occ_a
        out['grid_pt_occ_gt_seed'] = occ_m.repeat_interleave(NS, dim=0)

        # ---- coarse stage (:1236-1302)
        occ_emb = mlp_layer(sd, p + '.seed_agent_occ_embed', occ_a.transpose(0, 1).reshape(-1, self.G).float())   # row t * S + k
        seed_nodes = (torch.arange(T)[:, None] * N + A + torch.arange(S)[None, :]).reshape(-1)
        e_o = (torch.arange(T * S), seed_nodes)
        fs = cat([raw_a, raw_seed]).transpose(0, 1).reshape(T * N, -1)
        for i in range(3):
            fs = attention_layer(sd, f'{p}.occ2sa_attn_layers.{i}', fs, None, e_o[0], e_o[1], x_src_raw=occ_emb)
            fs = attention_layer(sd, f'{p}.pt2sa_attn_layers.{i}', fs, r_q, e_q[0], e_q[1], x_src_raw=x_pt_s)
            fs = attention_layer(sd, f'{p}.a2sa_attn_layers.{i}', fs, r_s, e_s[0], e_s[1])
        f_seed = fs.view(T, N, -1).transpose(0, 1)[A:]                                  # (S, T, D)
        st_seed_prob = mlp_layer(sd, p + '.seed_state_predict_head', f_seed)
        out['raw_next_state_prob_seed'] = st_seed_prob.clone()
        st_seed_idx = st_seed_prob.softmax(-1).argmax(-1, keepdim=True)
        type_prob = mlp_layer(sd, p + '.seed_type_predict_head', f_seed)
        type_idx = type_prob.softmax(-1).argmax(-1, keepdim=True)
        type_gt = atype[:, None].repeat(1, T)
        shape_seed = mlp_layer(sd, p + '.seed_shape_predict_head', f_seed)
        shape_gt = shape[:, None].repeat(1, T, 1)
        pos_prob = mlp_layer(sd, p + '.seed_pos_rel_token_predict_head', f_seed)
        pos_xy_gt = _t(ag['pos_xy']).float() / cfg.pl2seed_radius
        gp = _t(np.asarray(gidx)).clone()
        gp = cat([gp, gp[seed_of]])
        out['neighbor_agent_grid_index_gt'] = gp.transpose(0, 1).reshape(-1)[e_s[0]]
        out['neighbor_pt_grid_index_gt'] = pt_cells.reshape(-1)[e_q[0]]
        out['neighbor_agent_grid_idx'] = mlp_layer(sd, p + '.grid_index_head', r_s)
        out['neighbor_pt_grid_idx'] = mlp_layer(sd, p + '.grid_index_head', r_q)
        ma = torch.zeros(e_s[0].numel(), dtype=torch.bool)
        mp = torch.zeros(e_q[0].numel(), dtype=torch.bool)
        ma[torch.randperm(ma.shape[0])[:180]] = True
        mp[torch.randperm(mp.shape[0])[:600]] = True
        out['neighbor_agent_grid_index_eval_mask'], out['neighbor_pt_grid_index_eval_mask'] = ma, mp
        out['grid_agent_occ_seed'] = mlp_layer(sd, p + '.grid_agent_occ_head', f_seed)
        out['grid_pt_occ_seed'] = mlp_layer(sd, p + '.grid_pt_occ_head', f_seed)

        # ---- refine stage (:1304-1385)
        mask_sa = torch.zeros(A, T, dtype=torch.bool)
        for t in range(T):
            avail = ((state[:, t] != INVALID) & (gidx[:, t] != -1)).nonzero()[..., 0]
            mask_sa[avail[torch.randperm(avail.shape[0])[:B * 10]], t] = True
        mask_sa[is_bos] = True
        mask_sa[:, 0] = False
        mask_sa[av] = False
        state_sa = torch.full_like(state, INVALID)
        state_sa[mask_sa] = ENTER
        head_sa = head.clone()
        ego_of = av[graph_of]
        head_sa[mask_sa] = head[ego_of][mask_sa]
        mv_sa, hv_sa = build_vector(pos, head_sa, state_sa)
        mv_sa[mask_sa] = MOTION_GAP
        tok_sa = sd[p + '.no_token_emb.weight'][0].expand(A, T, -1).clone()
        tok_sa[state_sa == ENTER] = sd[p + '.bos_token_emb.weight'][0]
        f_sa = self.agent_feature(mv_sa, hv_sa, tok_sa, grid_tab[gidx], type_gt, shape_gt, state_sa)
        f_sa[~mask_sa] = raw_a[~mask_sa]
        xs_s, xs_d, xr = [], [], []
        ms_s, ms_d, mr = [], [], []
        inv_sa = state_sa == INVALID
        for t in range(T):
            for b in range(B):
                rows = torch.arange(int(ptr[b]), int(ptr[b + 1]))
                P, Hd, hvr = pos[rows, t], head_sa[rows, t], hv_sa[rows, t]
                sa = torch.nonzero(mask_sa[rows, t])[:, 0]
                if sa.numel() == 0:
                    continue
                yi, xi = radius_first_k(P, P[sa], cfg.a2sa_radius, 8)
                k = ~mask_sa[rows[xi], t] & imask[rows[xi], t]
                d, s = sa[yi[k]], xi[k]
                dp = P[s] - P[d]
                dth = wrap_angle(Hd[s] - Hd[d])
                xr.append(torch.stack([torch.norm(dp, p=2, dim=-1), angle_between(hvr[d], dp), dth], dim=-1))
                xs_s.append(t * A + rows[s]); xs_d.append(t * A + rows[d])
                m0, m1 = int(pt_ptr[b]), int(pt_ptr[b + 1])
                MP, MO = map_pos[m0:m1], map_orient[m0:m1]
                yi, xi = radius_first_k(MP, P[sa], cfg.pl2sa_radius, 32)
                d, s = sa[yi], xi
                dp = MP[s] - P[d]
                dth = wrap_angle(MO[s] - Hd[d])
                mr.append(torch.stack([torch.norm(dp, p=2, dim=-1), angle_between(hvr[d], dp), dth], dim=-1))
                ms_s.append(t * M + m0 + s); ms_d.append(t * A + rows[d])
        z = torch.zeros(0, dtype=torch.long)
        e_x = (cat(xs_s), cat(xs_d)) if xs_s else (z, z)
        e_y = (cat(ms_s), cat(ms_d)) if ms_s else (z, z)
        r_x = fourier_embedding(sd, p + '.r_a2a_emb', cat(xr)) if xs_s and e_x[0].numel() else torch.zeros(0, 128)
        r_y = fourier_embedding(sd, p + '.r_pt2a_emb', cat(mr)) if ms_s and e_y[0].numel() else torch.zeros(0, 128)
        out['_edges'].update(a2sa_refine=int(e_x[0].numel()), m2sa_refine=int(e_y[0].numel()))
        sel = torch.zeros(A * T, dtype=torch.bool)
        sel[torch.unique(e_x[1])] = True
        sel[torch.unique(e_y[1])] = True
        mask_sa[~sel.view(T, A).transpose(0, 1)] = False
        fs = f_sa.transpose(0, 1).reshape(T * A, -1)
        for i in range(3):
            fs = attention_layer(sd, f'{p}.pt2a_attn_layers.{i}', fs, r_y if e_y[0].numel() else None, e_y[0], e_y[1], x_src_raw=x_pt_s)
            fs = attention_layer(sd, f'{p}.a2a_attn_layers.{i}', fs, r_x if e_x[0].numel() else None, e_x[0], e_x[1])
        f_sa = fs.view(T, A, -1).transpose(0, 1)
        out['next_head_rel_prob_seed'] = mlp_layer(sd, p + '.seed_heading_rel_token_predict_head', f_sa)
        out['next_head_rel_index_gt_seed'] = _t(ag['heading_token_idx']).long()
        out['next_head_rel_theta_gt_seed'] = _t(ag['heading_theta']).float() / math.pi
        out['next_offset_xy_seed'] = torch.tanh(mlp_layer(sd, p + '.seed_offset_xy_predict_head', f_sa)) * 2
        out['next_offset_xy_gt_seed'] = _t(ag['grid_offset_xy']).float()

        # ---- evaluation masks (:1387-1420)
        tok_eval = mask & mask.roll(-1, 1) & mask.roll(1, 1)
        st_eval = tok_eval.clone()
        for a_, c_ in torch.nonzero(is_bos).tolist():
            tok_eval[a_, c_:c_ + 1] = True
            tok_eval[a_, c_ + 1:c_ + 2] = mask[a_, c_ + 2:c_ + 3]
            st_eval[a_, :c_] = False
            st_eval[a_, c_:c_ + 1] = True
            st_eval[a_, c_ + 1:c_ + 2] = mask[a_, c_ + 2:c_ + 3]
        eos_idx = torch.nonzero(is_eos)
        tok_eval[eos_idx[:, 0], eos_idx[:, 1]] = False
        for a_, c_ in eos_idx.tolist():
            st_eval[a_, c_ + 1:] = True
            st_eval[a_, c_:c_ + 1] = mask[a_, c_ - 1:c_]
        tok_eval[:, 0] = mask[:, 0] & mask[:, 1]
        st_eval[:, 0] = mask[:, 0] & mask[:, 1]
        tok_eval[:, -1] = False
        st_eval[:, -1] = False
        seed_st_eval = torch.ones(S, T, dtype=torch.bool)
        seed_st_eval[:, 0] = False

        # ---- seed rows against the entering agents in bearing order (:1451-1509)
        pred, gt = [], []
        for b in range(B):
            bs = sort_idx[ptr[b]:ptr[b + 1]]
            n_b = min(NS, bs.shape[0])
            pred.append((torch.arange(n_b) + b * NS)[:, None].repeat(1, T))
            gt.append(bs[:n_b] + ptr[b])
        pred, gt = cat(pred), cat(gt)                                                   # (n, T) each
        n = pred.shape[0]
        rest = []
        for t in range(T):
            used = torch.zeros(S, dtype=torch.bool)
            used[pred[:, t]] = True
            rest.append(torch.arange(S)[~used])
        padded = cat([pred, torch.stack(rest, dim=1)])                                  # (S, T)
        out['next_state_idx_seed'] = torch.gather(st_seed_idx, 0, padded[..., None])
        out['next_state_prob_seed'] = torch.gather(st_seed_prob, 0, padded[..., None].expand(-1, -1, 2))
        st_gt_seed = torch.gather(state, 0, gt)
        st_gt_seed = cat([st_gt_seed, torch.zeros(S - n, T, dtype=torch.long)])
        enter = st_gt_seed == ENTER
        out['next_state_idx_gt_seed'] = enter.long()          # seed_state_type = ['invalid', 'enter']
        out['next_type_idx_seed'] = torch.gather(type_idx, 0, pred[..., None])
        out['next_type_prob_seed'] = torch.gather(type_prob, 0, pred[..., None].expand(-1, -1, 3))
        out['next_type_idx_gt_seed'] = torch.gather(type_gt, 0, gt)
        out['next_pos_rel_prob_seed'] = torch.gather(pos_prob, 0, pred[..., None].expand(-1, -1, self.G))
        out['next_pos_rel_index_gt_seed'] = torch.gather(gidx, 0, gt)
        out['next_pos_rel_xy_gt_seed'] = torch.gather(pos_xy_gt, 0, gt[..., None].expand(-1, -1, 2))
        out['next_shape_seed'] = torch.gather(shape_seed, 0, pred[..., None].expand(-1, -1, 3))
        out['next_shape_gt_seed'] = torch.gather(shape_gt, 0, gt[..., None].expand(-1, -1, 3))
        attr_eval = enter[:n].clone()
        attr_eval[:, 0] = False
        attr_eval[out['next_pos_rel_index_gt_seed'] == self.G // 2] = False
        st_eval[av] = False
        state_gt = state_gt.clone()
        state_gt[state_gt == EXIT] = 2                        # valid_state_type.index('exit')
        occ_eval = torch.ones(S, T, self.G, dtype=torch.bool)
        occ_eval[:, 0] = False
        occ_eval[..., self.G // 2] = False
        tgt = pred.clone()
        tgt[~attr_eval] = -1
        out.update(next_token_eval_mask=tok_eval, next_state_eval_mask=st_eval, next_state_idx_gt=state_gt,
                   next_state_eval_mask_seed=seed_st_eval, next_attr_eval_mask_seed=attr_eval, next_head_eval_mask_seed=mask_sa,
                   grid_agent_occ_eval_mask_seed=occ_eval, grid_pt_occ_eval_mask_seed=occ_eval, target_indices=tgt)
        return out


def split_scenes(batch):
    """per-scene views of the map part of a batch (what oracle.rollout_oracle.map_encoder takes)"""
    pt = batch['pt_token']
    ptr = np.asarray(pt['ptr'])
    e = np.asarray(batch['pt_token__to__map_polygon']['edge_index'])
    for b in range(len(ptr) - 1):
        sl = slice(int(ptr[b]), int(ptr[b + 1]))
        yield {'pt_token': {k: (np.asarray(v)[sl] if isinstance(v, np.ndarray) and v.shape[:1] == (int(ptr[-1]),) else v)
                            for k, v in pt.items()},
               'map_polygon': batch['map_polygon'],
               'pt_token__to__map_polygon': {'edge_index': e[:, sl]}}


def run_forward(sd, batch, cfg, vocab, map_vocab, grid):
    """InfGenDecoder.forward (infgen_decoder.py:114-121) on a batch dict; torch's global CPU generator supplies the
    permutations (seed it like the fixture script)"""
    x_pt = torch.cat([map_encoder(sd, sc, cfg, map_vocab) for sc in split_scenes(batch)])
    out = ForwardOracle(sd, cfg, grid).forward(batch, x_pt, vocab)
    out['x_pt'] = x_pt
    return out
