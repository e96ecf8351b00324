"""CPU ORACLE (test infrastructure) — closed-loop rollout WITH scenario insertion.

Extends ``rollout_oracle.RolloutOracle`` with the insertion sub-loop of
``InfGenAgentDecoder.inference`` (reference infgen/modules/agent_decoder.py:1773-2105; SURVEY
Appendix A.6): per decode step t >= 1, up to ``insert_limit = 10`` iterations of
  seed query (occupancy / map / agent attention, 3 layers) -> enter? type, shape, grid cell ->
  reject if the cell is occupied -> append a row -> heading stage (the motion pt2a/a2a layers
  0..2 re-used on the new row) -> heading token + xy offset -> raw feature of the new row.
Greedy everywhere (``insert_beam_size = 1``): a rejected cell would be re-sampled identically, so a
rejection ends the step's insertion (A.6(d)).

PINNED by tests/test_oracle_golden.py against fixtures generated from the reference with
``DEBUG=1`` (forced "enter", agent_decoder.py:1888-1889) and with the natural seed-state head.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from .rollout_oracle import (RolloutOracle, INVALID, VALID, ENTER, EXIT, SEED_TYPE, INVALID_SHAPE, MOTION_GAP,
                             AGENT_SHAPE, angle_between, wrap_angle, fourier_embedding, mlp_embedding, mlp_layer,
                             attention_layer, radius_first_k, rot_right, encode_pos, _t)

INSERT_LIMIT = 10          # agent_decoder.py:1738
SEED_LAYERS = 3            # agent_decoder.py:235


class InsertionOracle(RolloutOracle):

    def __init__(self, sd, cfg, grid, prefix='agent_encoder', force_enter: bool = False, insert_k: int = 1,
                 insert_uniforms=None):
        super().__init__(sd, cfg, grid, prefix=prefix, live_state=True)
        self.force_enter = force_enter
        # cell of a new agent: arg-max, or (insert_k > 1) the reference's softmax -> topk(insert_beam_size) -> multinomial
        # (agent_decoder.py:1900-1904) as inverse CDF over the top-k probabilities with insert_uniforms[t][iteration]
        self.insert_k = int(insert_k)
        self.insert_uniforms = insert_uniforms

    # ------------------------------------------------------------------ pieces
    def _edgeless(self, name, i, x, x_src=None):
        z = torch.zeros(0, dtype=torch.long)
        return attention_layer(self.sd, f'{self.p}.{name}.{i}', x, None, z, z, x_src_raw=x_src)

    def seed_feature(self, st):
        """the all-invalid template row (agent_decoder.py:1814-1818, 462-509; A.6(b)): constant of the weights"""
        sd, p = self.sd, self.p
        mv = torch.full((1, 2), -2.0)
        hv = torch.tensor([[1.0, 0.0]])
        feat = torch.stack([torch.norm(mv, p=2, dim=-1), angle_between(hv, mv)], dim=-1)
        cat = [st['seed_type_emb'][None], st['seed_shape_emb'][None]]
        x_a = fourier_embedding(sd, p + '.x_a_emb', feat, cat)
        tok = sd[p + '.no_token_emb.weight']
        s_a = sd[p + '.state_a_emb.weight'][INVALID][None]
        g = st['grid_tab'][self.grid.shape[0] // 2][None]
        return mlp_embedding(sd, p + '.fusion_emb', torch.cat([tok, x_a, s_a, g], dim=-1))

    def _grow(self, st, key, row):
        st[key] = torch.cat([st[key], row], dim=0)

    # ------------------------------------------------------------------ one insertion attempt
    def try_insert(self, st, c, t, raw_c, num_new, it=0):
        """returns (inserted: True / False, or None when the chosen cell was occupied - the iteration is spent, :1906-1909 -, new
        raw_c).  SURVEY A.6 steps (1)-(7)."""
        sd, p, cfg = self.sd, self.p, self.cfg
        A = st['pos'].shape[0]
        av = st['av']
        pos_c, head_c, state_c = st['pos'][:, c], st['head'][:, c], st['state'][:, c]
        ego_pos, ego_head = pos_c[av], head_c[av]
        ego_hv = torch.stack([ego_head.cos(), ego_head.sin()])
        # ---- (2) edges into the seed node
        yi, xi = radius_first_k(torch.cat([pos_c, ego_pos[None]]), ego_pos[None], cfg.pl2seed_radius, 300)
        src_a = xi[(xi < A)]
        src_a = src_a[st['imask'][src_a, c]]
        dp = pos_c[src_a] - ego_pos
        dth = wrap_angle(head_c[src_a] - ego_head)
        r_a = torch.stack([torch.norm(dp, p=2, dim=-1), angle_between(ego_hv[None].expand(len(src_a), 2), dp), dth], dim=-1)
        r_a = fourier_embedding(sd, p + '.r_a2sa_emb', r_a) if len(src_a) else torch.zeros(0, 128)
        _, src_m = radius_first_k(st['map_pos'], ego_pos[None], cfg.pl2seed_radius, 2048)
        dpm = st['map_pos'][src_m] - ego_pos
        dthm = wrap_angle(st['map_orient'][src_m] - ego_head)
        r_m = torch.stack([torch.norm(dpm, p=2, dim=-1), angle_between(ego_hv[None].expand(len(src_m), 2), dpm), dthm], dim=-1)
        r_m = fourier_embedding(sd, p + '.r_pt2sa_emb', r_m) if len(src_m) else torch.zeros(0, 128)
        G = self.grid.shape[0]
        occ = torch.zeros(G, dtype=torch.long)
        g_c = st['gridtok'][:, c]
        occ[g_c[g_c != -1]] = 1
        occ_emb = mlp_layer(sd, p + '.seed_agent_occ_embed', occ.float()[None])
        # ---- (3) three (occ2sa, pt2sa, a2sa) layers; every node passes every layer, only the seed has edges
        xa = raw_c.clone()                       # agents' column-c features (sources of a2sa)
        xs = self.seed_feature(st)               # (1,128)
        z0 = torch.zeros(1, dtype=torch.long)
        for i in range(SEED_LAYERS):
            xs = attention_layer(sd, f'{p}.occ2sa_attn_layers.{i}', xs, None, z0, z0, x_src_raw=occ_emb)
            xa = self._edgeless('occ2sa_attn_layers', i, xa, x_src=occ_emb)
            xs = attention_layer(sd, f'{p}.pt2sa_attn_layers.{i}', xs, r_m, src_m, torch.zeros(len(src_m), dtype=torch.long),
                                 x_src_raw=st['x_pt'])
            xa = self._edgeless('pt2sa_attn_layers', i, xa, x_src=st['x_pt'])
            xin = torch.cat([xs, xa], dim=0)
            xo = attention_layer(sd, f'{p}.a2sa_attn_layers.{i}', xin, r_a, 1 + src_a, torch.zeros(len(src_a), dtype=torch.long))
            xs, xa = xo[:1], xo[1:]
        # ---- (4) heads on the seed node
        st_logit = mlp_layer(sd, p + '.seed_state_predict_head', xs)
        enter = int(st_logit.softmax(-1).argmax(-1)) == 1
        if self.force_enter:
            enter = True
        ty_logit = mlp_layer(sd, p + '.seed_type_predict_head', xs)
        ty = int(ty_logit.softmax(-1).argmax(-1))
        shape = mlp_layer(sd, p + '.seed_shape_predict_head', xs)[0]
        pos_logit = mlp_layer(sd, p + '.seed_pos_rel_token_predict_head', xs)
        pos_prob = torch.softmax(pos_logit, dim=-1)
        cell = int(torch.topk(pos_prob, k=1, dim=-1)[1][0, 0])
        if self.insert_k > 1:
            pk, ik = torch.topk(pos_prob[0], k=self.insert_k)
            cdf = torch.cumsum(pk / pk[0], 0)                  # (ratios to the largest: the device forms exp(v - v_max) sums)
            u = float(self.insert_uniforms[t][it]) * float(cdf[-1])
            cell = int(ik[min(int((u >= cdf).sum()), self.insert_k - 1)])
        top2 = torch.topk(pos_logit[0], k=2).values
        new_pos = rot_right(self.grid[cell][None, None], (ego_head - math.pi / 2)[None])[0, 0] + ego_pos
        st['seed_log'].append(dict(t=t, enter=enter, cell=cell, occupied=bool(occ[cell]), type=ty,
                                   state_margin=float((st_logit[0, 1] - st_logit[0, 0]).abs()),
                                   cell_margin=float(top2[0] - top2[1]), type_logits=ty_logit[0].tolist()))
        if bool(occ[cell]):
            # rejected; greedy would pick the same cell again -> no further insertion this step; sampled: draw again next iteration
            return (None if self.insert_k > 1 else False), raw_c
        if not enter or num_new + 1 > INSERT_LIMIT:
            return False, raw_c
        # ---- (5) append a row
        T = st['pos'].shape[1]
        z = lambda *s, **k: torch.zeros(*s, **k)
        npos = z(1, T, 2); npos[0, c] = new_pos
        nhead = z(1, T); nhead[0, c] = ego_head
        nstate = z(1, T, dtype=torch.long); nstate[0, c] = ENTER
        ngrid = torch.full((1, T), -1, dtype=torch.long); ngrid[0, c] = cell
        ntok = torch.full((1, T), -1, dtype=torch.long); ntok[0, c] = -2
        for k, v in (('pos', npos), ('head', nhead), ('state', nstate), ('gridtok', ngrid), ('token', ntok)):
            self._grow(st, k, v)
        st['type'] = torch.cat([st['type'], torch.tensor([ty])])
        tm = torch.ones(1, T, dtype=torch.bool)
        im = torch.ones(1, T, dtype=torch.bool); im[0, :c] = False
        self._grow(st, 'tmask', tm); self._grow(st, 'imask', im)
        te = st['seed_type_emb'][None, None].repeat(1, T, 1)
        se = st['seed_shape_emb'][None, None].repeat(1, T, 1)
        te[0, c:] = sd[p + '.type_a_emb.weight'][ty]
        se[0, c:] = mlp_embedding(sd, p + '.shape_emb', shape[None])[0]
        self._grow(st, 'type_emb', te); self._grow(st, 'shape_emb', se)
        for i in range(cfg.num_agent_layers):
            st['X'][i] = torch.cat([st['X'][i], z(1, T, cfg.hidden_dim)], dim=0)
        R = st['pred_traj'].shape[1]
        pt, ph, ps = z(1, R, 2), z(1, R), z(1, R)
        if t > 0:
            pt[0, (t - 1) * 5:t * 5] = new_pos
            ph[0, (t - 1) * 5:t * 5] = ego_head
            ps[0, (t - 1) * 5:t * 5] = float(ENTER)
        self._grow(st, 'pred_traj', pt); self._grow(st, 'pred_head', ph); self._grow(st, 'pred_state', ps)
        st['pred_type'] = torch.cat([st['pred_type'], torch.tensor([ty])])
        st['pred_shape'] = torch.cat([st['pred_shape'], shape[None]])
        st['agent_id'] = torch.cat([st['agent_id'], (st['agent_id'].max() + 1)[None]])
        st['tok_out'] = torch.cat([st['tok_out'], torch.full((1, st['tok_out'].shape[1]), -1, dtype=torch.long)])
        so = torch.zeros(1, st['st_out'].shape[1], dtype=torch.long); so[0, c] = ENTER
        st['st_out'] = torch.cat([st['st_out'], so])
        # ---- (6) heading stage: the new row attends map tokens / agents within 10 m through the MOTION
        #          pt2a / a2a layers 0..2; sources pass the preceding layers edgelessly
        new = A
        x_new = self.raw_feature_row(st, new, c)
        _, hm = radius_first_k(st['map_pos'], new_pos[None], cfg.pl2sa_radius, 128)
        hv_new = torch.stack([ego_head.cos(), ego_head.sin()])
        dpm = st['map_pos'][hm] - new_pos
        dthm = wrap_angle(st['map_orient'][hm] - ego_head)
        r_hm = torch.stack([torch.norm(dpm, p=2, dim=-1), angle_between(hv_new[None].expand(len(hm), 2), dpm), dthm], dim=-1)
        r_hm = fourier_embedding(sd, p + '.r_pt2a_emb', r_hm) if len(hm) else torch.zeros(0, 128)
        pos_all = st['pos'][:, c]
        _, ha = radius_first_k(pos_all, new_pos[None], cfg.a2sa_radius, 24)
        ha = ha[(ha != new)]
        ha = ha[st['imask'][ha, c]]
        dpa = pos_all[ha] - new_pos
        dtha = wrap_angle(st['head'][ha, c] - ego_head)
        r_ha = torch.stack([torch.norm(dpa, p=2, dim=-1), angle_between(hv_new[None].expand(len(ha), 2), dpa), dtha], dim=-1)
        r_ha = fourier_embedding(sd, p + '.r_a2a_emb', r_ha) if len(ha) else torch.zeros(0, 128)
        xa = raw_c.clone()
        xn = x_new
        for i in range(SEED_LAYERS):
            xn = attention_layer(sd, f'{p}.pt2a_attn_layers.{i}', xn, r_hm, hm, torch.zeros(len(hm), dtype=torch.long),
                                 x_src_raw=st['x_pt'])
            xa = self._edgeless('pt2a_attn_layers', i, xa, x_src=st['x_pt'])
            xin = torch.cat([xn, xa], dim=0)
            xo = attention_layer(sd, f'{p}.a2a_attn_layers.{i}', xin, r_ha, 1 + ha, torch.zeros(len(ha), dtype=torch.long))
            xn, xa = xo[:1], xo[1:]
        hidx = mlp_layer(sd, p + '.seed_heading_rel_token_predict_head', xn).softmax(-1).argmax(-1)
        dec = ((hidx * cfg.angle_interval - 180) / 360 * (2 * math.pi)).float()
        new_head = wrap_angle(dec + ego_head)[0]
        off = torch.tanh(mlp_layer(sd, p + '.seed_offset_xy_predict_head', xn))[0] * 2
        st['seed_log'][-1].update(offset=off.tolist(), heading_bin=int(hidx[0]), n_map_h=int(len(hm)), n_agent_h=int(len(ha)))
        st['head'][new, c] = new_head
        st['pos'][new, c] = st['pos'][new, c] + off
        # ---- (7) final raw feature of the new row; Q13 head-vector overwrite for this step's new rows
        x_new = self.raw_feature_row(st, new, c)
        st['hv_override'] = (c, st['first_new'], torch.stack([new_head.cos(), new_head.sin()]))
        return True, torch.cat([raw_c, x_new], dim=0)

    def raw_feature_row(self, st, row, j):
        """raw feature (A.2) of a single row"""
        sub = dict(st)
        for k in ('pos', 'head', 'state', 'token', 'gridtok', 'type', 'type_emb', 'shape_emb'):
            sub[k] = st[k][row:row + 1]
        return self.raw_feature(sub, j)

    # ------------------------------------------------------------------ full rollout
    @torch.no_grad()
    def rollout(self, scene, x_pt, vocab, teacher_tokens=None, teacher_states=None):
        sd, p, cfg = self.sd, self.p, self.cfg
        base = RolloutOracle.rollout        # reuse the setup by running zero decode steps of the base class
        cfg0 = type(cfg)(**{**cfg.__dict__})
        st, ctx = self._setup(scene, x_pt, vocab)
        hc = cfg.hist_columns
        T, R = cfg.num_columns, cfg.num_recurrent_steps_val
        self.run_stack(st, 0, self.raw_feature(st, 0), edgeless=True)
        raw_c = self.raw_feature(st, 1)
        logits_all, n_agents = [], []
        tabs = ctx['tabs']
        for t in range(cfg.num_decode_steps):
            c, n = hc - 1 + t, hc + t
            st['first_new'] = st['pos'].shape[0]
            st['hv_override'] = None
            num_new = 0
            if t > 0:
                for it in range(INSERT_LIMIT):               # `while True: p += 1; if p - 1 >= insert_limit: break` (:1774-1776)
                    ok, raw_c = self.try_insert(st, c, t, raw_c, num_new, it)
                    if ok is None:
                        continue
                    if not ok:
                        break
                    num_new += 1
            A = st['pos'].shape[0]
            n_agents.append(A)
            pos, head, state, token, gridtok = st['pos'], st['head'], st['state'], st['token'], st['gridtok']
            atype = st['type']
            av = st['av']
            x = self.run_stack(st, c, raw_c)
            st['hv_override'] = None
            logits = mlp_layer(sd, p + '.token_predict_head', x)
            logits_all.append(logits)
            next_tok = torch.topk(torch.softmax(logits, dim=-1), k=1, dim=-1)[1][:, 0]
            nstate = mlp_layer(sd, p + '.state_predict_head', x).softmax(dim=-1).argmax(dim=-1)
            nstate[nstate == 2] = EXIT
            nstate[av] = VALID
            if teacher_tokens is not None:
                tt = _t(teacher_tokens[:A, n]).long().clone()
                tt[tt < 0] = 0
                next_tok = tt
            if teacher_states is not None:
                nstate = _t(teacher_states[:A, n]).long().clone()
            contour = tabs[atype, next_tok]
            contour = rot_right(contour.view(A, 24, 2), head[:, c]).view(A, 6, 4, 2) + pos[:, None, None, c, :]
            diff = contour[:, 1:, 0, :] - contour[:, 1:, 3, :]
            st['pred_traj'][:, t * 5:(t + 1) * 5] = contour[:, 1:].mean(dim=2)
            st['pred_head'][:, t * 5:(t + 1) * 5] = torch.arctan2(diff[:, :, 1], diff[:, :, 0])
            st['pred_state'][:, t * 5:(t + 1) * 5] = nstate[:, None].float().repeat(1, 5)
            pos[:, n] = contour[:, -1].mean(dim=1)
            d = contour[:, -1, 0, :] - contour[:, -1, 3, :]
            th_n = torch.arctan2(d[:, 1], d[:, 0])
            head[:, n] = th_n
            state[:, n] = nstate
            gridtok[:, n] = encode_pos(self.grid, pos[:, n], pos[av, n][None].expand(A, 2), th_n[av])
            is_inv = nstate == INVALID
            next_tok = next_tok.clone()
            next_tok[is_inv] = -1
            pos[is_inv, n] = 0.0
            head[is_inv, n] = 0.0
            gridtok[is_inv, n] = -1
            st['imask'][is_inv, n] = False
            st['type_emb'][is_inv, n] = st['seed_type_emb']
            st['shape_emb'][is_inv, n] = st['seed_shape_emb']
            token[:, n] = next_tok
            st['tok_out'][:, n] = next_tok
            st['st_out'][:, n] = nstate
            raw_c = self.raw_feature(st, n)

        A = st['pos'].shape[0]
        A0 = ctx['A0']
        H = cfg.num_historical_steps
        pred_traj = torch.cat([torch.zeros(A, H, 2), st['pred_traj']], dim=1)
        pred_head = torch.cat([torch.zeros(A, H), st['pred_head']], dim=1)
        pred_state = torch.cat([torch.zeros(A, H), st['pred_state']], dim=1)
        ag, filt = ctx['ag'], ctx['filt']
        state0 = ctx['state0']
        pred_traj[:A0, 0] = _t(ag['position'])[filt][:, 0, :2].float()
        pred_head[:A0, 0] = _t(ag['heading'])[filt][:, 0].float()
        pred_state[:A0, 1:H] = state0[filt][:, :hc].repeat_interleave(cfg.shift, dim=1).float()
        htok = _t(ag['token_idx'])[filt][:, :hc].long().clone()
        htok[htok < 0] = 0
        at0 = st['type'][:A0]
        hcont = tabs[at0[:, None].expand(A0, hc), htok]
        hcont = rot_right(hcont.view(A0, hc * 24, 2), st['head'][:A0, 0]).view(A0, hc, 6, 4, 2) + st['pos'][:A0, 0][:, None, None, None, :]
        pred_traj[:A0, 1:H] = hcont[:, :, 1:].mean(dim=3).reshape(A0, -1, 2)
        dxy = hcont[..., 1:, 0, :] - hcont[..., 1:, 3, :]
        pred_head[:A0, 1:H] = torch.arctan2(dxy[..., 1], dxy[..., 0]).reshape(A0, -1)
        pred_valid = (pred_state != INVALID) & (pred_state != ENTER)
        return dict(
            ego_index=st['av'], agent_id=st['agent_id'], pos_a=st['pos'], head_a=st['head'], grid_a=st['gridtok'],
            pred_traj=pred_traj, pred_head=pred_head, pred_state=pred_state, pred_valid=pred_valid,
            pred_type=st['pred_type'], pred_shape=st['pred_shape'],
            next_token_idx=st['tok_out'], next_state_idx=st['st_out'],
            logits=logits_all, n_agents=np.asarray(n_agents), seed_log=st['seed_log'],
            edge_count=np.asarray(st['edge_count'], dtype=np.int64),
            layer_inputs=st['X'])      # X[i][row, column]: input of layer triple i (tests compare intermediate features)

    # ------------------------------------------------------------------ setup shared with the base class
    def _setup(self, scene, x_pt, vocab):
        """agent_decoder.py:1609-1719 (same as RolloutOracle.rollout's preamble), growable state"""
        sd, p, cfg = self.sd, self.p, self.cfg
        ag = scene['agent']
        state0 = _t(ag['state_idx']).long()
        filt = state0[:, 1] != INVALID
        av0 = int(np.asarray(ag['av_index']).reshape(-1)[0])
        av = av0 - int((~filt[:av0]).sum())
        T, R = cfg.num_columns, cfg.num_recurrent_steps_val

        def take(k, dtype):
            return _t(ag[k])[filt].clone().to(dtype)

        def pad(x, val):
            if x.shape[1] >= T:
                return x
            shp = (x.shape[0], T - x.shape[1]) + tuple(x.shape[2:])
            return torch.cat([x, torch.full(shp, val, dtype=x.dtype)], dim=1)
        pos = pad(take('token_pos', torch.float32), 0.0)
        head = pad(take('token_heading', torch.float32), 0.0)
        token = pad(take('token_idx', torch.long), -1)
        state = pad(state0[filt].clone(), INVALID)
        gridtok = pad(take('grid_token_idx', torch.long), -1)
        valid = pad(take('raw_agent_valid_mask', torch.bool), True)
        atype = take('type', torch.long)
        shape10 = _t(ag['shape'])[filt][:, cfg.num_historical_steps - 1].float()
        eval_mask = _t(ag['valid_mask'])[filt][:, cfg.num_historical_steps - 1]
        A = pos.shape[0]
        hc = cfg.hist_columns
        pos[:, hc:] = 0; head[:, hc:] = 0; token[:, hc:] = -1; state[:, hc:] = INVALID; gridtok[:, hc:] = -1
        valid[:, hc:] = True
        valid[~eval_mask] = False
        is_bos, is_eos = state == ENTER, state == EXIT
        bos = torch.where(is_bos.any(1), torch.argmax(is_bos.long(), 1), torch.tensor(0))
        eos = torch.where(is_eos.any(1), torch.argmax(is_eos.long(), 1), torch.tensor(T - 1))
        cols = torch.arange(T)[None, :]
        motion = (cols > bos[:, None]) & (cols <= eos[:, None])
        motion[:, cfg.num_historical_steps // cfg.shift:] = False
        tmask = torch.ones(A, T, dtype=torch.bool)
        tmask[motion] = valid[motion]
        imask = torch.ones(A, T, dtype=torch.bool)
        nonmotion = ~motion
        nonmotion[:, cfg.num_historical_steps // cfg.shift:] = False
        imask[nonmotion] = False
        imask[state == ENTER] = True
        imask[av] = True
        tmask[:, hc:] = True
        imask[:, hc:] = True
        seed_type_emb = sd[p + '.type_a_emb.weight'][SEED_TYPE]
        seed_shape_emb = mlp_embedding(sd, p + '.shape_emb', torch.full((1, 3), INVALID_SHAPE))[0]
        type_emb = sd[p + '.type_a_emb.weight'][atype][:, None, :].repeat(1, T, 1)
        shape_emb = mlp_embedding(sd, p + '.shape_emb', shape10)[:, None, :].repeat(1, T, 1)
        inv = state == INVALID
        type_emb[inv] = seed_type_emb
        shape_emb[inv] = seed_shape_emb
        tok_tab, grid_tab = self.tables(vocab)
        tok_out = torch.full((A, T), -1, dtype=torch.long)
        st_out = torch.zeros(A, T, dtype=torch.long)
        tok_out[:, :hc] = _t(ag['token_idx'])[filt][:, :hc].long()
        st_out[:, :hc] = state0[filt][:, :hc]
        st = dict(pos=pos, head=head, token=token, state=state, gridtok=gridtok, type=atype, tmask=tmask, imask=imask,
                  type_emb=type_emb, shape_emb=shape_emb, tok_tab=tok_tab, grid_tab=grid_tab, x_pt=x_pt,
                  map_pos=_t(scene['pt_token']['position'])[:, :2].contiguous().float(),
                  map_orient=_t(scene['pt_token']['orientation']).float(),
                  X=[torch.zeros(A, T, cfg.hidden_dim) for _ in range(cfg.num_agent_layers)], edge_count=[],
                  av=av, seed_type_emb=seed_type_emb, seed_shape_emb=seed_shape_emb,
                  pred_traj=torch.zeros(A, R, 2), pred_head=torch.zeros(A, R), pred_state=torch.zeros(A, R),
                  pred_type=atype.clone(), pred_shape=_t(ag['shape'])[filt][:, hc - 1].float(),
                  agent_id=_t(ag['id'])[filt].clone(), tok_out=tok_out, st_out=st_out, seed_log=[], hv_override=None)
        tabs = torch.stack([_t(vocab[k]).float() for k in ('veh', 'ped', 'cyc')])
        return st, dict(tabs=tabs, A0=A, ag=ag, filt=filt, state0=state0)


def run_scene_with_insertion(sd, scene, cfg, vocab, map_vocab, grid, force_enter=False, teacher=None, insert_k=1,
                             insert_uniforms=None):
    from .rollout_oracle import map_encoder
    with torch.no_grad():
        x_pt = map_encoder(sd, scene, cfg, map_vocab)
        orc = InsertionOracle(sd, cfg, grid, force_enter=force_enter, insert_k=insert_k, insert_uniforms=insert_uniforms)
        tt = ts = None
        if teacher is not None:
            tt, ts = teacher
        out = orc.rollout(scene, x_pt, vocab, teacher_tokens=tt, teacher_states=ts)
    out['x_pt'] = x_pt
    return out
