"""Device-side agent tokenisation - mirror of the reference's ``TokenProcessor`` (infgen/datasets/preprocess.py:236-653;
SURVEY section 8f rank 1): ``_tokenize_agent`` / ``forward`` (three launches through ``infgen_tokenize_agent``) and the
contour-matching core ``_match_agent_token`` (``infgen_match_agent_tokens``).  Same signatures, dict keys and in-place
side effects as the reference.  There is no CPU fallback."""
import pickle
from typing import Dict, Optional, Tuple

import numpy as np

import torch

from .. import _lib


class TokenProcessor(torch.nn.Module):
    """reference preprocess.py:236-287.  The token tables come from ``agent_tokens`` ({'veh' | 'ped' | 'cyc': (n_token, 6, 4,
    2)}) or ``agent_token_path`` (a pickle with the reference's layout, ``['token_all'][type]``); the reference reads its own
    ``tokens/agent_vocab_555_s2.pkl``, which is data of that repository and not shipped here."""

    def __init__(self, token_size: int = 2048, training: bool = False, predict_motion: bool = False,
                 predict_state: bool = False, predict_map: bool = False, state_token: Optional[Dict[str, int]] = None,
                 agent_tokens: Optional[Dict[str, np.ndarray]] = None, agent_token_path: Optional[str] = None, **kwargs):
        super().__init__()
        self.token_size, self.shift, self.noise, self.current_step = token_size, 5, False, 10
        self.training = False
        # False: skip the reference's per-agent copy of the tables (token_traj_all, 393 KB per agent; token_traj)
        self.materialize_token_traj_all = True
        self.disable_invalid = not predict_state
        self.predict_motion, self.predict_state, self.predict_map = predict_motion, predict_state, predict_map
        st = state_token or dict(invalid=0, valid=1, enter=2, exit=3)
        self.invalid_state, self.valid_state = int(st['invalid']), int(st['valid'])
        self.enter_state, self.exit_state = int(st['enter']), int(st['exit'])
        self.pl2seed_radius = kwargs.get('pl2seed_radius', None)
        if agent_tokens is None and agent_token_path is not None:
            with open(agent_token_path, 'rb') as f:
                agent_tokens = pickle.load(f)['token_all']
        if agent_tokens is not None:
            for name in ('veh', 'ped', 'cyc'):
                self.register_buffer(f'agent_token_all_{name}', torch.as_tensor(np.asarray(agent_tokens[name]),
                                                                                 dtype=torch.float32), persistent=False)

    def forward(self, data):
        """reference preprocess.py:289-306"""
        data['agent']['av_index'] = data['agent']['av_idx']
        data = self._tokenize_agent(data)
        if 'city' in data:
            del data['city']
        for k in ('polygon_is_intersection', 'route_type'):
            if 'map_polygon' in data and k in data['map_polygon']:
                del data['map_polygon'][k]
        av = int(data['agent']['av_idx'])
        data['ego_pos'] = data['agent']['token_pos'][[av]]
        data['ego_heading'] = data['agent']['token_heading'][[av]]
        return data

    @torch.no_grad()
    def _tokenize_agent(self, data):
        """reference preprocess.py:335-550.  Reads data['agent'][valid_mask (A, T) bool, heading (A, T), position (A, T, 3),
        velocity (A, T, 2), type (A,), shape (A, T, 3)] on the GPU; like the reference it cleans / extrapolates
        ``valid_mask``, ``heading`` and ``velocity`` in place and resets ``shape``; adds token_idx, state_idx, token_contour,
        token_pos, token_heading, agent_valid_mask, raw_agent_valid_mask, raw_height, token_traj(_all), traj_pos /
        traj_heading (None) and trajectory_token_{veh,ped,cyc}."""
        if not hasattr(self, 'agent_token_all_veh'):
            raise RuntimeError('TokenProcessor needs agent_tokens / agent_token_path')
        ag = data['agent']
        dev = ag['position'].device
        if dev.type != 'cuda':
            raise RuntimeError('TokenProcessor._tokenize_agent runs on the GPU only (no CPU fallback)')
        A, T = ag['valid_mask'].shape
        n_tok = T // self.shift
        valid = ag['valid_mask'].to(torch.uint8).contiguous()
        head = ag['heading'].to(torch.float32).contiguous()
        vel = ag['velocity'][..., :2].to(torch.float32).contiguous()
        pos = ag['position'][..., :2].to(torch.float32).contiguous()      # a private copy, like the reference's
        ty = ag['type'].to(torch.int32).contiguous()
        names = ('veh', 'ped', 'cyc')
        tables = [getattr(self, f'agent_token_all_{n}').to(dev) for n in names]
        tok_last = torch.stack([t[:, -1] for t in tables]).contiguous()
        shape_in = ag['shape'].to(torch.float32).contiguous()
        shape_out = torch.empty_like(shape_in)
        wl = torch.empty(A, 2, dtype=torch.float32, device=dev)
        idx = torch.empty(A, n_tok, dtype=torch.int32, device=dev)
        state = torch.empty_like(idx)
        contour = torch.empty(A, n_tok, 4, 2, dtype=torch.float32, device=dev)
        tpos = torch.empty(A, n_tok, 2, dtype=torch.float32, device=dev)
        thead = torch.empty(A, n_tok, dtype=torch.float32, device=dev)
        tv = torch.empty(A, n_tok, dtype=torch.uint8, device=dev)
        raw = torch.empty_like(tv)
        p = _lib.ptr
        _lib.check(_lib.load().infgen_tokenize_agent(
            p(valid), p(pos), p(head), p(vel), p(ty), p(tok_last), p(shape_in), p(shape_out), p(wl), A, T, self.shift,
            self.current_step, tok_last.shape[1], self.invalid_state, self.valid_state, self.enter_state, self.exit_state,
            int(not self.disable_invalid), p(idx), p(contour), p(state), p(tpos), p(thead), p(tv), p(raw),
            torch.cuda.current_stream(dev).cuda_stream), 'infgen_tokenize_agent')
        if (shape_out[:, 0] == 0).all(-1).any():
            raise ValueError('Found invalid shape values.')
        ag['valid_mask'].copy_(valid.bool())
        ag['heading'].copy_(head)
        ag['velocity'][..., :2] = vel
        ag['shape'].copy_(shape_out)
        height = ag['position'][:, self.current_step, 2]
        seen = raw[:, 1].bool()
        mean_z = {}
        for k, n in enumerate(names):
            m = height[(ty == k) & seen].mean()
            mean_z[n] = m if k == 0 else torch.where(torch.isnan(m), mean_z['veh'], m)
        token_traj_all = torch.stack(tables)[ty.long()] if self.materialize_token_traj_all else None   # (A, n_token, 6, 4, 2)
        ag.update(token_traj_all=token_traj_all, token_traj=token_traj_all[:, :, -1] if token_traj_all is not None else None,
                  token_idx=idx.long(),
                  state_idx=state.long(), token_contour=contour, traj_pos=None, traj_heading=None, token_pos=tpos,
                  token_heading=thead, agent_valid_mask=tv.bool(), raw_agent_valid_mask=raw.bool(), raw_height=mean_z)
        for n, t in zip(names, tables):
            ag[f'trajectory_token_{n}'] = t
        return data

    @torch.no_grad()
    def _match_agent_token(self, valid_mask: torch.Tensor, pos: torch.Tensor, heading: torch.Tensor,
                           shape: torch.Tensor, token_traj: torch.Tensor,
                           token_traj_all: Optional[torch.Tensor] = None,
                           agent_type: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, list]:
        """valid_mask (A, T) bool, pos (A, T, 2), heading (A, T), shape (A, 2) = (width, length),
        token_traj (A, n_token, 4, 2) per agent - or (n_type, n_token, 4, 2) together with ``agent_type`` (A,), which
        spares the 64 KB per agent copy the reference makes.  ``token_traj_all`` / noise are not supported.
        -> token_index (A, T // shift) int64, token_contour (A, T // shift, 4, 2), []"""
        if token_traj_all is not None:
            raise NotImplementedError('token_traj_all (the reference passes None, preprocess.py:407)')
        if self.noise:
            raise NotImplementedError('noise (np.random top-5 resampling, preprocess.py:624-633) is not reproducible')
        dev = pos.device
        if dev.type != 'cuda':
            raise RuntimeError('TokenProcessor._match_agent_token runs on the GPU only (no CPU fallback)')
        lib = _lib.load()
        A, T = valid_mask.shape
        n_token = token_traj.shape[1]
        valid = valid_mask.to(torch.uint8).contiguous()
        p = pos[..., :2].to(torch.float32).contiguous()
        h = heading.to(torch.float32).contiguous()
        sh = shape[..., :2].to(torch.float32).contiguous()
        tok = token_traj.to(torch.float32).contiguous()
        ty = agent_type.to(torch.int32).contiguous() if agent_type is not None else None
        if ty is None and tok.shape[0] != A:
            raise ValueError('token_traj must hold one table per agent unless agent_type is given')
        n_out = T // self.shift
        idx = torch.empty(A, n_out, dtype=torch.int32, device=dev)
        contour = torch.empty(A, n_out, 4, 2, dtype=torch.float32, device=dev)
        _lib.check(lib.infgen_match_agent_tokens(_lib.ptr(valid), _lib.ptr(p), _lib.ptr(h), _lib.ptr(sh), _lib.ptr(ty),
                                                 _lib.ptr(tok), n_token * 8, A, T, self.shift, n_token, _lib.ptr(idx),
                                                 _lib.ptr(contour), torch.cuda.current_stream(dev).cuda_stream),
                   'infgen_match_agent_tokens')
        return idx.long(), contour, []


@torch.no_grad()
def match_token_map(traj_pos: torch.Tensor, traj_theta: torch.Tensor, token_sample_pt: torch.Tensor) -> torch.Tensor:
    """The matching core of the reference's ``InfGen.match_token_map`` (infgen/model/infgen.py:918-936, noise off) on the
    GPU: traj_pos (P, 3, 2), traj_theta (P,), token_sample_pt (n_token, 3, 2) -> pt_token_id (P,) int64."""
    dev = traj_pos.device
    if dev.type != 'cuda':
        raise RuntimeError('match_token_map runs on the GPU only (no CPU fallback)')
    lib = _lib.load()
    P, n_token = traj_pos.shape[0], token_sample_pt.shape[0]
    tp = traj_pos.to(torch.float32).contiguous()
    th = traj_theta.to(torch.float32).contiguous()
    sp = token_sample_pt.to(device=dev, dtype=torch.float32).contiguous()
    out = torch.empty(P, dtype=torch.int32, device=dev)
    _lib.check(lib.infgen_match_map_tokens(_lib.ptr(tp), _lib.ptr(th), _lib.ptr(sp), P, n_token, _lib.ptr(out),
                                           torch.cuda.current_stream(dev).cuda_stream), 'infgen_match_map_tokens')
    return out.long()


@torch.no_grad()
def fetch_enterings(data, attr_tokenizer, pl2seed_radius: float, enter_state: int = 2, invalid_state: int = 0,
                    predict_occ: bool = False):
    """The reference's ``InfGen._fetch_enterings(self, data)`` (infgen/model/infgen.py:1008-1128) with the attributes it
    reads from ``self`` as arguments.  ``data`` is the (batched) scene dict on the GPU: data['agent'][state_idx, token_pos,
    token_heading, batch, av_index], data['pt_token'][token_idx, position, batch], ``data.num_graphs`` (or the length of
    av_index).  Adds the same keys as the reference to data['agent'] and returns data.  Two launches for all scenes."""
    ag = data['agent']
    dev = ag['token_pos'].device
    if dev.type != 'cuda':
        raise RuntimeError('fetch_enterings runs on the GPU only (no CPU fallback)')
    A, T = ag['state_idx'].shape
    av = torch.as_tensor(ag['av_index'], device=dev).reshape(-1).to(torch.int32).contiguous()
    B = int(getattr(data, 'num_graphs', av.numel()))
    batch = ag['batch'] if 'batch' in ag else torch.zeros(A, dtype=torch.long, device=dev)
    counts = torch.bincount(batch, minlength=B)
    ptr = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    ptr[1:] = torch.cumsum(counts, 0)
    pos = ag['token_pos'].to(torch.float32).contiguous()
    head = ag['token_heading'].to(torch.float32).contiguous()
    st = ag['state_idx'].to(torch.int32).contiguous()
    grid = attr_tokenizer.grid.to(dev, torch.float32).contiguous()
    i32 = lambda *s: torch.empty(*s, dtype=torch.int32, device=dev)
    f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    u8 = lambda *s: torch.empty(*s, dtype=torch.uint8, device=dev)
    cell, hbin, order = i32(A, T), i32(A, T), i32(A, T)
    off, rel, theta = f32(A, T, 2), f32(A, T, 2), f32(A, T)
    near, born = u8(A, T), u8(A, T)
    pt_pos = pt_ptr = pt_cell = None
    M = 0
    if predict_occ:
        pt = data['pt_token']
        pt_pos = pt['position'].to(torch.float32).contiguous()
        M = pt_pos.shape[0]
        pb = pt['batch'] if 'batch' in pt else torch.zeros(M, dtype=torch.long, device=dev)
        pt_ptr = torch.zeros(B + 1, dtype=torch.int32, device=dev)
        pt_ptr[1:] = torch.cumsum(torch.bincount(pb, minlength=B), 0)
        pt_cell = i32(T, M)
    p = _lib.ptr
    _lib.check(_lib.load().infgen_fetch_enterings(
        p(pos), p(head), p(st), p(ptr), p(av), B, int(counts.max()), T, p(grid), grid.shape[0], float(pl2seed_radius),
        float(attr_tokenizer.angle_interval), enter_state, invalid_state, p(cell), p(off), p(hbin), p(order), p(near), p(born),
        p(rel), p(theta), p(pt_pos), pt_pos.shape[1] if pt_pos is not None else 0, p(pt_ptr), M, p(pt_cell),
        torch.cuda.current_stream(dev).cuda_stream), 'infgen_fetch_enterings')
    ag.update(grid_token_idx=cell.long(), grid_offset_xy=off, heading_token_idx=hbin.long(), sort_indices=order.long(),
              inrange_mask=near.bool(), bos_mask=born.bool(), pos_xy=rel, heading_theta=theta)
    if predict_occ:
        ag['pt_grid_token_idx'] = pt_cell.long()
    return data
