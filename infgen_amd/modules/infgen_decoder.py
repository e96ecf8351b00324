"""InfGenDecoder: the drop-in boundary of the hot path (reference
infgen/modules/infgen_decoder.py:15-143).  Same constructor signature, same sub-module names
(``map_encoder`` / ``agent_encoder``), ``forward`` / ``inference`` / ``inference_no_map`` with the
reference's return dicts (``forward``: the teacher-forced open-loop pass, infgen_amd/forward_engine.py).  ``inference`` runs map encoder + closed-loop rollout on the GPU through
libinfgen_hip.so; ``inference_batch`` is the throughput entry (many scenes in lockstep).
"""
from __future__ import annotations

import weakref
from typing import Dict, List, Mapping, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from ..engine import InsertionHeadroomError, PackedWeights, RolloutEngine
from ..synth import RolloutConfig
from .agent_decoder import InfGenAgentDecoder
from .attr_tokenizer import Attr_Tokenizer
from .map_decoder import InfGenMapDecoder

_AGENT_KEYS = ('state_idx', 'valid_mask', 'id', 'raw_agent_valid_mask', 'token_pos', 'token_idx', 'token_heading',
               'shape', 'type', 'grid_token_idx', 'position', 'heading', 'av_index',
               'trajectory_token_veh', 'trajectory_token_ped', 'trajectory_token_cyc')


def _np(v):
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy()
    return np.asarray(v)


def scene_from_data(data) -> Dict[str, Dict[str, np.ndarray]]:
    """the reference's HeteroData-style ``data`` (dict access) -> host scene dict of numpy arrays"""
    ag = data['agent']
    agent = {k: _np(ag[k]) for k in _AGENT_KEYS}
    pt = data['pt_token']
    ptd = {k: _np(pt[k]) for k in ('position', 'orientation', 'type', 'pl_type', 'token_idx')}
    key = ('pt_token', 'to', 'map_polygon')
    try:
        e = data[key]['edge_index']
    except (KeyError, TypeError):
        e = data['pt_token__to__map_polygon']['edge_index']
    return {'agent': agent, 'pt_token': ptd, 'map_polygon': {'light_type': _np(data['map_polygon']['light_type'])},
            'pt_token__to__map_polygon': {'edge_index': _np(e)}}


_FWD_AGENT_KEYS = ('state_idx', 'raw_agent_valid_mask', 'token_pos', 'token_idx', 'token_heading', 'shape', 'type',
                    'grid_token_idx', 'grid_offset_xy', 'heading_token_idx', 'sort_indices', 'pos_xy', 'heading_theta',
                    'pt_grid_token_idx', 'av_index', 'trajectory_token_veh', 'trajectory_token_ped', 'trajectory_token_cyc')


def batch_from_data(data) -> Dict[str, Dict[str, np.ndarray]]:
    """the (batched) ``data`` the reference's forward reads (agent_decoder.py:1108-1127, map_decoder.py:71-93) -> host dict of
    numpy arrays; ``ptr`` defaults to one scene"""
    ag, pt = data['agent'], data['pt_token']
    agent = {k: _np(ag[k]) for k in _FWD_AGENT_KEYS}
    A = agent['state_idx'].shape[0]
    agent['ptr'] = _np(ag['ptr']) if 'ptr' in ag else np.array([0, A], np.int64)
    ptd = {k: _np(pt[k]) for k in ('position', 'orientation', 'type', 'pl_type', 'token_idx')}
    ptd['ptr'] = _np(pt['ptr']) if 'ptr' in pt else np.array([0, ptd['position'].shape[0]], np.int64)
    key = ('pt_token', 'to', 'map_polygon')
    try:
        e = data[key]['edge_index']
    except (KeyError, TypeError):
        e = data['pt_token__to__map_polygon']['edge_index']
    return {'agent': agent, 'pt_token': ptd, 'map_polygon': {'light_type': _np(data['map_polygon']['light_type'])},
            'pt_token__to__map_polygon': {'edge_index': _np(e)}}


def scenes_from_datas(datas) -> List[Dict[str, Dict[str, np.ndarray]]]:
    """``scene_from_data`` for many scenes with ONE device -> host copy per key instead of one per key and scene (a 512-scene
    batch is ~12,000 small synchronous copies otherwise): tensors of a key are concatenated on their device, copied once and
    split into per-scene views on the host"""
    if len(datas) <= 1:
        return [scene_from_data(d) for d in datas]
    key = ('pt_token', 'to', 'map_polygon')

    def edge(d):
        try:
            return d[key]['edge_index']
        except (KeyError, TypeError):
            return d['pt_token__to__map_polygon']['edge_index']

    def split(vals, axis=0):
        """list of equal-rank arrays / tensors -> list of numpy views of one host copy"""
        if not all(isinstance(v, torch.Tensor) for v in vals):
            return [_np(v) for v in vals]
        if vals[0].dim() == 0:
            return list(torch.stack(vals).detach().cpu().numpy())
        sizes = [int(v.shape[axis]) for v in vals]
        host = torch.cat([v.detach() for v in vals], dim=axis).cpu().numpy()
        offs = np.concatenate([[0], np.cumsum(sizes)])
        return [host[offs[i]:offs[i + 1]] if axis == 0 else host[:, offs[i]:offs[i + 1]] for i in range(len(vals))]
    shared = ('trajectory_token_veh', 'trajectory_token_ped', 'trajectory_token_cyc')
    first = {k: _np(datas[0]['agent'][k]) for k in shared}
    cols = {}
    for k in _AGENT_KEYS:
        if k in shared:
            continue
        vals = [d['agent'][k] for d in datas]
        if k == 'av_index':
            vals = [v.reshape(-1) if isinstance(v, torch.Tensor) else np.asarray(v).reshape(-1) for v in vals]
        cols[k] = split(vals)
    pcols = {k: split([d['pt_token'][k] for d in datas]) for k in ('position', 'orientation', 'type', 'pl_type', 'token_idx')}
    light = split([d['map_polygon']['light_type'] for d in datas])
    edges = split([edge(d) for d in datas], axis=1)
    out = []
    for i in range(len(datas)):
        agent = {k: cols[k][i] for k in cols}
        agent.update(first)
        out.append({'agent': agent, 'pt_token': {k: pcols[k][i] for k in pcols}, 'map_polygon': {'light_type': light[i]},
                    'pt_token__to__map_polygon': {'edge_index': edges[i]}})
    return out


_STACK_AGENT_KEYS = ('state_idx', 'valid_mask', 'id', 'raw_agent_valid_mask', 'token_pos', 'token_idx', 'token_heading', 'shape',
                     'type', 'grid_token_idx', 'position', 'heading', 'av_index')
_STACK_PT_KEYS = ('position', 'orientation', 'type', 'pl_type', 'token_idx')


def stack_datas(datas, min_scenes: int = 8, any_device: bool = False):
    """a batch of ``data`` objects whose arrays are DEVICE tensors of one shape -> one stacked device tensor per key (what
    ``RolloutEngine.reload_device`` takes), or None when the batch is not of that kind (host arrays, ragged shapes, several
    devices, fewer than ``min_scenes`` scenes): then ``scenes_from_datas`` + the host setup take it.  No host copy.
    ``any_device``: CPU tensors count too (the CPU test of the device-side setup)."""
    if len(datas) < min_scenes:
        return None
    key = ('pt_token', 'to', 'map_polygon')

    def edge(d):
        try:
            return d[key]['edge_index']
        except (KeyError, TypeError):
            return d['pt_token__to__map_polygon']['edge_index']

    def stack(vals):
        v0 = vals[0]
        if not all(isinstance(v, torch.Tensor) and v.device == v0.device and v.shape == v0.shape for v in vals):
            raise TypeError('not a one-shape batch of tensors on one device')
        return torch.stack(vals)
    try:
        agent = {k: stack([d['agent'][k].reshape(-1)[:1] if k == 'av_index' else d['agent'][k] for d in datas])
                 for k in _STACK_AGENT_KEYS}
        if agent['state_idx'].device.type != 'cuda' and not any_device:
            return None
        return {'agent': agent, 'pt_token': {k: stack([d['pt_token'][k] for d in datas]) for k in _STACK_PT_KEYS},
                'light_type': stack([d['map_polygon']['light_type'] for d in datas]),
                'edge_index': stack([edge(d) for d in datas])}
    except (KeyError, TypeError, AttributeError, RuntimeError):
        return None


class _LazyScenes:
    """``scenes_from_datas(datas)``, made when first read (the device-side reload of a reused engine never reads it)"""

    def __init__(self, datas):
        self._datas, self._scenes = datas, None

    def _get(self):
        if self._scenes is None:
            self._scenes = scenes_from_datas(self._datas)
        return self._scenes

    def __len__(self):
        return len(self._datas)

    def __getitem__(self, i):
        return self._get()[i]

    def __iter__(self):
        return iter(self._get())


class InfGenDecoder(nn.Module):

    def __init__(self, decoder_type: str, dataset: str, input_dim: int, hidden_dim: int, num_historical_steps: int,
                 pl2pl_radius: float, time_span: Optional[int], pl2a_radius: float, pl2seed_radius: float,
                 a2a_radius: float, a2sa_radius: float, pl2sa_radius: float, num_freq_bands: int, num_map_layers: int,
                 num_agent_layers: int, num_heads: int, head_dim: int, dropout: float, map_token: Dict,
                 token_size=512, attr_tokenizer: Attr_Tokenizer = None, predict_motion: bool = False,
                 predict_state: bool = False, predict_map: bool = False, predict_occ: bool = False,
                 use_grid_token: bool = True, use_head_token: bool = True, use_state_token: bool = True,
                 disable_insertion: bool = False, state_token: Dict[str, int] = None, seed_size: int = 5,
                 buffer_size: int = 32, num_recurrent_steps_val: int = -1, loss_weight: dict = None, logger=None) -> None:
        super().__init__()
        if decoder_type != 'agent_decoder':
            raise ValueError(f'Unsupport decoder type: {decoder_type} (the HIP path implements agent_decoder)')
        self.map_encoder = InfGenMapDecoder(dataset=dataset, input_dim=input_dim, hidden_dim=hidden_dim,
                                            num_historical_steps=num_historical_steps, pl2pl_radius=pl2pl_radius,
                                            num_freq_bands=num_freq_bands, num_layers=num_map_layers, num_heads=num_heads,
                                            head_dim=head_dim, dropout=dropout, map_token=map_token)
        self.agent_encoder = InfGenAgentDecoder(
            dataset=dataset, input_dim=input_dim, hidden_dim=hidden_dim, num_historical_steps=num_historical_steps,
            time_span=time_span, pl2a_radius=pl2a_radius, pl2seed_radius=pl2seed_radius, a2a_radius=a2a_radius,
            a2sa_radius=a2sa_radius, pl2sa_radius=pl2sa_radius, num_freq_bands=num_freq_bands, num_layers=num_agent_layers,
            num_heads=num_heads, head_dim=head_dim, dropout=dropout, token_size=token_size, attr_tokenizer=attr_tokenizer,
            predict_motion=predict_motion, predict_state=predict_state, predict_map=predict_map, predict_occ=predict_occ,
            state_token=state_token, use_grid_token=use_grid_token, use_head_token=use_head_token,
            use_state_token=use_state_token, disable_insertion=disable_insertion, seed_size=seed_size,
            buffer_size=buffer_size, num_recurrent_steps_val=num_recurrent_steps_val, loss_weight=loss_weight, logger=logger)
        ref = weakref.ref(self)
        self.map_encoder._owner = ref
        self.agent_encoder._owner = ref
        self.map_enc = None
        self.predict_motion, self.predict_state, self.predict_map, self.predict_occ = predict_motion, predict_state, predict_map, predict_occ
        self.data_keys = ["agent_valid_mask", "category", "valid_mask", "av_index", "scenario_id", "shape"]
        self._cfg_kw = dict(input_dim=input_dim, hidden_dim=hidden_dim, num_heads=num_heads, head_dim=head_dim,
                            num_freq_bands=num_freq_bands, num_map_layers=num_map_layers, num_agent_layers=num_agent_layers,
                            num_historical_steps=num_historical_steps, token_size=token_size, a2a_radius=a2a_radius,
                            pl2a_radius=pl2a_radius, pl2pl_radius=pl2pl_radius, a2sa_radius=a2sa_radius,
                            pl2sa_radius=pl2sa_radius, pl2seed_radius=pl2seed_radius,
                            time_span=time_span if time_span is not None else num_historical_steps,
                            seed_size=seed_size, buffer_size=buffer_size, disable_insertion=disable_insertion,
                            state_token=dict(state_token))
        # arithmetic of the rollout's GEMM kernels - the counterpart of the trainer's `precision` flag, which the reference's run.py
        # never passes (fp32): '32' (default) the fp32-accurate operand split; 'bf16' bf16 operands with fp32 accumulation (packs of
        # bf16 weights, InfgenOptions.gemm_terms = 2; BASELINE config C5); '16' fp16 operands (gemm_terms = 1).  Set it before a call.
        self.rollout_precision = '32'
        self._packed = None
        self._param_dicts = None
        self._last_w = None
        self._packed_ver = None
        self._engines = {}          # RolloutEngine per batch layout, reused across calls (RolloutEngine.reload)

    _PRECISIONS = {'32': None, 'bf16': {'gemm_terms': 2}, '16': {'gemm_terms': 1}}

    # ------------------------------------------------------------------ weights
    def _weights(self) -> PackedWeights:
        # every call checks that the packed copy still belongs to the module's parameters (their versions and storage); walking the
        # module tree for that took 3 ms of a 14 ms single-scene call, so the per-module parameter dicts are collected once (the
        # tree is built in __init__; a Parameter that is replaced inside its module is still seen - the dicts are read every call)
        if self._param_dicts is None:
            self._param_dicts = [m._parameters for m in self.modules() if m._parameters]
        ps = [p for d in self._param_dicts for p in d.values() if p is not None]
        dev = ps[0].device
        if dev.type != 'cuda':
            raise _lib.InfgenHipError('InfGenDecoder must live on a cuda device: the product path has no CPU fallback')
        tok = self.agent_encoder.attr_tokenizer
        R = self.agent_encoder.num_recurrent_steps_val
        # the rollout length and the tokenizer geometry are part of the key: changing num_recurrent_steps_val between calls
        # (80 -> 300 for long-term validation) must not be ignored
        prec = str(self.rollout_precision)
        if prec not in self._PRECISIONS:
            raise ValueError(f'rollout_precision must be one of {sorted(self._PRECISIONS)}, not {prec!r}')
        ver = (dev, tuple(p._version for p in ps), tuple(p.data_ptr() for p in ps), R, tok.grid_range, tok.grid_interval,
               tok.angle_interval, prec)
        if self._packed_ver != ver:
            cfg = RolloutConfig(num_recurrent_steps_val=R if R != -1 else 80, grid_range=tok.grid_range,
                                grid_interval=tok.grid_interval, angle_interval=tok.angle_interval, **self._cfg_kw)
            sd = {k: v.detach().cpu().numpy() for k, v in self.state_dict().items()}
            self._packed = PackedWeights(sd, cfg, dev, operand_bits=8 if prec == 'bf16' else 11)
            self._packed_ver = ver
            self._engines = {}
        return self._packed

    # ------------------------------------------------------------------ driver
    def _run(self, data, x_pt=None, map_only=False, batch: Optional[Sequence] = None, sample_uniforms=None,
             batch_seed_outputs: bool = False, copies: int = 1):
        ae = self.agent_encoder
        datas = list(batch) if batch is not None else [data]
        copies = int(copies)
        # copies = n: every scene of the batch is decoded n times in lockstep over ONE map encoding (RolloutEngine(copies=n));
        # the result list holds scene 0's n rollouts, then scene 1's, ...
        # a batch of device tensors of one shape is set up on the device (RolloutEngine.reload_device); its host form is only made
        # when something reads it (a new engine, a filtered row, the host-side outputs)
        stk = stack_datas(datas) if batch is not None and copies == 1 else None
        scenes = _LazyScenes(datas) if stk is not None else scenes_from_datas(datas)
        w = self._last_w = self._weights()
        ag0 = datas[0]['agent']
        if ae.num_recurrent_steps_val == -1:
            # sticky like the reference (agent_decoder.py:1633-1635)
            ae.num_recurrent_steps_val = int(ag0['position'].shape[1]) - ae.num_historical_steps
            w.cfg.num_recurrent_steps_val = ae.num_recurrent_steps_val
        vocab = {k: _np(ag0[f'trajectory_token_{k}']) for k in ('veh', 'ped', 'cyc')}
        map_vocab = _np(self.map_encoder.map_token['traj_src']).astype(np.float32)
        grid = ae.attr_tokenizer.grid.detach().cpu().numpy()
        xo = None
        if x_pt is not None:
            xo = [x_pt] if batch is None else list(x_pt)
        import os
        w.cfg.disable_insertion = bool(ae.disable_insertion)
        k = int(getattr(ae, 'motion_beam_size', 1))
        if k > 1 and sample_uniforms is None and not map_only:
            # stochastic decode like the reference's default (top-k multinomial), driven by torch's RNG: one uniform per decode
            # step and agent row - with insertion on for every row a scene can ever hold (a row without its own uniform would
            # be decoded greedily)
            amax = max(int(d_['agent']['state_idx'].shape[0]) for d_ in datas)
            ucols = amax if w.cfg.disable_insertion else int(_lib.load().infgen_layout_query(_lib.Q_MAX_AGENTS))
            sample_uniforms = torch.rand(w.cfg.num_decode_steps, len(scenes) * copies, ucols).numpy()
        ik = int(getattr(ae, 'insert_beam_size', 1))
        insert_uniforms = None
        if ik > 1 and not w.cfg.disable_insertion and not map_only:
            # the cell of an inserted agent from the insert_beam_size most probable ones (agent_decoder.py:1900-1904), torch's RNG
            insert_uniforms = torch.rand(w.cfg.num_decode_steps, 10, len(scenes) * copies).numpy()
        def make_engine(headroom=None):
            return RolloutEngine(w, scenes, vocab, map_vocab, grid, x_pt_override=xo, insert_headroom=headroom,
                                 force_enter=bool(int(os.getenv('DEBUG', 0))),    # DEBUG=1 forces 'enter' (agent_decoder.py:1888)
                                 sample_k=k if not map_only else 1, sample_uniforms=sample_uniforms,
                                 insert_k=ik if insert_uniforms is not None else 1, insert_uniforms=insert_uniforms,
                                 # the seed node's per-insertion outputs (plot inputs of the reference, 5.5 MB per scene): the
                                 # single-scene entry and the n-copies batch of inference_rollouts; the throughput entry
                                 # (inference_batch) returns the zero arrays the reference initialises them to
                                 seed_outputs=(batch is None or batch_seed_outputs) and not w.cfg.disable_insertion and not map_only,
                                 copies=copies, options=self._PRECISIONS[str(self.rollout_precision)])
        # one engine per batch layout is kept across calls: a second call of the same shape re-uploads the scene arrays into
        # the first call's device buffers instead of building (and allocating) an engine again
        ekey = (len(scenes), PackedWeights.tables_key(*(vocab[k_] for k_ in ('veh', 'ped', 'cyc')), grid, map_vocab),
                bool(w.cfg.disable_insertion), w.cfg.num_recurrent_steps_val, k if not map_only else 1,
                ik if insert_uniforms is not None else 1, bool(int(os.getenv('DEBUG', 0))), batch is None, map_only, xo is None,
                bool(batch_seed_outputs), copies)
        eng = self._engines.get(ekey)
        if (stk is not None and eng is not None and eng.fits_device(stk) and
                eng.reload_device(stk, scenes, sample_uniforms=sample_uniforms, insert_uniforms=insert_uniforms, x_pt_override=xo)):
            pass
        elif eng is not None and eng.fits(scenes):
            eng.reload(scenes, sample_uniforms=sample_uniforms, insert_uniforms=insert_uniforms, x_pt_override=xo)
        else:
            eng = make_engine()
            if len(self._engines) >= 2:
                self._engines.pop(next(iter(self._engines)))
            self._engines[ekey] = eng
        if map_only:
            eng.prologue(map_only=True)
            return eng.x_pt[:eng.hosts[0]['M']].clone()
        # the reference's agent arrays grow without bound; here rows are pre-allocated per scene.  If the inserted agents
        # outgrow them the (deterministic) rollout is repeated with twice the rows instead of dropping insertions
        while True:
            try:
                eng.rollout()
                break
            except InsertionHeadroomError as e:
                amax = max(h['A'] for h in eng.hosts)
                limit = eng.lib.infgen_layout_query(_lib.Q_MAX_AGENTS)
                if eng.A_cap >= limit:
                    raise
                eng = self._engines[ekey] = make_engine(headroom=min(2 * eng.A_cap, limit) - amax)
        # per-scene dicts of device tensors (no host round trip of the results), detached from the engine's buffers
        outs = eng.outputs_device(detach=True)
        dev = w.device
        res = []
        steps = w.cfg.num_decode_steps
        G = ae.grid_size
        zero = {}                              # shared (read-only) zero tensors of the seed outputs a batch does not record

        def z(*shape):
            if shape not in zero:
                zero[shape] = torch.zeros(*shape, device=dev)
            return zero[shape]
        T_cols = w.cfg.num_columns
        if copies > 1:                          # scene i's copies are adjacent in the engine's batch
            datas = [d_ for d_ in datas for _ in range(copies)]
        for i_, (d, o) in enumerate(zip(datas, outs)):
            r = o                                   # (a LazyOut: per-scene views are cut when a key is read, not here)
            # without insertion (or in the batched entry) these stay what the reference initialises them to (:1746-1750, :1730)
            for k_, shp_ in (('next_state_prob_seed', (11, steps)), ('next_pos_rel_prob_seed', (11, steps, G)),
                             ('grid_agent_occ_seed', (11, steps, G)), ('grid_pt_occ_seed', (11, steps, G)),
                             ('grid_agent_occ_gt_seed', (11, steps, G))):
                if k_ not in r:
                    r[k_] = z(*shp_)
            if 'agent_labels' not in r:
                r.set_lazy('agent_labels', (lambda n=eng.hosts[i_]['A'] + o['num_inserted']: [[None] * T_cols for _ in range(n)]))
            r['log_message'] = ('No agents inserted!' if o['num_inserted'] == 0 else
                                f"Number of total inserted agents: {o['num_inserted']}")
            # the callee mutates data['batch_size_a'] like the reference (agent_decoder.py:1649)
            try:
                filt = eng.hosts[i_]['filt']
                if not filt.all():
                    av0 = int(np.asarray(scenes[i_ // copies]['agent']['av_index']).reshape(-1)[0])
                    removed = int((~filt[:av0]).sum())
                    if removed and i_ % copies == 0:          # (once per data object)
                        d['batch_size_a'] -= removed
            except (KeyError, TypeError):
                pass
            res.append(r)
        return res if batch is not None else res[0]

    def get_agent_inputs(self, data):
        raise NotImplementedError('training-only helper (agent_decoder.py:933) — out of the hot path')

    @torch.no_grad()
    def forward(self, data) -> Dict[str, torch.Tensor]:
        """map encoder + teacher-forced forward over every token column of the batch (reference infgen_decoder.py:114-121,
        agent_decoder.py:1104-1603) on the GPU (infgen_amd/forward_engine.py).  Evaluation only (no autograd
        through the HIP kernels).  The candidate rows of the refine stage and the neighbour-grid evaluation masks are drawn
        with ``torch.randperm`` from torch's CPU generator in the reference's order."""
        from ..forward_engine import ForwardEngine
        w = self._weights()                 # (raises on a CPU module: no CPU fallback)
        batch = batch_from_data(data)
        vocab = {k: batch['agent'][f'trajectory_token_{k}'] for k in ('veh', 'ped', 'cyc')}
        map_vocab = _np(self.map_encoder.map_token['traj_src']).astype(np.float32)
        grid = self.agent_encoder.attr_tokenizer.grid.detach().cpu().numpy()
        out = ForwardEngine(w, batch, vocab, map_vocab, grid).run()
        dev = w.device
        map_enc = {'map_next_token_idx': torch.zeros(0, 10, dtype=torch.long, device=dev),
                   'map_next_token_prob': torch.zeros(0, self.map_encoder.token_size, device=dev),
                   'map_next_token_idx_gt': torch.zeros(0, dtype=torch.long, device=dev),
                   'map_next_token_eval_mask': torch.zeros(0, dtype=torch.bool, device=dev)}
        return {**map_enc, **out, **{k: data[k] for k in self.data_keys if k in data}}

    @torch.no_grad()
    def inference(self, data, sample_uniforms=None) -> Dict[str, torch.Tensor]:
        """map encoder + closed-loop rollout of one scene (reference infgen_decoder.py:123-130).
        Greedy unless ``agent_encoder.motion_beam_size > 1``; then tokens are drawn by inverse CDF over the
        top-k probabilities with ``sample_uniforms`` ([steps][1][A]) or torch.rand when omitted."""
        r = self._run(data, sample_uniforms=sample_uniforms)
        x_pt = r.pop('x_pt')
        map_enc = {'x_pt': x_pt, 'map_next_token_idx': torch.zeros(0, 10, dtype=torch.long, device=x_pt.device),
                   'map_next_token_prob': torch.zeros(0, self.map_encoder.token_size, device=x_pt.device),
                   'map_next_token_idx_gt': torch.zeros(0, dtype=torch.long, device=x_pt.device),
                   'map_next_token_eval_mask': torch.zeros(0, dtype=torch.bool, device=x_pt.device)}
        return r.merged(first=map_enc, last={k: data[k] for k in self.data_keys if k in data})

    @torch.no_grad()
    def inference_no_map(self, data, map_enc) -> Dict[str, torch.Tensor]:
        r = self._run(data, x_pt=map_enc['x_pt'])
        r.pop('x_pt')
        return r.merged(first=map_enc)

    @torch.no_grad()
    def inference_rollouts(self, data, n: int) -> List[Dict[str, torch.Tensor]]:
        """``n`` independent rollouts of ONE scene (the reference's ``n_rollout_close_val`` loop, infgen/model/infgen.py:704-706,
        which calls ``inference(data.clone())`` n times) as one batch of n copies decoded in lockstep: with
        ``motion_beam_size`` / ``insert_beam_size`` > 1 every copy draws its own uniforms from torch's RNG, so the results are n
        samples; greedy copies are identical.  ``data`` itself is not mutated (the copies are)."""
        # one engine batch of n copies of the scene over ONE map encoding (RolloutEngine(copies=n): the map-token graph, the map
        # encoder and the map K / V rows exist once - the reference offers inference_no_map(data, map_enc) for the same purpose);
        # every rollout carries what ``inference`` returns for it: the seed node's outputs and the map_next_token_* keys too
        return self.inference_batch([data.clone() if hasattr(data, 'clone') else dict(data)], seed_outputs=True, copies=int(n))

    @torch.no_grad()
    def inference_batch(self, datas: Sequence, seed_outputs: bool = False, copies: int = 1) -> List[Dict[str, torch.Tensor]]:
        """throughput entry: many independent scenes decoded in lockstep on this GPU.  Every dict has the key set of
        ``inference``; the seed node's per-insertion arrays (``*_seed``) are recorded only with ``seed_outputs=True`` (5.5 MB per
        scene), otherwise they are the zero arrays the reference initialises them to (agent_decoder.py:1746-1750)."""
        rs = self._run(None, batch=datas, batch_seed_outputs=seed_outputs, copies=copies)
        if copies > 1:                          # ``copies`` rollouts per scene over one map encoding: scene 0's first, then scene 1's ...
            datas = [d for d in datas for _ in range(int(copies))]
        out = []
        dev, ts = self._last_w.device, self.map_encoder.token_size
        map_keys = {'map_next_token_idx': torch.zeros(0, 10, dtype=torch.long, device=dev),
                    'map_next_token_prob': torch.zeros(0, ts, device=dev),
                    'map_next_token_idx_gt': torch.zeros(0, dtype=torch.long, device=dev),
                    'map_next_token_eval_mask': torch.zeros(0, dtype=torch.bool, device=dev)}
        for d, r in zip(datas, rs):
            out.append(r.merged(first=map_keys, last={k: d[k] for k in self.data_keys if k in d}))
        return out
