"""``InfGen``: the caller of the hot path, mirroring the reference's LightningModule (infgen/model/infgen.py) for the
close-loop validation branch - ``__init__(model_config, save_path, logger)``, ``set(mode)``, ``load_state_from_file``,
``match_token_map``, ``sample_pt_pred``, ``_fetch_enterings`` and ``validation_step`` (:573-842):

    raw scene -> TokenProcessor -> match_token_map -> _fetch_enterings -> InfGenDecoder.inference (the rollout)
              -> rollouts dict (pickled like the reference and / or kept on the GPU) -> MetricFeatures

Every stage runs in the HIP library; this class only sequences them.  It is a plain ``nn.Module`` (pytorch_lightning is
not a dependency): ``self.log`` / trainer hooks are not provided, ``global_rank`` comes from torch.distributed.
The open-loop branch (``val_open_loop`` / ``OPEN_LOOP=1``, reference :627-686) runs ``InfGenDecoder.forward`` - the teacher-forced pass
of SURVEY section 8f rank 3 (infgen_amd/forward_engine.py) - and leaves the token + state cross-entropy in ``self.val_loss``;
training (autograd) is not built.

The reference reads its token tables from ``infgen/tokens/*.pkl`` (data of that repository).  Here they are arguments:
``map_token_traj`` (n_token, 11, 2) or ``map_token_traj_path`` (pickle with ['traj_src']), and ``agent_tokens`` /
``agent_token_path`` for the TokenProcessor.
"""
import os
import pickle
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from ..metrics import compute_metrics
from ..modules import Attr_Tokenizer, InfGenDecoder, TokenProcessor, fetch_enterings
from ..modules.token_processor import match_token_map as _match_core


def _get(cfg, name, default=None):
    return getattr(cfg, name) if hasattr(cfg, name) else default


class InfGen(nn.Module):

    def __init__(self, model_config, save_path: os.PathLike = "", logger=None, **kwargs) -> None:
        super().__init__()
        mc, dc = model_config, model_config.decoder
        self.model_config = mc
        self.dataset, self.input_dim, self.hidden_dim = mc.dataset, mc.input_dim, mc.hidden_dim
        self.num_historical_steps, self.num_freq_bands = mc.num_historical_steps, mc.num_freq_bands
        self.save_path, self.local_logger = save_path, logger
        self.noise = True                     # reference :46 (random top-8 resampling of the map tokens)
        self.max_epochs = kwargs.get('max_epochs', 0)
        self._map_token_traj = kwargs.get('map_token_traj')
        self.map_token_traj_path = kwargs.get('map_token_traj_path')
        self.init_map_token()
        self.predict_motion, self.predict_state = mc.predict_motion, mc.predict_state
        self.predict_map, self.predict_occ = mc.predict_map, mc.predict_occ
        self.pl2seed_radius, self.token_size = dc.pl2seed_radius, dc.token_size
        self.use_grid_token = not _get(mc, 'disable_grid_token', False)
        if not self.use_grid_token:
            self.predict_occ = False
        self.use_head_token = not _get(mc, 'disable_head_token', False)
        self.use_state_token = not _get(mc, 'disable_state_token', False)
        self.disable_insertion = bool(_get(mc, 'disable_insertion', False))
        st = mc.state_token
        self.token_processer = TokenProcessor(self.token_size, training=self.training, predict_motion=self.predict_motion,
                                              predict_state=self.predict_state, predict_map=self.predict_map, state_token=st,
                                              pl2seed_radius=self.pl2seed_radius, agent_tokens=kwargs.get('agent_tokens'),
                                              agent_token_path=kwargs.get('agent_token_path'))
        self.token_processer.materialize_token_traj_all = False       # the rollout reads the three type tables
        self.attr_tokenizer = Attr_Tokenizer(grid_range=mc.grid_range, grid_interval=mc.grid_interval,
                                             radius=dc.pl2seed_radius, angle_interval=mc.angle_interval)
        self.invalid_state, self.valid_state = int(st['invalid']), int(st['valid'])
        self.enter_state, self.exit_state = int(st['enter']), int(st['exit'])
        self.seed_size = int(dc.seed_size)
        self.encoder = InfGenDecoder(
            decoder_type=mc.decoder_type, dataset=mc.dataset, input_dim=mc.input_dim, hidden_dim=mc.hidden_dim,
            num_historical_steps=mc.num_historical_steps, num_freq_bands=mc.num_freq_bands, num_heads=mc.num_heads,
            head_dim=mc.head_dim, dropout=mc.dropout, num_map_layers=dc.num_map_layers, num_agent_layers=dc.num_agent_layers,
            pl2pl_radius=dc.pl2pl_radius, pl2a_radius=dc.pl2a_radius, pl2seed_radius=dc.pl2seed_radius,
            a2a_radius=dc.a2a_radius, a2sa_radius=dc.a2sa_radius, pl2sa_radius=dc.pl2sa_radius, time_span=dc.time_span,
            map_token={'traj_src': self.map_token['traj_src']}, token_size=self.token_size,
            attr_tokenizer=self.attr_tokenizer, predict_motion=self.predict_motion, predict_state=self.predict_state,
            predict_map=self.predict_map, predict_occ=self.predict_occ, state_token=st, use_grid_token=self.use_grid_token,
            use_head_token=self.use_head_token, use_state_token=self.use_state_token,
            disable_insertion=self.disable_insertion, seed_size=self.seed_size, buffer_size=dc.buffer_size,
            num_recurrent_steps_val=mc.num_recurrent_steps_val, loss_weight=_get(mc, 'loss_weight'), logger=logger)
        self.val_open_loop = bool(_get(mc, 'val_open_loop', False))
        self.loss_weight = _get(mc, 'loss_weight')
        self.val_close_loop = bool(_get(mc, 'val_close_loop', True))
        self.n_rollout_close_val = int(_get(mc, 'n_rollout_close_val', 1))
        self._mode = 'training'
        self._long_metrics = None
        self._online_metric = self._save_validate_reuslts = self._plot_rollouts = False
        self.scenario_rollouts, self.scenario_features = [], []

    # ------------------------------------------------------------------ reference :188-215
    def set(self, mode: str = 'train'):
        self._mode = mode
        if mode == 'validation':
            self._online_metric = self._save_validate_reuslts = True
        elif mode == 'test':
            self._save_validate_reuslts = True
        elif mode == 'plot_rollouts':
            raise NotImplementedError('plotting is not part of the HIP path')

    def init_map_token(self):
        src = self._map_token_traj
        if src is None:
            if not self.map_token_traj_path:
                raise ValueError('InfGen needs map_token_traj or map_token_traj_path (the reference reads tokens/map_traj_token5.pkl)')
            with open(self.map_token_traj_path, 'rb') as f:
                src = pickle.load(f)['traj_src']
        src = np.asarray(src, dtype=np.float32)
        self.argmin_sample_len = 3
        idx = torch.linspace(0, src.shape[1] - 1, steps=self.argmin_sample_len).long()
        end = np.arctan2(src[:, -1, 1] - src[:, -2, 1], src[:, -1, 0] - src[:, -2, 0])
        self.map_token = {'traj_src': torch.from_numpy(src), 'sample_pt': torch.from_numpy(src[:, idx.numpy()]).float(),
                          'traj_end_theta': torch.from_numpy(end).float()}

    def get_agent_inputs(self, data):
        return self.encoder.get_agent_inputs(data)

    def forward(self, data):
        """reference infgen/model/infgen.py:217-219"""
        return self.encoder(data)

    # ------------------------------------------------------------------ reference :875-916
    def load_state_from_file(self, filename, to_cpu=False):
        if not os.path.isfile(filename):
            raise FileNotFoundError
        checkpoint = torch.load(filename, map_location=torch.device('cpu') if to_cpu else None, weights_only=False)
        disk, model = checkpoint['state_dict'], self.state_dict()
        keep = {k: v for k, v in disk.items() if k in model and v.shape == model[k].shape}
        for k, v in disk.items():
            if k not in keep:
                print(f'Ignore key in disk ({"not found in model" if k not in model else "shape does not match"}): {k}, '
                      f'shape={tuple(v.shape)}')
        missing, unexpected = self.load_state_dict(keep, strict=False)
        if self.local_logger is not None:
            self.local_logger.info(f'Missing keys: {missing}')
            self.local_logger.info('==> Done (total keys %d)' % len(model))
        return checkpoint.get('it', 0.0), checkpoint.get('epoch', -1)

    # ------------------------------------------------------------------ reference :918-984
    @torch.no_grad()
    def match_token_map(self, data):
        ms, pt = data['map_save'], data['pt_token']
        traj_pos, traj_theta = ms['traj_pos'].to(torch.float), ms['traj_theta'].to(torch.float)
        dev = traj_pos.device
        pl_idx = ms['pl_idx_list'].long()
        sample_pt = self.map_token['sample_pt'].to(dev)
        token_id = _match_core(traj_pos, traj_theta, sample_pt)
        if self.noise:        # not reproducible in the reference either (torch.randint): one of the 8 nearest tokens
            cos, sin = traj_theta.cos(), traj_theta.sin()
            rot = torch.stack([torch.stack([cos, -sin], -1), torch.stack([sin, cos], -1)], -2)
            local = torch.bmm(traj_pos - traj_pos[:, 0:1], rot)
            near = ((sample_pt[None] - local[:, None]) ** 2).sum((-2, -1)).argsort(1)[:, :8]
            token_id = near.gather(1, torch.randint(0, 8, (near.shape[0], 1), device=dev))[:, 0]
        P = traj_pos.shape[0]
        token2pl = torch.stack([torch.arange(P, device=dev), pl_idx])
        pls, inv = torch.unique(pl_idx, return_inverse=True)
        side = pt['side'].long().clamp(0, 2)
        known = (pt['side'] >= 0) & (pt['side'] <= 2)
        counts = torch.zeros(pls.numel() * 3, device=dev).index_add_(0, (inv * 3 + side)[known], torch.ones(int(known.sum()), device=dev))
        counts = counts.reshape(-1, 3)
        width = int(counts.max().item())
        pt['traj_mask'] = torch.arange(width, device=dev)[None, None, :] < counts[..., None]
        pt['position'] = torch.cat([traj_pos[:, 0, :], torch.zeros(pt['num_nodes'], 1, device=dev)], -1)
        pt['orientation'] = traj_theta.clone()
        pt['height'] = pt['position'][:, -1]
        data[('pt_token', 'to', 'map_polygon')] = {'edge_index': token2pl}
        pt['token_idx'] = token_id
        return data

    # ------------------------------------------------------------------ reference :986-1006 (random masks, map pre-training)
    @torch.no_grad()
    def sample_pt_pred(self, data):
        tm = data['pt_token']['traj_mask']
        n_pl, n_side, n_pt = tm.shape
        k = (n_pt - 1) // 3
        raw = torch.arange(1, n_pt, device=tm.device).repeat(n_pl, n_side, 1)
        pick = raw.reshape(-1)[torch.randperm(raw.numel(), device=tm.device)[:n_pl * n_side * k]].reshape(n_pl, n_side, k)
        pick = pick.sort(-1)[0]
        valid = tm.clone().scatter_(2, pick, False)
        pred = tm.clone().scatter_(2, pick, False)
        keep = torch.ones_like(tm).scatter_(2, pick - 1, False)
        pred.masked_fill_(keep, False)
        pred = pred & torch.roll(tm, shifts=-1, dims=2)
        target = torch.roll(pred, shifts=1, dims=2)
        pt = data['pt_token']
        pt['pt_valid_mask'], pt['pt_pred_mask'], pt['pt_target_mask'] = valid[tm], pred[tm], target[tm]
        return data

    # ------------------------------------------------------------------ reference :1008-1128
    def _fetch_enterings(self, data, plot: bool = False):
        return fetch_enterings(data, self.attr_tokenizer, self.pl2seed_radius, self.enter_state, self.invalid_state,
                               self.predict_occ)

    # ------------------------------------------------------------------ reference :573-842, close-loop branch
    @torch.no_grad()
    def validation_step(self, data, batch_idx):
        rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
        rollouts_path = os.path.join(self.save_path, f'idx_{rank}_{batch_idx}_rollouts.pkl')
        if self._save_validate_reuslts and os.path.exists(rollouts_path):
            return
        data = self.token_processer(data)
        data = self.match_token_map(data)
        data = self.sample_pt_pred(data)
        data = self._fetch_enterings(data)
        ag, pt = data['agent'], data['pt_token']
        dev = ag['token_pos'].device
        if 'ptr' not in ag:
            ag['ptr'] = torch.tensor([0, ag['token_pos'].shape[0]], device=dev)
        if 'ptr' not in pt:
            pt['ptr'] = torch.tensor([0, pt['position'].shape[0]], device=dev)
        data['batch_size_a'] = ag['ptr'][1:] - ag['ptr'][:-1]
        data['batch_size_pl'] = pt['ptr'][1:] - pt['ptr'][:-1]
        if self.val_open_loop or int(os.getenv('OPEN_LOOP', 0)):
            # reference :627-686: teacher-forced forward, token + state cross-entropies as 'val_loss' (the occupancy plots of
            # that branch are not part of the HIP path)
            if isinstance(ag['av_index'], torch.Tensor) and ag['av_index'].numel() > 1:
                ag['av_index'] = ag['av_index'] + ag['ptr'][:-1]                    # batched graphs (:610-611)
            pred = self(data)
            self.open_loop_pred = pred
            loss = torch.zeros((), device=pred['next_token_prob'].device)
            if self.predict_motion:
                m_ = pred['next_token_eval_mask']
                loss = loss + torch.nn.functional.cross_entropy(pred['next_token_prob'][m_], pred['next_token_idx_gt'][m_],
                                                                label_smoothing=0.1)
            if self.predict_state:
                m_ = pred['next_state_eval_mask']
                sw = torch.tensor(self.loss_weight['state_weight'], device=loss.device) if self.loss_weight else None
                loss = loss + torch.nn.functional.cross_entropy(pred['next_state_prob'][m_], pred['next_state_idx_gt'][m_], weight=sw)
            self.val_loss = loss
            if not (self.val_close_loop and (self.predict_motion or self.predict_state)):
                return loss
        if not (self.val_close_loop and (self.predict_motion or self.predict_state)):
            return
        if self.n_rollout_close_val > 1:
            # the reference's loop (:704-706) as one batch of n copies of the scene with their own sampling uniforms
            self.last_rollouts = self.encoder.inference_rollouts(data, self.n_rollout_close_val)
            rollout = self.last_rollouts[-1]
        else:
            rollout = self.encoder.inference(data.clone() if hasattr(data, 'clone') else data)
            self.last_rollouts = [rollout]
        rollouts = [rollout]                                   # the reference appends outside its loop (:704-706): the last one
        if not (self._online_metric or self._save_validate_reuslts):
            return rollout
        formatted = compute_metrics.format_rollouts(data, rollouts)
        if self._save_validate_reuslts:
            os.makedirs(self.save_path or '.', exist_ok=True)
            with open(rollouts_path, 'wb') as f:
                pickle.dump({k: v.cpu() if torch.is_tensor(v) else v for k, v in formatted.items()}, f)
        if self._online_metric:
            sims = compute_metrics.output_to_rollouts(formatted)
            self.scenario_rollouts.extend(sims)
            feats = [compute_metrics.compute_metric_features(s.joint_scenes[0]) for s in sims]
            self.scenario_features.extend(feats)
            if self._long_metrics is not None:          # an infgen_amd.metrics.LongMetric the caller assigned (the reference
                for f in feats:                         # builds its own from data/waymo_processed/log_features, :193-196)
                    self._long_metrics.update(features=f)
        return rollout

    def on_validation_start(self):
        self.scenario_rollouts, self.scenario_features = [], []
