"""InfGenAgentDecoder with the reference's interface (infgen/modules/agent_decoder.py:98-314,
1605-2389): same constructor signature, the same parameter tree (``state_dict`` keys of a
Lightning checkpoint's ``encoder.agent_encoder.*`` load unchanged) and
``inference(data, map_enc) -> dict`` with the reference's 23 output keys.

The rollout itself runs in libinfgen_hip.so (infgen_amd/engine.py), including the scenario-
insertion sub-loop when ``disable_insertion`` is False.  Motion tokens: greedy for
``motion_beam_size = 1``, otherwise top-k inverse-CDF sampling on supplied uniforms (the reference's
top-5 multinomial, reproducible); the cell of an inserted agent: arg-max for ``insert_beam_size = 1`` (the default here),
otherwise drawn from the top-k cells the same way (the reference's top-10 multinomial, agent_decoder.py:1900-1909, incl. its
retry after an occupied cell).  The seed-loop outputs (``next_state_prob_seed``, ``next_pos_rel_prob_seed``, ``grid_*_occ_seed``,
``agent_labels``) are filled like the reference's by ``inference`` / ``inference_rollouts``; ``inference_batch`` returns zeros
there unless called with ``seed_outputs=True``.
"""
from __future__ import annotations

from typing import Dict, Mapping, Optional

import torch
import torch.nn as nn

from .attr_tokenizer import Attr_Tokenizer
from .layers import AttentionLayer, FourierEmbedding, MLPEmbedding, MLPLayer
from ..synth import AGENT_TYPE
from ..utils.func import weight_init


class InfGenAgentDecoder(nn.Module):

    def __init__(self, dataset: str, input_dim: int, hidden_dim: int, num_historical_steps: int,
                 time_span: Optional[int], pl2a_radius: float, pl2seed_radius: float, a2a_radius: float,
                 a2sa_radius: float, pl2sa_radius: float, num_freq_bands: int, num_layers: int, num_heads: int,
                 head_dim: int, dropout: float, token_size: int, attr_tokenizer: Attr_Tokenizer = None,
                 predict_motion: bool = False, predict_state: bool = False, predict_map: bool = False,
                 predict_occ: bool = False, state_token: Dict[str, int] = None, use_grid_token: bool = True,
                 use_head_token: bool = True, use_state_token: bool = True, disable_insertion: bool = False,
                 seed_size: int = 5, buffer_size: int = 32, num_recurrent_steps_val: int = -1,
                 loss_weight: dict = None, logger=None) -> None:
        super().__init__()
        self.dataset, self.input_dim, self.hidden_dim = dataset, input_dim, hidden_dim
        self.num_historical_steps = num_historical_steps
        self.time_span = time_span if time_span is not None else num_historical_steps
        self.pl2a_radius, self.pl2seed_radius, self.a2a_radius = pl2a_radius, pl2seed_radius, a2a_radius
        self.a2sa_radius, self.pl2sa_radius = a2sa_radius, pl2sa_radius
        self.num_freq_bands, self.num_layers, self.num_heads, self.head_dim = num_freq_bands, num_layers, num_heads, head_dim
        self.dropout = dropout
        self.predict_motion, self.predict_state, self.predict_map, self.predict_occ = predict_motion, predict_state, predict_map, predict_occ
        self.use_grid_token, self.use_head_token, self.use_state_token = use_grid_token, use_head_token, use_state_token
        self.disable_insertion = disable_insertion
        self.num_recurrent_steps_val = num_recurrent_steps_val
        self.loss_weight, self.logger = loss_weight, logger
        self.attr_tokenizer = attr_tokenizer
        if not (use_grid_token and use_head_token and use_state_token):
            raise ValueError('the HIP path implements the full-token model (use_grid/head/state_token = True)')
        self.state_type = list(state_token.keys())
        self.state_token = state_token
        self.invalid_state, self.valid_state = int(state_token['invalid']), int(state_token['valid'])
        self.enter_state, self.exit_state = int(state_token['enter']), int(state_token['exit'])
        self.seed_state_type = ['invalid', 'enter']
        self.valid_state_type = ['invalid', 'valid', 'exit']
        self.seed_size, self.buffer_size = seed_size, buffer_size

        self.type_a_emb = nn.Embedding(len(AGENT_TYPE), hidden_dim)
        self.shape_emb = MLPEmbedding(input_dim=3, hidden_dim=hidden_dim)
        self.state_a_emb = nn.Embedding(len(self.state_type), hidden_dim)
        self.x_a_emb = FourierEmbedding(input_dim=2, hidden_dim=hidden_dim, num_freq_bands=num_freq_bands)
        self.r_t_emb = FourierEmbedding(input_dim=4, hidden_dim=hidden_dim, num_freq_bands=num_freq_bands)
        self.r_pt2a_emb = FourierEmbedding(input_dim=3, hidden_dim=hidden_dim, num_freq_bands=num_freq_bands)
        self.r_a2a_emb = FourierEmbedding(input_dim=3, hidden_dim=hidden_dim, num_freq_bands=num_freq_bands)
        self.r_pt2sa_emb = FourierEmbedding(input_dim=3, hidden_dim=hidden_dim, num_freq_bands=num_freq_bands)
        self.r_a2sa_emb = FourierEmbedding(input_dim=3, hidden_dim=hidden_dim, num_freq_bands=num_freq_bands)
        self.token_emb_veh = MLPEmbedding(input_dim=8, hidden_dim=hidden_dim)
        self.token_emb_ped = MLPEmbedding(input_dim=8, hidden_dim=hidden_dim)
        self.token_emb_cyc = MLPEmbedding(input_dim=8, hidden_dim=hidden_dim)
        self.token_emb_grid = MLPEmbedding(input_dim=2, hidden_dim=hidden_dim)
        self.no_token_emb = nn.Embedding(1, hidden_dim)
        self.bos_token_emb = nn.Embedding(1, hidden_dim)
        self.invalid_offset_token_emb = nn.Embedding(1, hidden_dim)
        self.fusion_emb = MLPEmbedding(input_dim=hidden_dim * 4, hidden_dim=hidden_dim)

        def layers(n, bipartite, pos=True):
            return nn.ModuleList([AttentionLayer(hidden_dim=hidden_dim, num_heads=num_heads, head_dim=head_dim,
                                                 dropout=dropout, bipartite=bipartite, has_pos_emb=pos) for _ in range(n)])
        self.t_attn_layers = layers(num_layers, False)
        self.pt2a_attn_layers = layers(num_layers, True)
        self.a2a_attn_layers = layers(num_layers, False)
        self.seed_layers = 3
        self.pt2sa_attn_layers = layers(self.seed_layers, True)
        self.a2sa_attn_layers = layers(self.seed_layers, False)
        self.occ2sa_attn_layers = layers(self.seed_layers, True, pos=False)

        self.token_size = token_size
        self.token_predict_head = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim, output_dim=token_size)
        self.state_predict_head = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim, output_dim=len(self.valid_state_type))
        self.seed_state_predict_head = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim, output_dim=len(self.seed_state_type))
        self.seed_type_predict_head = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim, output_dim=len(AGENT_TYPE) - 1)
        self.seed_shape_predict_head = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim, output_dim=3)
        self.grid_size = self.attr_tokenizer.grid_size
        self.angle_size = self.attr_tokenizer.angle_size
        self.seed_pos_rel_token_predict_head = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim, output_dim=self.grid_size)
        self.seed_offset_xy_predict_head = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim, output_dim=2)
        self.seed_agent_occ_embed = MLPLayer(input_dim=self.grid_size, hidden_dim=hidden_dim, output_dim=hidden_dim)
        self.seed_heading_rel_token_predict_head = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim, output_dim=self.angle_size)
        if self.predict_occ:
            self.grid_agent_occ_head = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim, output_dim=self.grid_size)
            self.grid_pt_occ_head = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim, output_dim=self.grid_size)
        self.grid_index_head = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim, output_dim=self.grid_size)
        self.num_seed_feature = 10
        self.apply(weight_init)
        self.shift = 5
        self.motion_beam_size = 1          # greedy (the reference's default 5 samples with torch RNG)
        self.insert_beam_size = 1
        assert self.num_recurrent_steps_val % self.shift == 0 or self.num_recurrent_steps_val == -1, \
            f"Invalid num_recurrent_steps_val: {num_recurrent_steps_val}."

    @torch.no_grad()
    def inference(self, data, map_enc: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """reference agent_decoder.py:1605-2389"""
        owner = getattr(self, '_owner', None)
        if owner is None:
            raise RuntimeError('InfGenAgentDecoder.inference is driven through InfGenDecoder (shared packed weights)')
        return owner()._run(data, x_pt=map_enc['x_pt'])
