"""InfGenMapDecoder with the reference's interface (infgen/modules/map_decoder.py:13-130): same
constructor, same parameter names; ``forward(data)`` runs the map-token encoder on the GPU through
the HIP kernels (radius graph -> Fourier embedding -> 3 attention layers)."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn as nn

from .layers import AttentionLayer, FourierEmbedding, MLPEmbedding, MLPLayer
from ..utils.func import weight_init


class InfGenMapDecoder(nn.Module):

    def __init__(self, dataset: str, input_dim: int, hidden_dim: int, num_historical_steps: int, pl2pl_radius: float,
                 num_freq_bands: int, num_layers: int, num_heads: int, head_dim: int, dropout: float, map_token) -> None:
        super().__init__()
        self.dataset, self.input_dim, self.hidden_dim = dataset, input_dim, hidden_dim
        self.num_historical_steps, self.pl2pl_radius, self.num_freq_bands = num_historical_steps, pl2pl_radius, num_freq_bands
        self.num_layers, self.num_heads, self.head_dim, self.dropout = num_layers, num_heads, head_dim, dropout
        if input_dim != 2:
            raise ValueError('the HIP path implements input_dim == 2 (configs/ours_*.yaml)')
        self.type_pt_emb = nn.Embedding(17, hidden_dim)
        self.side_pt_emb = nn.Embedding(4, hidden_dim)
        self.polygon_type_emb = nn.Embedding(4, hidden_dim)
        self.light_pl_emb = nn.Embedding(4, hidden_dim)
        self.r_pt2pt_emb = FourierEmbedding(input_dim=3, hidden_dim=hidden_dim, num_freq_bands=num_freq_bands)
        self.pt2pt_layers = nn.ModuleList([AttentionLayer(hidden_dim=hidden_dim, num_heads=num_heads, head_dim=head_dim,
                                                          dropout=dropout, bipartite=False, has_pos_emb=True)
                                           for _ in range(num_layers)])
        self.token_size = 1024
        self.token_predict_head = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim, output_dim=self.token_size)
        self.token_emb = MLPEmbedding(input_dim=22, hidden_dim=hidden_dim)
        self.map_token = map_token
        self.apply(weight_init)
        self.mask_pt = False

    @torch.no_grad()
    def forward(self, data) -> Dict[str, torch.Tensor]:
        """returns {'x_pt': (M,128), 'map_next_token_*': ...} like the reference; the map-token head
        only serves the training target (``predict_map``), so its outputs are empty here."""
        owner = getattr(self, '_owner', None)
        if owner is None:
            raise RuntimeError('InfGenMapDecoder.forward is driven through InfGenDecoder (shared packed weights)')
        x_pt = owner()._run(data, map_only=True)
        dev = x_pt.device
        pt = data['pt_token']
        tgt = torch.as_tensor(pt['token_idx'])[torch.as_tensor(pt['pt_target_mask']).bool()]
        return {'x_pt': x_pt,
                'map_next_token_idx': torch.zeros(0, 10, dtype=torch.long, device=dev),
                'map_next_token_prob': torch.zeros(0, self.token_size, device=dev),
                'map_next_token_idx_gt': tgt.to(dev),
                'map_next_token_eval_mask': torch.zeros(0, dtype=torch.bool, device=dev)}
