"""Attr_Tokenizer with the reference's interface (infgen/modules/attr_tokenizer.py:8-110).

Holds the polar-masked square grid as buffers (``grid``, ``dist``, ``dir``: state_dict
compatible).  ``encode_pos`` inside the rollout runs in the HIP kernel ``k_integrate``; the
methods here are the host-side utilities the reference exposes to its callers
(``decode_pos`` / ``decode_heading`` / ``pad_square`` are used by pre/post-processing).
"""
import numpy as np
import torch
import torch.nn as nn

from ..synth import build_grid
from ..utils.func import angle_between_2d_vectors, wrap_angle


class Attr_Tokenizer(nn.Module):

    def __init__(self, grid_range, grid_interval, radius, angle_interval):
        super().__init__()
        self.grid_range, self.grid_interval, self.radius, self.angle_interval = grid_range, grid_interval, radius, angle_interval
        self.heading = torch.pi / 2
        grid = torch.from_numpy(build_grid(grid_range, grid_interval, radius))
        self.register_buffer('grid', grid)
        self.register_buffer('dist', torch.norm(grid, p=2, dim=-1))
        head_vector = torch.stack([torch.tensor(self.heading).cos(), torch.tensor(self.heading).sin()])
        self.register_buffer('dir', angle_between_2d_vectors(ctr_vector=head_vector.unsqueeze(0), nbr_vector=grid))
        self.num_grid = int(grid_range / grid_interval) + 1
        n = self.num_grid
        x = np.arange(n, dtype=np.float32)
        gx, gy = np.meshgrid(x, x, indexing='xy')
        sq = np.stack([gx.reshape(-1), gy.reshape(-1)], -1).reshape(n, n, 2)[::-1].reshape(-1, 2)
        sq = (sq - np.float32(n // 2)) * np.float32(grid_interval)
        self.square_mask = np.sqrt((sq ** 2).sum(-1)) <= radius
        self.grid_size = self.grid.shape[0]
        self.angle_size = int(360. / self.angle_interval)
        assert torch.all(self.grid[self.grid_size // 2] == 0.)

    def _apply_rot(self, x, theta):
        cos, sin = theta.cos(), theta.sin()
        rot_mat = torch.zeros((theta.shape[0], 2, 2), device=theta.device)
        rot_mat[:, 0, 0] = cos
        rot_mat[:, 0, 1] = sin
        rot_mat[:, 1, 0] = -sin
        rot_mat[:, 1, 1] = cos
        return torch.bmm(x, rot_mat)

    def pad_square(self, prob, indices=None):
        pad_prob = np.zeros((*prob.shape[:-1], self.square_mask.shape[0]))
        pad_prob[..., self.square_mask] = prob
        square_indices = np.arange(self.square_mask.shape[0])
        circle_indices = np.concatenate([square_indices[self.square_mask], [-1]])
        if indices is not None:
            indices = circle_indices[indices]
        return pad_prob, indices

    def get_grid(self, x, theta=None):
        x = x.reshape(-1, 2)
        grid = self.grid[None, ...].to(x.device)
        if theta is not None:
            grid = self._apply_rot(grid, (theta - self.heading).expand(x.shape[0]))
        return x[:, None] + grid

    def encode_pos(self, x, y, theta_y=None):
        assert x.dim() == y.dim() and x.shape[-1] == 2 and y.shape[-1] == 2
        centered_x = x - y
        if theta_y is not None:
            centered_x = self._apply_rot(centered_x[:, None], -(theta_y - self.heading).expand(x.shape[0]))[:, 0]
        distance = ((centered_x[:, None] - self.grid.to(x.device)[None, ...]) ** 2).sum(-1).sqrt()
        index = torch.argmin(distance, dim=-1)
        return index.long(), centered_x - self.grid.to(x.device)[index]

    def decode_pos(self, index, y=None, theta_y=None):
        assert torch.all((index >= 0) & (index < self.grid_size))
        centered_x = self.grid.to(index.device)[index.long()]
        if y is not None:
            if theta_y is not None:
                centered_x = self._apply_rot(centered_x[:, None], (theta_y - self.heading).expand(centered_x.shape[0]))[:, 0]
            return (centered_x + y).float()
        return centered_x.float()

    def encode_heading(self, heading):
        heading = (wrap_angle(heading) + torch.pi) / (2 * torch.pi) * 360
        return (heading // self.angle_interval).long()

    def decode_heading(self, index):
        assert torch.all(index >= 0) and torch.all(index < (360 / self.angle_interval))
        angles = index * self.angle_interval - 180
        return (angles / 360 * (2 * torch.pi)).float()
