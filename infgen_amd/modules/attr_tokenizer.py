"""Attr_Tokenizer with the reference's interface (infgen/modules/attr_tokenizer.py:8-110): the ego-centric grid of
position cells (a disc of radius ``radius`` cut out of a ``grid_range`` square at ``grid_interval`` spacing) and the
``angle_interval``-degree heading bins.

The grid itself comes from ``synth.build_grid`` (one definition for the engine, the oracle inputs and this module) and is
held as the buffers ``grid`` / ``dist`` / ``dir`` the reference's checkpoints carry.  Inside the rollout the tokenisation
runs on the device (``k_integrate`` / ``k_fetch_enterings``); the methods below are the host-side utilities of the
reference's public surface, written against the frame convention  world = cell . R(theta_ego - pi/2) + ego  with
R(phi) = [[cos, sin], [-sin, cos]] applied on the right (SURVEY a-Q12).
"""
import numpy as np
import torch
import torch.nn as nn

from ..synth import build_grid
from ..utils.func import angle_between_2d_vectors, wrap_angle


def _to_frame(points: torch.Tensor, phi: torch.Tensor) -> torch.Tensor:
    """rows of ``points`` (..., 2) times R(phi), phi broadcast over the leading dimension"""
    c, s = torch.cos(phi), torch.sin(phi)
    while c.dim() < points.dim() - 1:
        c, s = c.unsqueeze(-1), s.unsqueeze(-1)
    px, py = points[..., 0], points[..., 1]
    return torch.stack((px * c - py * s, px * s + py * c), dim=-1)


class Attr_Tokenizer(nn.Module):

    def __init__(self, grid_range, grid_interval, radius, angle_interval):
        super().__init__()
        self.grid_range, self.grid_interval = grid_range, grid_interval
        self.radius, self.angle_interval = radius, angle_interval
        self.heading = torch.pi / 2                      # the grid's +y axis is the ego's forward direction
        cells = torch.from_numpy(build_grid(grid_range, grid_interval, radius))
        self.register_buffer('grid', cells)
        self.register_buffer('dist', cells.square().sum(-1).sqrt())
        # float32 cos(pi/2) = -4.4e-8, not 0: the sign of that residue decides +pi / -pi for the cells straight behind the ego
        half_pi = torch.tensor(self.heading, dtype=torch.float32)
        forward = torch.stack([half_pi.cos(), half_pi.sin()]).unsqueeze(0)
        self.register_buffer('dir', angle_between_2d_vectors(ctr_vector=forward, nbr_vector=cells))
        # which cells of the full square lattice (row-major from the top row down) lie inside the disc
        self.num_grid = n = int(grid_range / grid_interval) + 1
        axis = (np.arange(n, dtype=np.float32) - np.float32(n // 2)) * np.float32(grid_interval)
        lattice_r = np.hypot(axis[None, :], axis[::-1][:, None]).reshape(-1)
        self.square_mask = lattice_r <= radius
        self.grid_size = int(cells.shape[0])
        self.angle_size = int(360. / angle_interval)
        assert self.square_mask.sum() == self.grid_size and bool((cells[self.grid_size // 2] == 0).all())

    def _cells(self, like: torch.Tensor) -> torch.Tensor:
        return self.grid.to(like.device)

    def _frame_angle(self, theta, n):
        return (theta - self.heading).expand(n)

    # -- reference :57-68
    def pad_square(self, prob, indices=None):
        """disc-indexed values -> the full square lattice (for plotting); disc indices -> lattice indices (-1 stays -1)"""
        square = np.zeros(prob.shape[:-1] + (self.square_mask.size,))
        square[..., self.square_mask] = prob
        if indices is not None:
            lookup = np.append(np.flatnonzero(self.square_mask), -1)
            indices = lookup[indices]
        return square, indices

    # -- reference :70-75
    def get_grid(self, x, theta=None):
        """world positions of every cell around each centre x (N, 2), optionally in the frame of heading theta"""
        x = x.reshape(-1, 2)
        cells = self._cells(x).unsqueeze(0)
        if theta is not None:
            cells = _to_frame(cells.expand(x.shape[0], -1, -1), self._frame_angle(theta, x.shape[0]))
        return cells + x.unsqueeze(1)

    # -- reference :77-89
    def encode_pos(self, x, y, theta_y=None):
        """nearest cell (first index on ties, like torch.argmin over the euclidean distances) of x seen from y"""
        assert x.dim() == y.dim() and x.shape[-1] == 2 and y.shape[-1] == 2, f'bad shapes {x.shape} {y.shape}'
        rel = x - y
        if theta_y is not None:
            rel = _to_frame(rel, -self._frame_angle(theta_y, x.shape[0]))
        cells = self._cells(x)
        index = (rel.unsqueeze(1) - cells.unsqueeze(0)).square().sum(-1).sqrt().argmin(dim=-1)
        return index.long(), rel - cells[index]

    # -- reference :91-99
    def decode_pos(self, index, y=None, theta_y=None):
        assert bool(((index >= 0) & (index < self.grid_size)).all())
        rel = self._cells(index)[index.long()]
        if y is None:
            return rel.float()
        if theta_y is not None:
            rel = _to_frame(rel, self._frame_angle(theta_y, rel.shape[0]))
        return (rel + y).float()

    # -- reference :101-110
    def encode_heading(self, heading):
        degrees = (wrap_angle(heading) + torch.pi) / (2 * torch.pi) * 360
        return torch.div(degrees, self.angle_interval, rounding_mode='floor').long()

    def decode_heading(self, index):
        assert bool((index >= 0).all()) and bool((index < 360 / self.angle_interval).all())
        return ((index * self.angle_interval - 180) / 360 * (2 * torch.pi)).float()
