"""CPU restatement of the reference's InfGen._fetch_enterings (infgen/model/infgen.py:1008-1128) - TEST INFRASTRUCTURE ONLY
(imported by tests/, never by the product path).  Pinned against the reference's own output
(tests/golden/make_golden_enterings.py -> tests/golden/enterings_*.npz; tests/test_oracle_golden.py).

Per scene and token step: which agents are within pl2seed_radius of the ego, their cell of the polar-cropped grid in the
ego frame (Attr_Tokenizer.encode_pos, attr_tokenizer.py:77-89) with the offset to the cell centre, the heading bin
(encode_heading :101-104), the order of the entering agents by bearing from the ego's heading, and the grid cell of every
map token per step (predict_occ)."""
import math

import torch

from .rollout_oracle import angle_between, rot_right, wrap_angle


def _encode(grid, x, ego, theta):
    cx = rot_right((x - ego)[:, None], (-(theta - math.pi / 2)).expand(x.shape[0]))[:, 0]
    idx = ((cx[:, None] - grid[None]) ** 2).sum(-1).sqrt().argmin(-1)
    return idx, cx - grid[idx]


def fetch_enterings(token_pos, token_heading, state_idx, batch, av_index, grid, radius, angle_interval, enter_state=2,
                    invalid_state=0, pt_pos=None, pt_batch=None):
    """token_pos (A, T, 2), token_heading (A, T), state_idx (A, T), batch (A,) scene of every agent, av_index (B,) row of
    the ego inside its scene; pt_pos (M, 2) / pt_batch (M,) for the map-token cells."""
    A, T = state_idx.shape
    out = dict(grid_token_idx=torch.zeros(A, T, dtype=torch.long), grid_offset_xy=torch.zeros(A, T, 2),
               heading_token_idx=torch.zeros(A, T, dtype=torch.long), sort_indices=torch.zeros(A, T, dtype=torch.long),
               inrange_mask=torch.zeros(A, T, dtype=torch.bool), bos_mask=torch.zeros(A, T, dtype=torch.bool),
               pos_xy=torch.zeros(A, T, 2), heading_theta=torch.zeros(A, T))
    if pt_pos is not None:
        out['pt_grid_token_idx'] = torch.zeros(T, pt_pos.shape[0], dtype=torch.long)
    for b in range(len(av_index)):
        rows = torch.nonzero(batch == b)[:, 0]
        pos, head, st = token_pos[rows], token_heading[rows], state_idx[rows]
        av = int(av_index[b])
        n = len(rows)
        for t in range(T):
            ego, th = pos[av, t][None], head[av, t][None]
            born = st[:, t] == enter_state
            near = ((pos[:, t] - ego) ** 2).sum(-1).sqrt() <= radius
            use = near & (st[:, t] != invalid_state)
            cell = torch.full((n,), -1, dtype=torch.long)
            off = torch.zeros(n, 2)
            rel = torch.zeros(n, 2)
            cell[use], off[use] = _encode(grid, pos[use, t], ego, th)
            rel[use] = pos[use, t] - ego
            bearing = angle_between(torch.stack([th.cos(), th.sin()], -1), pos[:, t] - ego)
            bearing[~(born & near)] = math.inf
            key, order = bearing.sort()
            order[torch.isinf(key)] = av
            out['grid_token_idx'][rows, t] = cell
            out['grid_offset_xy'][rows, t] = off
            out['pos_xy'][rows, t] = rel
            out['sort_indices'][rows, t] = order
            out['inrange_mask'][rows, t] = near
            out['bos_mask'][rows, t] = born
            if pt_pos is not None:
                cols = torch.nonzero(pt_batch == b)[:, 0]
                pnear = ((pt_pos[cols, :2] - ego) ** 2).sum(-1).sqrt() <= radius
                pc = torch.full((len(cols),), -1, dtype=torch.long)
                pc[pnear] = _encode(grid, pt_pos[cols][pnear, :2], ego, th)[0]
                out['pt_grid_token_idx'][t, cols] = pc
        dh = head - head[av][None]
        out['heading_token_idx'][rows] = ((wrap_angle(dh) + math.pi) / (2 * math.pi) * 360 // angle_interval).long()
        out['heading_theta'][rows] = wrap_angle(dh)
    return out
