"""Golden vectors for Attr_Tokenizer (infgen/modules/attr_tokenizer.py:8-110): the REFERENCE's own class on seeded inputs.
Build container only.

    python tests/golden/make_golden_tokenizer.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import _standins  # noqa: E402

_standins.install()
sys.path.insert(0, '/root/reference')
from infgen.modules.attr_tokenizer import Attr_Tokenizer  # noqa: E402

_standins.assert_reference(Attr_Tokenizer)


def main():
    tok = Attr_Tokenizer(grid_range=150., grid_interval=3., radius=75., angle_interval=3.)
    rng = np.random.default_rng(9102)
    n = 96
    y = rng.uniform(-40, 40, (n, 2)).astype(np.float32)
    x = (y + rng.uniform(-90, 90, (n, 2))).astype(np.float32)          # some beyond the disc
    x[:4] = y[:4] + np.float32([[1.5, 1.5], [-1.5, 1.5], [0, 0], [3.0, 0]])   # ties between cells, the centre, a cell centre
    theta = rng.uniform(-np.pi, np.pi, 1).astype(np.float32)
    tx, ty, tt = torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(theta)
    idx_r, off_r = tok.encode_pos(tx, ty, tt)
    idx_n, off_n = tok.encode_pos(tx, ty)
    dec_r = tok.decode_pos(idx_r, ty, tt)
    dec_n = tok.decode_pos(idx_r, ty)
    dec_0 = tok.decode_pos(idx_r)
    head = torch.from_numpy(rng.uniform(-7, 7, 200).astype(np.float32))
    hbin = tok.encode_heading(head)
    hdec = tok.decode_heading(hbin)
    grid_w = tok.get_grid(ty[:1], tt)          # (the reference supports one centre per heading)
    prob = rng.uniform(0, 1, (2, tok.grid_size))
    pad, pidx = tok.pad_square(prob, np.array([0, 5, tok.grid_size - 1, -1]))
    np.savez_compressed(os.path.join(HERE, 'attr_tokenizer.npz'), x=x, y=y, theta=theta, idx_r=idx_r.numpy(), off_r=off_r.numpy(),
                        idx_n=idx_n.numpy(), off_n=off_n.numpy(), dec_r=dec_r.numpy(), dec_n=dec_n.numpy(), dec_0=dec_0.numpy(),
                        head=head.numpy(), hbin=hbin.numpy(), hdec=hdec.numpy(), grid_w=grid_w.numpy(), prob=prob, pad=pad,
                        pidx=pidx, grid=tok.grid.numpy(), dist=tok.dist.numpy(), dir=tok.dir.numpy(),
                        square_mask=tok.square_mask)
    print('cells', tok.grid_size, 'encode_pos rotated', idx_r[:6].tolist())


if __name__ == '__main__':
    main()
