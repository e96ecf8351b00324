"""Device-side agent tokenisation - mirror of the reference's ``TokenProcessor._match_agent_token``
(infgen/datasets/preprocess.py:552-653).  Same signature and return values; the work
runs in ``k_match_tokens`` through the C ABI (``infgen_match_agent_tokens``).  There is no CPU fallback."""
from typing import Optional, Tuple

import torch

from .. import _lib


class TokenProcessor(torch.nn.Module):
    """Only the contour-matching core is provided (the rest of ``_tokenize_agent`` is bookkeeping on its outputs)."""

    def __init__(self, token_size: int = 2048, shift: int = 5):
        super().__init__()
        self.token_size, self.shift, self.noise = token_size, shift, False

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
