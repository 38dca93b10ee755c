"""Batched vector / matrix products of the reference's API (pypose/function/linalg.py:5-110: ``bvv``, ``bmv``, ``bvmv``).

Thin, broadcastable contractions over trailing dimensions -- plain tensor algebra on whatever device the operands live on
(the reference's ``optim/optimizer.py:2`` imports ``bmv``); LieTensor operands are read through their plain values."""
import torch

from ..lietensor import LieTensor


def _plain(t):
    return t.tensor() if isinstance(t, LieTensor) else t


def bvv(lvec, rvec, *, out=None):
    """outer products ``lvec[..., :, None] * rvec[..., None, :]`` with broadcasting batch dimensions"""
    lvec, rvec = _plain(lvec), _plain(rvec)
    return torch.matmul(lvec.unsqueeze(-1), rvec.unsqueeze(-2), out=out)


def bmv(mat, vec, *, out=None):
    """matrix-vector products ``mat @ vec`` over broadcasting batch dimensions"""
    assert mat.ndim >= 2 and vec.ndim >= 1, 'Input arguments invalid'
    assert mat.shape[-1] == vec.shape[-1], 'matrix-vector shape invalid'
    mat, vec = _plain(mat), _plain(vec)
    return torch.matmul(mat, vec.unsqueeze(-1), out=out).squeeze_(-1)


def bvmv(lvec, mat, rvec):
    """bilinear forms ``lvec^T mat rvec`` over broadcasting batch dimensions (at least 1-D)"""
    assert mat.ndim >= 2 and lvec.ndim >= 1 and rvec.ndim >= 1, 'Shape invalid'
    assert lvec.shape[-1] == mat.shape[-2] and mat.shape[-1] == rvec.shape[-1]
    lvec, mat, rvec = _plain(lvec), _plain(mat), _plain(rvec)
    return torch.atleast_1d((lvec.unsqueeze(-2) @ mat @ rvec.unsqueeze(-1)).squeeze(-1).squeeze(-1))
