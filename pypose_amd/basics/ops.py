"""``pm`` and the cumulative (scan) operators of the reference (pypose/basics/ops.py).

``cumprod`` / ``cumprod_`` on SO3 / SE3 / Sim3 / RxSO3 LieTensors run as a single-pass HIP scan kernel
(``pplie_scan_*``: one wavefront per sequence, wave-level inclusive scan with the group product, O(L) work),
``cumprod_`` of a plain stack of small square matrices ``[..., L, d, d]`` along L as one walk per sequence
(``pplie_scan_mat``; no gradient recorded); every other ``ops`` callable falls
back to the reference's generic Hillis-Steele formulation built from torch index ops (the user's
``ops`` is an arbitrary Python callable, so it cannot be fused).
"""
import math

import torch


def pm(input, *, out=None):
    """+1 where ``input >= 0``, -1 elsewhere (reference basics/ops.py:4-24)."""
    out = torch.sign(torch.sign(input) * 2 + 1, out=out)
    return out


def cumops_(input, dim, ops):
    """In-place inclusive scan with a user-supplied associative ``ops(a, b)`` along ``dim``
    (reference basics/ops.py:27-36: log2(L) rounds of shift-and-combine)."""
    L, v = input.shape[dim], input
    assert dim != -1 or dim != v.shape[-1], "Invalid dim"
    step = 1
    while step < L:
        hi = torch.arange(step, L, device=v.device, dtype=torch.int64)
        v.index_copy_(dim, hi, ops(v.index_select(dim, hi - step), v.index_select(dim, hi)))
        step *= 2
    return v


def cummul_(input, dim, left=True):
    from .scan import try_scan_
    done = try_scan_(input, dim, left)      # on group LieTensors ``*`` is the group product
    if done is not None:
        return done
    return cumops_(input, dim, (lambda a, b: b * a) if left else (lambda a, b: a * b))


def cumprod_(input, dim, left=True):
    from .scan import try_scan_
    done = try_scan_(input, dim, left, matmul=True)      # (plain [..., L, d, d] stacks: pplie_scan_mat)
    if done is not None:
        return done
    return cumops_(input, dim, (lambda a, b: b @ a) if left else (lambda a, b: a @ b))


def cumops(input, dim, ops):
    return cumops_(input.clone(), dim, ops)


def cummul(input, dim, left=True):
    return cummul_(input.clone(), dim, left)


def cumprod(input, dim, left=True):
    return cumprod_(input.clone(), dim, left)
