"""A stand-in for the HIP launcher built from the oracle -- TESTS ONLY.

Lets the CPU-only suite exercise the host-side logic of pypose_amd (LieTensor dispatch,
broadcasting, autograd wiring, vmap rules, LM control flow, torch.distributed sharding) in a
container without a GPU.  Installed with ``pypose_amd._C.set_backend_for_testing``; the product
never installs it (its ops raise without a HIP device).
"""
import contextlib

import numpy as np
import torch

from oracle import lie_np
from pypose_amd import _C


def oracle_row_op(name, ins, out_widths, prm=None):
    from oracle import convert_np
    if name in convert_np.OPS:
        arrs = [t.detach().cpu().numpy() for t in ins]
        if arrs[0].shape[0] == 0:
            return tuple(torch.empty((0, w), dtype=ins[0].dtype) for w in out_widths)
        outs = convert_np.OPS[name](*arrs) if prm is None else convert_np.OPS[name](*arrs, prm)
        return tuple(torch.from_numpy(np.ascontiguousarray(o)).to(ins[0].dtype) for o in outs)
    if name.startswith("block_"):
        from oracle import optim_np
        outs = getattr(optim_np, name)(*[t.detach().cpu().numpy() for t in ins])
        res = tuple(torch.from_numpy(np.ascontiguousarray(o)) for o in outs)
        return res if name == "block_normal_eq" else res[0]
    fn = lie_np.OPS[name] if name in lie_np.OPS else (lie_np.EXTRA_OPS[name] if name in lie_np.EXTRA_OPS else lie_np.COMPOSED_OPS[name])
    arrs = [t.detach().cpu().numpy() for t in ins]
    n = arrs[0].shape[0]
    if n == 0:
        return tuple(torch.empty((0, w), dtype=ins[0].dtype) for w in out_widths)
    outs = fn(*arrs)
    res = tuple(torch.from_numpy(np.ascontiguousarray(o)).to(ins[0].dtype) for o in outs)
    for r, w in zip(res, out_widths):
        assert r.shape == (n, w), (name, r.shape, w)
    return res


@contextlib.contextmanager
def oracle_backend():
    _C.set_backend_for_testing(oracle_row_op)
    try:
        yield
    finally:
        _C.set_backend_for_testing(None)
