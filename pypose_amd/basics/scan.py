"""Dispatch of ``cumprod_`` / ``cummul_`` on group LieTensors to the single-pass HIP scan
(csrc/scan.hip: one wavefront per sequence, O(L) work, one launch) -- replaces the
Hillis-Steele formulation of pypose/basics/ops.py:27-36 when no gradient is required."""
import ctypes

import torch

from .. import _C

_SIG = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
_KEY = {"SO3Type": "so3", "SE3Type": "se3", "Sim3Type": "sim3", "RxSO3Type": "rxso3"}


def try_scan_(input, dim, left):
    """Scan ``input`` (a group LieTensor) in place along ``dim`` on the GPU; None if not applicable."""
    ltype = getattr(input, "ltype", None)
    key = _KEY.get(type(ltype).__name__) if ltype is not None else None
    if key is None or _C._test_backend is not None or not input.is_cuda:
        return None
    if input.dtype not in (torch.float32, torch.float64) or not input.is_contiguous():
        return None
    if torch.is_grad_enabled() and input.requires_grad:
        return None          # the differentiable route goes through the Mul Functions
    nd = input.dim()
    dim = dim % nd
    if dim == nd - 1:
        return None
    L = input.shape[dim]
    outer = 1
    for s in input.shape[:dim]:
        outer *= s
    inner = 1
    for s in input.shape[dim + 1:-1]:
        inner *= s
    fn = _C.library().symbol(f"pplie_scan_{key}" + ("_f32" if input.dtype == torch.float32 else "_f64"), _SIG)
    with _C._on_device(input.device):
        code = fn(input.data_ptr(), outer * inner, L, inner, 1 if left else 0, _C.stream_ptr(input.device))
    _C.check(code, f"pplie_scan_{key}")
    _C.mark_written(input)                # the scan wrote through the raw pointer
    return input
