"""Dispatch of ``cumprod_`` / ``cummul_`` on group LieTensors to the single-pass HIP scan (csrc/scan.hip: one wavefront
per sequence, O(L) work, one launch) -- replaces the Hillis-Steele formulation of pypose/basics/ops.py:27-36, WITH or
WITHOUT a gradient: the differentiable route is one autograd node whose backward is the one-pass kernel
``pplie_scan_<g>_bwd`` (a reverse sum of adjoint-transported cotangents read off the scan's output), instead of the
reference's log2(L) rounds of index_select / Mul / index_copy_ nodes."""
import ctypes

import torch

from .. import _C

_SIG = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
_BSIG = [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
_KEY = {"SO3Type": "so3", "SE3Type": "se3", "Sim3Type": "sim3", "RxSO3Type": "rxso3"}
_ADJ = {"so3": "SO3_Adj", "se3": "SE3_Adj", "sim3": "Sim3_Adj", "rxso3": "RxSO3_Adj"}
_plain = lambda t: torch.Tensor.as_subclass(t, torch.Tensor)
DIFFERENTIABLE_SCAN = True       # False: gradients go through the reference's Hillis-Steele rounds (tests compare the two routes)


def _geometry(shape, dim):
    L = shape[dim]
    outer = 1
    for s in shape[:dim]:
        outer *= s
    inner = 1
    for s in shape[dim + 1:-1]:
        inner *= s
    return outer, L, inner


def _launch_fwd(t, key, dim, left):
    outer, L, inner = _geometry(t.shape, dim)
    fn = _C.library().symbol(f"pplie_scan_{key}" + ("_f32" if t.dtype == torch.float32 else "_f64"), _SIG)
    with _C._on_device(t.device):
        code = fn(t.data_ptr(), outer * inner, L, inner, 1 if left else 0, _C.stream_ptr(t.device))
    _C.check(code, f"pplie_scan_{key}")


def _launch_bwd(xy, g, key, dim, left):
    """xy: the scan's input (left products) or output (right products) -- what the kernel reads, csrc/scan.hip"""
    outer, L, inner = _geometry(xy.shape, dim)
    gx = torch.empty_like(xy)
    fn = _C.library().symbol(f"pplie_scan_{key}_bwd" + ("_f32" if xy.dtype == torch.float32 else "_f64"), _BSIG)
    with _C._on_device(xy.device):
        code = fn(xy.data_ptr() if left else None, None if left else xy.data_ptr(), g.data_ptr(), gx.data_ptr(), outer * inner,
                  L, inner, 1 if left else 0, _C.stream_ptr(xy.device))
    _C.check(code, f"pplie_scan_{key}_bwd")
    return gx


def _composed_bwd(y, g, key, dim, left):
    """the same closed form from differentiable torch ops -- only when the backward itself is being recorded
    (``create_graph=True``): double backward through a scan"""
    from ..lietensor import matrices
    adj = getattr(matrices, _ADJ[key])
    D = y.shape[-1] - 1
    rowvec = lambda v, Y: (v.unsqueeze(-2) @ adj(Y)).squeeze(-2)            # v @ Adj(Y) = Adj(Y)^T v
    rsum = lambda v: v.flip(dim).cumsum(dim).flip(dim)
    gt = g[..., :D]
    if left:
        from ..lietensor import lietensor as lt
        Y = lt._wrap(y, getattr(lt, {"so3": "SO3_type", "se3": "SE3_type", "sim3": "Sim3_type", "rxso3": "RxSO3_type"}[key]))
        out = rowvec(rsum(rowvec(gt, y)), _plain(Y.Inv()))
    else:
        ident = torch.zeros_like(y.narrow(dim, 0, 1))
        ident[..., [3] if key in ("so3", "rxso3") else [6]] = 1
        if key in ("sim3", "rxso3"):
            ident[..., -1] = 1
        yprev = torch.cat([ident, y.narrow(dim, 0, y.shape[dim] - 1)], dim=dim)
        out = rowvec(rsum(gt), yprev)
    # last (padding) component: zero, except the first element's, which passes through (its factor enters no Mul node on the
    # reference's route either: basics/ops.py:27-36 never touches element 0) -- same values as the kernel
    L = y.shape[dim]
    last = torch.cat([g.narrow(dim, 0, 1)[..., D:], torch.zeros_like(out.narrow(dim, 1, L - 1)[..., :1])], dim=dim) if L > 1 \
        else g[..., D:]
    return torch.cat([out, last], dim=-1)


class _GroupScan(torch.autograd.Function):
    """in-place product scan of a non-leaf group tensor as ONE autograd node"""

    @staticmethod
    def forward(ctx, x, key, dim, left):
        ctx.mark_dirty(x)
        # left products: the backward transports cotangents through the scan's own FACTORS (kept: the scan overwrites them);
        # right products: through the output
        x_in = _plain(x).clone() if left else None
        _launch_fwd(_plain(x), key, dim, left)
        ctx.key, ctx.dim, ctx.left = key, dim, left
        ctx.save_for_backward(x, x_in)
        return x

    @staticmethod
    def backward(ctx, g):
        y, x_in = ctx.saved_tensors
        y = _plain(y)
        g = _plain(g)
        if torch.is_grad_enabled():
            return _composed_bwd(y, g, ctx.key, ctx.dim, ctx.left), None, None, None
        return _launch_bwd(x_in if ctx.left else y, g.contiguous(), ctx.key, ctx.dim, ctx.left), None, None, None


NATIVE_NODE = True               # False: the Python autograd.Function instead of the C++ node (same kernels; tests compare)
_rules, _addr = {}, {}


def _native_scan(input, key, dim, left):
    """_GroupScan as a C++ autograd node (csrc_torch/pplie_autograd.cpp ScanOp): a Python Function's backward costs ~100 us of host
    time on the autograd engine's thread, more than the backward kernel of most scans.  None when the extension is not there."""
    from ..lietensor import operation as _op
    nat = _op._native() if NATIVE_NODE else None
    if nat is None or not hasattr(nat, "scan_op"):
        return None
    rule = _rules.get(key)
    if rule is None:
        _op._native_state["rules"] += 1
        rule = _rules[key] = _op._native_state["rules"]
        nat.set_rule(rule, lambda y, g, d, lf, _k=key: _composed_bwd(_plain(y), _plain(g), _k, d, lf))
    outer, L, inner = _geometry(input.shape, dim)
    fns = _addr.get((key, input.dtype))
    if fns is None:
        sfx, lib = "_f32" if input.dtype == torch.float32 else "_f64", _C.library()
        fns = _addr[(key, input.dtype)] = (lib.address(f"pplie_scan_{key}{sfx}"), lib.address(f"pplie_scan_{key}_bwd{sfx}"))
    return nat.scan_op(input, fns[0], fns[1], outer * inner, L, inner, bool(left), rule, dim)


_MAT_SIG = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
_MAT_DIMS = (2, 3, 4, 6, 7, 9)


def _try_mat_scan_(input, dim, left):
    """``cumprod_`` of a plain tensor of square matrices ``[..., L, d, d]`` along the axis right in front of the matrices (the
    reference's one use: the [B, F + 1, 9, 9] propagation matrices of the IMU covariance, module/imu_preintegrator.py:462) as ONE
    launch of ``pplie_scan_mat`` instead of ceil(log2 L) rounds of index_select / bmm / index_copy_; None if not that case (another
    axis, another matrix size, a gradient being recorded: the reference formulation in torch is differentiable)."""
    if type(input) is not torch.Tensor or _C._test_backend is not None or not input.is_cuda or input.dim() < 3:
        return None
    d = input.shape[-1]
    if input.shape[-2] != d or d not in _MAT_DIMS or dim % input.dim() != input.dim() - 3:
        return None
    if input.dtype not in (torch.float32, torch.float64) or not input.is_contiguous() or (torch.is_grad_enabled() and input.requires_grad):
        return None
    L = input.shape[-3]
    nseq = input.numel() // (L * d * d) if L else 0
    fn = _C.library().symbol("pplie_scan_mat" + ("_f32" if input.dtype == torch.float32 else "_f64"), _MAT_SIG)
    with _C._on_device(input.device):
        code = fn(input.data_ptr(), nseq, L, d, 1 if left else 0, _C.stream_ptr(input.device))
    _C.check(code, "pplie_scan_mat")
    _C.mark_written(input)
    return input


def try_scan_(input, dim, left, matmul=False):
    """Scan ``input`` (a group LieTensor; with ``matmul`` -- cumprod_, whose operator is ``@`` -- also a plain stack of small square
    matrices) in place along ``dim`` on the GPU; None if not applicable."""
    ltype = getattr(input, "ltype", None)
    key = _KEY.get(type(ltype).__name__) if ltype is not None else None
    if key is None and ltype is None:
        return _try_mat_scan_(input, dim, left) if matmul else None
    if key is None or _C._test_backend is not None or not input.is_cuda:
        return None
    if input.dtype not in (torch.float32, torch.float64) or not input.is_contiguous():
        return None
    nd = input.dim()
    dim = dim % nd
    if dim == nd - 1:
        return None
    if torch.is_grad_enabled() and input.requires_grad:
        if not DIFFERENTIABLE_SCAN or torch._C._are_functorch_transforms_active() or input.is_leaf:
            return None       # transforms trace the composed route; a leaf raises there exactly as in the reference
        nat = _native_scan(input, key, dim, left)
        return nat if nat is not None else _GroupScan.apply(input, key, dim, left)
    _launch_fwd(input, key, dim, left)
    _C.mark_written(input)                # the scan wrote through the raw pointer
    return input
