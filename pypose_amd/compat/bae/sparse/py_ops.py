"""``diagonal_op_(A, op)``: apply ``op`` to the diagonal of a sparse CSR matrix, in place
(pypose/optim/optimizer.py:643 clamps it, :664 scales it by 1 + damping)."""
import torch


def _hip_diag(A, op):
    """the two operations the reference applies (functools.partial of torch.clamp_ with min / max, of torch.mul with `other`) as ONE
    kernel over the CSR rows (csrc/csr_pcg.hip pplie_csr_diag_op); False when this call is something else"""
    import ctypes
    import functools
    from pypose_amd import _C          # (absolute: this namespace is imported as the top-level package `bae`)
    if not isinstance(op, functools.partial) or op.args or A.layout != torch.sparse_csr or not A.is_cuda \
            or A.dtype not in (torch.float32, torch.float64) or _C._test_backend is not None:
        return False
    kw = op.keywords
    if op.func in (torch.clamp_, torch.clamp) and set(kw) <= {"min", "max"} and all(isinstance(v, (int, float)) for v in kw.values()):
        lo, hi, scale = float(kw.get("min", -float("inf"))), float(kw.get("max", float("inf"))), 1.0
    elif op.func is torch.mul and set(kw) == {"other"} and isinstance(kw["other"], (int, float)):
        lo, hi, scale = -float("inf"), float("inf"), float(kw["other"])
    else:
        return False
    crow, col, val = A.crow_indices(), A.col_indices(), A.values()
    if crow.dtype != col.dtype or crow.dtype not in (torch.int64, torch.int32) or not (crow.is_contiguous() and col.is_contiguous()
                                                                                       and val.is_contiguous()):
        return False
    sig = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_void_p]
    fn = _C.library().symbol("pplie_csr_diag_op" + ("_f32" if A.dtype == torch.float32 else "_f64"), sig)
    with _C._on_device(val.device):
        _C.check(fn(crow.data_ptr(), col.data_ptr(), val.data_ptr(), A.shape[0], lo, hi, scale, 1 if crow.dtype == torch.int64 else 0,
                    _C.stream_ptr(val.device)), "pplie_csr_diag_op")
    return True


def diagonal_op_(A, op):
    if _hip_diag(A, op):
        return A
    crow, col, val = A.crow_indices(), A.col_indices(), A.values()
    n = A.shape[0]
    row = torch.repeat_interleave(torch.arange(n, device=col.device), crow[1:] - crow[:-1])
    on = (row == col).nonzero().reshape(-1)
    d = val[on]
    res = op(d)
    val[on] = d if res is None else res
    return A
