"""``chspline`` and ``bspline`` (host-side mirror of pypose/function/spline.py:4-102, 105-225).

Same arguments, assertions and output layout as the reference.  Both interpolators run as ONE HIP kernel each
(csrc/spline.hip) when nothing has to be differentiated; under autograd / functorch the B-spline is evaluated as
the same product of exponentials out of the differentiable Lie kernels (a composition of hot-path ops, like the
reference's), and the Hermite spline as plain tensor arithmetic.
"""
import ctypes
import functools

import torch

from .. import _C
from ..lietensor.lietensor import LieTensor, SE3_type
from .checking import is_SE3

_BSPLINE_SIG = [ctypes.c_void_p] * 3 + [ctypes.c_int64] * 3 + [ctypes.c_void_p]
_CHSPLINE_SIG = [ctypes.c_void_p] * 4 + [ctypes.c_int64] * 4 + [ctypes.c_void_p]


def _suffix(t):
    if t.dtype == torch.float32:
        return "_f32"
    if t.dtype == torch.float64:
        return "_f64"
    raise TypeError(f"pypose_amd: splines support float32/float64, got {t.dtype}")


def _fused_ok(*tensors):
    """The single-kernel path: device tensors, nothing recording a graph, no functorch level, no test backend."""
    if _C._test_backend is not None or not all(t.is_cuda for t in tensors):
        return False
    if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
        return False
    from ..lietensor.operation import _transforms_active
    return not _transforms_active()


# ---------------------------------------------------------------------------------------------------------------
# cubic Hermite spline
# ---------------------------------------------------------------------------------------------------------------
@functools.lru_cache(maxsize=64)
def _hermite_tables(N, interval, dtype, device):
    """Sample times, their segment index and the four Hermite basis values per sample (spline.py:78-96): shared by
    every trajectory and channel (and cached: ~15 tiny launches otherwise precede a ~0.1 ms kernel)."""
    steps = torch.arange(0, 1, interval, dtype=dtype, device=device)
    knots = torch.arange(0, N, dtype=dtype, device=device)
    times = (knots.unsqueeze(-1) + steps).view(-1)[:-(steps.shape[0] - 1)]     # 0 ... N-1, closing knot included
    seg = torch.searchsorted(knots[1:], times)                                  # sample at knot j>0 -> segment j-1
    u = (times - knots[seg]) / (knots[seg + 1] - knots[seg])
    powers = u.unsqueeze(0) ** torch.arange(4, dtype=dtype, device=device).unsqueeze(-1)      # [4, M]
    basis = torch.tensor([[1, 0, -3, 2], [0, 1, -2, 1], [0, 0, 3, -2], [0, 0, -1, 1]], dtype=dtype, device=device)
    return seg, (basis @ powers).mT.contiguous()                                # [M], [M, 4]


def chspline(points, interval=0.1):
    """Cubic Hermite spline through ``points [..., N, C]`` at unit knot spacing, sampled every ``interval``
    (values and finite-difference tangents are matched at the knots).  Returns ``[..., (N-1)*K + 1, C]`` with
    ``K = len(arange(0, 1, interval))`` (reference spline.py:4-102)."""
    assert points.dim() >= 2, "Dimension of points should be [..., N, C]"
    assert interval < 1.0, "The interval should be smaller than 1."
    batch, N, C = points.shape[:-2], points.shape[-2], points.shape[-1]
    seg, hh = _hermite_tables(N, interval, points.dtype, points.device)
    M = seg.shape[0]
    if _fused_ok(points) and points.dtype in (torch.float32, torch.float64) and N >= 2:
        pts = points.detach().reshape(-1, N, C).contiguous()
        out = torch.empty((pts.shape[0], M, C), dtype=pts.dtype, device=pts.device)
        fn = _C.library().symbol("pplie_chspline" + _suffix(pts), _CHSPLINE_SIG)
        with _C._on_device(pts.device):
            code = fn(pts.data_ptr(), hh.data_ptr(), seg.data_ptr(), out.data_ptr(), pts.shape[0], N, C, M,
                      _C.stream_ptr(pts.device))
        _C.check(code, "pplie_chspline")
        return out.view(batch + (M, C))
    # differentiable route: the same blend with tensor ops
    rise = points[..., 1:, :] - points[..., :-1, :]
    tangent = torch.cat([rise[..., :1, :], (rise[..., 1:, :] + rise[..., :-1, :]) / 2, rise[..., -1:, :]], dim=-2)
    out = hh[:, 0:1] * points[..., seg, :]
    out = out + hh[:, 1:2] * tangent[..., seg, :]
    out = out + hh[:, 2:3] * points[..., seg + 1, :]
    out = out + hh[:, 3:4] * tangent[..., seg + 1, :]
    return out


# ---------------------------------------------------------------------------------------------------------------
# cumulative B-spline on SE3
# ---------------------------------------------------------------------------------------------------------------
@functools.lru_cache(maxsize=64)
def _bspline_weights(interval, dtype, device):
    """``w [3, K+1]``: the cumulative basis at u = 0, interval, 2 interval, ... (spline.py:206-212) and, in the last
    column, at u = 1 -- the row sums of the basis matrix, used for the closing pose (:216)."""
    u = torch.arange(0, 1, interval, dtype=dtype, device=device)
    powers = u ** torch.arange(4, dtype=dtype, device=device).view(-1, 1)
    basis = torch.tensor([[5, 3, -3, 1], [1, 3, 3, -2], [0, 0, 0, 1]], dtype=dtype, device=device) / 6
    return torch.cat([basis @ powers, basis.sum(dim=1, keepdim=True)], dim=1).contiguous()


def _bspline_composed(data, w):
    """The spline as a composition of the differentiable Lie kernels."""
    N = data.shape[-2]
    K = w.shape[1] - 1
    twist = (data[..., :-1, :].Inv() * data[..., 1:, :]).Log()                 # [.., N-1, 6] between neighbours
    win = torch.arange(N - 3, device=data.device).unsqueeze(-1) + torch.arange(3, device=data.device)
    xi = twist[..., win, :]                                                     # [.., N-3, 3, 6]
    first = data[..., : N - 3, :]
    steps = (xi.unsqueeze(-3) * w[:, :K].mT.unsqueeze(-1)).Exp()               # [.., N-3, K, 3, 7]
    inner = first.unsqueeze(-2) * (steps[..., 0, :] * steps[..., 1, :] * steps[..., 2, :])
    tail = (xi[..., -1, :, :] * w[:, K:]).Exp()                                 # [.., 3, 7] last segment at u = 1
    close = first[..., -1:, :] * (tail[..., 0:1, :] * tail[..., 1:2, :] * tail[..., 2:3, :])
    return torch.cat((inner.reshape(data.shape[:-2] + ((N - 3) * K, 7)), close), dim=-2)


def bspline(data, interval=0.1, extrapolate=False):
    """Cumulative cubic B-spline through SE3 poses ``data [..., N, 7]`` sampled every ``interval`` (knot spacing
    1): ``(N-3)*K + 1`` poses per trajectory; ``extrapolate=True`` repeats the end poses twice so that the curve
    spans the whole input (reference spline.py:105-225)."""
    assert is_SE3(data), "The input poses are not SE3Type."
    assert data.dim() >= 2, "Dimension of data should be [..., N, C]."
    assert interval < 1.0, "The interval should be smaller than 1."
    batch = data.shape[:-2]
    if extrapolate:
        head = data[..., :1, :].expand(batch + (2, -1))
        tail = data[..., -1:, :].expand(batch + (2, -1))
        data = torch.cat((head, data, tail), dim=-2)
    else:
        assert data.shape[-2] >= 4, "Number of poses is less than 4."
    N = data.shape[-2]
    w = _bspline_weights(interval, data.dtype, data.device)
    K = w.shape[1] - 1
    if not (_fused_ok(data) and K >= 2):
        return _bspline_composed(data, w)
    poses = data.tensor().detach().reshape(-1, N, 7).contiguous()
    nb, L = poses.shape[0], (N - 3) * K + 1
    out = torch.empty((nb, L, 7), dtype=poses.dtype, device=poses.device)
    fn = _C.library().symbol("pplie_se3_bspline" + _suffix(poses), _BSPLINE_SIG)
    with _C._on_device(poses.device):
        code = fn(poses.data_ptr(), w.data_ptr(), out.data_ptr(), nb, N, K, _C.stream_ptr(poses.device))
    _C.check(code, "pplie_se3_bspline")
    return LieTensor(out.view(batch + (L, 7)), ltype=SE3_type)
