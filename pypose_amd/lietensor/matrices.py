"""The matrix-valued helpers of ``pypose/lietensor/operation.py:7-301`` as Python callables.

The HIP kernels never build these matrices (csrc/lie_math.h applies them to a vector in registers); user code and the
reference's own modules call them by name (``so3_Jl``, ``se3_Jl_inv``, ``SE3_Adj``, ``SO3_Matrix``, ...), so they exist
here too, with two routes that return the same values:

* kernel route (device tensors, nothing to differentiate): every row / column of a matrix is one application of an existing
  row kernel to a basis vector -- ``g @ se3_Jl(x)`` IS ``pplie_se3_exp_bwd`` (operation.py:416-417), ``g @ se3_Jl_inv(y)`` is
  ``pplie_se3_log_bwd`` (:394), ``Adj(X) a`` is ``pplie_se3_adj_fwd`` (:756-757) -- one launch over ``rows x basis`` rows;
* composed route (gradients required, or host tensors): closed forms written with differentiable torch ops.  This is also
  what makes DOUBLE backward through the Lie Functions work: under ``create_graph=True`` their backward is evaluated as
  ``g @ M(saved)`` with these helpers (``operation._composed_backward``), exactly the structure of the reference's
  backward passes, instead of the (non-differentiable) backward kernel.

Closed forms (K = [phi]x, theta = |phi|):  Jl = I + B K + C K^2,  Jl^-1 = I - K/2 + F K^2  with
B = (1 - cos)/theta^2, C = (theta - sin)/theta^3, F = (1 - theta cos(theta/2) / (2 sin(theta/2)))/theta^2 and their Taylor
series below a threshold where the closed forms cancel (the reference switches at theta <= eps, operation.py:12, 27; the
values agree to rounding, the series is the more accurate of the two in between).
"""
from __future__ import annotations

import torch

from .basics import vec2skew

__all__ = ["so3_Jl", "so3_Jl_inv", "so3_adj", "calcQ", "se3_Jl", "se3_Jl_inv", "se3_adj", "rxso3_Ws", "rxso3_Jl", "rxso3_Jl_inv",
           "rxso3_adj", "sim3_adj", "sim3_Jl", "sim3_Jl_inv", "SO3_Adj", "SO3_Matrix", "SO3_Act_Jacobian", "SO3_Matrix4x4",
           "SO3_Act4_Jacobian", "SE3_Adj", "SE3_Matrix", "SE3_Act_Jacobian", "SE3_Matrix4x4", "SE3_Act4_Jacobian", "RxSO3_Adj",
           "RxSO3_Matrix", "RxSO3_Rotation", "RxSO3_Act_Jacobian", "RxSO3_Matrix4x4", "RxSO3_Act4_Jacobian", "Sim3_Adj", "Sim3_Matrix",
           "Sim3_Act_Jacobian", "Sim3_Matrix4x4", "Sim3_Act4_Jacobian"]


def _plain(t):
    return torch.Tensor.as_subclass(t, torch.Tensor) if type(t) is not torch.Tensor else t


def _eye(k, ref, lead=None):
    I = torch.eye(k, dtype=ref.dtype, device=ref.device)
    return I if lead is None else I.expand(tuple(lead) + (k, k))


def _kernel_route(*ts):
    """device tensors of a kernel dtype and nothing to record"""
    from .. import _C
    return _C._test_backend is None and all(t.is_cuda and t.dtype in (torch.float32, torch.float64) for t in ts) \
        and not (torch.is_grad_enabled() and any(t.requires_grad for t in ts)) and not torch._C._are_functorch_transforms_active()


def _rows_from(kernel, x, in_widths, out_width, n_rows, take):
    """[..., n_rows, take]: row i = kernel(x, e_i)[:take] -- the row-vector products the backward kernels compute
    (``in_widths`` = (saved operand, cotangent), ``out_width`` = what the kernel writes per row)"""
    from . import operation as _op
    E = torch.zeros((n_rows, in_widths[1]), dtype=x.dtype, device=x.device)
    E[:, :n_rows] = torch.eye(n_rows, dtype=x.dtype, device=x.device)
    out = _op._launch(kernel, (x.unsqueeze(-2), E), in_widths, (out_width,))[0]
    return out[..., :take]


# ---------------------------------------------------------------------------------------------------------------------
# coefficient functions (differentiable; series where the closed forms cancel)
# ---------------------------------------------------------------------------------------------------------------------
def _small(dtype):
    return 1e-2 if dtype == torch.float32 else 1e-5          # theta^2 below which the series are used


def _coef_BC(th2):
    """B = (1 - cos t)/t^2, C = (t - sin t)/t^3 at t^2 = th2 [..., 1, 1]"""
    small = th2 < _small(th2.dtype)
    s2 = torch.where(small, torch.ones_like(th2), th2)          # (keeps the unused closed-form lane finite: no NaN gradients)
    t = s2.sqrt()
    B = torch.where(small, 0.5 - th2 / 24 + th2 * th2 / 720, (1 - t.cos()) / s2)
    C = torch.where(small, 1.0 / 6 - th2 / 120 + th2 * th2 / 5040, (t - t.sin()) / (s2 * t))
    return B, C


def _coef_F(th2):
    """F = (1 - t cos(t/2) / (2 sin(t/2))) / t^2"""
    small = th2 < _small(th2.dtype)
    s2 = torch.where(small, torch.ones_like(th2), th2)
    t = s2.sqrt()
    return torch.where(small, 1.0 / 12 + th2 / 720 + th2 * th2 / 30240, (1 - t * (0.5 * t).cos() / (2 * (0.5 * t).sin())) / s2)


def _coef_Q(th2):
    """the three coefficients of calcQ (operation.py:44-55): (t - sin)/t^3, (t^2 + 2 cos - 2)/(2 t^4), (2t - 3 sin + t cos)/(2 t^5)"""
    small = th2 < (0.25 if th2.dtype == torch.float32 else 1e-3)
    s2 = torch.where(small, torch.ones_like(th2), th2)
    t = s2.sqrt()
    s, c = t.sin(), t.cos()
    c1 = torch.where(small, 1.0 / 6 - th2 / 120 + th2 ** 2 / 5040 - th2 ** 3 / 362880, (t - s) / (s2 * t))
    c2 = torch.where(small, 1.0 / 24 - th2 / 720 + th2 ** 2 / 40320 - th2 ** 3 / 3628800, (s2 + 2 * c - 2) / (2 * s2 * s2))
    c3 = torch.where(small, 1.0 / 120 - th2 / 2520 + th2 ** 2 / 120960 - th2 ** 3 / 9979200, (2 * t - 3 * s + t * c) / (2 * s2 * s2 * t))
    return c1, c2, c3


def _th2(phi):
    return phi.square().sum(-1, keepdim=True).unsqueeze(-1)


# ---------------------------------------------------------------------------------------------------------------------
# so3 / se3
# ---------------------------------------------------------------------------------------------------------------------
def so3_Jl(x):
    x = _plain(x)
    if _kernel_route(x):
        return _rows_from("so3_exp_bwd", x, (3, 4), 3, 3, 3)
    K = vec2skew(x)
    B, C = _coef_BC(_th2(x))
    return _eye(3, x) + B * K + C * (K @ K)


def so3_Jl_inv(x):
    x = _plain(x)
    if _kernel_route(x):
        return _rows_from("so3_log_bwd", x, (3, 3), 4, 3, 3)
    K = vec2skew(x)
    return _eye(3, x) - 0.5 * K + _coef_F(_th2(x)) * (K @ K)


def so3_adj(x):
    return vec2skew(_plain(x))


def calcQ(x):
    x = _plain(x)
    if _kernel_route(x):
        return se3_Jl(x)[..., :3, 3:]
    T, P = vec2skew(x[..., :3]), vec2skew(x[..., 3:])
    c1, c2, c3 = _coef_Q(_th2(x[..., 3:]))
    PT, TP = P @ T, T @ P
    PTP = PT @ P
    return 0.5 * T + c1 * (PT + TP + PTP) + c2 * (P @ PT + TP @ P - 3 * PTP) + c3 * (PTP @ P + P @ PTP)


def _blocks(rows):
    return torch.cat([torch.cat(r, dim=-1) for r in rows], dim=-2)


def se3_Jl(x):
    x = _plain(x)
    if _kernel_route(x):
        return _rows_from("se3_exp_bwd", x, (6, 7), 6, 6, 6)
    J = so3_Jl(x[..., 3:])
    return _blocks([[J, calcQ(x)], [torch.zeros_like(J), J]])


def se3_Jl_inv(x):
    x = _plain(x)
    if _kernel_route(x):
        return _rows_from("se3_log_bwd", x, (6, 6), 7, 6, 6)
    Ji = so3_Jl_inv(x[..., 3:])
    return _blocks([[Ji, -Ji @ calcQ(x) @ Ji], [torch.zeros_like(Ji), Ji]])


def se3_adj(x):
    x = _plain(x)
    P, T = vec2skew(x[..., 3:]), vec2skew(x[..., :3])
    return _blocks([[P, T], [torch.zeros_like(P), P]])


# ---------------------------------------------------------------------------------------------------------------------
# rxso3 / sim3
# ---------------------------------------------------------------------------------------------------------------------
def rxso3_Ws(x):
    """W(phi, sigma) with t = W tau in sim3_Exp (operation.py:85-129): A K + B K^2 + C I."""
    x = _plain(x)
    if _kernel_route(x):
        from . import operation as _op
        E = torch.zeros((3, 7), dtype=x.dtype, device=x.device)
        E[:, :3] = torch.eye(3, dtype=x.dtype, device=x.device)
        xi = torch.cat([E.expand(tuple(x.shape[:-1]) + (3, 7))[..., :3], x.unsqueeze(-2).expand(tuple(x.shape[:-1]) + (3, 4))], -1)
        return _op._launch("sim3_exp_fwd", (xi.contiguous(),), (7,), (8,))[0][..., :3].transpose(-1, -2)
    phi, sigma = x[..., :3], x[..., 3:4].unsqueeze(-1)
    th2 = _th2(phi)
    K = vec2skew(phi)
    eps = torch.finfo(x.dtype).eps
    sm_s, sm_t = sigma.abs() <= eps, th2.sqrt() <= eps
    ss = torch.where(sm_s, torch.ones_like(sigma), sigma)
    st2 = torch.where(sm_t, torch.ones_like(th2), th2)
    th = st2.sqrt()
    scale = sigma.exp()
    sn, cs = th.sin(), th.cos()
    # C = (e^s - 1)/s ; general A, B (both sigma and theta regular)
    C = torch.where(sm_s, torch.ones_like(sigma), (scale - 1) / ss)
    den = ss * ss + st2
    a_gen = (ss * scale * sn + (1 - scale * cs) * th) / (th * den)
    b_gen = (C - ((scale * cs - 1) * ss + scale * sn * th) / den) / st2
    # sigma small, theta regular
    a_s0 = (1 - cs) / st2
    b_s0 = (th - sn) / (st2 * th)
    # theta small, sigma regular
    s2 = ss * ss
    a_t0 = ((ss - 1) * scale + 1) / s2
    b_t0 = (0.5 * s2 * scale + scale - 1 - s2 * scale) / (s2 * ss)      # (as the reference writes it, operation.py:115)
    A = torch.where(sm_t, torch.where(sm_s, torch.full_like(sigma, 0.5), a_t0), torch.where(sm_s, a_s0, a_gen))
    Bc = torch.where(sm_t, torch.where(sm_s, torch.full_like(sigma, 1.0 / 6), b_t0), torch.where(sm_s, b_s0, b_gen))
    return A * K + Bc * (K @ K) + C * _eye(3, x)


def _embed(M3, k, fill_eye=True):
    out = (_eye(k, M3, M3.shape[:-2]) if fill_eye else torch.zeros(tuple(M3.shape[:-2]) + (k, k), dtype=M3.dtype, device=M3.device)).clone()
    out[..., :3, :3] = M3
    return out


def rxso3_Jl(x):
    return _embed(so3_Jl(_plain(x)[..., :3]), 4)


def rxso3_Jl_inv(x):
    return _embed(so3_Jl_inv(_plain(x)[..., :3]), 4)


def rxso3_adj(x):
    return _embed(vec2skew(_plain(x)[..., :3]), 4, fill_eye=False)


def sim3_adj(x):
    x = _plain(x)
    tau, phi, sigma = x[..., :3], x[..., 3:6], x[..., 6:]
    T, P = vec2skew(tau), vec2skew(phi)
    top = torch.cat([P + sigma.unsqueeze(-1) * _eye(3, x), T, -tau.unsqueeze(-1)], dim=-1)
    mid = torch.cat([torch.zeros_like(P), P, torch.zeros_like(tau).unsqueeze(-1)], dim=-1)
    return torch.cat([top, mid, torch.zeros(tuple(x.shape[:-1]) + (1, 7), dtype=x.dtype, device=x.device)], dim=-2)


def sim3_Jl(x):
    """the reference's truncated series in the 7x7 adjoint (operation.py:159-165): sum_{k<=5} Xi^k / (k+1)!"""
    x = _plain(x)
    if _kernel_route(x):
        return _rows_from("sim3_exp_bwd", x, (7, 8), 7, 7, 7)
    Xi = sim3_adj(x)
    Xi2 = Xi @ Xi
    Xi4 = Xi2 @ Xi2
    return _eye(7, x) + Xi / 2 + Xi2 / 6 + (Xi @ Xi2) / 24 + Xi4 / 120 + (Xi @ Xi4) / 720


def sim3_Jl_inv(x):
    """I - Xi/2 + Xi^2/12 - Xi^4/720 (operation.py:168-172)"""
    x = _plain(x)
    if _kernel_route(x):
        return _rows_from("sim3_log_bwd", x, (7, 7), 8, 7, 7)
    Xi = sim3_adj(x)
    Xi2 = Xi @ Xi
    return _eye(7, x) - Xi / 2 + Xi2 / 12 - (Xi2 @ Xi2) / 720


# ---------------------------------------------------------------------------------------------------------------------
# group elements: Adj, Matrix, Act Jacobians
# ---------------------------------------------------------------------------------------------------------------------
def _cols_from(kernel, X, wg, wa):
    """Adj(X) [..., wa, wa]: column j = adj kernel(X, e_j)"""
    from . import operation as _op
    E = torch.eye(wa, dtype=X.dtype, device=X.device)
    return _op._launch(kernel, (X.unsqueeze(-2), E), (wg, wa), (wa,))[0].transpose(-1, -2)


def SO3_Adj(X):
    X = _plain(X)
    if _kernel_route(X):
        return _cols_from("so3_adj_fwd", X, 4, 3)
    v, w = X[..., :3], X[..., 3:].unsqueeze(-1)
    I = _eye(3, X)
    return 2 * w * (w * I + vec2skew(v)) - I + 2 * v.unsqueeze(-1) * v.unsqueeze(-2)


def SO3_Matrix(X):
    return SO3_Adj(X)


def SO3_Act_Jacobian(p):
    return vec2skew(-_plain(p))


def SO3_Matrix4x4(X):
    return _embed(SO3_Matrix(X), 4)


def _pad_rows(J, rows):
    return torch.cat([J, torch.zeros(tuple(J.shape[:-2]) + (rows - J.shape[-2], J.shape[-1]), dtype=J.dtype, device=J.device)], dim=-2)


def SO3_Act4_Jacobian(p):
    return _pad_rows(SO3_Act_Jacobian(_plain(p)[..., :3]), 4)


def SE3_Adj(X):
    X = _plain(X)
    if _kernel_route(X):
        return _cols_from("se3_adj_fwd", X, 7, 6)
    R = SO3_Adj(X[..., 3:])
    return _blocks([[R, vec2skew(X[..., :3]) @ R], [torch.zeros_like(R), R]])


def _homogeneous(M3, t):
    top = torch.cat([M3, t.unsqueeze(-1)], dim=-1)
    bottom = torch.zeros(tuple(M3.shape[:-2]) + (1, 4), dtype=M3.dtype, device=M3.device)
    bottom[..., 3] = 1
    return torch.cat([top, bottom], dim=-2)


def SE3_Matrix(X):
    X = _plain(X)
    return _homogeneous(SO3_Matrix(X[..., 3:]), X[..., :3])


def SE3_Act_Jacobian(p):
    p = _plain(p)
    return torch.cat([_eye(3, p, p.shape[:-1]), vec2skew(-p)], dim=-1)


def SE3_Matrix4x4(X):
    return SE3_Matrix(X)


def SE3_Act4_Jacobian(p):
    p = _plain(p)
    return _pad_rows(torch.cat([_eye(3, p, p.shape[:-1]) * p[..., 3:].unsqueeze(-1), vec2skew(-p[..., :3])], dim=-1), 4)


def RxSO3_Adj(X):
    return _embed(SO3_Adj(_plain(X)[..., :4]), 4)


def RxSO3_Matrix(X):
    X = _plain(X)
    return X[..., 4:].unsqueeze(-1) * SO3_Adj(X[..., :4])


def RxSO3_Rotation(X):
    return SO3_Adj(_plain(X)[..., :4])


def RxSO3_Act_Jacobian(p):
    p = _plain(p)
    return torch.cat([vec2skew(-p), p.unsqueeze(-1)], dim=-1)


def RxSO3_Matrix4x4(X):
    return _embed(RxSO3_Matrix(X), 4)


def RxSO3_Act4_Jacobian(p):
    p = _plain(p)
    return _pad_rows(torch.cat([vec2skew(-p[..., :3]), p[..., :3].unsqueeze(-1)], dim=-1), 4)


def Sim3_Adj(X):
    X = _plain(X)
    if _kernel_route(X):
        return _cols_from("sim3_adj_fwd", X, 8, 7)
    t = X[..., :3]
    R, sR = RxSO3_Rotation(X[..., 3:]), RxSO3_Matrix(X[..., 3:])
    z3 = torch.zeros_like(R)
    zc = torch.zeros_like(t).unsqueeze(-1)
    top = torch.cat([sR, vec2skew(t) @ R, -t.unsqueeze(-1)], dim=-1)
    mid = torch.cat([z3, R, zc], dim=-1)
    last = torch.zeros(tuple(X.shape[:-1]) + (1, 7), dtype=X.dtype, device=X.device)
    last[..., 6] = 1
    return torch.cat([top, mid, last], dim=-2)


def Sim3_Matrix(X):
    X = _plain(X)
    return _homogeneous(RxSO3_Matrix(X[..., 3:]), X[..., :3])


def Sim3_Act_Jacobian(p):
    p = _plain(p)
    return torch.cat([SE3_Act_Jacobian(p), p.unsqueeze(-1)], dim=-1)


def Sim3_Matrix4x4(X):
    return Sim3_Matrix(X)


def Sim3_Act4_Jacobian(p):
    p = _plain(p)
    J = torch.cat([SE3_Act4_Jacobian(p)[..., :3, :], p[..., :3].unsqueeze(-1)], dim=-1)
    return _pad_rows(J, 4)
