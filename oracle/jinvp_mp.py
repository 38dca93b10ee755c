"""ORACLE (test infrastructure only) -- the gradient of ``X.Jinvp(p)`` in 40-digit arithmetic (mpmath).

Why it exists: the reference obtains Jinvp's backward by plain autograd through the closed forms of ``so3_Jl_inv`` / ``calcQ``
(/root/reference/pypose/lietensor/lietensor.py:422-429, operation.py:23-32, 37-58, 68-75).  Those coefficients divide by
theta^2 ... theta^5 after cancelling to that order, so below theta ~ 1e-3 neither the reference's fp64 autograd nor a
finite-difference oracle in fp64 (oracle/lie_np.py ``_central_diff``: error ~ eps64 / (theta^4 h)) is an anchor: they
disagree with each other at 1e-4 ... 1e-2 there.  This file evaluates the SAME closed forms (no series) with 40 digits and
differentiates them by central differences with h = 1e-15 in that arithmetic: the truth both are approximating, good to
better than 1e-12 down to theta = 1e-6.  ``tests/test_jinvp_small_angle.py`` compares the kernels' arithmetic with it (host
build and GPU), and shows where the reference itself stops being accurate.

Only SE3 (the group whose Q-term makes the problem acute) and SO3.  Pure-Python loops: use it for a few hundred rows.
"""
import mpmath as mp
import numpy as np

mp.mp.dps = 40


def _skew(v):
    return mp.matrix([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def _so3_Jl_inv(phi):
    """operation.py:23-32 (closed form only)"""
    K = _skew(phi)
    th = mp.sqrt(phi[0] ** 2 + phi[1] ** 2 + phi[2] ** 2)
    c2 = (1 - th * mp.cos(th / 2) / (2 * mp.sin(th / 2))) / (th * th)
    return mp.eye(3) - K / 2 + c2 * (K * K)


def _calcQ(tau, phi):
    """operation.py:37-58 (closed form only)"""
    T, P = _skew(tau), _skew(phi)
    th = mp.sqrt(phi[0] ** 2 + phi[1] ** 2 + phi[2] ** 2)
    t2 = th * th
    t4 = t2 * t2
    c1 = (th - mp.sin(th)) / (t2 * th)
    c2 = (t2 + 2 * mp.cos(th) - 2) / (2 * t4)
    c3 = (2 * th - 3 * mp.sin(th) + th * mp.cos(th)) / (2 * t4 * th)
    return T / 2 + c1 * (P * T + T * P + P * T * P) + c2 * (P * P * T + T * P * P - 3 * P * T * P) \
        + c3 * (P * T * P * P + P * P * T * P)


def _se3_jinvp(x, p):
    """se3_Jl_inv(x) p, operation.py:68-75: [[Ji, -Ji Q Ji], [0, Ji]]"""
    tau, phi = x[:3], x[3:]
    Ji = _so3_Jl_inv(phi)
    Q = _calcQ(tau, phi)
    pt, pr = mp.matrix(p[:3]), mp.matrix(p[3:])
    a = Ji * pr
    top = Ji * pt - Ji * (Q * a)
    return [top[i] for i in range(3)] + [a[i] for i in range(3)]


def _so3_jinvp(x, p):
    a = _so3_Jl_inv(x) * mp.matrix(p)
    return [a[i] for i in range(3)]


def jinvp_algebra_grad(group, x, p, g, h="1e-15"):
    """d/dx sum_i g_i (Jl_inv(x) p)_i for rows x [n, 6 | 3] (algebra coordinates), float64 in / float64 out, and the gradient
    with respect to p (= g @ Jl_inv(x)); everything in between at mp.dps digits."""
    f = {"se3": _se3_jinvp, "so3": _so3_jinvp}[group]
    h = mp.mpf(h)
    n, w = x.shape
    gx, gp = np.zeros((n, w)), np.zeros((n, w))
    for r in range(n):
        xr = [mp.mpf(float(v)) for v in x[r]]
        pr = [mp.mpf(float(v)) for v in p[r]]
        gr = [mp.mpf(float(v)) for v in g[r]]
        for k in range(w):
            xp, xm = list(xr), list(xr)
            xp[k] += h
            xm[k] -= h
            fp, fm = f(xp, pr), f(xm, pr)
            gx[r, k] = float(sum(gi * (a - b) for gi, a, b in zip(gr, fp, fm)) / (2 * h))
            e = [mp.mpf(0)] * w
            e[k] = mp.mpf(1)
            gp[r, k] = float(sum(gi * a for gi, a in zip(gr, f(xr, e))))
    return gx, gp
