"""ORACLE (test infrastructure only) -- numpy restatement of pypose/lietensor/operation.py.

This module restates, on the CPU and with the reference's own matrix formulation (3x3 / 6x6 /
7x7 matrices, the same branch masks, the same closed-form coefficient expressions, the same
``nan_to_num`` masking), every function of the reference's batched Lie-group arithmetic that
the HIP kernels in ``pypose_amd/csrc`` replace.  It exists so that parity tests have an
independent implementation to compare against on machines where ``/root/reference`` does not
exist (the GPU box).  It is **not** part of the product: only ``tests/``, ``bench.py``'s
``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may import it.

Pinning: ``tests/golden/lie_golden.npz`` holds inputs/outputs produced by the real reference
(imported from /root/reference by ``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py``
checks every function here against them (fp64 to 1e-12, fp32 to the reference's own noise).

Every op takes/returns 2-D arrays ``[N, W]`` (dtype preserved: float32 stays float32).
Citations ``op.py:a-b`` are to ``/root/reference/pypose/lietensor/operation.py``.
"""
from __future__ import annotations

import numpy as np

_ERR = dict(divide="ignore", invalid="ignore", over="ignore")


def _eps(a):
    return np.finfo(a.dtype).eps


def _mv(M, v):
    """batched matrix @ vector"""
    return np.einsum("nij,nj->ni", M, v)


def _vm(v, M):
    """batched row-vector @ matrix (the ``g.unsqueeze(-2) @ M`` of every backward)"""
    return np.einsum("ni,nij->nj", v, M)


def _eye(n, k, dtype):
    return np.broadcast_to(np.eye(k, dtype=dtype), (n, k, k)).copy()


def _zero_col(a):
    return np.zeros((a.shape[0], 1), dtype=a.dtype)


def pm(x):
    """basics/ops.py:24 -- sign with +1 at 0"""
    return np.sign(np.sign(x) * 2 + 1)


def vec2skew(v):
    """lietensor/basics.py:38-41"""
    n = v.shape[0]
    K = np.zeros((n, 3, 3), dtype=v.dtype)
    K[:, 0, 1], K[:, 0, 2] = -v[:, 2], v[:, 1]
    K[:, 1, 0], K[:, 1, 2] = v[:, 2], -v[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -v[:, 1], v[:, 0]
    return K


# --------------------------------------------------------------------------- Jacobian helpers
def so3_Jl(x):
    """op.py:7-20"""
    with np.errstate(**_ERR):
        K = vec2skew(x)
        theta = np.linalg.norm(x, axis=-1)[:, None, None].astype(x.dtype)
        theta2 = theta ** 2
        idx = theta > _eps(x)
        c1 = np.where(idx, (1 - np.cos(theta)) / theta2, 0.5 - (1.0 / 24.0) * theta2)
        c2 = np.where(idx, (theta - np.sin(theta)) / (theta * theta2), 1.0 / 6.0 - (1.0 / 120) * theta2)
        return (_eye(len(x), 3, x.dtype) + c1 * K + c2 * (K @ K)).astype(x.dtype)


def so3_Jl_inv(x):
    """op.py:23-32"""
    with np.errstate(**_ERR):
        K = vec2skew(x)
        theta = np.linalg.norm(x, axis=-1)[:, None, None].astype(x.dtype)
        idx = theta > _eps(x)
        c2 = idx * np.nan_to_num((1.0 - theta * np.cos(0.5 * theta) / (2.0 * np.sin(0.5 * theta))) / (theta * theta))
        c2 = c2 + (~idx) * 1.0 / 12.0
        return (_eye(len(x), 3, x.dtype) - 0.5 * K + c2 * (K @ K)).astype(x.dtype)


def calcQ(x):
    """op.py:37-58"""
    with np.errstate(**_ERR):
        tau, phi = x[:, :3], x[:, 3:]
        Tau, Phi = vec2skew(tau), vec2skew(phi)
        theta = np.linalg.norm(phi, axis=-1)[:, None, None].astype(x.dtype)
        theta2 = theta ** 2
        theta4 = theta2 ** 2
        idx = theta > _eps(x)
        c1 = idx * np.nan_to_num((theta - np.sin(theta)) / (theta2 * theta)) + (~idx) * (1.0 / 6.0 - (1.0 / 120.0) * theta2)
        c2 = idx * np.nan_to_num((theta2 + 2 * np.cos(theta) - 2) / (2 * theta4)) + (~idx) * (1.0 / 24.0 - (1.0 / 720.0) * theta2)
        c3 = idx * np.nan_to_num((2 * theta - 3 * np.sin(theta) + theta * np.cos(theta)) / (2 * theta4 * theta)) \
            + (~idx) * (1.0 / 120.0 - (1.0 / 2520.0) * theta2)
        Q = 0.5 * Tau + c1 * (Phi @ Tau + Tau @ Phi + Phi @ Tau @ Phi) \
            + c2 * (Phi @ Phi @ Tau + Tau @ Phi @ Phi - 3 * Phi @ Tau @ Phi) \
            + c3 * (Phi @ Tau @ Phi @ Phi + Phi @ Phi @ Tau @ Phi)
        return Q.astype(x.dtype)


def _blk(a, b, c, d):
    return np.concatenate([np.concatenate([a, b], -1), np.concatenate([c, d], -1)], -2)


def se3_Jl(x):
    """op.py:61-65"""
    J = so3_Jl(x[:, 3:])
    return _blk(J, calcQ(x), np.zeros_like(J), J)


def se3_Jl_inv(x):
    """op.py:68-75"""
    Ji, Q = so3_Jl_inv(x[:, 3:]), calcQ(x)
    return _blk(Ji, -Ji @ Q @ Ji, np.zeros_like(Ji), Ji)


def so3_adj(x):
    """op.py:34"""
    return vec2skew(x)


def se3_adj(x):
    """op.py:77-83"""
    Phi = vec2skew(x[:, 3:])
    return _blk(Phi, vec2skew(x[:, :3]), np.zeros_like(Phi), Phi)


def rxso3_adj(x):
    """op.py:142-145"""
    A = np.zeros((len(x), 4, 4), dtype=x.dtype)
    A[:, :3, :3] = vec2skew(x[:, :3])
    return A


def sim3_adj(x):
    """op.py:147-156"""
    tau, phi, sigma = x[:, :3], x[:, 3:6], x[:, 6:]
    ad = np.zeros((len(x), 7, 7), dtype=x.dtype)
    ad[:, :3, :3] = vec2skew(phi) + sigma[:, :, None] * _eye(len(x), 3, x.dtype)
    ad[:, :3, 3:6] = vec2skew(tau)
    ad[:, :3, 6] = -tau
    ad[:, 3:6, 3:6] = vec2skew(phi)
    return ad


def rxso3_Ws(x):
    """op.py:85-129 (four-way sigma/theta branch; condition-3 ``B`` kept exactly as written)"""
    with np.errstate(**_ERR):
        rotation, sigma = x[:, :3], x[:, 3]
        theta = np.linalg.norm(rotation, axis=-1).astype(x.dtype)
        A, B, C = np.zeros_like(theta), np.zeros_like(theta), np.zeros_like(theta)
        sl = np.abs(sigma) > _eps(x)
        tl = theta > _eps(x)
        c1, c2, c3, c4 = (~sl) & (~tl), (~sl) & tl, sl & (~tl), sl & tl
        scale, sigma2, theta2 = np.exp(sigma), sigma * sigma, theta * theta
        theta2_inv = 1.0 / theta2
        C[~sl], A[c1], B[c1] = 1.0, 0.5, 1.0 / 6
        A[c2] = (1.0 - np.cos(theta[c2])) * theta2_inv[c2]
        B[c2] = (theta[c2] - np.sin(theta[c2])) / (theta2[c2] * theta[c2])
        C[sl] = (scale[sl] - 1.0) / sigma[sl]
        A[c3] = (1.0 + (sigma[c3] - 1.0) * scale[c3]) / sigma2[c3]
        B[c3] = (0.5 * sigma2[c3] * scale[c3] + scale[c3] - 1.0 - sigma2[c3] * scale[c3]) / (sigma2[c3] * sigma[c3])
        a4, b4, cc4 = scale[c4] * np.sin(theta[c4]), scale[c4] * np.cos(theta[c4]), theta2[c4] + sigma2[c4]
        A[c4] = (a4 * sigma[c4] + (1 - b4) * theta[c4]) / (theta[c4] * cc4)
        B[c4] = (C[c4] - ((b4 - 1) * sigma[c4] + a4 * theta[c4]) / cc4) * theta2_inv[c4]
        K = vec2skew(rotation)
        A, B, C = A[:, None, None], B[:, None, None], C[:, None, None]
        return (A * K + B * (K @ K) + C * _eye(len(x), 3, x.dtype)).astype(x.dtype)


def rxso3_Jl(x):
    """op.py:132-135"""
    J = _eye(len(x), 4, x.dtype)
    J[:, :3, :3] = so3_Jl(x[:, :3])
    return J


def rxso3_Jl_inv(x):
    """op.py:137-140"""
    J = _eye(len(x), 4, x.dtype)
    J[:, :3, :3] = so3_Jl_inv(x[:, :3])
    return J


def sim3_Jl(x):
    """op.py:159-164 (truncated series in the 7x7 adjoint)"""
    Xi = sim3_adj(x)
    Xi2 = Xi @ Xi
    Xi4 = Xi2 @ Xi2
    I = _eye(len(x), 7, x.dtype)
    return (I + (1.0 / 2.0) * Xi + (1.0 / 6.0) * Xi2 + (1.0 / 24.0) * Xi @ Xi2 + (1.0 / 120.0) * Xi4
            + (1.0 / 720.0) * Xi @ Xi4).astype(x.dtype)


def sim3_Jl_inv(x):
    """op.py:167-172"""
    Xi = sim3_adj(x)
    Xi2 = Xi @ Xi
    Xi4 = Xi2 @ Xi2
    I = _eye(len(x), 7, x.dtype)
    return (I - (1.0 / 2.0) * Xi + (1.0 / 12.0) * Xi2 - (1.0 / 720.0) * Xi4).astype(x.dtype)


# --------------------------------------------------------------------------- group matrices
def SO3_Adj(X):
    """op.py:175-179 (also SO3_Matrix :182-183)"""
    I = _eye(len(X), 3, X.dtype)
    Xv, Xw = X[:, :3], X[:, 3:]
    return (2.0 * Xw[:, :, None] * (Xw[:, :, None] * I + vec2skew(Xv)) - I
            + 2.0 * Xv[:, :, None] * Xv[:, None, :]).astype(X.dtype)


def SE3_Adj(X):
    """op.py:202-210"""
    R = SO3_Adj(X[:, 3:])
    return _blk(R, vec2skew(X[:, :3]) @ R, np.zeros_like(R), R)


def SE3_Matrix(X):
    """op.py:213-217"""
    T = _eye(len(X), 4, X.dtype)
    T[:, :3, :3] = SO3_Adj(X[:, 3:])
    T[:, :3, 3] = X[:, :3]
    return T


def RxSO3_Adj(X):
    """op.py:237-240"""
    A = _eye(len(X), 4, X.dtype)
    A[:, :3, :3] = SO3_Adj(X[:, :4])
    return A


def RxSO3_Matrix(X):
    """op.py:243-244"""
    return X[:, 4:, None] * SO3_Adj(X[:, :4])


def Sim3_Adj(X):
    """op.py:268-276"""
    A = _eye(len(X), 7, X.dtype)
    R = SO3_Adj(X[:, 3:7])
    A[:, :3, :3] = RxSO3_Matrix(X[:, 3:])
    A[:, :3, 3:6] = vec2skew(X[:, :3]) @ R
    A[:, :3, 6] = -X[:, :3]
    A[:, 3:6, 3:6] = R
    return A


def Sim3_Matrix(X):
    """op.py:279-283"""
    T = _eye(len(X), 4, X.dtype)
    T[:, :3, :3] = RxSO3_Matrix(X[:, 3:])
    T[:, :3, 3] = X[:, :3]
    return T


def _mat4(M3):
    T = _eye(len(M3), 4, M3.dtype)
    T[:, :3, :3] = M3
    return T


# Act Jacobians (op.py:186-187, 196-199, 220-234, 251-265, 286-301)
def SO3_Act_Jacobian(p):
    return vec2skew(-p)


def SO3_Act4_Jacobian(p):
    J = np.zeros((len(p), 4, 3), dtype=p.dtype)
    J[:, :3, :3] = SO3_Act_Jacobian(p[:, :3])
    return J


def SE3_Act_Jacobian(p):
    return np.concatenate([_eye(len(p), 3, p.dtype), vec2skew(-p)], -1)


def SE3_Act4_Jacobian(p):
    J = np.zeros((len(p), 4, 6), dtype=p.dtype)
    J[:, :3, :3] = _eye(len(p), 3, p.dtype) * p[:, 3:, None]
    J[:, :3, 3:] = vec2skew(-p[:, :3])
    return J


def RxSO3_Act_Jacobian(p):
    return np.concatenate([vec2skew(-p), p[:, :, None]], -1)


def RxSO3_Act4_Jacobian(p):
    J = np.zeros((len(p), 4, 4), dtype=p.dtype)
    J[:, :3, :3] = SO3_Act_Jacobian(p[:, :3])
    J[:, :3, 3] = p[:, :3]
    return J


def Sim3_Act_Jacobian(p):
    return np.concatenate([SE3_Act_Jacobian(p), p[:, :, None]], -1)


def Sim3_Act4_Jacobian(p):
    J = np.zeros((len(p), 4, 7), dtype=p.dtype)
    J[:, :, :6] = SE3_Act4_Jacobian(p)
    J[:, :3, 6] = p[:, :3]
    return J


# --------------------------------------------------------------------------- SO3
def so3_exp_fwd(x):
    """so3_Exp.forward op.py:343-357"""
    with np.errstate(**_ERR):
        theta = np.linalg.norm(x, axis=-1, keepdims=True).astype(x.dtype)
        th_half, th2 = 0.5 * theta, theta * theta
        th4 = th2 * th2
        idx = theta > _eps(x)
        imag = np.where(idx, np.sin(th_half) / theta, 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4)
        real = np.where(idx, np.cos(th_half), 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4)
        return (np.concatenate([x * imag, real], -1).astype(x.dtype),)


def so3_exp_bwd(x, g):
    """so3_Exp.backward op.py:366-370"""
    return (_vm(g[:, :-1], so3_Jl(x)),)


def so3_log_fwd(X):
    """SO3_Log.forward op.py:307-324"""
    with np.errstate(**_ERR):
        eps = _eps(X)
        v, w = X[:, :3], X[:, 3:]
        vn = np.linalg.norm(v, axis=-1, keepdims=True).astype(X.dtype)
        vl, wl = vn > eps, np.abs(w) > eps
        idx1, idx2, idx3 = vl & wl, vl & (~wl), ~vl
        f = idx1 * np.nan_to_num(2.0 * np.arctan(vn / w) / vn)
        f = f + idx2 * np.nan_to_num(pm(w) * X.dtype.type(np.pi) / vn)
        f = f + idx3 * np.nan_to_num(2.0 * (1.0 / w - vn * vn / (3 * w ** 3)))
        return ((f * v).astype(X.dtype),)


def so3_log_bwd(y, g):
    """SO3_Log.backward op.py:332-337"""
    return (np.concatenate([_vm(g, so3_Jl_inv(y)), _zero_col(y)], -1),)


def so3_act_fwd(X, p):
    """SO3_Act.forward op.py:519-525"""
    Xv, Xw = X[:, :3], X[:, 3:]
    uv = np.cross(Xv, p)
    uv = uv + uv
    return ((p + Xw * uv + np.cross(Xv, uv)).astype(X.dtype),)


def so3_act_bwd(X, out, g):
    """SO3_Act.backward op.py:535-542"""
    m = SO3_Adj(X)
    return (np.concatenate([_vm(g, SO3_Act_Jacobian(out)), _zero_col(X)], -1), _vm(g, m[:, :3, :3]))


def so3_act4_fwd(X, p):
    """op.py:626-629"""
    return (np.concatenate([so3_act_fwd(X, p[:, :3])[0], p[:, 3:]], -1),)


def so3_act4_bwd(X, out, g):
    """op.py:639-645"""
    return (np.concatenate([_vm(g, SO3_Act4_Jacobian(out)), _zero_col(X)], -1), _vm(g, _mat4(SO3_Adj(X))))


def so3_mul_fwd(X, Y):
    """SO3_Mul.forward op.py:832-837"""
    Xv, Xw, Yv, Yw = X[:, :3], X[:, 3:], Y[:, :3], Y[:, 3:]
    Zv = Xw * Yv + Xv * Yw + np.cross(Xv, Yv)
    Zw = Xw * Yw - (Xv * Yv).sum(-1, keepdims=True)
    return (np.concatenate([Zv, Zw], -1).astype(X.dtype),)


def so3_mul_bwd(X, g):
    """op.py:846-852"""
    z = _zero_col(X)
    return (np.concatenate([g[:, :-1], z], -1), np.concatenate([_vm(g[:, :-1], SO3_Adj(X)), z], -1))


def so3_inv_fwd(X):
    """op.py:933-936"""
    return (np.concatenate([-X[:, :3], X[:, 3:]], -1),)


def so3_inv_bwd(Y, g):
    """op.py:945-949"""
    return (np.concatenate([-_vm(g[:, :-1], SO3_Adj(Y)), _zero_col(Y)], -1),)


def so3_adj_fwd(X, a):
    """SO3_AdjXa.forward op.py:728-732"""
    return (_mv(SO3_Adj(X), a),)


def so3_adj_bwd(X, out, g):
    """op.py:743-748 (the saved adj_matrix is SO3_Adj(X), recomputed here)"""
    return (np.concatenate([-_vm(g, so3_adj(out)), _zero_col(X)], -1), _vm(g, SO3_Adj(X)))


def so3_adjt_fwd(X, a):
    """SO3_AdjTXa.forward op.py:1027-1030"""
    return so3_adj_fwd(so3_inv_fwd(X)[0], a)


def so3_adjt_bwd(X, a, g):
    """op.py:1039-1044"""
    a_grad = so3_adj_fwd(X, g)[0]
    return (np.concatenate([-_vm(a, so3_adj(a_grad)), _zero_col(X)], -1), a_grad)


def so3_jinvp_fwd(X, p):
    """SO3Type.Jinvp lietensor.py:257-264"""
    return (_mv(so3_Jl_inv(so3_log_fwd(X)[0]), p),)


def so3_jr_fwd(x):
    """so3Type.Jr lietensor.py:343-351 -> [N,9] row-major"""
    with np.errstate(**_ERR):
        K = vec2skew(x)
        theta = np.linalg.norm(x, axis=-1)[:, None, None].astype(x.dtype)
        I = _eye(len(x), 3, x.dtype)
        Jr = I - (1 - np.cos(theta)) / theta ** 2 * K + (theta - np.sin(theta)) / theta ** 3 * (K @ K)
        return (np.where(theta > _eps(x), Jr, I).astype(x.dtype).reshape(len(x), 9),)


# --------------------------------------------------------------------------- SE3
def se3_exp_fwd(x):
    """se3_Exp.forward op.py:401-405"""
    t = _mv(so3_Jl(x[:, 3:]), x[:, :3])
    return (np.concatenate([t, so3_exp_fwd(x[:, 3:])[0]], -1),)


def se3_exp_bwd(x, g):
    """op.py:413-418"""
    return (_vm(g[:, :-1], se3_Jl(x)),)


def se3_log_fwd(X):
    """SE3_Log.forward op.py:376-382"""
    phi = so3_log_fwd(X[:, 3:])[0]
    tau = _mv(so3_Jl_inv(phi), X[:, :3])
    return (np.concatenate([tau, phi], -1),)


def se3_log_bwd(y, g):
    """op.py:389-395"""
    return (np.concatenate([_vm(g, se3_Jl_inv(y)), _zero_col(y)], -1),)


def se3_act_fwd(X, p):
    """op.py:548-551"""
    return (X[:, :3] + so3_act_fwd(X[:, 3:], p)[0],)


def se3_act_bwd(X, out, g):
    """op.py:561-568"""
    m = SE3_Matrix(X)
    return (np.concatenate([_vm(g, SE3_Act_Jacobian(out)), _zero_col(X)], -1), _vm(g, m[:, :3, :3]))


def se3_act4_fwd(X, p):
    """op.py:651-655"""
    t = so3_act_fwd(X[:, 3:], p[:, :3])[0] + X[:, :3] * p[:, 3:]
    return (np.concatenate([t, p[:, 3:]], -1),)


def se3_act4_bwd(X, out, g):
    """op.py:665-671"""
    return (np.concatenate([_vm(g, SE3_Act4_Jacobian(out)), _zero_col(X)], -1), _vm(g, SE3_Matrix(X)))


def se3_mul_fwd(X, Y):
    """op.py:858-862"""
    t = X[:, :3] + so3_act_fwd(X[:, 3:], Y[:, :3])[0]
    return (np.concatenate([t, so3_mul_fwd(X[:, 3:], Y[:, 3:])[0]], -1),)


def se3_mul_bwd(X, g):
    """op.py:871-877"""
    z = _zero_col(X)
    return (np.concatenate([g[:, :-1], z], -1), np.concatenate([_vm(g[:, :-1], SE3_Adj(X)), z], -1))


def se3_inv_fwd(X):
    """op.py:955-960"""
    q_inv = so3_inv_fwd(X[:, 3:])[0]
    return (np.concatenate([-so3_act_fwd(q_inv, X[:, :3])[0], q_inv], -1),)


def se3_inv_bwd(Y, g):
    """op.py:969-973"""
    return (np.concatenate([-_vm(g[:, :-1], SE3_Adj(Y)), _zero_col(Y)], -1),)


def se3_adj_fwd(X, a):
    """op.py:754-758"""
    return (_mv(SE3_Adj(X), a),)


def se3_adj_bwd(X, out, g):
    """op.py:769-774"""
    return (np.concatenate([-_vm(g, se3_adj(out)), _zero_col(X)], -1), _vm(g, SE3_Adj(X)))


def se3_adjt_fwd(X, a):
    """op.py:1050-1053"""
    return se3_adj_fwd(se3_inv_fwd(X)[0], a)


def se3_adjt_bwd(X, a, g):
    """op.py:1062-1067"""
    a_grad = se3_adj_fwd(X, g)[0]
    return (np.concatenate([-_vm(a, se3_adj(a_grad)), _zero_col(X)], -1), a_grad)


def se3_jinvp_fwd(X, p):
    """SE3Type.Jinvp lietensor.py:422-429"""
    return (_mv(se3_Jl_inv(se3_log_fwd(X)[0]), p),)


# --------------------------------------------------------------------------- RxSO3
def rxso3_exp_fwd(x):
    """op.py:447-451"""
    return (np.concatenate([so3_exp_fwd(x[:, :3])[0], np.exp(x[:, 3:])], -1),)


def rxso3_exp_bwd(x, g):
    """op.py:460-464"""
    return (_vm(g[:, :-1], rxso3_Jl(x)),)


def rxso3_log_fwd(X):
    """op.py:424-428"""
    with np.errstate(**_ERR):
        return (np.concatenate([so3_log_fwd(X[:, :4])[0], np.log(X[:, 4:])], -1),)


def rxso3_log_bwd(y, g):
    """op.py:436-441"""
    return (np.concatenate([_vm(g, rxso3_Jl_inv(y)), _zero_col(y)], -1),)


def rxso3_act_fwd(X, p):
    """op.py:574-577"""
    return (X[:, 4:] * so3_act_fwd(X[:, :4], p)[0],)


def rxso3_act_bwd(X, out, g):
    """op.py:587-594"""
    m = RxSO3_Matrix(X)
    return (np.concatenate([_vm(g, RxSO3_Act_Jacobian(out)), _zero_col(X)], -1), _vm(g, m[:, :3, :3]))


def rxso3_act4_fwd(X, p):
    """op.py:677-680"""
    return (np.concatenate([rxso3_act_fwd(X, p[:, :3])[0], p[:, 3:]], -1),)


def rxso3_act4_bwd(X, out, g):
    """op.py:690-696"""
    return (np.concatenate([_vm(g, RxSO3_Act4_Jacobian(out)), _zero_col(X)], -1), _vm(g, _mat4(RxSO3_Matrix(X))))


def rxso3_mul_fwd(X, Y):
    """op.py:883-887"""
    return (np.concatenate([so3_mul_fwd(X[:, :4], Y[:, :4])[0], X[:, 4:] * Y[:, 4:]], -1),)


def rxso3_mul_bwd(X, g):
    """op.py:896-902"""
    z = _zero_col(X)
    return (np.concatenate([g[:, :-1], z], -1), np.concatenate([_vm(g[:, :-1], RxSO3_Adj(X)), z], -1))


def rxso3_inv_fwd(X):
    """op.py:979-984"""
    with np.errstate(**_ERR):
        return (np.concatenate([so3_inv_fwd(X[:, :4])[0], 1.0 / X[:, 4:]], -1).astype(X.dtype),)


def rxso3_inv_bwd(Y, g):
    """op.py:993-997"""
    return (np.concatenate([-_vm(g[:, :-1], RxSO3_Adj(Y)), _zero_col(Y)], -1),)


def rxso3_adj_fwd(X, a):
    """op.py:780-784"""
    return (_mv(RxSO3_Adj(X), a),)


def rxso3_adj_bwd(X, out, g):
    """op.py:795-800"""
    return (np.concatenate([-_vm(g, rxso3_adj(out)), _zero_col(X)], -1), _vm(g, RxSO3_Adj(X)))


def rxso3_adjt_fwd(X, a):
    """op.py:1073-1076"""
    return rxso3_adj_fwd(rxso3_inv_fwd(X)[0], a)


def rxso3_adjt_bwd(X, a, g):
    """op.py:1085-1090"""
    a_grad = rxso3_adj_fwd(X, g)[0]
    return (np.concatenate([-_vm(a, rxso3_adj(a_grad)), _zero_col(X)], -1), a_grad)


def rxso3_jinvp_fwd(X, p):
    """RxSO3Type.Jinvp lietensor.py:700-707"""
    return (_mv(rxso3_Jl_inv(rxso3_log_fwd(X)[0]), p),)


# --------------------------------------------------------------------------- Sim3
def sim3_exp_fwd(x):
    """op.py:495-500"""
    t = _mv(rxso3_Ws(x[:, 3:]), x[:, :3])
    return (np.concatenate([t, rxso3_exp_fwd(x[:, 3:])[0]], -1),)


def sim3_exp_bwd(x, g):
    """op.py:509-513"""
    return (_vm(g[:, :-1], sim3_Jl(x)),)


def sim3_log_fwd(X):
    """op.py:470-476 (batched 3x3 inverse of Ws)"""
    ps = rxso3_log_fwd(X[:, 3:])[0]
    Ws_inv = np.linalg.inv(rxso3_Ws(ps)).astype(X.dtype)
    return (np.concatenate([_mv(Ws_inv, X[:, :3]), ps], -1),)


def sim3_log_bwd(y, g):
    """op.py:484-489"""
    return (np.concatenate([_vm(g, sim3_Jl_inv(y)), _zero_col(y)], -1),)


def sim3_act_fwd(X, p):
    """op.py:600-603"""
    return (X[:, :3] + rxso3_act_fwd(X[:, 3:], p)[0],)


def sim3_act_bwd(X, out, g):
    """op.py:613-620"""
    m = Sim3_Matrix(X)
    return (np.concatenate([_vm(g, Sim3_Act_Jacobian(out)), _zero_col(X)], -1), _vm(g, m[:, :3, :3]))


def sim3_act4_fwd(X, p):
    """op.py:702-706"""
    t = rxso3_act_fwd(X[:, 3:], p[:, :3])[0] + X[:, :3] * p[:, 3:]
    return (np.concatenate([t, p[:, 3:]], -1),)


def sim3_act4_bwd(X, out, g):
    """op.py:716-722"""
    return (np.concatenate([_vm(g, Sim3_Act4_Jacobian(out)), _zero_col(X)], -1), _vm(g, Sim3_Matrix(X)))


def sim3_mul_fwd(X, Y):
    """op.py:908-912"""
    t = X[:, :3] + rxso3_act_fwd(X[:, 3:], Y[:, :3])[0]
    return (np.concatenate([t, rxso3_mul_fwd(X[:, 3:], Y[:, 3:])[0]], -1),)


def sim3_mul_bwd(X, g):
    """op.py:921-927"""
    z = _zero_col(X)
    return (np.concatenate([g[:, :-1], z], -1), np.concatenate([_vm(g[:, :-1], Sim3_Adj(X)), z], -1))


def sim3_inv_fwd(X):
    """op.py:1003-1008"""
    qs_inv = rxso3_inv_fwd(X[:, 3:])[0]
    return (np.concatenate([-rxso3_act_fwd(qs_inv, X[:, :3])[0], qs_inv], -1),)


def sim3_inv_bwd(Y, g):
    """op.py:1017-1021"""
    return (np.concatenate([-_vm(g[:, :-1], Sim3_Adj(Y)), _zero_col(Y)], -1),)


def sim3_adj_fwd(X, a):
    """op.py:806-810"""
    return (_mv(Sim3_Adj(X), a),)


def sim3_adj_bwd(X, out, g):
    """op.py:821-826"""
    return (np.concatenate([-_vm(g, sim3_adj(out)), _zero_col(X)], -1), _vm(g, Sim3_Adj(X)))


def sim3_adjt_fwd(X, a):
    """op.py:1096-1099"""
    return sim3_adj_fwd(sim3_inv_fwd(X)[0], a)


def sim3_adjt_bwd(X, a, g):
    """op.py:1108-1113"""
    a_grad = sim3_adj_fwd(X, g)[0]
    return (np.concatenate([-_vm(a, sim3_adj(a_grad)), _zero_col(X)], -1), a_grad)


def sim3_jinvp_fwd(X, p):
    """Sim3Type.Jinvp lietensor.py:556-563"""
    return (_mv(sim3_Jl_inv(sim3_log_fwd(X)[0]), p),)


# --------------------------------------------------------------------------- Jinvp / Jr backward
def _central_diff(f, x, g, h):
    """sum_i g_i d f_i / d x_k by central differences (oracle only: O(h^2) accurate)."""
    out = np.zeros_like(x)
    for k in range(x.shape[1]):
        e = np.zeros_like(x)
        e[:, k] = h
        out[:, k] = ((f(x + e) - f(x - e)) * g).sum(-1) / (2 * h)
    return out


def _make_jinvp_bwd(g):
    Jl_inv = {"so3": so3_Jl_inv, "se3": se3_Jl_inv, "sim3": sim3_Jl_inv, "rxso3": rxso3_Jl_inv}[g]
    log_fwd, log_bwd = globals()[f"{g}_log_fwd"], globals()[f"{g}_log_bwd"]

    def bwd(X, p, gr):
        """Jinvp backward as autograd sees it (lietensor.py:257-264 etc.): through Jl_inv(x) p by
        (here) central differences in float64, then through <Group>_Log.backward."""
        x = log_fwd(X.astype(np.float64))[0]
        p64, g64 = p.astype(np.float64), gr.astype(np.float64)
        h = _central_diff(lambda xx: _mv(Jl_inv(xx), p64), x, g64, 1e-6)
        gX = log_bwd(x, h)[0]
        gp = _vm(g64, Jl_inv(x))
        return gX.astype(X.dtype), gp.astype(X.dtype)
    bwd.__name__ = f"{g}_jinvp_bwd"
    return bwd


so3_jinvp_bwd, se3_jinvp_bwd, sim3_jinvp_bwd, rxso3_jinvp_bwd = (_make_jinvp_bwd(g) for g in ("so3", "se3", "sim3", "rxso3"))


def so3_jr_bwd(x, G):
    """so3.Jr backward (autograd through lietensor.py:343-351), by central differences in float64."""
    x64 = x.astype(np.float64)
    return (_central_diff(lambda xx: so3_jr_fwd(xx)[0], x64, G.astype(np.float64), 1e-6).astype(x.dtype),)


# --------------------------------------------------------------------------- registry
GROUPS = {"so3": (3, 4), "se3": (6, 7), "sim3": (7, 8), "rxso3": (4, 5)}   # (algebra, group) widths
_OPNAMES = ["exp_fwd", "exp_bwd", "log_fwd", "log_bwd", "inv_fwd", "inv_bwd", "mul_fwd", "mul_bwd",
            "act_fwd", "act_bwd", "act4_fwd", "act4_bwd", "adj_fwd", "adj_bwd", "adjt_fwd", "adjt_bwd", "jinvp_fwd",
            "jinvp_bwd"]
OPS = {f"{g}_{o}": globals()[f"{g}_{o}"] for g in GROUPS for o in _OPNAMES}
OPS["so3_jr_fwd"] = so3_jr_fwd
OPS["so3_jr_bwd"] = so3_jr_bwd
# compositions of the above (no golden of their own): p.add_(d) of the optimizers (lietensor.py:60-65),
# Exp(d[:da]) * p with the step d zero-padded to the group width
# ---- pinhole reprojection with closed-form blocks (function/geometry.py:37-57 homo2cart, :60-113 point2pixel, :171-226 reprojerr)
def _reproj_parts(X, p, cam):
    q = se3_act_fwd(X, p)[0]
    K = cam[:, :9].reshape(-1, 3, 3)
    h = _mv(K, q)
    hz = h[:, 2]
    tiny = np.finfo(X.dtype).tiny
    az = np.abs(hz)
    clamped = az < tiny
    den = np.where(hz < 0, -1.0, 1.0).astype(X.dtype) * np.maximum(az, tiny)     # pm(0) = +1
    return q, K, h, den, clamped


def se3_reproj_fwd(X, p, cam):
    """reprojerr(points, pixels, K, pose, reduction='none') for one (pose, point) pair per row: geometry.py:224-226"""
    q, K, h, den, _ = _reproj_parts(X, p, cam)
    return (h[:, :2] / den[:, None] - cam[:, 9:11],)


def se3_reproj_lin(X, p, cam):
    """the residual and its Jacobian blocks [d r/d pose (2x6) | d r/d point (2x3)] in closed form: chain rule through
    homo2cart (abs().clamp_() passes no gradient where it clamps), K, and SE3_Act.backward's [I | -skew(q)] / R (op.py:561-568)"""
    q, K, h, den, clamped = _reproj_parts(X, p, cam)
    n = X.shape[0]
    inv = 1.0 / den
    pix = h[:, :2] * inv[:, None]
    D = np.zeros((n, 2, 3), dtype=X.dtype)
    D[:, 0, 0] = inv
    D[:, 1, 1] = inv
    D[:, :, 2] = -pix * np.where(clamped, 0.0, inv)[:, None]
    M = D @ K                                                     # [n, 2, 3]
    Jq = np.concatenate([np.broadcast_to(np.eye(3, dtype=X.dtype), (n, 3, 3)), -vec2skew(q)], -1)    # [n, 3, 6]
    R = SE3_Matrix(X)[:, :3, :3]
    J = np.concatenate([M @ Jq, M @ R], -1)                        # [n, 2, 9]
    return (pix - cam[:, 9:11], J.reshape(n, 18))


def reproj_vjp(J, g):
    Jm = J.reshape(-1, 2, 9)
    v = _vm(g, Jm)
    return (np.concatenate([v[:, :6], _zero_col(J)], -1), v[:, 6:9])


EXTRA_OPS = {"se3_reproj_fwd": se3_reproj_fwd, "se3_reproj_lin": se3_reproj_lin, "reproj_vjp": reproj_vjp}

COMPOSED_OPS = {}
for _g, (_da, _dg) in GROUPS.items():
    COMPOSED_OPS[f"{_g}_retract"] = (lambda g, da: lambda d, X: (
        OPS[f"{g}_mul_fwd"](OPS[f"{g}_exp_fwd"](np.ascontiguousarray(d[:, :da]))[0], X)[0],))(_g, _da)


def op_signature(name):
    """(input widths, output widths) of op ``name`` -- the shapes the C ABI uses."""
    if name == "so3_jr_fwd":
        return (3,), (9,)
    if name == "so3_jr_bwd":
        return (3, 9), (3,)
    g, o = name.split("_", 1)
    da, dg = GROUPS[g]
    return {
        "exp_fwd": ((da,), (dg,)), "exp_bwd": ((da, dg), (da,)),
        "log_fwd": ((dg,), (da,)), "log_bwd": ((da, da), (dg,)),
        "inv_fwd": ((dg,), (dg,)), "inv_bwd": ((dg, dg), (dg,)),
        "mul_fwd": ((dg, dg), (dg,)), "mul_bwd": ((dg, dg), (dg, dg)),
        "act_fwd": ((dg, 3), (3,)), "act_bwd": ((dg, 3, 3), (dg, 3)),
        "act4_fwd": ((dg, 4), (4,)), "act4_bwd": ((dg, 4, 4), (dg, 4)),
        "adj_fwd": ((dg, da), (da,)), "adj_bwd": ((dg, da, da), (dg, da)),
        "adjt_fwd": ((dg, da), (da,)), "adjt_bwd": ((dg, da, da), (dg, da)),
        "jinvp_fwd": ((dg, da), (da,)), "jinvp_bwd": ((dg, da, da), (dg, da)),
    }[o]
