"""ORACLE (test infrastructure only) -- numpy restatement of IMUPreintegrator
(pypose/module/imu_preintegrator.py:314-465) and of cumprod (pypose/basics/ops.py:153-204),
written as explicit sequential loops.  Pinned against tests/golden/imu_golden.npz (outputs of
the real reference) by tests/test_imu_host.py."""
import numpy as np

from . import lie_np as L


def quat_mul(a, b):
    return L.so3_mul_fwd(a, b)[0]


def quat_matrix(q):
    """rows = R e_j transposed back, i.e. the rotation matrix (lietensor.py:281-285 via Act)."""
    n = len(q)
    cols = [L.so3_act_fwd(q, np.tile(e, (n, 1)).astype(q.dtype))[0] for e in np.eye(3)]
    return np.stack(cols, -1)


def cumprod(X, mul, left=True):
    """inclusive product scan along axis 1 of [B, L, W] (basics/ops.py:49-56 semantics)."""
    out = X.copy()
    for i in range(1, X.shape[1]):
        out[:, i] = mul(X[:, i], out[:, i - 1]) if left else mul(out[:, i - 1], X[:, i])
    return out


def integrate(dt, gyro, acc, gravity, rot=None, init_rot=None):
    """imu_preintegrator.py:359-384"""
    B, F = dt.shape[:2]
    dr = L.so3_exp_fwd((gyro * dt).reshape(-1, 3))[0].reshape(B, F, 4)
    ident = np.zeros((B, 1, 4), dt.dtype); ident[..., 3] = 1
    w = np.concatenate([ident, dr], 1)
    incre_r = cumprod(w, quat_mul, left=False)
    g = np.tile(gravity, (B * F, 1)).astype(dt.dtype)
    if rot is not None:
        Rw = rot.reshape(-1, 4)
    else:
        r0 = np.broadcast_to(init_rot, (B, 1, 4)).repeat(F, 1).reshape(-1, 4) if init_rot is not None else None
        Rw = incre_r[:, 1:].reshape(-1, 4) if r0 is None else quat_mul(r0, incre_r[:, 1:].reshape(-1, 4))
    a = acc - L.so3_act_fwd(L.so3_inv_fwd(Rw)[0], g)[0].reshape(B, F, 3)
    Ra = L.so3_act_fwd(incre_r[:, :F].reshape(-1, 4), a.reshape(-1, 3))[0].reshape(B, F, 3)
    z = np.zeros((B, 1, 3), dt.dtype)
    incre_v = np.cumsum(np.concatenate([z, Ra * dt], 1), 1)
    incre_p = np.cumsum(np.concatenate([z, incre_v[:, :F] * dt + Ra * 0.5 * dt ** 2], 1), 1)
    incre_t = np.concatenate([np.zeros((B, 1, 1), dt.dtype), np.cumsum(dt, 1)], 1)
    return dict(a=a, Dp=incre_p[:, 1:], Dv=incre_v[:, 1:], Dr=incre_r[:, 1:], Dt=incre_t[:, 1:], w=w[:, 1:])


def predict(r0, v0, p0, st):
    """imu_preintegrator.py:422-426"""
    B, F = st["Dr"].shape[:2]
    R0 = np.broadcast_to(r0, (B, 1, 4)).repeat(F, 1).reshape(-1, 4)
    rot = quat_mul(R0, st["Dr"].reshape(-1, 4)).reshape(B, F, 4)
    vel = v0 + L.so3_act_fwd(R0, st["Dv"].reshape(-1, 3))[0].reshape(B, F, 3)
    pos = p0 + L.so3_act_fwd(R0, st["Dp"].reshape(-1, 3))[0].reshape(B, F, 3) + v0 * st["Dt"]
    return rot, vel, pos


def propagate_cov(dt, Rk, Rij, a, init_cov, gyro_cov, acc_cov):
    """imu_preintegrator.py:431-465: cov = sum_k P_k Bc_k P_k^T, P_k = A_k ... A_F (A_F = I),
    restated literally (suffix products), not as the recurrence the kernel uses."""
    B, F = dt.shape[:2]
    A = np.tile(np.eye(9, dtype=dt.dtype), (B, F + 1, 1, 1))
    Rkm = quat_matrix(Rk.reshape(-1, 4)).reshape(B, F, 3, 3)
    Rjm = quat_matrix(Rij.reshape(-1, 4)).reshape(B, F, 3, 3)
    Ha = L.vec2skew(a.reshape(-1, 3)).reshape(B, F, 3, 3)
    h = dt[..., None]
    A[:, :-1, 0:3, 0:3] = np.swapaxes(Rkm, -1, -2)
    A[:, :-1, 3:6, 0:3] = -(Rjm @ Ha) * h
    A[:, :-1, 6:9, 0:3] = -0.5 * (Rjm @ Ha) * h ** 2
    A[:, :-1, 6:9, 3:6] = np.eye(3, dtype=dt.dtype) * h
    Bg = np.zeros((B, F, 9, 3), dt.dtype); Ba = np.zeros((B, F, 9, 3), dt.dtype)
    phi = L.so3_log_fwd(Rk.reshape(-1, 4))[0]
    Bg[..., 0:3, :] = L.so3_jr_fwd(phi)[0].reshape(B, F, 3, 3) * h
    Ba[..., 3:6, :] = Rjm * h
    Ba[..., 6:9, :] = 0.5 * Rjm * h ** 2
    Cg = np.zeros((B, gyro_cov.shape[1], 3, 3), dt.dtype); Ca = np.zeros((B, acc_cov.shape[1], 3, 3), dt.dtype)
    for i in range(3):
        Cg[..., i, i] = np.broadcast_to(gyro_cov, (B,) + gyro_cov.shape[1:])[..., i]
        Ca[..., i, i] = np.broadcast_to(acc_cov, (B,) + acc_cov.shape[1:])[..., i]
    Bc = (Bg @ Cg @ np.swapaxes(Bg, -1, -2) + Ba @ Ca @ np.swapaxes(Ba, -1, -2)) / h
    Bc = np.concatenate([np.broadcast_to(init_cov, (B, 9, 9))[:, None], Bc], 1)
    cov = np.zeros((B, 9, 9), dt.dtype)
    P = np.tile(np.eye(9, dtype=dt.dtype), (B, 1, 1))
    for k in range(F, -1, -1):
        P = A[:, k] @ P
        cov += P @ Bc[:, k] @ np.swapaxes(P, -1, -2)
    return cov


def preintegrate(dt, gyro, acc, gravity=9.81007, r0=None, v0=None, p0=None, rot=None, init_cov=None,
                 gyro_cov=None, acc_cov=None, Rij0=None):
    B, F = dt.shape[:2]
    D = dt.dtype
    r0 = np.array([[[0, 0, 0, 1.0]]], D) if r0 is None else r0
    v0 = np.zeros((1, 1, 3), D) if v0 is None else v0
    p0 = np.zeros((1, 1, 3), D) if p0 is None else p0
    # the reference builds its gravity / covariance buffers in float32 and casts the module
    # (imu_preintegrator.py:104-116): defaults are float32-rounded values
    st = integrate(dt, gyro, acc, np.array([0, 0, np.float32(gravity)], D), rot=rot, init_rot=r0)
    rotp, vel, pos = predict(r0, v0, p0, st)
    gyro_cov = np.full((B, 1, 3), np.float32((3.2e-3) ** 2), D) if gyro_cov is None else gyro_cov
    acc_cov = np.full((B, 1, 3), np.float32((8e-2) ** 2), D) if acc_cov is None else acc_cov
    init_cov = np.zeros((1, 9, 9), D) if init_cov is None else init_cov
    Rij = st["Dr"] if Rij0 is None else quat_mul(np.broadcast_to(Rij0, (B, 1, 4)).repeat(F, 1).reshape(-1, 4),
                                                 st["Dr"].reshape(-1, 4)).reshape(B, F, 4)
    cov = propagate_cov(dt, st["w"], Rij, st["a"], init_cov, gyro_cov, acc_cov)
    return dict(rot=rotp, vel=vel, pos=pos, cov=cov, Rij=Rij[:, -1:])
