"""ORACLE (test infrastructure only) -- gradients through the reference's product scans and IMU pre-integration.

* ``scan_bwd``: numpy, sequential.  The reference differentiates cumprod (pypose/basics/ops.py:27-56, 153-204) through the
  ``<G>_Mul`` backward rule (pypose/lietensor/operation.py:846-852, 871-877, 896-902, 921-927: for Z = A*B, A_grad = [g[:-1], 0],
  B_grad = [g[:-1] @ Adj(A), 0]).  The same rule applied to the SEQUENTIAL chain y_i = y_{i-1} x_i (or x_i y_{i-1}) walked
  backwards gives the same gradient; this restatement shares neither the reference's Hillis-Steele tree nor the kernel's
  anchored closed form.
* ``imu_grads``: plain torch fp64 restatement of pypose/module/imu_preintegrator.py:359-384, 422-426 (quaternions as [x,y,z,w]
  tensors, an explicit loop for the rotation products) differentiated by autograd.  The reference's gradients of group
  elements are LEFT-TANGENT vectors (operation.py header convention); the restatement gets them as ordinary gradients by
  parametrising r0 <- Exp(eps) r0 at eps = 0 and by pairing the rotation cotangent with the first-order log of
  rot_f * stopgrad(rot_f)^-1.

Pinned against tests/golden/grad_golden.npz (gradients recorded from the real reference by tests/golden/make_grad_golden.py)
in tests/test_grad_oracle_host.py.  Nothing under pypose_amd/ imports this."""
import numpy as np
import torch

from . import lie_np as L

_W = {"so3": 4, "se3": 7, "sim3": 8, "rxso3": 5}


def scan_bwd(group, x, y, g, left):
    """x, y, g: [B, L, W] (scan along axis 1); returns gx [B, L, W] (last component 0)."""
    mul_bwd = getattr(L, f"{group}_mul_bwd")
    B, n, W = x.shape
    gx = np.zeros_like(x)
    G = np.zeros((B, W), x.dtype)                       # total cotangent on y_i, carried backwards
    for i in range(n - 1, -1, -1):
        G = G + np.concatenate([g[:, i, :W - 1], np.zeros((B, 1), x.dtype)], -1)
        if left:                                        # y_i = x_i * y_{i-1}: x_i is the LEFT factor
            gx[:, i] = G
            G = mul_bwd(x[:, i], G)[1] if i > 0 else G
        else:                                           # y_i = y_{i-1} * x_i: x_i is the RIGHT factor
            if i > 0:
                gx[:, i] = mul_bwd(y[:, i - 1], G)[1]
            else:
                gx[:, i] = G
    # element 0 is never an OUTPUT of a Mul in the reference's rounds (ops.py:34-35 only overwrites indices >= step): y_0 IS x_0,
    # so its cotangent reaches x_0 as it stands, last embedding component included
    gx[:, 0, W - 1] = g[:, 0, W - 1]
    return gx


# ---- IMU: plain-torch restatement ---------------------------------------------------------------------------------
def _qmul(a, b):
    av, aw, bv, bw = a[..., :3], a[..., 3:], b[..., :3], b[..., 3:]
    return torch.cat([aw * bv + bw * av + torch.linalg.cross(av, bv), aw * bw - (av * bv).sum(-1, keepdim=True)], -1)


def _qrot(q, p):
    v, w = q[..., :3], q[..., 3:]
    uv = 2 * torch.linalg.cross(v, p)
    return p + w * uv + torch.linalg.cross(v, uv)


def _qinv(q):
    return torch.cat([-q[..., :3], q[..., 3:]], -1)


def _qexp(phi):
    th = phi.norm(dim=-1, keepdim=True)
    small = th < 1e-8
    ths = torch.where(small, torch.ones_like(th), th)
    imag = torch.where(small, 0.5 - th ** 2 / 48, torch.sin(0.5 * ths) / ths)
    real = torch.where(small, 1 - th ** 2 / 8, torch.cos(0.5 * ths))
    return torch.cat([imag * phi, real], -1)


def imu_states(dt, gyro, acc, r0, v0, p0, gravity, rot_known=None):
    """[B,F,*] inputs, r0 [B,1,4], v0/p0 [B,1,3] -> rot [B,F,4], vel, pos (imu_preintegrator.py:359-384, 422-426)"""
    B, F = dt.shape[:2]
    dr = _qexp(gyro * dt)                                                     # :360
    P = [torch.zeros(B, 4, dtype=dt.dtype) + torch.tensor([0, 0, 0, 1.0], dtype=dt.dtype)]
    for f in range(F):                                                        # :361-362 (cumprod, right products)
        P.append(_qmul(P[-1], dr[:, f]))
    incre_r = torch.stack(P, 1)                                               # [B, F+1, 4]
    grav = gravity.reshape(1, 1, 3).expand(B, F, 3)
    if rot_known is not None:
        a = acc - _qrot(_qinv(rot_known), grav)                               # :364-365
    else:
        a = acc - _qrot(_qinv(_qmul(r0.expand(B, F, 4), incre_r[:, 1:])), grav)   # :367-370
    Ra = _qrot(incre_r[:, :F], a)
    z = torch.zeros(B, 1, 3, dtype=dt.dtype)
    incre_v = torch.cumsum(torch.cat([z, Ra * dt], 1), 1)                     # :372-374
    incre_p = torch.cumsum(torch.cat([z, incre_v[:, :F] * dt + Ra * 0.5 * dt ** 2], 1), 1)   # :376-378
    incre_t = torch.cumsum(dt, 1)                                             # :380-381
    rot = _qmul(r0.expand(B, F, 4), incre_r[:, 1:])                           # :422-426
    vel = v0 + _qrot(r0.expand(B, F, 4), incre_v[:, 1:])
    pos = p0 + _qrot(r0.expand(B, F, 4), incre_p[:, 1:]) + v0 * incre_t
    return rot, vel, pos


def imu_grads(dt, gyro, acc, r0, v0, p0, gravity, Gr, Gv, Gp, rot_known=None):
    """numpy in / numpy out: gradients of  sum(Gr . d rot) + sum(Gv * vel) + sum(Gp * pos)  in the reference's convention
    (Gr: left-tangent cotangent [B,F,3] of rot; returned g_r0 is a left-tangent vector [B,1,3])."""
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).double()
    dt, gyro, acc, v0, p0 = (T(a).requires_grad_(True) for a in (dt, gyro, acc, v0, p0))
    eps = torch.zeros(r0.shape[0], 1, 3, dtype=torch.float64, requires_grad=True)
    r0e = _qmul(_qexp(eps), T(r0))
    rot, vel, pos = imu_states(dt, gyro, acc, r0e, v0, p0, T(gravity), None if rot_known is None else T(rot_known))
    rel = _qmul(rot, _qinv(rot.detach()))                    # = Exp(delta_f) at first order
    loss = (T(Gr) * (2 * rel[..., :3] / rel[..., 3:])).sum() + (T(Gv) * vel).sum() + (T(Gp) * pos).sum()
    g = torch.autograd.grad(loss, [dt, gyro, acc, eps, v0, p0])
    return dict(zip(("dt", "gyro", "acc", "r0", "v0", "p0"), (t.numpy() for t in g)))
