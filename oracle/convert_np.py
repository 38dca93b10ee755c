"""ORACLE (test infrastructure only -- never imported by the product) -- numpy restatement of the
reference's data-format conversions either side of the Lie hot path:

  mat2so3_fwd     pypose/lietensor/convert.py:97-146   (mat2SO3: 3x3 rotation -> [x,y,z,w])
  euler2so3_fwd   pypose/lietensor/convert.py:650-663  (euler2SO3)
  so3_euler_fwd   pypose/lietensor/lietensor.py:1151-1173 (LieTensor.euler)
  *_bwd           vector-Jacobian products by central differences of the forward in fp64 (the reference
                  differentiates the same formulas with autograd; goldens from it pin both)
  read_g2o        examples/module/pgo/pgo_dataset.py:22-51 (line-by-line g2o parser, info2mat)

Pinned against tests/golden/convert_golden.npz, generated from the real reference by
tests/golden/make_convert_golden.py.
"""
import numpy as np


def mat2so3_fwd(m, atol=1e-5):
    m = np.asarray(m)
    R = m.reshape(-1, 3, 3)
    rt = np.swapaxes(R, -1, -2)
    d2 = rt[:, 2, 2] < atol
    d0_d1 = rt[:, 0, 0] > rt[:, 1, 1]
    d0_nd1 = rt[:, 0, 0] < -rt[:, 1, 1]
    t0 = 1 + rt[:, 0, 0] - rt[:, 1, 1] - rt[:, 2, 2]
    q0 = np.stack([rt[:, 1, 2] - rt[:, 2, 1], t0, rt[:, 0, 1] + rt[:, 1, 0], rt[:, 2, 0] + rt[:, 0, 2]], -1)
    t1 = 1 - rt[:, 0, 0] + rt[:, 1, 1] - rt[:, 2, 2]
    q1 = np.stack([rt[:, 2, 0] - rt[:, 0, 2], rt[:, 0, 1] + rt[:, 1, 0], t1, rt[:, 1, 2] + rt[:, 2, 1]], -1)
    t2 = 1 - rt[:, 0, 0] - rt[:, 1, 1] + rt[:, 2, 2]
    q2 = np.stack([rt[:, 0, 1] - rt[:, 1, 0], rt[:, 2, 0] + rt[:, 0, 2], rt[:, 1, 2] + rt[:, 2, 1], t2], -1)
    t3 = 1 + rt[:, 0, 0] + rt[:, 1, 1] + rt[:, 2, 2]
    q3 = np.stack([t3, rt[:, 1, 2] - rt[:, 2, 1], rt[:, 2, 0] - rt[:, 0, 2], rt[:, 0, 1] - rt[:, 1, 0]], -1)
    c = [d2 & d0_d1, d2 & ~d0_d1, ~d2 & d0_nd1, ~d2 & ~d0_nd1]
    c = [x[:, None].astype(m.dtype) for x in c]
    q = q0 * c[0] + q1 * c[1] + q2 * c[2] + q3 * c[3]
    with np.errstate(invalid="ignore"):
        q = q / (2 * np.sqrt(t0[:, None] * c[0] + t1[:, None] * c[1] + t2[:, None] * c[2] + t3[:, None] * c[3]))
    return (q[:, [1, 2, 3, 0]].astype(m.dtype),)


def euler2so3_fwd(e):
    e = np.asarray(e)
    roll, pitch, yaw = e[:, 0], e[:, 1], e[:, 2]
    cy, sy = np.cos(yaw * 0.5), np.sin(yaw * 0.5)
    cp, sp = np.cos(pitch * 0.5), np.sin(pitch * 0.5)
    cr, sr = np.cos(roll * 0.5), np.sin(roll * 0.5)
    q = np.stack([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
                  cr * cp * cy + sr * sp * sy], -1)
    return (q.astype(e.dtype),)


def so3_euler_fwd(q, eps=2e-4):
    q = np.asarray(q)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    xx, yy, zz, ww = x * x, y * y, z * z, w * w
    t0 = 2 * (w * x + y * z)
    t1 = (ww + zz) - (xx + yy)
    t2 = 2 * (w * y - z * x) / (xx + yy + zz + ww)
    t3 = 2 * (w * z + x * y)
    t4 = (ww + xx) - (yy + zz)
    flag = np.abs(t2) < 1. - eps
    pm = np.sign(np.sign(t2) * 2 + 1)
    roll = np.where(flag, np.arctan2(t0, t1), 0.0)
    yaw = np.where(flag, np.arctan2(t3, t4), -2 * pm * np.arctan2(x, w))
    pitch = np.arcsin(np.clip(t2, -1, 1))
    return (np.stack([roll, pitch, yaw], -1).astype(q.dtype),)


def _vjp_fd(fwd, a, g, h=1e-6, **kw):
    a64, g64 = np.asarray(a, np.float64), np.asarray(g, np.float64)
    out = np.zeros_like(a64)
    for k in range(a64.shape[1]):
        d = np.zeros_like(a64)
        d[:, k] = h
        out[:, k] = ((fwd(a64 + d, **kw)[0] - fwd(a64 - d, **kw)[0]) / (2 * h) * g64).sum(-1)
    return (out.astype(np.asarray(a).dtype),)


def mat2so3_bwd(m, g, atol=1e-5):
    return _vjp_fd(mat2so3_fwd, m, g, atol=atol)


def euler2so3_bwd(e, g):
    return _vjp_fd(euler2so3_fwd, e, g)


def so3_euler_bwd(q, g, eps=2e-4):
    """analytic chain rule through lietensor.py:1151-1173 (torch.where / clamp pass gradients of the taken branch)"""
    q, g = np.asarray(q, np.float64), np.asarray(g, np.float64)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    n = x * x + y * y + z * z + w * w
    t0, t1 = 2 * (w * x + y * z), (w * w + z * z) - (x * x + y * y)
    u = 2 * (w * y - z * x)
    t2 = u / n
    t3, t4 = 2 * (w * z + x * y), (w * w + x * x) - (y * y + z * z)
    flag = np.abs(t2) < 1. - eps
    pm = np.sign(np.sign(t2) * 2 + 1)
    Z = np.zeros_like(x)
    d = lambda dx, dy, dz, dw: np.stack([dx, dy, dz, dw], -1)
    dt0, dt1 = d(2 * w, 2 * z, 2 * y, 2 * x), d(-2 * x, -2 * y, 2 * z, 2 * w)
    dt3, dt4 = d(2 * y, 2 * x, 2 * w, 2 * z), d(2 * x, -2 * y, -2 * z, 2 * w)
    du, dn = d(-2 * z, 2 * w, -2 * x, 2 * y), d(2 * x, 2 * y, 2 * z, 2 * w)
    dt2 = du / n[:, None] - (u / n ** 2)[:, None] * dn
    droll = np.where(flag[:, None], (t1[:, None] * dt0 - t0[:, None] * dt1) / (t0 ** 2 + t1 ** 2)[:, None], 0.0)
    dyaw_reg = (t4[:, None] * dt3 - t3[:, None] * dt4) / (t3 ** 2 + t4 ** 2)[:, None]
    dyaw_sing = (-2 * pm / (x * x + w * w))[:, None] * d(w, Z, Z, -x)
    dyaw = np.where(flag[:, None], dyaw_reg, dyaw_sing)
    with np.errstate(divide="ignore", invalid="ignore"):
        dpitch = np.where((np.abs(t2) <= 1)[:, None], dt2 / np.sqrt(1 - t2 ** 2)[:, None], 0.0)
    out = g[:, 0:1] * droll + g[:, 1:2] * dpitch + g[:, 2:3] * dyaw
    return (out,)


OPS = {"mat2so3_fwd": mat2so3_fwd, "mat2so3_bwd": mat2so3_bwd, "euler2so3_fwd": euler2so3_fwd,
       "euler2so3_bwd": euler2so3_bwd, "so3_euler_fwd": so3_euler_fwd, "so3_euler_bwd": so3_euler_bwd}


def info2mat(info):
    """21 upper-triangular entries, row by row -> symmetric 6x6 (pgo_dataset.py:22-29)"""
    mat = np.zeros((6, 6))
    ix = 0
    for i in range(6):
        mat[i, i:] = info[ix:ix + (6 - i)]
        mat[i:, i] = info[ix:ix + (6 - i)]
        ix += 6 - i
    return mat


def read_g2o(path):
    """ids [N], nodes [N,7], edges [E,2], poses [E,7], infos [E,6,6] (float64 / int64), file order
    (pgo_dataset.py:33-44: VERTEX_SE3:QUAT id x y z qx qy qz qw; EDGE_SE3:QUAT i j x y z qx qy qz qw + 21 info)"""
    ids, nodes, edges, poses, infos = [], [], [], [], []
    with open(path) as f:
        for line in f:
            line = line.split()
            if not line:
                continue
            if line[0] == 'VERTEX_SE3:QUAT':
                ids.append(int(line[1]))
                nodes.append(np.array(line[2:], dtype=np.float64))
            elif line[0] == 'EDGE_SE3:QUAT':
                edges.append(np.array(line[1:3], dtype=np.int64))
                poses.append(np.array(line[3:10], dtype=np.float64))
                infos.append(info2mat(np.array(line[10:], dtype=np.float64)))
    return (np.array(ids, dtype=np.int64), np.stack(nodes), np.stack(edges), np.stack(poses), np.stack(infos))
