"""tests/golden/grad_golden.npz: GRADIENTS through the reference's product scans and through its IMUPreintegrator, recorded
from the REAL reference (build container only):
    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_grad_golden.py
(the reference differentiates log2(L) Hillis-Steele rounds, basics/ops.py:27-36, and the composed IMU graph,
module/imu_preintegrator.py:359-384, 422-426; pypose_amd's one-pass backward kernels must reproduce these numbers)
"""
import os, sys
import numpy as np
import torch
sys.dont_write_bytecode = True
import pypose as pp  # the reference

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "grad_golden.npz")
D = torch.float64
S = {}
torch.manual_seed(11)
gen = torch.Generator().manual_seed(12)

# ---- scans: loss = sum(W * cumprod(X)) with random embedding-space weights
RAND = {"SO3": pp.randn_SO3, "SE3": pp.randn_SE3, "Sim3": pp.randn_Sim3, "RxSO3": pp.randn_RxSO3}
SHAPES = {"a": ((2, 133), 1), "b": ((70, 3), 0), "c": ((2, 5, 3), 1), "d": ((1, 1), 1), "e": ((3, 2), 1), "f": ((200,), 0)}
for gname, rnd in RAND.items():
    for tag, (shape, dim) in SHAPES.items():
        X0 = rnd(*shape, sigma=0.7, dtype=D)
        W = torch.randn(X0.shape, dtype=D, generator=gen)
        S[f"scan/{gname}/{tag}/X"] = X0.tensor().numpy()
        S[f"scan/{gname}/{tag}/W"] = W.numpy()
        for left in (True, False):
            X = X0.clone().requires_grad_(True)
            Y = pp.cumprod(X, dim=dim, left=left)
            (Y.tensor() * W).sum().backward()
            S[f"scan/{gname}/{tag}/{'L' if left else 'R'}/Y"] = Y.detach().tensor().numpy()
            S[f"scan/{gname}/{tag}/{'L' if left else 'R'}/gX"] = X.grad.tensor().numpy() if hasattr(X.grad, "tensor") else X.grad.numpy()
# a long, drifting SE3 trajectory (translations grow to ~100): the case absolute-pose transports lose digits on
X0 = pp.randn_SE3(1, 1500, sigma=0.2, dtype=D)
X0 = pp.SE3(torch.cat([X0.tensor()[..., :3] + torch.tensor([0.3, 0, 0], dtype=D), X0.tensor()[..., 3:]], -1))
W = torch.randn(X0.shape, dtype=D, generator=gen)
S["scan/SE3/long/X"], S["scan/SE3/long/W"] = X0.tensor().numpy(), W.numpy()
for left in (True, False):
    X = X0.clone().requires_grad_(True)
    Y = pp.cumprod(X, dim=1, left=left)
    (Y.tensor() * W).sum().backward()
    S[f"scan/SE3/long/{'L' if left else 'R'}/Y"] = Y.detach().tensor().numpy()
    S[f"scan/SE3/long/{'L' if left else 'R'}/gX"] = X.grad.tensor().numpy() if hasattr(X.grad, "tensor") else X.grad.numpy()

# ---- IMU: loss = sum(Wr * rot) + sum(Wv * vel) + sum(Wp * pos)
B, F = 3, 150
dt0 = 0.005 + 0.001 * torch.rand(B, F, 1, dtype=D, generator=gen)
gyro0 = 0.3 * torch.randn(B, F, 3, dtype=D, generator=gen)
acc0 = torch.randn(B, F, 3, dtype=D, generator=gen) + torch.tensor([0, 0, 9.81], dtype=D)
Wr = torch.randn(B, F, 4, dtype=D, generator=gen)
Wv = torch.randn(B, F, 3, dtype=D, generator=gen)
Wp = torch.randn(B, F, 3, dtype=D, generator=gen)
r00 = pp.randn_SO3(B, 1, dtype=D); p00 = torch.randn(B, 1, 3, dtype=D); v00 = torch.randn(B, 1, 3, dtype=D)
rotk = pp.randn_SO3(B, F, dtype=D)
for k, v in dict(dt=dt0, gyro=gyro0, acc=acc0, Wr=Wr, Wv=Wv, Wp=Wp, r0=r00.tensor(), p0=p00, v0=v00, rotk=rotk.tensor()).items():
    S[f"imu/{k}"] = v.numpy()


def mk(**kw):
    return pp.module.IMUPreintegrator(pos=torch.zeros(3, dtype=D), rot=pp.identity_SO3(dtype=D), vel=torch.zeros(3, dtype=D), **kw).to(D)


def run(tag, init, known, prop_cov):
    dt, gyro, acc = (t.clone().requires_grad_(True) for t in (dt0, gyro0, acc0))
    kw = {}
    leaves = {"dt": dt, "gyro": gyro, "acc": acc}
    if init:
        r0 = r00.clone().requires_grad_(True); p0 = p00.clone().requires_grad_(True); v0 = v00.clone().requires_grad_(True)
        kw["init_state"] = {"pos": p0, "rot": r0, "vel": v0}
        leaves.update(r0=r0, p0=p0, v0=v0)
    if known:
        kw["rot"] = rotk
    o = mk(reset=True, prop_cov=prop_cov)(dt, gyro, acc, **kw)
    loss = (o["rot"].tensor() * Wr).sum() + (o["vel"] * Wv).sum() + (o["pos"] * Wp).sum()
    loss.backward()
    S[f"imu/{tag}/loss"] = loss.detach().numpy()
    for k, v in leaves.items():
        g = v.grad
        S[f"imu/{tag}/g_{k}"] = (g.tensor() if hasattr(g, "tensor") else g).numpy()


run("plain", False, False, False)
run("init", True, False, False)
run("known", True, True, False)
run("cov", True, False, True)          # with covariance propagation (a function of detached states)
# the example's loss (examples/module/imu/imu_corrector.py:69-74) on case "init"
dt, gyro, acc = dt0.clone(), gyro0.clone().requires_grad_(True), acc0.clone().requires_grad_(True)
o = mk(reset=True, prop_cov=False)(dt, gyro, acc, init_state={"pos": p00, "rot": r00, "vel": v00})
gt_pos = torch.randn(B, F, 3, dtype=D, generator=gen)
S["imu/example/gt_pos"] = gt_pos.numpy()
loss = torch.nn.functional.mse_loss(o["pos"], gt_pos) + 5e2 * (rotk * o["rot"].Inv()).Log().norm(dim=-1).mean()
loss.backward()
S["imu/example/loss"] = loss.detach().numpy()
S["imu/example/g_gyro"], S["imu/example/g_acc"] = gyro.grad.numpy(), acc.grad.numpy()
np.savez_compressed(OUT, **S)
print("wrote", OUT, len(S), os.path.getsize(OUT))
