"""tests/golden/imu_golden.npz from the REAL reference's IMUPreintegrator (build container only):
    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_imu_golden.py
"""
import os, sys
import numpy as np
import torch
sys.dont_write_bytecode = True
import pypose as pp  # the reference

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "imu_golden.npz")
D = torch.float64
S = {}
g = torch.Generator().manual_seed(4)
B, F = 3, 150
dt = 0.005 + 0.001 * torch.rand(B, F, 1, dtype=D, generator=g)
gyro = 0.3 * torch.randn(B, F, 3, dtype=D, generator=g)
acc = torch.randn(B, F, 3, dtype=D, generator=g) + torch.tensor([0, 0, 9.81], dtype=D)
S["dt"], S["gyro"], S["acc"] = dt.numpy(), gyro.numpy(), acc.numpy()
torch.manual_seed(2)
r0 = pp.randn_SO3(B, 1, dtype=D); p0 = torch.randn(B, 1, 3, dtype=D); v0 = torch.randn(B, 1, 3, dtype=D)
S["r0"], S["p0"], S["v0"] = r0.numpy(), p0.numpy(), v0.numpy()


def mk(**kw):
    return pp.module.IMUPreintegrator(pos=torch.zeros(3, dtype=D), rot=pp.identity_SO3(dtype=D), vel=torch.zeros(3, dtype=D), **kw).to(D)

# (1) reset=True, covariance, default init
m = mk(reset=True, prop_cov=True)
o = m(dt, gyro, acc)
for k in ("rot", "vel", "pos", "cov"):
    S[f"case1/{k}"] = o[k].detach().numpy()
# (2) explicit init state
o = m(dt, gyro, acc, init_state={"pos": p0, "rot": r0, "vel": v0})
for k in ("rot", "vel", "pos", "cov"):
    S[f"case2/{k}"] = o[k].detach().numpy()
# (3) known rotations supplied
rot_known = pp.randn_SO3(B, F, dtype=D)
S["rot_known"] = rot_known.numpy()
o = m(dt, gyro, acc, rot=rot_known)
for k in ("rot", "vel", "pos", "cov"):
    S[f"case3/{k}"] = o[k].detach().numpy()
# (4) reset=False: two consecutive windows accumulate state and covariance
m2 = mk(reset=False, prop_cov=True)
o1 = m2(dt[:, :70], gyro[:, :70], acc[:, :70])
o2 = m2(dt[:, 70:], gyro[:, 70:], acc[:, 70:])
for k in ("rot", "vel", "pos", "cov"):
    S[f"case4a/{k}"] = o1[k].detach().numpy(); S[f"case4b/{k}"] = o2[k].detach().numpy()
# (5) no covariance
m3 = mk(reset=True, prop_cov=False)
o = m3(dt, gyro, acc)
S["case5/pos"] = o["pos"].detach().numpy()
# (6) per-batch covariances
gc = (1e-3 + torch.rand(B, 1, 3, dtype=D, generator=g) * 1e-3) ** 2
ac = (5e-2 + torch.rand(B, 1, 3, dtype=D, generator=g) * 5e-2) ** 2
S["gc"], S["ac"] = gc.numpy(), ac.numpy()
o = m(dt, gyro, acc, gyro_cov=gc, acc_cov=ac)
S["case6/cov"] = o["cov"].detach().numpy()
# scans (basics/ops.py): cumprod left/right on SE3 and SO3
X = pp.randn_SE3(2, 133, dtype=D); S["scan/X"] = X.numpy()
S["scan/se3_left"] = pp.cumprod(X, dim=1, left=True).numpy(); S["scan/se3_right"] = pp.cumprod(X, dim=1, left=False).numpy()
Q = pp.randn_SO3(70, 3, dtype=D); S["scan/Q"] = Q.numpy()
S["scan/so3_dim0"] = pp.cumprod(Q, dim=0, left=True).numpy()
np.savez_compressed(OUT, **S)
print("wrote", OUT, len(S), os.path.getsize(OUT))
