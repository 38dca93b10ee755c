"""Goldens for the pinhole helpers from the REAL reference:
    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_geometry_golden.py"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
import pypose as pp  # noqa: E402

D = torch.float64
torch.manual_seed(2)
f, H, W = 2.0, 3.0, 4.0
K = torch.tensor([[f, 0.1, H / 2], [0, 1.5 * f, W / 2], [0, 0, 1]], dtype=D)
pts = torch.randn(2, 6, 3, dtype=D) + torch.tensor([0, 0, 4.0], dtype=D)
T = pp.randn_SE3(2, sigma=0.3, dtype=D)
px = pp.point2pixel(pts, K, T)
S = {"K": K, "pts": pts, "T": T.tensor(), "px": px, "px_noext": pp.point2pixel(pts, K),
     "homo": pp.cart2homo(pts), "cart": pp.homo2cart(torch.cat([pts, torch.tensor([0., -0., 2, -3, 1e-320, 1]).view(1, 6, 1).expand(2, 6, 1)], -1))}
depth = torch.rand(2, 6, dtype=D) + 1
S["depth"], S["back"] = depth, pp.pixel2point(px, depth, K)
obs = px + 0.1 * torch.randn_like(px)
S["obs"] = obs
for red in ("none", "norm", "sum"):
    S["err_" + red] = pp.reprojerr(pts, obs, K, T, reduction=red)
# gradient of the summed squared error w.r.t. points and pose
p = pts.clone().requires_grad_(True)
Tp = pp.Parameter(T.clone())
loss = pp.reprojerr(p, obs, K, Tp, reduction='none').square().sum()
gp, gT = torch.autograd.grad(loss, [p, Tp])
S["g_pts"], S["g_T"] = gp, gT
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "geometry_golden.npz"),
                    **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in S.items()})
print("ok", len(S))
