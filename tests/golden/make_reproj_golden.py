"""Goldens for the closed-form reprojection linearisation from the REAL reference:
    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_reproj_golden.py

* per-observation residuals and Jacobian blocks of pp.reprojerr (function/geometry.py:171-226): d r / d pose by the
  reference's autograd (left tangent, the [tangent, 0] gradient convention) and d r / d point, one pair per row, with a
  general intrinsic matrix (skew, non-unit last row), a point behind the camera and a depth below the clamp;
* the trajectory of the reference's dense LM on a small bundle-adjustment problem whose model is
  ``reprojerr(points[pidx], pixels, K, poses[cidx])`` (every observation its own (pose, point) pair)."""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
import pypose as pp  # noqa: E402

D = torch.float64
torch.manual_seed(5)
n = 40
X = pp.randn_SE3(n, sigma=0.4, dtype=D)
p = torch.randn(n, 3, dtype=D) + torch.tensor([0, 0, 5.0], dtype=D)
p[3, 2] = -4.0                                   # behind the camera
K = torch.tensor([[520.0, 0.7, 321.0], [0.0, 515.0, 242.0], [1e-3, -2e-3, 1.0]], dtype=D)
uv = torch.randn(n, 2, dtype=D) * 3
S = {"X": X.tensor(), "p": p, "K": K, "uv": uv}
Xp = pp.Parameter(X.clone())
pr = p.clone().requires_grad_(True)
r = pp.reprojerr(pr.unsqueeze(-2), uv.unsqueeze(-2), K, Xp).squeeze(-2)          # [n, 2]
S["r"] = r
Jx, Jp = [], []
for a in range(2):
    gX, gP = torch.autograd.grad(r[:, a].sum(), [Xp, pr], retain_graph=True)
    Jx.append(gX[:, :6]), Jp.append(gP)
    assert float(gX[:, 6].abs().max()) == 0
S["J_pose"], S["J_point"] = torch.stack(Jx, 1), torch.stack(Jp, 1)               # [n, 2, 6], [n, 2, 3]
# the clamp of homo2cart: camera-frame depth exactly zero / below tiny (identity pose so that the depth is the point's z)
Xi = pp.identity_SE3(3, dtype=D)
pc = torch.tensor([[0.3, -0.2, 0.0], [0.3, -0.2, 1e-320], [0.3, -0.2, -1e-320]], dtype=D)
Kc = torch.tensor([[2.0, 0, 0.5], [0, 3.0, 0.25], [0, 0, 1.0]], dtype=D)
Xc = pp.Parameter(Xi.clone())
pcr = pc.clone().requires_grad_(True)
rc = pp.reprojerr(pcr.unsqueeze(-2), torch.zeros(3, 1, 2, dtype=D), Kc, Xc).squeeze(-2)
S["clamp_p"], S["clamp_K"], S["clamp_r"] = pc, Kc, rc
Jc = []
for a in range(2):
    gP, = torch.autograd.grad(rc[:, a].sum(), [pcr], retain_graph=True)
    Jc.append(gP)
S["clamp_J_point"] = torch.stack(Jc, 1)


# ---- bundle adjustment through reprojerr, dense LM of the reference
class BA(torch.nn.Module):
    def __init__(self, poses, points):
        super().__init__()
        self.poses = pp.Parameter(poses)
        self.points = torch.nn.Parameter(points)

    def forward(self, cidx, pidx, pixels, K):
        return pp.reprojerr(self.points[pidx].unsqueeze(-2), pixels.unsqueeze(-2), K, self.poses[cidx]).squeeze(-2)


C, N = 4, 24
g = torch.Generator().manual_seed(9)
pts = torch.randn(N, 3, generator=g, dtype=D) + torch.tensor([0, 0, 6.0], dtype=D)
poses = pp.se3(0.15 * torch.randn(C, 6, generator=g, dtype=D)).Exp()
Kb = torch.tensor([[400.0, 0, 160.0], [0, 400.0, 120.0], [0, 0, 1.0]], dtype=D)
vis = torch.rand(C, N, generator=g) < 0.8
cidx, pidx = vis.nonzero(as_tuple=True)
pix = pp.point2pixel(pts[pidx].unsqueeze(-2), Kb, poses[cidx]).squeeze(-2) + 0.3 * torch.randn(len(cidx), 2, generator=g, dtype=D)
poses0 = poses * pp.se3(0.03 * torch.randn(C, 6, generator=g, dtype=D)).Exp()
pts0 = pts + 0.05 * torch.randn(N, 3, generator=g, dtype=D)
S.update({"ba/poses0": poses0.tensor(), "ba/points0": pts0, "ba/cidx": cidx, "ba/pidx": pidx, "ba/pixels": pix, "ba/K": Kb})
for name, kw in (("plain", {}), ("huber", {"kernel": pp.optim.kernel.Huber(delta=1.0)})):
    model = BA(poses0.clone(), pts0.clone())
    opt = pp.optim.LM(model, strategy=pp.optim.strategy.TrustRegion(radius=1e4), min=1e-6, vectorize=True, **kw)
    losses = [float(opt.step((cidx, pidx, pix, Kb))) for _ in range(4)]
    S[f"ba/{name}/loss"] = np.asarray(losses)
    S[f"ba/{name}/poses"] = model.poses.detach().tensor().clone()
    S[f"ba/{name}/points"] = model.points.detach().clone()
    print(name, losses)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reproj_golden.npz"),
                    **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in S.items()})
print("ok", len(S))
