"""Goldens for chspline / bspline / geodesic_loss from the REAL reference:
    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_spline_golden.py"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
import pypose as pp  # noqa: E402

D = torch.float64
torch.manual_seed(5)
S = {}
# the reference docstring's B-spline input (spline.py:171-182) and a random batch of trajectories
a1 = pp.euler2SO3(torch.tensor([0., 0., 0.], dtype=D))
a2 = pp.euler2SO3(torch.tensor([torch.pi / 4., torch.pi / 3., torch.pi / 2.], dtype=D))
doc = pp.SE3(torch.tensor([[[0., 4., 0., *a1.tolist()], [0., 3., 0., *a1.tolist()], [0., 2., 0., *a1.tolist()],
                            [0., 1., 0., *a1.tolist()], [1., 0., 1., *a2.tolist()], [2., 0., 1., *a2.tolist()],
                            [3., 0., 1., *a2.tolist()], [4., 0., 1., *a2.tolist()]]], dtype=D))
S["bs_doc_in"], S["bs_doc_out"] = doc.tensor(), pp.bspline(doc, 0.1).tensor()
S["bs_doc_extra"] = pp.bspline(doc, 0.1, extrapolate=True).tensor()
traj = pp.randn_SE3(3, 2, 9, sigma=0.7, dtype=D)
S["bs_in"] = traj.tensor()
for name, iv in (("01", 0.1), ("03", 0.3), ("06", 0.6)):
    S["bs_out_" + name] = pp.bspline(traj, iv).tensor()
S["bs_out_extra"] = pp.bspline(traj, 0.25, extrapolate=True).tensor()
S["bs_min_in"] = traj[0, 0, :4].tensor()
S["bs_min_out"] = pp.bspline(traj[0, 0, :4], 0.2).tensor()
# gradient of a scalar of the spline w.r.t. the control poses
ctrl = pp.Parameter(traj[0].clone())
coef = torch.randn(2, (9 - 3) * 4 + 1, 7, dtype=D)
S["bs_coef"] = coef
S["bs_grad"] = torch.autograd.grad((pp.bspline(ctrl, 0.25).tensor() * coef).sum(), ctrl)[0]
# Hermite spline: the docstring's points (spline.py:43-59) and random batches
pts = torch.tensor([[[0., 0., 0.], [1., .5, 0.1], [0., 1., 0.2], [1., 1.5, 0.4], [1.5, 0., 0.], [2., 1.5, 0.4], [2.5, 0., 0.],
                     [1.75, 0.75, 0.2], [2.25, 0.75, 0.2], [3., 1.5, 0.4], [3., 0., 0.], [4., 0., 0.], [4., 1.5, 0.4],
                     [5., 1., 0.2], [4., 0.75, 0.2], [5., 0., 0.]]], dtype=D)
S["ch_doc_in"], S["ch_doc_out"] = pts, pp.chspline(pts, 0.1)
rnd = torch.randn(2, 3, 7, 5, dtype=D)
S["ch_in"] = rnd
for name, iv in (("02", 0.2), ("03", 0.3), ("07", 0.7)):
    S["ch_out_" + name] = pp.chspline(rnd, iv)
S["ch_two_in"] = rnd[0, 0, :2]
S["ch_two_out"] = pp.chspline(rnd[0, 0, :2], 0.25)
# geodesic loss on every LieTensor type (module/loss.py:6-38; exported as pp.geodesic_loss, pp.module.GeodesicLoss)
for kind in ("SO3", "SE3", "Sim3", "RxSO3", "so3", "se3", "sim3", "rxso3"):
    x, y = getattr(pp, "randn_" + kind)(6, dtype=D), getattr(pp, "randn_" + kind)(6, dtype=D)
    S[f"geo_{kind}_x"], S[f"geo_{kind}_y"] = x.tensor(), y.tensor()
    for red in ("none", "mean", "sum"):
        S[f"geo_{kind}_{red}"] = pp.geodesic_loss(x, y, reduction=red)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "spline_golden.npz"),
                    **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in S.items()})
print("ok", len(S))
