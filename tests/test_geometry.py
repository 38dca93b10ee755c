"""Pinhole helpers (pypose_amd.function.geometry) against goldens recorded from the real reference: on the CPU
through the oracle backend, on the GPU through the HIP SE3_Act kernel."""
import os

import numpy as np
import pytest
import torch

import pypose_amd as pp
from tests.oracle_backend import oracle_backend

G = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "geometry_golden.npz")))


def _check(dev, tol):
    t = lambda k: torch.from_numpy(G[k]).to(dev)
    K, pts, T = t("K"), t("pts"), pp.SE3(t("T"))
    close = lambda a, k: np.testing.assert_allclose(a.detach().cpu().numpy(), G[k], rtol=0, atol=tol)
    close(pp.cart2homo(pts), "homo")
    bottom = torch.tensor([0., -0., 2, -3, 1e-320, 1], dtype=torch.float64, device=dev).view(1, 6, 1).expand(2, 6, 1)
    got, want = pp.homo2cart(torch.cat([pts, bottom], -1)).cpu().numpy(), G["cart"]
    np.testing.assert_allclose(got, want, rtol=1e-12)                   # (values up to 1e308: relative)
    close(pp.point2pixel(pts, K, T), "px")
    close(pp.point2pixel(pts, K), "px_noext")
    close(pp.pixel2point(t("px"), t("depth"), K), "back")
    for red in ("none", "norm", "sum"):
        close(pp.reprojerr(pts, t("obs"), K, T, reduction=red), "err_" + red)
    p = pts.clone().requires_grad_(True)
    Tp = pp.Parameter(T.clone())
    loss = pp.reprojerr(p, t("obs"), K, Tp).square().sum()
    gp, gT = torch.autograd.grad(loss, [p, Tp])
    close(gp, "g_pts"), close(gT, "g_T")
    with pytest.raises(AssertionError, match="Points shape"):
        pp.point2pixel(pts[..., :2], K)
    with pytest.raises(AssertionError, match="Type incorrect"):
        pp.point2pixel(pts, K, T.tensor())
    with pytest.raises(AssertionError, match="Reduction method"):
        pp.reprojerr(pts, t("obs"), K, T, reduction="mean")
    with pytest.raises(AssertionError, match="fx Cannot"):
        pp.pixel2point(t("px"), t("depth"), torch.zeros(3, 3, dtype=torch.float64, device=dev))


def test_geometry_host_logic():
    with oracle_backend():
        _check("cpu", 1e-12)


@pytest.mark.gpu
def test_geometry_gpu():
    _check("cuda:0", 1e-11)
