"""Closed-form reprojection linearisation on the MI355X (csrc/reproj.hip through the C ABI): against the reference's
recorded residuals / autograd blocks, against the oracle at a size it finishes in seconds, and the LM trajectory of the
reference on a bundle-adjustment problem written with pp.reprojerr."""
import numpy as np
import pytest
import torch

import pypose_amd as pp
from pypose_amd import _C
from tests.reproj_util import G, t, check_ops, run_ba, cam_rows

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dtype,rtol", [(torch.float64, 1e-11), (torch.float32, 5e-6)])
def test_kernels_match_the_reference(dtype, rtol):
    assert _C._test_backend is None
    check_ops(DEV, dtype, rtol)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 2e-5)])
def test_kernels_match_the_oracle_at_size(dtype, tol):
    from oracle import lie_np
    n = 200_003                                             # ragged against every tile shape
    g = torch.Generator().manual_seed(3)
    X = pp.se3(0.5 * torch.randn(n, 6, generator=g, dtype=torch.float64)).to(DEV).Exp().tensor().to(dtype).contiguous()
    p = (torch.randn(n, 3, generator=g, dtype=torch.float64) + torch.tensor([0, 0, 6.0])).to(DEV, dtype).contiguous()
    K = torch.tensor([[520.0, 0.7, 321.0], [0.0, 515.0, 242.0], [1e-3, -2e-3, 1.0]], dtype=dtype, device=DEV)
    cam = cam_rows(K, torch.randn(n, 2, generator=g, dtype=torch.float64).to(DEV, dtype))
    r, J = _C.row_op("se3_reproj_lin", [X, p, cam], (2, 18))
    (r2,) = _C.row_op("se3_reproj_fwd", [X, p, cam], (2,))
    assert torch.equal(r, r2)
    wr, wJ = lie_np.se3_reproj_lin(X.cpu().double().numpy(), p.cpu().double().numpy(), cam.cpu().double().numpy())
    # fp32: a depth close to zero amplifies the input rounding; compare where the oracle's own conditioning is sane
    ok = np.abs(wJ).max(-1) < 1e4
    assert ok.mean() > 0.9
    sr, sJ = np.abs(wr[ok]).max(), np.abs(wJ[ok]).max()
    assert np.abs(r.cpu().double().numpy()[ok] - wr[ok]).max() <= tol * sr * 50
    assert np.abs(J.cpu().double().numpy()[ok] - wJ[ok]).max() <= tol * sJ * 50
    gg = torch.randn(n, 2, generator=g, dtype=torch.float64).to(DEV, dtype)
    gX, gp = _C.row_op("reproj_vjp", [J, gg], (7, 3))
    wX, wp = lie_np.reproj_vjp(J.cpu().double().numpy(), gg.cpu().double().numpy())
    assert np.abs(gX.cpu().double().numpy() - wX)[ok].max() <= tol * np.abs(wX[ok]).max()
    assert np.abs(gp.cpu().double().numpy() - wp)[ok].max() <= tol * np.abs(wp[ok]).max()


def test_reprojerr_autograd_on_the_device():
    D = torch.float64
    X = pp.Parameter(pp.SE3(t("X", DEV)))
    p = t("p", DEV).clone().requires_grad_(True)
    K = t("K", DEV).clone().requires_grad_(True)
    r = pp.reprojerr(p.unsqueeze(-2), t("uv", DEV).unsqueeze(-2), K, X).squeeze(-2)
    np.testing.assert_allclose(r.detach().cpu().numpy(), G["r"], rtol=1e-11, atol=1e-9)
    w = torch.randn(r.shape, dtype=D, generator=torch.Generator().manual_seed(0)).to(DEV)
    gX, gp, gK = torch.autograd.grad((r * w).sum(), [X, p, K])
    wn = w.cpu().numpy()
    np.testing.assert_allclose(gX[:, :6].cpu().numpy(), np.einsum("na,nac->nc", wn, G["J_pose"]), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(gp.cpu().numpy(), np.einsum("na,nac->nc", wn, G["J_point"]), rtol=1e-9, atol=1e-9)
    K2 = t("K", DEV).clone().requires_grad_(True)
    r2 = pp.homo2cart(pp.SE3(t("X", DEV)).unsqueeze(-2).Act(t("p", DEV).unsqueeze(-2)) @ K2.mT).squeeze(-2) - t("uv", DEV)
    (gK2,) = torch.autograd.grad((r2 * w).sum(), [K2])
    np.testing.assert_allclose(gK.cpu().numpy(), gK2.cpu().numpy(), rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("case", ["plain", "huber"])
def test_bundle_adjustment_closed_form_on_the_device(case):
    kernel = (lambda: pp.optim.kernel.Huber(delta=1.0)) if case == "huber" else (lambda: None)
    opt, model, losses = run_ba(DEV, torch.float64, True, kernel())
    assert opt.linearization == "multigraph" and opt._last_blocks == "closed-form"
    np.testing.assert_allclose(losses, G[f"ba/{case}/loss"], rtol=1e-8)
    np.testing.assert_allclose(model.poses.detach().tensor().cpu().numpy(), G[f"ba/{case}/poses"], rtol=1e-6, atol=1e-8)
    opt2, _, losses2 = run_ba(DEV, torch.float64, False, kernel())
    assert opt2._last_blocks == "autograd"
    np.testing.assert_allclose(losses2, losses, rtol=1e-10)
    _, _, l32 = run_ba(DEV, torch.float32, True, kernel())
    np.testing.assert_allclose(l32, G[f"ba/{case}/loss"], rtol=1e-3)     # fp32 pixels ~3e2 against sub-pixel residuals
