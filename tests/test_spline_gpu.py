"""Spline kernels (csrc/spline.hip) on the MI355X through the C ABI: goldens recorded from the real reference, the
numpy oracle on seeded inputs, the differentiable composition, and a closed form at full size."""
import os

import numpy as np
import pytest
import torch

import pypose_amd as pp
from oracle import spline_np

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "spline_golden.npz"))


def T(k, dtype=torch.float64):
    return torch.from_numpy(G[k]).to(dtype).to(DEV)


def pose_close(a, b, tol):
    """rows are SE3 elements: compare as transforms (q and -q are the same rotation)"""
    a, b = a.double().cpu(), b.double().cpu()
    sign = torch.sign((a[..., 3:] * b[..., 3:]).sum(-1, keepdim=True))
    err = torch.maximum((a[..., :3] - b[..., :3]).abs().max(), (a[..., 3:] - sign * b[..., 3:]).abs().max()).item()
    assert err <= tol, err


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 2e-5)])
def test_bspline_kernel_matches_reference_goldens(dtype, tol):
    from pypose_amd import _C
    assert _C._test_backend is None
    doc = pp.SE3(T("bs_doc_in", dtype))
    pose_close(pp.bspline(doc, 0.1).tensor(), T("bs_doc_out"), tol)
    pose_close(pp.bspline(doc, 0.1, extrapolate=True).tensor(), T("bs_doc_extra"), tol)
    traj = pp.SE3(T("bs_in", dtype))
    for name, iv in (("01", 0.1), ("03", 0.3), ("06", 0.6)):
        out = pp.bspline(traj, iv)
        assert pp.is_SE3(out) and out.shape == G["bs_out_" + name].shape and out.dtype == dtype
        pose_close(out.tensor(), T("bs_out_" + name), tol)
    pose_close(pp.bspline(traj, 0.25, extrapolate=True).tensor(), T("bs_out_extra"), tol)
    pose_close(pp.bspline(pp.SE3(T("bs_min_in", dtype)), 0.2).tensor(), T("bs_min_out"), tol)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 2e-5)])
@pytest.mark.parametrize("shape,interval", [((1, 4), 0.5), ((7, 5), 0.1), ((3, 2, 131), 0.07), ((600, 9), 0.3), ((2, 1030), 0.45)])
def test_bspline_kernel_matches_oracle(dtype, tol, shape, interval):
    torch.manual_seed(sum(shape))
    data = pp.randn_SE3(*shape, sigma=0.8, dtype=dtype, device=DEV)
    out = pp.bspline(data, interval)
    ref = spline_np.bspline(data.tensor().cpu().numpy(), interval)
    assert out.shape == ref.shape
    pose_close(out.tensor(), torch.from_numpy(ref), tol)
    # the differentiable composition (batched Lie kernels) agrees with the single kernel
    comp = pp.bspline(pp.Parameter(data), interval)
    assert comp.requires_grad
    pose_close(comp.detach().tensor(), out.tensor(), tol)


def test_bspline_gradient_matches_reference():
    ctrl = pp.Parameter(pp.SE3(T("bs_in")[0].clone()))
    g = torch.autograd.grad((pp.bspline(ctrl, 0.25).tensor() * T("bs_coef")).sum(), ctrl)[0]
    torch.testing.assert_close(g, T("bs_grad"), rtol=1e-8, atol=1e-10)


def test_bspline_closed_form_at_full_size():
    """Control poses on a one-parameter subgroup, P_i = Exp(i xi): every relative twist is xi, the cumulative
    weights sum to 1 + u, so sample k of segment i is Exp((i + 1 + u_k) xi) -- checked on 2*10^6 output poses."""
    torch.manual_seed(0)
    nb, N, interval = 2000, 104, 0.1
    xi = 0.02 * torch.randn(nb, 1, 6, dtype=torch.float64, device=DEV)
    steps = torch.arange(N, dtype=torch.float64, device=DEV).view(1, N, 1)
    ctrl = pp.se3(xi * steps).Exp()
    for dtype, tol in ((torch.float64, 1e-10), (torch.float32, 5e-5)):
        out = pp.bspline(ctrl.to(dtype), interval)
        K = 10
        assert out.shape == (nb, (N - 3) * K + 1, 7)
        u = torch.arange(0, 1, interval, dtype=torch.float64, device=DEV)
        when = (torch.arange(N - 3, dtype=torch.float64, device=DEV).view(-1, 1) + 1 + u).reshape(-1)
        when = torch.cat([when, when.new_tensor([N - 3 + 1.0])])              # closing pose: last segment at u = 1
        expect = pp.se3(xi * when.view(1, -1, 1)).Exp()
        pose_close(out.tensor(), expect.tensor(), tol)


def test_bspline_stream_and_noncontiguous_input():
    data = pp.randn_SE3(5, 12, dtype=torch.float64, device=DEV)
    ref = pp.bspline(data, 0.2)
    wide = torch.zeros(5, 12, 9, dtype=torch.float64, device=DEV)
    wide[..., 1:8] = data.tensor()
    pose_close(pp.bspline(pp.SE3(wide[..., 1:8]), 0.2).tensor(), ref.tensor(), 0)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        out = pp.bspline(data, 0.2)
    s.synchronize()
    pose_close(out.tensor(), ref.tensor(), 0)
    assert pp.bspline(pp.SE3(torch.zeros(0, 5, 7, device=DEV)), 0.25).shape == (0, 9, 7)           # empty batch: no launch
    assert pp.chspline(torch.zeros(0, 5, 3, device=DEV), 0.25).shape == (0, 17, 3)
    host = pp.bspline(data.cpu(), 0.2)                     # host tensors are staged through the GPU kernels
    assert host.device.type == "cpu"
    pose_close(host.tensor(), ref.tensor(), 1e-12)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 2e-6)])
def test_chspline_kernel(dtype, tol):
    torch.testing.assert_close(pp.chspline(T("ch_doc_in", dtype), 0.1).double(), T("ch_doc_out"), rtol=tol, atol=tol)
    for name, iv in (("02", 0.2), ("03", 0.3), ("07", 0.7)):
        out = pp.chspline(T("ch_in", dtype), iv)
        assert out.dtype == dtype
        torch.testing.assert_close(out.double(), T("ch_out_" + name), rtol=tol, atol=10 * tol)
    torch.testing.assert_close(pp.chspline(T("ch_two_in", dtype), 0.25).double(), T("ch_two_out"), rtol=tol, atol=10 * tol)
    torch.manual_seed(3)
    pts = torch.randn(33, 257, 3, dtype=dtype, device=DEV)
    out = pp.chspline(pts, 0.13)
    ref = spline_np.chspline(pts.cpu().numpy(), 0.13)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=20 * tol)
    # knots are interpolated exactly; the differentiable route agrees with the kernel
    K = len(np.arange(0, 1, 0.13))
    torch.testing.assert_close(out[:, ::K], pts, rtol=0, atol=10 * tol)
    diff = pp.chspline(pts.clone().requires_grad_(True), 0.13)
    assert diff.requires_grad
    torch.testing.assert_close(diff.detach(), out, rtol=tol, atol=10 * tol)


@pytest.mark.parametrize("kind", ("SO3", "SE3", "Sim3", "RxSO3", "so3", "se3", "sim3", "rxso3"))
def test_geodesic_loss_on_device(kind):
    make = getattr(pp, kind)
    for dtype, tol in ((torch.float64, 1e-10), (torch.float32, 2e-5)):
        x, y = make(T(f"geo_{kind}_x", dtype)), make(T(f"geo_{kind}_y", dtype))
        for red in ("none", "mean", "sum"):
            torch.testing.assert_close(pp.geodesic_loss(x, y, reduction=red).double(), T(f"geo_{kind}_{red}"), rtol=tol, atol=tol)
    x = pp.Parameter(make(T(f"geo_{kind}_x")))
    loss = pp.module.GeodesicLoss()(x, make(T(f"geo_{kind}_y")))
    loss.backward()
    assert torch.isfinite(x.grad).all()
