"""chspline / bspline / geodesic_loss: the oracle against goldens recorded from the real reference
(tests/golden/make_spline_golden.py), and the host-side composition (autograd route) through the oracle backend."""
import os

import numpy as np
import pytest
import torch

import pypose_amd as pp
from oracle import spline_np
from tests.oracle_backend import oracle_backend

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "spline_golden.npz"))
T = lambda k: torch.from_numpy(G[k])
KINDS = ("SO3", "SE3", "Sim3", "RxSO3", "so3", "se3", "sim3", "rxso3")


@pytest.fixture(autouse=True)
def _backend():
    with oracle_backend():
        yield


def test_oracle_bspline_matches_reference():
    np.testing.assert_allclose(spline_np.bspline(G["bs_doc_in"], 0.1), G["bs_doc_out"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(spline_np.bspline(G["bs_doc_in"], 0.1, True), G["bs_doc_extra"], rtol=1e-12, atol=1e-13)
    for name, iv in (("01", 0.1), ("03", 0.3), ("06", 0.6)):
        np.testing.assert_allclose(spline_np.bspline(G["bs_in"], iv), G["bs_out_" + name], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(spline_np.bspline(G["bs_in"], 0.25, True), G["bs_out_extra"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(spline_np.bspline(G["bs_min_in"], 0.2), G["bs_min_out"], rtol=1e-11, atol=1e-12)


def test_oracle_chspline_matches_reference():
    np.testing.assert_allclose(spline_np.chspline(G["ch_doc_in"], 0.1), G["ch_doc_out"], rtol=1e-12, atol=1e-13)
    for name, iv in (("02", 0.2), ("03", 0.3), ("07", 0.7)):
        np.testing.assert_allclose(spline_np.chspline(G["ch_in"], iv), G["ch_out_" + name], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(spline_np.chspline(G["ch_two_in"], 0.25), G["ch_two_out"], rtol=1e-12, atol=1e-13)


def test_bspline_composition_matches_reference():
    doc = pp.SE3(T("bs_doc_in"))
    torch.testing.assert_close(pp.bspline(doc, 0.1).tensor(), T("bs_doc_out"), rtol=1e-11, atol=1e-12)
    torch.testing.assert_close(pp.bspline(doc, 0.1, extrapolate=True).tensor(), T("bs_doc_extra"), rtol=1e-11, atol=1e-12)
    traj = pp.SE3(T("bs_in"))
    for name, iv in (("01", 0.1), ("03", 0.3), ("06", 0.6)):
        out = pp.bspline(traj, iv)
        assert pp.is_SE3(out) and out.shape == T("bs_out_" + name).shape
        torch.testing.assert_close(out.tensor(), T("bs_out_" + name), rtol=1e-10, atol=1e-12)
    torch.testing.assert_close(pp.bspline(traj, 0.25, extrapolate=True).tensor(), T("bs_out_extra"), rtol=1e-10, atol=1e-12)
    torch.testing.assert_close(pp.bspline(pp.SE3(T("bs_min_in")), 0.2).tensor(), T("bs_min_out"), rtol=1e-10, atol=1e-12)


def test_bspline_gradient_matches_reference():
    ctrl = pp.Parameter(pp.SE3(T("bs_in")[0].clone()))
    g = torch.autograd.grad((pp.bspline(ctrl, 0.25).tensor() * T("bs_coef")).sum(), ctrl)[0]
    torch.testing.assert_close(g, T("bs_grad"), rtol=1e-8, atol=1e-10)


def test_bspline_argument_checks():
    with pytest.raises(AssertionError, match="not SE3Type"):
        pp.bspline(pp.randn_SO3(2, 5))
    with pytest.raises(AssertionError, match="less than 4"):
        pp.bspline(pp.randn_SE3(2, 3))
    with pytest.raises(AssertionError, match="smaller than 1"):
        pp.bspline(pp.randn_SE3(2, 5), interval=1.0)
    assert pp.bspline(pp.randn_SE3(3), extrapolate=True).shape == (4 * 10 + 1, 7)       # 3 + 4 padded poses
    assert pp.bspline(pp.SE3(torch.zeros(0, 5, 7)), 0.25).shape == (0, 2 * 4 + 1, 7)    # empty batch
    assert pp.chspline(torch.zeros(0, 5, 3), 0.25).shape == (0, 17, 3)


def test_chspline_matches_reference():
    torch.testing.assert_close(pp.chspline(T("ch_doc_in"), 0.1), T("ch_doc_out"), rtol=1e-12, atol=1e-13)
    for name, iv in (("02", 0.2), ("03", 0.3), ("07", 0.7)):
        torch.testing.assert_close(pp.chspline(T("ch_in"), iv), T("ch_out_" + name), rtol=1e-12, atol=1e-13)
    torch.testing.assert_close(pp.chspline(T("ch_two_in"), 0.25), T("ch_two_out"), rtol=1e-12, atol=1e-13)
    pts = T("ch_in").clone().requires_grad_(True)
    out = pp.chspline(pts, 0.3)
    out.square().sum().backward()
    assert pts.grad is not None and torch.isfinite(pts.grad).all()
    with pytest.raises(AssertionError, match="smaller than 1"):
        pp.chspline(T("ch_in"), 1.5)


@pytest.mark.parametrize("kind", KINDS)
def test_geodesic_loss_matches_reference(kind):
    make = getattr(pp, kind)
    x, y = make(T(f"geo_{kind}_x")), make(T(f"geo_{kind}_y"))
    for red in ("none", "mean", "sum"):
        torch.testing.assert_close(pp.geodesic_loss(x, y, reduction=red), T(f"geo_{kind}_{red}"), rtol=1e-10, atol=1e-12)
    torch.testing.assert_close(pp.module.GeodesicLoss(reduction="sum")(x, y), T(f"geo_{kind}_sum"), rtol=1e-10, atol=1e-12)
    with pytest.raises(AssertionError):
        pp.geodesic_loss(x.tensor(), y)
    with pytest.raises(AssertionError):
        pp.geodesic_loss(x, y, reduction="max")
