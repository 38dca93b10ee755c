"""IMU pre-integration and scans on the MI355X: fused kernels vs the real reference's goldens
(fp64), vs the oracle at a larger size, and BASELINE configs[4] (4096 x 1024, fp32)."""
import os

import numpy as np
import pytest
import torch

import pypose_amd as pp
from oracle import imu_np, lie_np

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def G():
    return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "imu_golden.npz")))


def quat_close(a, b, tol):
    d = np.minimum(np.abs(a - b).max(-1), np.abs(a + b).max(-1))
    assert d.max() < tol, d.max()


def _module(dtype=torch.float64, **kw):
    return pp.module.IMUPreintegrator(pos=torch.zeros(3, dtype=dtype), rot=pp.identity_SO3(dtype=dtype),
                                      vel=torch.zeros(3, dtype=dtype), **kw).to(dtype).to(DEV)


def test_fused_route_matches_reference(G):
    T = lambda k: torch.from_numpy(G[k].copy()).to(DEV)
    dt, gyro, acc = T("dt"), T("gyro"), T("acc")
    m = _module(reset=True, prop_cov=True)
    assert m._fused_ok(dt, gyro, acc, None, {'pos': m.pos, 'rot': m.rot, 'vel': m.vel})
    o = m(dt, gyro, acc)
    quat_close(o["rot"].cpu().numpy(), G["case1/rot"], 1e-11)
    np.testing.assert_allclose(o["vel"].cpu().numpy(), G["case1/vel"], atol=1e-10)
    np.testing.assert_allclose(o["pos"].cpu().numpy(), G["case1/pos"], atol=1e-10)
    np.testing.assert_allclose(o["cov"].cpu().numpy(), G["case1/cov"], rtol=1e-8, atol=1e-18)
    o = m(dt, gyro, acc, init_state={"pos": T("p0"), "rot": pp.SO3(T("r0")), "vel": T("v0")})
    quat_close(o["rot"].cpu().numpy(), G["case2/rot"], 1e-11)
    np.testing.assert_allclose(o["pos"].cpu().numpy(), G["case2/pos"], atol=1e-10)
    np.testing.assert_allclose(o["cov"].cpu().numpy(), G["case2/cov"], rtol=1e-8, atol=1e-18)
    o = m(dt, gyro, acc, rot=pp.SO3(T("rot_known")))
    np.testing.assert_allclose(o["vel"].cpu().numpy(), G["case3/vel"], atol=1e-10)
    np.testing.assert_allclose(o["cov"].cpu().numpy(), G["case3/cov"], rtol=1e-8, atol=1e-18)
    o = m(dt, gyro, acc, gyro_cov=T("gc"), acc_cov=T("ac"))
    np.testing.assert_allclose(o["cov"].cpu().numpy(), G["case6/cov"], rtol=1e-8, atol=1e-18)
    m2 = _module(reset=False, prop_cov=True)
    o1 = m2(dt[:, :70], gyro[:, :70], acc[:, :70])
    o2 = m2(dt[:, 70:], gyro[:, 70:], acc[:, 70:])
    np.testing.assert_allclose(o1["cov"].cpu().numpy(), G["case4a/cov"], rtol=1e-8, atol=1e-18)
    np.testing.assert_allclose(o2["pos"].cpu().numpy(), G["case4b/pos"], atol=1e-10)
    np.testing.assert_allclose(o2["cov"].cpu().numpy(), G["case4b/cov"], rtol=1e-8, atol=1e-18)
    o = _module(reset=True, prop_cov=False)(dt, gyro, acc)
    assert o["cov"] is None
    np.testing.assert_allclose(o["pos"].cpu().numpy(), G["case5/pos"], atol=1e-10)


def test_composed_route_on_gpu_and_gradients(G):
    T = lambda k: torch.from_numpy(G[k].copy()).to(DEV)
    gyro = T("gyro").requires_grad_(True)
    m = _module(reset=True, prop_cov=True)
    o = m(T("dt"), gyro, T("acc"))                      # requires_grad -> composed route through the HIP Lie ops
    np.testing.assert_allclose(o["pos"].detach().cpu().numpy(), G["case1/pos"], atol=1e-9)
    np.testing.assert_allclose(o["cov"].cpu().numpy(), G["case1/cov"], rtol=1e-7, atol=1e-18)
    o["pos"].sum().backward()
    assert torch.isfinite(gyro.grad).all() and gyro.grad.abs().sum() > 0


def test_scan_kernels_match_reference(G):
    X = pp.SE3(torch.from_numpy(G["scan/X"].copy()).to(DEV))
    np.testing.assert_allclose(pp.cumprod(X, dim=1, left=True).cpu().numpy(), G["scan/se3_left"], atol=1e-10)
    np.testing.assert_allclose(pp.cumprod(X, dim=1, left=False).cpu().numpy(), G["scan/se3_right"], atol=1e-10)
    Q = pp.SO3(torch.from_numpy(G["scan/Q"].copy()).to(DEV))
    np.testing.assert_allclose(pp.cumprod(Q, dim=0).cpu().numpy(), G["scan/so3_dim0"], atol=1e-10)       # inner = 3
    Y = X.clone()
    assert pp.cumprod_(Y, dim=1) is Y
    np.testing.assert_allclose(Y.cpu().numpy(), G["scan/se3_left"], atol=1e-10)
    # fp32, long ragged sequences, every group, against a float64 sequential product
    for name, W, rnd in (("so3", 4, pp.randn_SO3), ("se3", 7, pp.randn_SE3), ("sim3", 8, pp.randn_Sim3), ("rxso3", 5, pp.randn_RxSO3)):
        torch.manual_seed(3)
        Z = rnd(5, 1025, sigma=0.2, device=DEV)
        out = pp.cumprod(Z, dim=1, left=False).cpu().numpy().astype(np.float64)
        mul = lambda a, b: lie_np.OPS[f"{name}_mul_fwd"](a, b)[0]
        ref = imu_np.cumprod(Z.cpu().numpy().astype(np.float64), mul, left=False)
        assert np.abs(out - ref).max() < 2e-4 * max(1.0, np.abs(ref).max()), name


def test_c5_full_size_fp32():
    """BASELINE configs[4]: 4096 sequences x 1024 steps, fp32, covariance on."""
    B, F = 4096, 1024
    g = torch.Generator(device=DEV).manual_seed(0)
    dt = torch.full((B, F, 1), 0.005, device=DEV)
    gyro = 0.1 * torch.randn(B, F, 3, device=DEV, generator=g)
    acc = torch.randn(B, F, 3, device=DEV, generator=g) + torch.tensor([0, 0, 9.81], device=DEV)
    m = _module(torch.float32, reset=True, prop_cov=True)
    o = m(dt, gyro, acc)
    assert o["rot"].shape == (B, F, 4) and o["cov"].shape == (B, 9, 9)
    assert torch.isfinite(o["pos"]).all() and torch.isfinite(o["cov"]).all()
    # parity on a 64-sequence slice against the float64 oracle
    sl = slice(100, 164)
    ref = imu_np.preintegrate(dt[sl].cpu().numpy().astype(np.float64), gyro[sl].cpu().numpy().astype(np.float64),
                              acc[sl].cpu().numpy().astype(np.float64))
    quat_close(o["rot"][sl].cpu().numpy(), ref["rot"], 2e-5)
    assert np.abs(o["vel"][sl].cpu().numpy() - ref["vel"]).max() < 1e-5 * max(1, np.abs(ref["vel"]).max()) + 1e-6 * F
    assert np.abs(o["pos"][sl].cpu().numpy() - ref["pos"]).max() < 1e-5 * max(1, np.abs(ref["pos"]).max()) + 1e-6 * F
    assert np.abs(o["cov"][sl].cpu().numpy() - ref["cov"]).max() < 1e-4 * np.abs(ref["cov"]).max()
    # covariance is symmetric positive semi-definite
    C = o["cov"].double()
    assert (C - C.mT).abs().max().item() < 1e-6 * C.abs().max().item()
    assert torch.linalg.eigvalsh((C + C.mT) / 2).min().item() > -1e-6 * C.abs().max().item()


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 2e-5)])
@pytest.mark.parametrize("F", [256, 257, 389, 1024])
@pytest.mark.parametrize("known", [False, True])
def test_two_steps_per_lane_kernel_matches_oracle(dtype, tol, F, known):
    """states only (no covariance) and F >= 256 take imu_integrate_multi_kernel (two consecutive steps per lane, one
    wave scan per 128 steps); ragged lengths exercise the padded tail; one-step-per-lane kernel on the same inputs
    (through the covariance route) gives the same states."""
    B = 37
    g = torch.Generator(device=DEV).manual_seed(F)
    dt = (0.004 + 0.002 * torch.rand(B, F, 1, device=DEV, generator=g, dtype=torch.float64)).to(dtype)
    gyro = (0.3 * torch.randn(B, F, 3, device=DEV, generator=g, dtype=torch.float64)).to(dtype)
    acc = (torch.randn(B, F, 3, device=DEV, generator=g, dtype=torch.float64) + torch.tensor([0, 0, 9.81], device=DEV)).to(dtype)
    ref = imu_np.preintegrate(dt.cpu().numpy().astype(np.float64), gyro.cpu().numpy().astype(np.float64),
                              acc.cpu().numpy().astype(np.float64))
    rot = pp.SO3(torch.from_numpy(ref["rot"]).to(dtype).to(DEV)) if known else None
    if known:       # the oracle with the orientations given (they equal its own, so the states are the same)
        ref = imu_np.preintegrate(dt.cpu().numpy().astype(np.float64), gyro.cpu().numpy().astype(np.float64),
                                  acc.cpu().numpy().astype(np.float64), rot=ref["rot"])
    o = _module(dtype, reset=True, prop_cov=False)(dt, gyro, acc, rot=rot)
    quat_close(o["rot"].cpu().numpy(), ref["rot"], tol)
    for key in ("vel", "pos"):
        assert np.abs(o[key].cpu().numpy() - ref[key]).max() < tol * max(1, np.abs(ref[key]).max()) + 0.1 * tol * F, key
    o1 = _module(dtype, reset=True, prop_cov=True)(dt, gyro, acc, rot=rot)
    quat_close(o1["rot"].cpu().numpy(), o["rot"].cpu().numpy(), tol)
    for key in ("vel", "pos"):
        assert (o1[key] - o[key]).abs().max().item() < tol * max(1, o[key].abs().max().item()) + 0.1 * tol * F, key


@pytest.mark.parametrize("left", [True, False])
@pytest.mark.parametrize("L", [256, 300, 1025])
def test_two_elements_per_lane_scan(left, L):
    """L >= 256 takes scan_kernel<K = 2>: fp64 against a sequential float64 product, both orders, ragged lengths, a
    scan dimension that is not the last batch dimension (inner = 3), and the first element left untouched."""
    for name, rnd in (("so3", pp.randn_SO3), ("se3", pp.randn_SE3), ("sim3", pp.randn_Sim3), ("rxso3", pp.randn_RxSO3)):
        torch.manual_seed(L)
        Z = rnd(2, L, 3, sigma=0.3, device=DEV, dtype=torch.float64)
        out = pp.cumprod(Z, dim=1, left=left)
        mul = lambda a, b: lie_np.OPS[f"{name}_mul_fwd"](a, b)[0]
        zn = Z.cpu().numpy()
        ref = np.stack([imu_np.cumprod(zn[:, :, i], mul, left=left) for i in range(3)], axis=2)
        assert np.abs(out.cpu().numpy() - ref).max() < 1e-11 * max(1.0, np.abs(ref).max()), name
        assert torch.equal(out.tensor()[:, 0], Z.tensor()[:, 0])
