"""IMU pre-integration + scans on the CPU: (a) the numpy oracle against the real reference's
golden outputs, (b) pypose_amd's composed (differentiable) route with the oracle stand-in backend."""
import os

import numpy as np
import pytest
import torch

import pypose_amd as pp
from oracle import imu_np, lie_np
from tests.oracle_backend import oracle_backend


@pytest.fixture(scope="module")
def G():
    return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "imu_golden.npz")))


def quat_close(a, b, tol):
    # q and -q are the same rotation
    d = np.minimum(np.abs(a - b).max(-1), np.abs(a + b).max(-1))
    assert d.max() < tol, d.max()


def test_oracle_imu_matches_reference(G):
    o = imu_np.preintegrate(G["dt"], G["gyro"], G["acc"])
    quat_close(o["rot"], G["case1/rot"], 1e-12)
    np.testing.assert_allclose(o["vel"], G["case1/vel"], atol=1e-11)
    np.testing.assert_allclose(o["pos"], G["case1/pos"], atol=1e-11)
    np.testing.assert_allclose(o["cov"], G["case1/cov"], rtol=1e-9, atol=1e-18)
    o = imu_np.preintegrate(G["dt"], G["gyro"], G["acc"], r0=G["r0"], v0=G["v0"], p0=G["p0"])
    quat_close(o["rot"], G["case2/rot"], 1e-12)
    np.testing.assert_allclose(o["pos"], G["case2/pos"], atol=1e-11)
    np.testing.assert_allclose(o["cov"], G["case2/cov"], rtol=1e-9, atol=1e-18)
    o = imu_np.preintegrate(G["dt"], G["gyro"], G["acc"], rot=G["rot_known"])
    np.testing.assert_allclose(o["vel"], G["case3/vel"], atol=1e-11)
    np.testing.assert_allclose(o["cov"], G["case3/cov"], rtol=1e-9, atol=1e-18)
    o = imu_np.preintegrate(G["dt"], G["gyro"], G["acc"], gyro_cov=G["gc"], acc_cov=G["ac"])
    np.testing.assert_allclose(o["cov"], G["case6/cov"], rtol=1e-9, atol=1e-18)


def test_oracle_scan_matches_reference(G):
    se3mul = lambda a, b: lie_np.se3_mul_fwd(a, b)[0]
    np.testing.assert_allclose(imu_np.cumprod(G["scan/X"], se3mul, left=True), G["scan/se3_left"], atol=1e-10)
    np.testing.assert_allclose(imu_np.cumprod(G["scan/X"], se3mul, left=False), G["scan/se3_right"], atol=1e-10)


def _module(**kw):
    D = torch.float64
    return pp.module.IMUPreintegrator(pos=torch.zeros(3, dtype=D), rot=pp.identity_SO3(dtype=D), vel=torch.zeros(3, dtype=D), **kw).to(D)


def test_composed_route_matches_reference(G):
    T = lambda k: torch.from_numpy(G[k].copy())
    dt, gyro, acc = T("dt"), T("gyro"), T("acc")
    with oracle_backend():
        m = _module(reset=True, prop_cov=True)
        o = m(dt, gyro, acc)
        quat_close(o["rot"].numpy(), G["case1/rot"], 1e-10)
        np.testing.assert_allclose(o["pos"].numpy(), G["case1/pos"], atol=1e-10)
        np.testing.assert_allclose(o["cov"].numpy(), G["case1/cov"], rtol=1e-8, atol=1e-18)
        o = m(dt, gyro, acc, init_state={"pos": T("p0"), "rot": pp.SO3(T("r0")), "vel": T("v0")})
        np.testing.assert_allclose(o["vel"].numpy(), G["case2/vel"], atol=1e-10)
        np.testing.assert_allclose(o["cov"].numpy(), G["case2/cov"], rtol=1e-8, atol=1e-18)
        o = m(dt, gyro, acc, rot=pp.SO3(T("rot_known")))
        np.testing.assert_allclose(o["pos"].numpy(), G["case3/pos"], atol=1e-10)
        m2 = _module(reset=False, prop_cov=True)
        o1 = m2(dt[:, :70], gyro[:, :70], acc[:, :70])
        o2 = m2(dt[:, 70:], gyro[:, 70:], acc[:, 70:])
        np.testing.assert_allclose(o1["cov"].numpy(), G["case4a/cov"], rtol=1e-8, atol=1e-18)
        np.testing.assert_allclose(o2["pos"].numpy(), G["case4b/pos"], atol=1e-10)
        np.testing.assert_allclose(o2["cov"].numpy(), G["case4b/cov"], rtol=1e-8, atol=1e-18)
        o = _module(reset=True, prop_cov=False)(dt, gyro, acc)
        assert o["cov"] is None
        np.testing.assert_allclose(o["pos"].numpy(), G["case5/pos"], atol=1e-10)
        with pytest.raises(RuntimeError):
            _module(reset=False, prop_cov=False)


def test_composed_route_is_differentiable(G):
    T = lambda k: torch.from_numpy(G[k].copy())
    with oracle_backend():
        gyro = T("gyro")[:, :20].clone().requires_grad_(True)
        m = _module(reset=True, prop_cov=False)
        o = m(T("dt")[:, :20], gyro, T("acc")[:, :20])
        o["pos"].sum().backward()
        assert gyro.grad is not None and torch.isfinite(gyro.grad).all() and gyro.grad.abs().sum() > 0


def test_cumprod_generic_path(G):
    with oracle_backend():
        X = pp.SE3(torch.from_numpy(G["scan/X"].copy()))
        np.testing.assert_allclose(pp.cumprod(X, dim=1, left=True).numpy(), G["scan/se3_left"], atol=1e-9)
        np.testing.assert_allclose(pp.cumprod(X, dim=1, left=False).numpy(), G["scan/se3_right"], atol=1e-9)
        Q = pp.SO3(torch.from_numpy(G["scan/Q"].copy()))
        np.testing.assert_allclose(pp.cumprod(Q, dim=0).numpy(), G["scan/so3_dim0"], atol=1e-9)
        assert pp.cumprod(X, dim=1).ltype is pp.SE3_type
    x = torch.arange(1.0, 6.0)
    torch.testing.assert_close(pp.cumops(x, 0, lambda a, b: a + b), torch.cumsum(x, 0))
    torch.testing.assert_close(pp.pm(torch.tensor([-2.0, 0.0, 3.0])), torch.tensor([-1.0, 1.0, 1.0]))
