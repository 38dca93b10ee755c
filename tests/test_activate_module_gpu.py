"""activate(pypose, module=True) on REAL kernels (missing 5): the reference package's IMUPreintegrator,
scans, Jinvp and Jr take the fused HIP kernels when their data lives on the GPU -- against the un-activated reference on
the CPU -- and BASELINE configs[4] runs through the activated reference in at most 3 launches of the library."""
import pytest
import torch

from oracle import ref_loader

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_loader.available(), reason="oracle/_ref not shipped")]
DEV = "cuda:0"


@pytest.fixture(scope="module")
def rpp():
    return ref_loader.load()


def _plain(t):
    return torch.Tensor.as_subclass(t, torch.Tensor) if isinstance(t, torch.Tensor) else t


def _imu_inputs(B, F, D, dev):
    g = torch.Generator().manual_seed(0)
    dt = torch.full((B, F, 1), 0.005, dtype=D)
    gyro = 0.1 * torch.randn(B, F, 3, dtype=D, generator=g)
    acc = torch.randn(B, F, 3, dtype=D, generator=g) + torch.tensor([0., 0., 9.81], dtype=D)
    return dt.to(dev), gyro.to(dev), acc.to(dev)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-8), (torch.float32, 3e-4)])
@pytest.mark.parametrize("reset", [True, False])
def test_activated_reference_imu_equals_the_reference_on_the_cpu(rpp, dtype, tol, reset):
    from pypose_amd import _C, activate
    B, F = 16, 200
    cpu = rpp.module.IMUPreintegrator(prop_cov=True, reset=reset).to(dtype)
    want = [cpu(*_imu_inputs(B, F, dtype, "cpu")) for _ in range(2)]             # two calls: the carried state matters when not reset
    activate.activate(rpp, module=True)
    try:
        gpu = rpp.module.IMUPreintegrator(prop_cov=True, reset=reset).to(dtype).to(DEV)
        launches = []
        real = _C.stream_ptr
        _C.stream_ptr = lambda d: (launches.append(1), real(d))[1]
        try:
            got = [gpu(*_imu_inputs(B, F, dtype, DEV)) for _ in range(2)]
        finally:
            _C.stream_ptr = real
        assert type(got[0]["rot"]).__module__.startswith(rpp.__name__) and got[0]["rot"].ltype is rpp.SO3_type
        host_again = rpp.module.IMUPreintegrator(prop_cov=True, reset=reset).to(dtype)(*_imu_inputs(B, F, dtype, "cpu"))
    finally:
        activate.deactivate()
    assert len(launches) <= 2 * 5, launches            # integrate + covariance + Rij product (+ r0^-1 and its product when the state moved) per forward
    for g, w in zip(got, want):
        for k in ("rot", "vel", "pos", "cov"):
            a, b = _plain(g[k]).double().cpu(), _plain(w[k]).double()
            if k == "rot":                               # q and -q are the same rotation
                b = torch.where((a * b).sum(-1, keepdim=True) < 0, -b, b)
            assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max())), (k, float((a - b).abs().max()))
    for k in ("rot", "vel", "pos", "cov"):
        assert torch.equal(_plain(host_again[k]), _plain(want[0][k]))              # host tensors keep the reference's own code


def test_configs4_through_the_activated_reference_is_three_launches(rpp):
    from pypose_amd import _C, activate
    dt, gyro, acc = _imu_inputs(4096, 1024, torch.float32, DEV)
    activate.activate(rpp, module=True)
    try:
        integ = rpp.module.IMUPreintegrator(prop_cov=True, reset=True).to(DEV)
        integ(dt=dt, gyro=gyro, acc=acc)                  # (first call: broadcast copies of the initial state, r0^-1)
        launches = []
        real = _C.stream_ptr
        _C.stream_ptr = lambda d: (launches.append(1), real(d))[1]
        try:
            out = integ(dt=dt, gyro=gyro, acc=acc)
        finally:
            _C.stream_ptr = real
    finally:
        activate.deactivate()
    assert len(launches) <= 3, launches
    assert out["rot"].shape == (4096, 1024, 4) and out["cov"].shape == (4096, 9, 9) and bool(torch.isfinite(out["cov"]).all())


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 2e-5)])
def test_activated_scans_jinvp_and_jr(rpp, dtype, tol):
    from pypose_amd import _C, activate
    torch.manual_seed(2)
    X = rpp.randn_SE3(6, 300, dtype=dtype, sigma=0.2)
    S = rpp.randn_SO3(257, dtype=dtype)
    a = rpp.randn_se3(257, dtype=dtype)
    Y = rpp.randn_SE3(257, dtype=dtype)
    w = rpp.randn_so3(64, dtype=dtype)
    b = rpp.randn_so3(257, dtype=dtype)
    want = [rpp.cumprod(X, dim=1, left=False).tensor(), rpp.cummul(X, dim=1).tensor(), X.cumprod(dim=1).tensor(),
            Y.Jinvp(a).tensor(), S.Jinvp(b).tensor(), w.Jr()]
    Xc = X.clone().requires_grad_(True)                    # the reference's own gradient through its Hillis-Steele rounds (CPU)
    Wc = torch.randn(X.shape, dtype=dtype)
    (rpp.cumprod(Xc, dim=1).tensor() * Wc).sum().backward()
    want_grad = Xc.grad.tensor() if hasattr(Xc.grad, "tensor") else Xc.grad
    activate.activate(rpp, module=True)
    try:
        Xg, Sg, ag, Yg, wg = X.to(DEV), S.to(DEV), a.to(DEV), Y.to(DEV), w.to(DEV)
        got = [rpp.cumprod(Xg, dim=1, left=False).tensor(), rpp.cummul(Xg, dim=1).tensor(), Xg.cumprod(dim=1).tensor(),
               Yg.Jinvp(ag).tensor(), Sg.Jinvp(b.to(DEV)).tensor(), wg.Jr()]
        Xr = Xg.clone().requires_grad_(True)               # gradients: one node, backward = pplie_scan_se3_bwd
        (rpp.cumprod(Xr, dim=1).tensor() * Wc.to(DEV)).sum().backward()
        got_grad = (Xr.grad.tensor() if hasattr(Xr.grad, "tensor") else Xr.grad).double().cpu()
        assert float((got_grad - want_grad.double()).abs().max()) <= tol * float(want_grad.abs().max())
    finally:
        activate.deactivate()
    for g, wv in zip(got, want):
        assert float((g.double().cpu() - wv.double()).abs().max()) <= tol * max(1.0, float(wv.abs().max()))


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 3e-5)])
def test_training_through_the_activated_reference_imu(rpp, dtype, tol):
    """gradients w.r.t. gyro / acc through the ACTIVATED reference module (fused forward + pplie_imu_integrate_bwd) against the
    reference's own autograd on the CPU in fp64"""
    from pypose_amd import activate
    dt, gyro, acc = _imu_inputs(6, 200, torch.float64, "cpu")
    Wp, Wr = torch.randn(6, 200, 3, dtype=torch.float64), torch.randn(6, 200, 4, dtype=torch.float64)

    def grads(dev, D):
        g, a = gyro.detach().clone().to(D).to(dev).requires_grad_(True), acc.detach().clone().to(D).to(dev).requires_grad_(True)
        integ = rpp.module.IMUPreintegrator(pos=torch.zeros(3, dtype=D), rot=rpp.identity_SO3(dtype=D),
                                            vel=torch.zeros(3, dtype=D), prop_cov=False, reset=True).to(D).to(dev)
        o = integ(dt=dt.to(D).to(dev), gyro=g, acc=a)
        ((o["pos"] * Wp.to(D).to(dev)).sum() + (o["rot"].tensor() * Wr.to(D).to(dev)).sum()).backward()
        return g.grad.double().cpu(), a.grad.double().cpu()
    want = grads("cpu", torch.float64)
    activate.activate(rpp, module=True)
    try:
        got = grads(DEV, dtype)
    finally:
        activate.deactivate()
    for x, y in zip(got, want):
        assert float((x - y).abs().max()) <= tol * float(y.abs().max())
