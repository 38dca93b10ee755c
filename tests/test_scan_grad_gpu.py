"""Row a19 / a28 with gradients: the one-pass backward kernels (pplie_scan_<g>_bwd, pplie_imu_integrate_bwd) against gradients
recorded from the REAL reference (tests/golden/grad_golden.npz), against the sequential oracle at larger sizes, and against the
composed route (the reference's Hillis-Steele formulation on the HIP Mul kernels) at BASELINE configs[4]'s shape."""
import os

import numpy as np
import pytest
import torch

import pypose_amd as pp
from oracle import grad_ref
from pypose_amd.basics import scan as scan_mod

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "grad_golden.npz")))
CTOR = {"SO3": pp.SO3, "SE3": pp.SE3, "Sim3": pp.Sim3, "RxSO3": pp.RxSO3}
KEY = {"SO3": "so3", "SE3": "se3", "Sim3": "sim3", "RxSO3": "rxso3"}
DIMS = {"a": 1, "b": 0, "c": 1, "d": 1, "e": 1, "f": 0, "long": 1}


def cases():
    for g in CTOR:
        for tag in ("a", "b", "c", "d", "e", "f") + (("long",) if g == "SE3" else ()):
            for side in "LR":
                yield g, tag, side


def _scan_grad(gname, Xn, Wn, dim, left, dtype, inplace=False):
    X = CTOR[gname](torch.from_numpy(Xn.copy()).to(dtype).to(DEV)).requires_grad_(True)
    W = torch.from_numpy(Wn.copy()).to(dtype).to(DEV)
    if inplace:
        Y = X.clone()
        Y.cumprod_(dim=dim, left=left)
    else:
        Y = pp.cumprod(X, dim=dim, left=left)
    (Y.tensor() * W).sum().backward()
    return Y.detach().tensor().cpu().numpy(), X.grad.tensor().cpu().numpy() if hasattr(X.grad, "tensor") else X.grad.cpu().numpy()


@pytest.mark.parametrize("gname,tag,side", list(cases()))
def test_scan_backward_kernel_matches_reference_gradients(gname, tag, side, monkeypatch):
    launches = []
    real = scan_mod._launch_bwd
    monkeypatch.setattr(scan_mod, "_launch_bwd", lambda *a: (launches.append(a[2]), real(*a))[1])
    Xn, Wn, ref = G[f"scan/{gname}/{tag}/X"], G[f"scan/{gname}/{tag}/W"], G[f"scan/{gname}/{tag}/{side}/gX"]
    scale = np.abs(ref).max() + 1e-300
    monkeypatch.setattr(scan_mod, "NATIVE_NODE", False)               # the Python autograd.Function around the kernel pair
    Y, gX = _scan_grad(gname, Xn, Wn, DIMS[tag], side == "L", torch.float64)
    assert launches == [KEY[gname]], "the backward did not run on pplie_scan_*_bwd"
    monkeypatch.setattr(scan_mod, "NATIVE_NODE", True)                # the same pair as a C++ node: same kernels, same bits
    Yn, gXn = _scan_grad(gname, Xn, Wn, DIMS[tag], side == "L", torch.float64)
    assert launches == [KEY[gname]] and np.array_equal(Yn, Y) and np.array_equal(gXn, gX)
    assert np.abs(Y - G[f"scan/{gname}/{tag}/{side}/Y"]).max() < 1e-9 * max(1.0, np.abs(Y).max())
    assert np.abs(gX - ref).max() <= 1e-9 * scale, (np.abs(gX - ref).max(), scale)
    _, gX32 = _scan_grad(gname, Xn, Wn, DIMS[tag], side == "L", torch.float32)
    # fp32: 2e-5 of the gradient's scale -- or, where the product itself is ill-conditioned in fp32 (compounding Sim3 / RxSO3
    # scales, the 1500-pose drifting trajectory), no worse than twice what the reference's own formulation (the Hillis-Steele
    # rounds on the fp32 Mul kernels) loses on the same input
    e32 = np.abs(gX32 - ref).max() / scale
    if e32 > 2e-5:
        scan_mod.DIFFERENTIABLE_SCAN = False
        try:
            _, gC32 = _scan_grad(gname, Xn, Wn, DIMS[tag], side == "L", torch.float32)
        finally:
            scan_mod.DIFFERENTIABLE_SCAN = True
        ec = np.abs(gC32 - ref).max() / scale
        assert e32 <= 2 * ec, (e32, ec)
    _, gXi = _scan_grad(gname, Xn, Wn, DIMS[tag], side == "L", torch.float64, inplace=True)
    assert np.abs(gXi - ref).max() <= 1e-9 * scale


@pytest.mark.parametrize("gname", list(CTOR))
@pytest.mark.parametrize("left", [True, False])
def test_scan_backward_kernel_vs_oracle_and_composed_route(gname, left):
    torch.manual_seed(3)
    B, L = 5, 1000                                     # 8 chunks of 128, the last one partial
    rnd = {"SO3": pp.randn_SO3, "SE3": pp.randn_SE3, "Sim3": pp.randn_Sim3, "RxSO3": pp.randn_RxSO3}[gname]
    X0 = rnd(B, L, sigma=0.3 if gname in ("Sim3", "RxSO3") else 0.8, dtype=torch.float64, device=DEV)
    if gname in ("Sim3", "RxSO3"):                     # keep the running scale bounded over 1000 factors
        t = X0.tensor().clone(); t[..., -1] = 1 + 0.01 * torch.randn(B, L, dtype=torch.float64, device=DEV); X0 = CTOR[gname](t)
    W = torch.randn(B, L, X0.shape[-1], dtype=torch.float64, device=DEV)

    def grad(x0, w):
        X = x0.clone().requires_grad_(True)
        Y = pp.cumprod(X, dim=1, left=left)
        (Y.tensor() * w).sum().backward()
        return Y.detach().tensor(), X.grad.tensor() if hasattr(X.grad, "tensor") else X.grad
    Y, gk = grad(X0, W)
    want = grad_ref.scan_bwd(KEY[gname], X0.tensor().cpu().numpy(), Y.cpu().numpy(), W.cpu().numpy(), left)
    scale = np.abs(want).max()
    assert np.abs(gk.cpu().numpy() - want).max() <= 1e-9 * scale
    scan_mod.DIFFERENTIABLE_SCAN = False
    try:
        _, gc = grad(X0, W)
    finally:
        scan_mod.DIFFERENTIABLE_SCAN = True
    assert np.abs(gc.cpu().numpy() - want).max() <= 1e-9 * scale
    # fp32 at the same size: kernel vs fp64 truth no worse than 4x the composed route's own fp32 error (+ 1e-5 of scale)
    _, gk32 = grad(CTOR[gname](X0.tensor().float()), W.float())
    scan_mod.DIFFERENTIABLE_SCAN = False
    try:
        _, gc32 = grad(CTOR[gname](X0.tensor().float()), W.float())
    finally:
        scan_mod.DIFFERENTIABLE_SCAN = True
    ek = np.abs(gk32.double().cpu().numpy() - want).max() / scale
    ec = np.abs(gc32.double().cpu().numpy() - want).max() / scale
    assert ek <= 4 * ec + 1e-5, (ek, ec)


def test_scan_autograd_contract():
    X = pp.randn_SE3(4, 50, dtype=torch.float64, device=DEV, requires_grad=True)
    with pytest.raises(RuntimeError):                  # in-place on a leaf that requires grad: as in the reference
        X.cumprod_(dim=1)
    Y = pp.cumprod(X, dim=1, left=False)
    assert Y.requires_grad and isinstance(Y, pp.LieTensor) and Y.ltype == pp.SE3_type
    # double backward: the closed form from differentiable ops (create_graph=True)
    W = torch.randn(4, 50, 7, dtype=torch.float64, device=DEV)
    g1, = torch.autograd.grad((Y.tensor() * W).sum(), X, create_graph=True)
    g1 = g1.tensor() if hasattr(g1, "tensor") else g1
    assert g1.requires_grad
    g0, = torch.autograd.grad((Y.tensor() * W).sum(), X, retain_graph=True)    # the kernel's values
    torch.testing.assert_close(g1.detach(), g0.tensor() if hasattr(g0, "tensor") else g0, rtol=1e-9, atol=1e-9)
    for left in (True, False):                                                 # (and for left products)
        Yl = pp.cumprod(X, dim=1, left=left)
        a, = torch.autograd.grad((Yl.tensor() * W).sum(), X, create_graph=True)
        b, = torch.autograd.grad((Yl.tensor() * W).sum(), X)
        tt = lambda t: t.tensor() if hasattr(t, "tensor") else t
        torch.testing.assert_close(tt(a).detach(), tt(b), rtol=1e-9, atol=1e-9)
    g1.square().sum().backward()
    assert torch.isfinite(X.grad.tensor() if hasattr(X.grad, "tensor") else X.grad).all()
    # a non-contiguous cotangent and a scan along dim 0 with trailing batch dims
    X2 = pp.randn_SO3(60, 3, 2, dtype=torch.float64, device=DEV, requires_grad=True)
    Y2 = pp.cumprod(X2, dim=0)
    Wn = torch.randn(4, 2, 3, 60, dtype=torch.float64, device=DEV).permute(3, 2, 1, 0)
    (Y2.tensor() * Wn).sum().backward()
    ref = grad_ref.scan_bwd("so3", X2.detach().tensor().permute(1, 2, 0, 3).reshape(6, 60, 4).cpu().numpy(),
                            Y2.detach().tensor().permute(1, 2, 0, 3).reshape(6, 60, 4).cpu().numpy(),
                            Wn.permute(1, 2, 0, 3).reshape(6, 60, 4).cpu().numpy(), True)
    got = (X2.grad.tensor() if hasattr(X2.grad, "tensor") else X2.grad).permute(1, 2, 0, 3).reshape(6, 60, 4).cpu().numpy()
    assert np.abs(got - ref).max() < 1e-10 * np.abs(ref).max()


# ---- IMU ------------------------------------------------------------------------------------------------------------
def _imu(dtype, **kw):
    return pp.module.IMUPreintegrator(pos=torch.zeros(3, dtype=dtype), rot=pp.identity_SO3(dtype=dtype),
                                      vel=torch.zeros(3, dtype=dtype), **kw).to(dtype).to(DEV)


def _imu_grads(tag, dtype, fused=True, native=True):
    T = lambda k: torch.from_numpy(G[k].copy()).to(dtype).to(DEV)
    dt, gyro, acc = (T(k).requires_grad_(True) for k in ("imu/dt", "imu/gyro", "imu/acc"))
    leaves, kw = {"dt": dt, "gyro": gyro, "acc": acc}, {}
    if tag != "plain":
        r0 = pp.SO3(T("imu/r0")).requires_grad_(True); p0 = T("imu/p0").requires_grad_(True); v0 = T("imu/v0").requires_grad_(True)
        kw["init_state"] = {"pos": p0, "rot": r0, "vel": v0}
        leaves.update(r0=r0, p0=p0, v0=v0)
    if tag == "known":
        kw["rot"] = pp.SO3(T("imu/rotk"))
    m = _imu(dtype, reset=True, prop_cov=(tag == "cov"))
    m.fused_backward, m.native_backward = fused, native
    o = m(dt, gyro, acc, **kw)
    loss = (o["rot"].tensor() * T("imu/Wr")).sum() + (o["vel"] * T("imu/Wv")).sum() + (o["pos"] * T("imu/Wp")).sum()
    loss.backward()
    out = {k: (v.grad.tensor() if hasattr(v.grad, "tensor") else v.grad).double().cpu().numpy() for k, v in leaves.items()}
    return float(loss.detach()), out, o


@pytest.mark.parametrize("tag", ["plain", "init", "known", "cov"])
def test_imu_backward_kernel_matches_reference_gradients(tag, monkeypatch):
    from pypose_amd.module import imu_preintegrator as im
    used = []
    real = im._ImuIntegrate.backward
    monkeypatch.setattr(im._ImuIntegrate, "backward", staticmethod(lambda ctx, *g: (used.append(1), real(ctx, *g))[1]))
    loss, got, o = _imu_grads(tag, torch.float64, native=False)
    assert used, "the gradient did not flow through pplie_imu_integrate_bwd"
    # the same node in C++ (csrc_torch/pplie_autograd.cpp ImuOp) whenever the initial state needs no gradient: same kernels, same bits
    del used[:]
    loss_n, got_n, o_n = _imu_grads(tag, torch.float64)
    if tag == "plain":                                   # (the other cases ask for the initial state's gradients: Python node)
        assert not used and "ImuOp" in o_n["pos"].grad_fn.name(), (used, o_n["pos"].grad_fn.name())
    assert loss_n == loss and all(np.array_equal(got_n[k], got[k]) for k in got)
    assert abs(loss - float(G[f"imu/{tag}/loss"])) <= 1e-9 * abs(float(G[f"imu/{tag}/loss"]))
    for k, v in got.items():
        ref = G[f"imu/{tag}/g_{k}"]
        assert np.abs(v - ref).max() <= 1e-9 * np.abs(ref).max(), (k, np.abs(v - ref).max(), np.abs(ref).max())
    if tag == "cov":
        assert o["cov"] is not None and not o["cov"].requires_grad
    _, got32, _ = _imu_grads(tag, torch.float32)
    for k, v in got32.items():
        ref = G[f"imu/{tag}/g_{k}"]
        assert np.abs(v - ref).max() <= 2e-5 * np.abs(ref).max(), (k, np.abs(v - ref).max(), np.abs(ref).max())
    _, gotc, _ = _imu_grads(tag, torch.float64, fused=False)          # the composed route, same goldens
    for k, v in gotc.items():
        ref = G[f"imu/{tag}/g_{k}"]
        assert np.abs(v - ref).max() <= 1e-9 * np.abs(ref).max(), (k, "composed")


def test_imu_example_loss_gradients():
    """the loss of examples/module/imu/imu_corrector.py:69-74 (mse on pos + 5e2 * geodesic rotation error)"""
    for dtype, tol in ((torch.float64, 1e-9), (torch.float32, 2e-5)):
        T = lambda k: torch.from_numpy(G[k].copy()).to(dtype).to(DEV)
        gyro, acc = T("imu/gyro").requires_grad_(True), T("imu/acc").requires_grad_(True)
        o = _imu(dtype, reset=True, prop_cov=False)(T("imu/dt"), gyro, acc,
                                                    init_state={"pos": T("imu/p0"), "rot": pp.SO3(T("imu/r0")), "vel": T("imu/v0")})
        loss = torch.nn.functional.mse_loss(o["pos"], T("imu/example/gt_pos")) + \
            5e2 * (pp.SO3(T("imu/rotk")) * o["rot"].Inv()).Log().norm(dim=-1).mean()
        loss.backward()
        assert abs(float(loss) - float(G["imu/example/loss"])) <= max(tol, 1e-6 if dtype == torch.float32 else 0) * abs(float(G["imu/example/loss"]))
        for k, v in (("gyro", gyro), ("acc", acc)):
            ref = G[f"imu/example/g_{k}"]
            assert np.abs(v.grad.double().cpu().numpy() - ref).max() <= tol * np.abs(ref).max(), (k, dtype)


def test_imu_backward_at_configs4_shape_follows_the_composed_route():
    """BASELINE configs[4] shape (sequences x 1024 steps, fp32): fused backward vs the composed graph on 256 sequences, both
    against the fp64 fused result"""
    torch.manual_seed(0)
    B, F = 256, 1024
    dt = torch.full((B, F, 1), 0.005, device=DEV)
    gyro = 0.1 * torch.randn(B, F, 3, device=DEV)
    acc = torch.randn(B, F, 3, device=DEV) + torch.tensor([0, 0, 9.81], device=DEV)
    Wp, Wv, Wr = torch.randn(B, F, 3, device=DEV), torch.randn(B, F, 3, device=DEV), torch.randn(B, F, 4, device=DEV)

    def run(dtype, fused):
        g, a = gyro.to(dtype).requires_grad_(True), acc.to(dtype).requires_grad_(True)
        m = _imu(dtype, reset=True, prop_cov=False)
        m.fused_backward = fused
        o = m(dt.to(dtype), g, a)
        ((o["rot"].tensor() * Wr.to(dtype)).sum() + (o["vel"] * Wv.to(dtype)).sum() + (o["pos"] * Wp.to(dtype)).sum()).backward()
        return g.grad.double(), a.grad.double()
    g64, a64 = run(torch.float64, True)
    gc64, ac64 = run(torch.float64, False)
    assert (g64 - gc64).abs().max() <= 1e-9 * g64.abs().max() and (a64 - ac64).abs().max() <= 1e-9 * a64.abs().max()
    gf, af = run(torch.float32, True)
    gc, ac = run(torch.float32, False)
    for x, xc, x64 in ((gf, gc, g64), (af, ac, a64)):
        ef, ec = float((x - x64).abs().max() / x64.abs().max()), float((xc - x64).abs().max() / x64.abs().max())
        assert ef <= 2e-5 or ef <= 2 * ec, (ef, ec)


def test_native_imu_node_equals_the_python_node_on_partial_losses():
    """known rotations, a loss on ONE output (the other cotangents arrive undefined), dt with and without a gradient, an expanded
    (stride-0) cotangent out of sum(): C++ node vs Python node, same kernels -> same bits"""
    torch.manual_seed(3)
    B, F = 7, 300
    for dtype in (torch.float32, torch.float64):
        dt0 = torch.full((B, F, 1), 0.01, dtype=dtype, device=DEV)
        gy0 = 0.2 * torch.randn(B, F, 3, dtype=dtype, device=DEV)
        ac0 = torch.randn(B, F, 3, dtype=dtype, device=DEV)
        rk = pp.randn_SO3(B, F, dtype=dtype, device=DEV)
        for known in (False, True):
            for which in ("pos", "vel", "rot"):
                for dt_grad in (False, True):
                    res = []
                    for native in (True, False):
                        dt, gy, ac = dt0.clone().requires_grad_(dt_grad), gy0.clone().requires_grad_(True), ac0.clone().requires_grad_(True)
                        m = _imu(dtype, reset=True, prop_cov=False)
                        m.native_backward = native
                        o = m(dt, gy, ac, rot=rk if known else None)
                        assert ("ImuOp" in o["pos"].grad_fn.name()) == native, o["pos"].grad_fn.name()
                        (o[which].tensor() if which == "rot" else o[which]).sum().backward()
                        res.append([t.grad.clone() for t in ((dt, gy, ac) if dt_grad else (gy, ac))])
                    for a, b in zip(*res):
                        assert torch.equal(a, b), (dtype, known, which, dt_grad)
    with pytest.raises(RuntimeError, match="not differentiable a second time"):
        gy = gy0.clone().requires_grad_(True)
        o = _imu(torch.float64, reset=True, prop_cov=False)(dt0.double(), gy.double(), ac0.double())
        torch.autograd.grad(o["pos"].sum(), gy, create_graph=True)
