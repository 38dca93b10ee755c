"""pypose_amd.lietensor.operation.{so3_Jl ... Sim3_Act4_Jacobian}: the reference's matrix-valued helpers
(pypose/lietensor/operation.py:7-301) as Python callables -- composed (differentiable torch) route on the host, kernel
route on the GPU -- against the reference's own functions."""
import pytest
import torch

from oracle import ref_loader
from pypose_amd.lietensor import matrices as M, operation as O

needs_ref = pytest.mark.skipif(not ref_loader.available(), reason="oracle/_ref not present (make -C oracle)")
D = torch.float64


def _inputs(rpp, dev="cpu"):
    torch.manual_seed(0)
    n = 33
    alg = {"so3": rpp.randn_so3(n, dtype=D).tensor(), "se3": rpp.randn_se3(n, dtype=D).tensor(),
           "rxso3": rpp.randn_rxso3(n, dtype=D).tensor(), "sim3": rpp.randn_sim3(n, dtype=D).tensor()}
    for k in alg:                                   # the small-angle / small-scale branches
        alg[k][0] = 0
        alg[k][1] *= 1e-9
        alg[k][2] *= 1e-3
    alg["rxso3"][3, 3] = 0
    alg["sim3"][3, 6] = 0
    alg["rxso3"][4, :3] = 0
    grp = {"SO3": rpp.randn_SO3(n, dtype=D).tensor(), "SE3": rpp.randn_SE3(n, dtype=D).tensor(),
           "RxSO3": rpp.randn_RxSO3(n, dtype=D).tensor(), "Sim3": rpp.randn_Sim3(n, dtype=D).tensor()}
    pts = {3: torch.randn(n, 3, dtype=D), 4: torch.randn(n, 4, dtype=D)}
    mv = lambda d: {k: v.to(dev) for k, v in d.items()}
    return mv(alg), mv(grp), mv(pts)


CASES = [("so3_Jl", "so3"), ("so3_Jl_inv", "so3"), ("so3_adj", "so3"), ("calcQ", "se3"), ("se3_Jl", "se3"), ("se3_Jl_inv", "se3"),
         ("se3_adj", "se3"), ("rxso3_Ws", "rxso3"), ("rxso3_Jl", "rxso3"), ("rxso3_Jl_inv", "rxso3"), ("rxso3_adj", "rxso3"),
         ("sim3_adj", "sim3"), ("sim3_Jl", "sim3"), ("sim3_Jl_inv", "sim3"),
         ("SO3_Adj", "SO3"), ("SO3_Matrix", "SO3"), ("SO3_Matrix4x4", "SO3"), ("SE3_Adj", "SE3"), ("SE3_Matrix", "SE3"),
         ("SE3_Matrix4x4", "SE3"), ("RxSO3_Adj", "RxSO3"), ("RxSO3_Matrix", "RxSO3"), ("RxSO3_Rotation", "RxSO3"),
         ("RxSO3_Matrix4x4", "RxSO3"), ("Sim3_Adj", "Sim3"), ("Sim3_Matrix", "Sim3"), ("Sim3_Matrix4x4", "Sim3"),
         ("SO3_Act_Jacobian", 3), ("SO3_Act4_Jacobian", 4), ("SE3_Act_Jacobian", 3), ("SE3_Act4_Jacobian", 4),
         ("RxSO3_Act_Jacobian", 3), ("RxSO3_Act4_Jacobian", 4), ("Sim3_Act_Jacobian", 3), ("Sim3_Act4_Jacobian", 4)]


def test_every_helper_of_the_reference_is_exported():
    assert {c[0] for c in CASES} == set(M.__all__)
    assert all(getattr(O, name) is getattr(M, name) for name in M.__all__)


@needs_ref
@pytest.mark.parametrize("name,arg", CASES)
def test_composed_route_equals_the_reference(name, arg):
    rpp = ref_loader.load()
    alg, grp, pts = _inputs(rpp)
    x = {**alg, **grp, **pts}[arg]
    want = getattr(rpp.lietensor.operation, name)(x)
    got = getattr(M, name)(x)
    assert got.shape == want.shape
    # (rows 0-2 sit on the reference's theta <= eps switch / inside the range where its closed forms cancel)
    torch.testing.assert_close(got, want, rtol=1e-9, atol=1e-9)


@needs_ref
@pytest.mark.parametrize("name,arg", [c for c in CASES if c[0] in ("so3_Jl", "so3_Jl_inv", "se3_Jl", "se3_Jl_inv", "calcQ", "sim3_Jl", "SE3_Adj", "rxso3_Ws")])
def test_composed_route_is_differentiable_like_the_reference(name, arg):
    rpp = ref_loader.load()
    alg, grp, pts = _inputs(rpp)
    x = {**alg, **grp}[arg][5:12].clone()
    w = torch.randn_like(getattr(M, name)(x))
    g1 = torch.autograd.grad((getattr(M, name)(x.requires_grad_(True)) * w).sum(), x)[0]
    x2 = x.detach().clone().requires_grad_(True)
    g2 = torch.autograd.grad((getattr(rpp.lietensor.operation, name)(x2) * w).sum(), x2)[0]
    torch.testing.assert_close(g1, g2, rtol=1e-7, atol=1e-8)


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("name,arg", CASES)
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 2e-5)])
def test_kernel_route_equals_the_reference(name, arg, dtype, tol):
    rpp = ref_loader.load()
    alg, grp, pts = _inputs(rpp)
    x = {**alg, **grp, **pts}[arg]
    want = getattr(rpp.lietensor.operation, name)(x)
    got = getattr(M, name)(x.to(dtype).cuda())
    assert got.is_cuda and got.shape == want.shape
    scale = max(1.0, float(want.abs().max()))
    if name == "rxso3_Ws":          # the reference's general (sigma, theta) branch cancels to ~1e-8 at |x| ~ 1e-9 (row 1); the kernel does not
        tol = max(tol, 1e-7)
    assert float((got.double().cpu() - want).abs().max()) <= tol * scale
