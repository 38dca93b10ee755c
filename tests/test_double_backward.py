"""Double backward through the Lie Functions (missing 6): under create_graph=True the backward kernels'
own derivative is taken through the differentiable compositions of lietensor/matrices.py, the structure of the reference's
backward passes (pypose/lietensor/operation.py:366-370, 389-395, ...).  Second derivatives against the reference's, fp64."""
import pytest
import torch

import pypose_amd as pp
from oracle import ref_loader
from tests.oracle_backend import oracle_backend

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="oracle/_ref not present (make -C oracle)")
D = torch.float64


def _scalar(P, tag, x, y, p, a):
    """one scalar that runs every Function of one group: Exp, Log, Inv, Mul, Act, Act4, Adj, AdjT, Jinvp"""
    mk = {"so3": P.so3, "se3": P.se3, "sim3": P.sim3, "rxso3": P.rxso3}[tag]
    X, Y = mk(x).Exp(), mk(y).Exp()
    Z = X.Inv() @ Y
    out = Z.Log().tensor().square().sum() + (X @ p).square().sum() + (Y @ torch.cat([p, torch.ones_like(p[..., :1])], -1)).sum()
    alg = mk(a)
    out = out + (X.Adj(alg).tensor() * Y.AdjT(alg).tensor()).sum() + Z.Jinvp(alg).tensor().sum()
    return out


@pytest.mark.parametrize("tag,w", [("so3", 3), ("se3", 6), ("sim3", 7), ("rxso3", 4)])
def test_hessian_vector_products_equal_the_reference(tag, w):
    rpp = ref_loader.load()
    torch.manual_seed(4)
    n = 5
    x0, y0, a0 = (0.7 * torch.randn(n, w, dtype=D) for _ in range(3))
    p0 = torch.randn(n, 3, dtype=D)
    v = torch.randn(n, w, dtype=D)

    def hvp(P):
        x = x0.clone().requires_grad_(True)
        f = _scalar(P, tag, x, y0, p0, a0)
        (g,) = torch.autograd.grad(f, x, create_graph=True)
        (h,) = torch.autograd.grad((g * v).sum(), x)
        return g.detach(), h
    g_ref, h_ref = hvp(rpp)
    with oracle_backend():
        g, h = hvp(pp)
    torch.testing.assert_close(g, g_ref, rtol=1e-8, atol=1e-9)
    torch.testing.assert_close(h, h_ref, rtol=1e-6, atol=1e-7)


def test_modjac_create_graph_and_so3_jr():
    rpp = ref_loader.load()
    torch.manual_seed(5)
    x0 = torch.randn(4, 3, dtype=D)

    def second(P):
        x = x0.clone().requires_grad_(True)
        J = P.so3(x).Jr()
        (g,) = torch.autograd.grad(J.square().sum(), x, create_graph=True)
        (h,) = torch.autograd.grad(g.sum(), x)
        return h
    with oracle_backend():
        h = second(pp)
    torch.testing.assert_close(h, second(rpp), rtol=1e-6, atol=1e-8)
