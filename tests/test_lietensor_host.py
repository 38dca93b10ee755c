"""Host-side logic of pypose_amd.lietensor on the CPU (oracle stand-in backend):
dispatch, broadcasting, ltype propagation, autograd wiring, vmap rules, error behaviour.
Modelled on the reference's tests/lietensor/test_lietensor.py (invariants, not golden values).
"""
import numpy as np
import pytest
import torch

import pypose_amd as pp
from tests.oracle_backend import oracle_backend


@pytest.fixture(autouse=True)
def _backend():
    with oracle_backend():
        yield


GROUPS = [("SO3", pp.randn_SO3, pp.randn_so3), ("SE3", pp.randn_SE3, pp.randn_se3),
          ("Sim3", pp.randn_Sim3, pp.randn_sim3), ("RxSO3", pp.randn_RxSO3, pp.randn_rxso3)]


def test_types_and_shapes():
    assert pp.SE3_type.dimension == torch.Size([7]) and pp.se3_type.dimension == torch.Size([6])
    assert pp.Sim3_type.manifold == torch.Size([7]) and pp.RxSO3_type.embedding == torch.Size([5])
    assert pp.so3_type.on_manifold and not pp.SO3_type.on_manifold
    x = pp.randn_se3(2, 3)
    assert x.lshape == (2, 3) and x.shape == (2, 3, 6) and x.ltype is pp.se3_type
    X = x.Exp()
    assert X.ltype is pp.SE3_type and X.shape == (2, 3, 7)
    assert X.Log().ltype is pp.se3_type
    assert type(X.ltype).__name__ == "SE3Type" and repr(X).startswith("SE3Type LieTensor")
    with pytest.raises(AssertionError):
        pp.SE3(torch.zeros(3, 6))


def test_exp_log_errors():
    with pytest.raises(AttributeError):
        pp.randn_SE3(2).Exp()
    with pytest.raises(AttributeError):
        pp.randn_se3(2).Log()
    with pytest.raises(AttributeError):
        pp.randn_se3(2).Retr(pp.randn_se3(2))
    with pytest.raises(NotImplementedError):
        pp.randn_SE3(2) * 2.0


@pytest.mark.parametrize("name,rG,ra", GROUPS)
def test_inv_log_commute(name, rG, ra):
    X = rG(2, 5, dtype=torch.float64)
    torch.testing.assert_close(X.Inv().Log().tensor(), X.Log().Inv().tensor(), rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("name,rG,ra", GROUPS)
def test_adj_identity(name, rG, ra):
    # Exp(Adj_X a) * X == X * Exp(a)     (reference tests/lietensor/test_lietensor.py:108-117)
    X, a = rG(4, dtype=torch.float64), ra(4, sigma=0.3, dtype=torch.float64)
    lhs, rhs = X.Adj(a).Exp() * X, X * a.Exp()
    torch.testing.assert_close((lhs.Inv() * rhs).Log().tensor(), torch.zeros_like(a.tensor()), atol=1e-6, rtol=0)
    lhs, rhs = a.Exp() * X, X * X.AdjT(a).Exp()
    torch.testing.assert_close((lhs.Inv() * rhs).Log().tensor(), torch.zeros_like(a.tensor()), atol=1e-6, rtol=0)


def test_broadcast_mul_act():
    X, Y = pp.randn_SE3(1, 5), pp.randn_SE3(5, 1)
    assert (X * Y).shape == (5, 5, 7) and (X @ Y).ltype is pp.SE3_type
    p = torch.randn(5, 3)
    assert (pp.randn_SE3(2, 1) @ p).shape == (2, 5, 3)
    p4 = torch.randn(5, 4)
    assert pp.randn_Sim3(5).Act(p4).shape == (5, 4)
    assert (pp.randn_SO3() * pp.randn_SO3()).shape == (4,)


def test_ltype_propagation():
    x = pp.randn_SE3(4, 2)
    assert torch.cat([x, x]).ltype is pp.SE3_type
    assert torch.stack([x, x]).ltype is pp.SE3_type
    assert x[0].ltype is pp.SE3_type and x.view(8, 7).ltype is pp.SE3_type
    assert x.to(torch.float64).ltype is pp.SE3_type and x.clone().ltype is pp.SE3_type
    a, b = x.split([1, 3], 0)
    assert a.ltype is pp.SE3_type and b.shape == (3, 2, 7)
    assert not isinstance(x.tensor(), pp.LieTensor)
    assert not hasattr(x.sum(), "ltype") or not isinstance(x + 0, type(None))


def test_parameter():
    p = pp.Parameter(pp.randn_SE3(3))
    assert isinstance(p, torch.nn.Parameter) and p.ltype is pp.SE3_type and p.requires_grad
    m = torch.nn.Module()
    m.pose = p
    assert len(list(m.parameters())) == 1
    import copy
    q = copy.deepcopy(p)
    assert q.ltype is pp.SE3_type and torch.equal(q.tensor(), p.tensor())
    with torch.no_grad():
        before = p.clone()
        p.add_(torch.zeros(3, 7))
        torch.testing.assert_close(p.tensor(), before.tensor())


def test_add_is_left_retraction():
    X = pp.randn_SE3(3, dtype=torch.float64)
    d = torch.randn(3, 7, dtype=torch.float64) * 0.1
    Y = X + d
    Z = pp.se3(d[..., :6]).Exp() * X
    torch.testing.assert_close(Y.tensor(), Z.tensor())
    a = pp.randn_se3(3, dtype=torch.float64)
    torch.testing.assert_close((a + d).tensor(), a.tensor() + d[..., :6])


@pytest.mark.parametrize("name,rG,ra", GROUPS)
def test_autograd_matches_finite_difference_structure(name, rG, ra):
    # d/dx sum(Log(Exp(x))) == ones (the two custom backwards compose to the identity map)
    x = ra(6, sigma=0.5, dtype=torch.float64, requires_grad=True)
    y = x.Exp().Log()
    y.sum().backward()
    # Sim3's Jacobians are truncated series in the reference (operation.py:159-172): not exact inverses
    torch.testing.assert_close(x.grad, torch.ones_like(x.grad), atol=2e-2 if name == "Sim3" else 1e-7, rtol=0)


def test_grad_wrt_group_is_tangent_padded():
    X = pp.randn_SE3(4, dtype=torch.float64, requires_grad=True)
    p = torch.randn(4, 3, dtype=torch.float64, requires_grad=True)
    (X.Act(p)).sum().backward()
    assert X.grad.shape == (4, 7) and torch.all(X.grad[:, 6] == 0)
    assert p.grad.shape == (4, 3)


def test_vmap_and_vectorized_jacobian():
    x = pp.randn_se3(3, dtype=torch.float64)

    def f(t):
        return pp.se3(t).Exp().Log().tensor()
    J = torch.autograd.functional.jacobian(f, x.tensor(), vectorize=True)
    Jl = torch.autograd.functional.jacobian(f, x.tensor(), vectorize=False)
    torch.testing.assert_close(J, Jl)
    assert J.shape == (3, 6, 3, 6)
    eye = torch.eye(6, dtype=torch.float64)
    for i in range(3):
        torch.testing.assert_close(J[i, :, i, :], eye, atol=1e-7, rtol=0)
    # vmap over a forward op
    X = pp.randn_SE3(5, dtype=torch.float64).tensor()
    out = torch.vmap(lambda t: pp.SE3(t).Inv().tensor())(X)
    torch.testing.assert_close(out, pp.SE3(X).Inv().tensor())


def test_matrix_rotation_translation():
    X = pp.randn_SE3(3, dtype=torch.float64)
    T = X.matrix()
    assert T.shape == (3, 4, 4)
    torch.testing.assert_close(T[:, :3, 3], X.translation())
    R = X.rotation().matrix()
    torch.testing.assert_close(T[:, :3, :3], R)
    torch.testing.assert_close(R @ R.mT, torch.eye(3, dtype=torch.float64).expand(3, 3, 3), atol=1e-9, rtol=0)
    S = pp.randn_Sim3(2, dtype=torch.float64)
    torch.testing.assert_close(S.matrix()[:, :3, :3], S.scale()[..., None] * S.rotation().matrix())


def test_identity_and_randn_seeded():
    assert torch.equal(pp.identity_SE3(2).tensor(), torch.tensor([[0., 0, 0, 0, 0, 0, 1]] * 2))
    assert torch.equal(pp.identity_Sim3(1).tensor(), torch.tensor([[0., 0, 0, 0, 0, 0, 1, 1]]))
    assert torch.equal(pp.identity_rxso3(2, 2).tensor(), torch.zeros(2, 2, 4))
    torch.manual_seed(0)
    a = pp.randn_se3(4)
    torch.manual_seed(0)
    b = pp.randn_se3(4)
    assert torch.equal(a.tensor(), b.tensor())
    x = pp.randn_SO3(2)
    x.identity_()
    assert torch.equal(x.tensor(), torch.tensor([[0., 0, 0, 1]] * 2))


def test_empty_and_noncontiguous():
    X = pp.SE3(torch.zeros(0, 7))
    assert X.Log().shape == (0, 6) and X.Inv().shape == (0, 7)
    assert (X * X).shape == (0, 7)
    Y = pp.randn_SE3(6, dtype=torch.float64)
    Z = Y[::2]
    torch.testing.assert_close(Z.Log().tensor(), Y.Log().tensor()[::2])
    big = torch.randn(4, 9, dtype=torch.float64)
    v = pp.se3(big[:, 1:7])
    torch.testing.assert_close(v.Exp().tensor(), pp.se3(big[:, 1:7].contiguous()).Exp().tensor())


def test_jinvp_and_jr():
    X = pp.randn_SE3(3, dtype=torch.float64)
    p = pp.randn_se3(3, dtype=torch.float64)
    assert X.Jinvp(p).ltype is pp.se3_type and X.Jinvp(p).shape == (3, 6)
    x = pp.randn_so3(4, dtype=torch.float64)
    J = x.Jr()
    assert J.shape == (4, 3, 3)
    torch.testing.assert_close(pp.so3(x.tensor()).Exp().Jr(), J, atol=1e-9, rtol=0)
