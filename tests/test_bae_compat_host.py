"""The `bae` plugin stand-in on CPU: the reference's sparse LM (LM(..., solver=PCG(), sparse=True) over
pp.Parameter(sjac=True) + @psjac) runs unmodified through pypose_amd/compat/bae; its sparse Jacobian equals the reference's
dense modjac Jacobian and its iterates equal the reference's dense LM iterates."""
import warnings

import pytest
import torch

from oracle import ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="oracle/_ref not shipped")
warnings.filterwarnings("ignore", message="Sparse CSR tensor support is in beta")

from tests.bae_compat_util import load_reference, models, chain_problem, reproj_problem, ba_example   # noqa: E402


@pytest.fixture(scope="module")
def pp():
    return load_reference()


def test_plugin_resolution(pp):
    import bae
    import pypose.optim.solver as ppos
    from pypose.autograd.function import psjac, parallel_for_sparse_jacobian
    from pypose_amd.compat.bae.utils import pysolvers
    assert bae.__version__ == "0.2.1"
    assert ppos.PCG.__module__ == "bae.utils.pysolvers" and ppos.PCG.__name__ == pysolvers.PCG.__name__
    assert psjac is parallel_for_sparse_jacobian and psjac.__module__ == "bae.autograd.function"
    p = pp.Parameter(pp.randn_SE3(3), sjac=True)
    assert isinstance(p, pp.LieTensor) and isinstance(p, torch.nn.Parameter) and p.requires_grad
    assert type(p.tensor()) is torch.Tensor


def test_reference_sparse_identity_converges(pp):            # tests/optim/test_sparse_lm.py:43-85, on CPU
    import pypose.optim.solver as ppos
    Identity, _, _ = models(pp)
    torch.manual_seed(0)
    xt = torch.randn(8, 1, dtype=torch.float64)
    x0 = xt + 0.1 * torch.randn_like(xt)
    m = Identity(x0)
    opt = pp.optim.LM(m, solver=ppos.PCG(), strategy=pp.optim.strategy.Constant(damping=1e-6), sparse=True)
    with torch.no_grad():
        loss0 = opt.model.loss(input=(), target=xt).item()
    for _ in range(6):
        loss = opt.step(input=(), target=xt).item()
    assert loss < loss0
    torch.testing.assert_close(m.x.tensor(), xt, rtol=1e-4, atol=1e-4)


def test_reference_sparse_chain_pgo_converges(pp):           # tests/optim/test_sparse_lm.py:87-150, on CPU
    import pypose.optim.solver as ppos
    _, Chain, _ = models(pp)
    torch.manual_seed(0)
    dt = torch.float64
    gt = pp.SE3(torch.tensor([[0., 0, 0, 0, 0, 0, 1], [1., 0, 0, 0, 0, 0, 1], [2., 0, 0, 0, 0, 0, 1]], dtype=dt))
    edges = torch.tensor([[0, 1], [1, 2]])
    rel = gt[edges[:, 0]].Inv() @ gt[edges[:, 1]]
    init = gt[1:] * pp.randn_SE3(2, sigma=0.1, dtype=dt)
    m = Chain(gt[:1], init)
    opt = pp.optim.LM(m, solver=ppos.PCG(), strategy=pp.optim.strategy.Constant(damping=1e-4), sparse=True)
    with torch.no_grad():
        loss0 = opt.model.loss(input=(edges, rel), target=None).item()
    for _ in range(5):
        loss = opt.step(input=(edges, rel)).item()
        if loss < 1e-5:
            break
    assert loss < loss0 and loss < 1e-5
    torch.testing.assert_close(pp.SE3(m.nodes).translation(), gt[1:].translation(), rtol=1e-3, atol=1e-3)


def _dense_J(pp, model, input):
    """the reference's dense Jacobian of the same model, by its modjac (optimizer.py:646-650)"""
    from pypose.optim.functional import modjac
    J = modjac(model, input=input, flatten=True, vectorize=True)
    return J


@pytest.mark.parametrize("case", ["chain", "reproj"])
def test_sparse_jacobian_equals_reference_modjac(pp, case):
    from bae.autograd.graph import jacobian
    _, Chain, Reproj = models(pp)
    if case == "chain":
        gt, edges, rel, init = chain_problem(pp)
        ms, md = Chain(gt[:1], init.clone(), sjac=True), Chain(gt[:1], init.clone(), sjac=False)
        inp = (edges, rel)
    else:
        poses, pts, cam, pt, pixel, f = reproj_problem(pp)
        ms, md = Reproj(poses.clone(), pts.clone(), sjac=True), Reproj(poses.clone(), pts.clone(), sjac=False)
        inp = (cam, pt, pixel, f)
    with torch.no_grad():                                    # the reference calls it under no_grad (optimizer.py:498)
        R = ms(*inp)
        Js = jacobian(R, list(ms.parameters()))
    Js = torch.cat([j.to_dense() for j in Js], -1)
    Jd = _dense_J(pp, md, inp)
    lie = next(iter(md.parameters()))                        # the dense Jacobian has the 7 ambient columns of a pose, the 7th zero
    nl = lie.numel()
    Jl = Jd[:, :nl].reshape(Jd.shape[0], -1, 7)
    assert Jl[..., 6].abs().max() == 0
    Jd = torch.cat([Jl[..., :6].reshape(Jd.shape[0], -1), Jd[:, nl:]], -1)
    assert Js.shape == Jd.shape
    torch.testing.assert_close(Js, Jd, rtol=1e-10, atol=1e-10)
    torch.testing.assert_close(torch.Tensor.as_subclass(R.detach(), torch.Tensor), md(*inp).detach(), rtol=0, atol=0)


@pytest.mark.parametrize("case", ["chain", "reproj"])
def test_sparse_lm_iterates_equal_dense_lm(pp, case):
    import pypose.optim.solver as ppos
    _, Chain, Reproj = models(pp)
    if case == "chain":
        gt, edges, rel, init = chain_problem(pp)
        ms, md = Chain(gt[:1], init.clone(), sjac=True), Chain(gt[:1], init.clone(), sjac=False)
        inp = (edges, rel)
    else:
        poses, pts, cam, pt, pixel, f = reproj_problem(pp)
        ms, md = Reproj(poses.clone(), pts.clone(), sjac=True), Reproj(poses.clone(), pts.clone(), sjac=False)
        inp = (cam, pt, pixel, f)
    strat = lambda: pp.optim.strategy.Constant(damping=1e-3)
    os_ = pp.optim.LM(ms, solver=ppos.PCG(maxiter=2000, tol=1e-14), strategy=strat(), sparse=True, min=1e-6)
    od = pp.optim.LM(md, solver=ppos.Cholesky(), strategy=strat(), vectorize=True, min=1e-6)
    for _ in range(3):
        ls, ld = os_.step(input=inp).item(), od.step(input=inp).item()
        assert ls == pytest.approx(ld, rel=1e-6, abs=1e-18)
    for a, b in zip(ms.parameters(), md.parameters()):
        torch.testing.assert_close(torch.Tensor.as_subclass(a.detach(), torch.Tensor), torch.Tensor.as_subclass(b.detach(), torch.Tensor), rtol=1e-7, atol=1e-9)


def test_diagonal_op_and_pcg():
    from pypose_amd.compat import install_bae
    install_bae()
    from bae.sparse.py_ops import diagonal_op_
    from bae.utils.pysolvers import PCG
    from functools import partial
    g = torch.Generator().manual_seed(3)
    B = torch.randn(20, 12, generator=g, dtype=torch.float64)
    B[B.abs() < 0.8] = 0
    A = (B.T @ B + 0.5 * torch.eye(12, dtype=torch.float64))
    S = A.to_sparse_csr()
    diagonal_op_(S, op=partial(torch.clamp_, min=2.0, max=5.0))
    ref = A.clone()
    ref.diagonal().clamp_(2.0, 5.0)
    torch.testing.assert_close(S.to_dense(), ref)
    diagonal_op_(S, op=partial(torch.mul, other=1.5))
    ref.diagonal().mul_(1.5)
    torch.testing.assert_close(S.to_dense(), ref)
    b = torch.randn(12, 1, generator=g, dtype=torch.float64)
    x = PCG(tol=1e-12)(A=S, b=b)
    torch.testing.assert_close(x, torch.linalg.solve(ref, b), rtol=1e-8, atol=1e-10)


def test_reference_sparse_call_site_on_the_structured_optimizer(pp):
    """install_bae() + activate(pypose, optim=True): the reference's sparse-LM call site as its users write it
    (pp.Parameter(sjac=True), @psjac, solver.PCG, LM(sparse=True)) lands on pypose_amd's pose-graph linearisation and
    walks the trajectory recorded from the reference's dense LM (tests/golden/lm_golden.npz, pgo40/noweight)."""
    import numpy as np
    from pypose_amd import activate
    from tests.optim_models import load_lm_golden
    from tests.oracle_backend import oracle_backend
    from pypose.autograd.function import psjac
    G = load_lm_golden()

    @psjac
    def err(n1, n2, poses):
        return (poses.Inv() @ n1.Inv() @ n2).Log().tensor()

    class PoseGraph(torch.nn.Module):
        def __init__(self, nodes):
            super().__init__()
            self.nodes = pp.Parameter(nodes, sjac=True)

        def forward(self, edges, poses):
            return err(self.nodes[edges[..., 0]], self.nodes[edges[..., 1]], poses)

    with oracle_backend():
        activate.activate(pp, force=True, optim=True)
        try:
            edges = torch.from_numpy(G["pgo40/edges"])
            poses = pp.SE3(torch.from_numpy(G["pgo40/poses"]))
            graph = PoseGraph(pp.SE3(torch.from_numpy(G["pgo40/init"])))
            opt = pp.optim.LM(graph, solver=pp.optim.solver.PCG(tol=1e-12), strategy=pp.optim.strategy.TrustRegion(radius=1e4),
                              min=1e-6, sparse=True)
            assert type(opt).__module__.startswith("pypose_amd")
            losses = [float(opt.step((edges, poses))) for _ in range(4)]
            assert opt.linearization == "graph"
            np.testing.assert_allclose(losses, G["pgo40/noweight/loss"][:4], rtol=1e-7)
        finally:
            activate.deactivate()


def test_reference_ba_example_model(pp):
    """the reference's bundle-adjustment example model and optimizer lines (examples/module/ba/bundle_adjustment.py:16-43,
    71-73) unmodified: (a) through the reference's own sparse LM on the stand-in == the reference's dense LM;
    (b) under activate(optim=True) on pypose_amd's multi-parameter graph linearisation == the same iterates"""
    from pypose.optim import LM
    from pypose.optim.solver import PCG, Cholesky
    from pypose_amd import activate
    from tests.oracle_backend import oracle_backend
    strat = lambda: pp.optim.strategy.TrustRegion(up=2.0, down=0.5 ** 4)
    ms, inp = ba_example(pp, True)
    md, _ = ba_example(pp, False)
    os_ = LM(ms, solver=PCG(tol=1e-14, maxiter=5000), strategy=strat(), reject=30, sparse=True)
    od = LM(md, solver=Cholesky(), strategy=strat(), reject=30, vectorize=True)
    want = []
    for _ in range(3):
        ls, ld = os_.step(inp).item(), od.step(inp).item()
        assert ls == pytest.approx(ld, rel=1e-6, abs=1e-16)
        want.append(ld)
    assert want[-1] < 1e-8 * want[0]
    with oracle_backend():
        activate.activate(pp, force=True, optim=True)
        try:
            ma, inp = ba_example(pp, True)
            oa = pp.optim.LM(ma, solver=pp.optim.solver.PCG(tol=1e-14, maxiter=5000), strategy=strat(), reject=30, sparse=True)
            assert type(oa).__module__.startswith("pypose_amd")
            got = [oa.step(inp).item() for _ in range(3)]
            assert oa.linearization == "multigraph"
        finally:
            activate.deactivate()
    for g, w in zip(got, want):
        assert g == pytest.approx(w, rel=1e-6, abs=1e-16)
