"""The reference's sparse LM (tests/optim/test_sparse_lm.py cases + a two-parameter reprojection problem) on the MI355X:
the `bae` plugin stand-in (pypose_amd/compat/bae) drives the reference's own optimizer code, the model's Lie ops are
rebound to the HIP kernels by activate(pypose); iterates must equal the un-activated reference's on the CPU."""
import warnings

import pytest
import torch

from oracle import ref_loader

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_loader.available(), reason="oracle/_ref not shipped")]
warnings.filterwarnings("ignore", message="Sparse CSR tensor support is in beta")
DEV = "cuda:0"

from tests.bae_compat_util import load_reference, models, chain_problem, reproj_problem, ba_example   # noqa: E402


def _run(pp, case, dev, steps=3):
    import pypose.optim.solver as ppos
    _, Chain, Reproj = models(pp)
    if case == "chain":
        gt, edges, rel, init = chain_problem(pp, device=dev)
        m, inp = Chain(gt[:1], init.clone()), (edges, rel)
    else:
        poses, pts, cam, pt, pixel, f = reproj_problem(pp, device=dev)
        m, inp = Reproj(poses.clone(), pts.clone()), (cam, pt, pixel, f)
    m = m.to(dev)
    opt = pp.optim.LM(m, solver=ppos.PCG(maxiter=2000, tol=1e-14), strategy=pp.optim.strategy.Constant(damping=1e-3),
                      sparse=True, min=1e-6)
    losses = [opt.step(input=inp).item() for _ in range(steps)]
    return losses, [torch.Tensor.as_subclass(p.detach(), torch.Tensor).cpu() for p in m.parameters()]


@pytest.mark.parametrize("case", ["chain", "reproj"])
def test_reference_sparse_lm_on_hip_kernels(case):
    from pypose_amd import _C, activate
    pp = load_reference()
    want_l, want_p = _run(pp, case, "cpu")
    activate.activate(pp)
    try:
        launched = []
        real = _C.row_op
        from pypose_amd.lietensor import operation as _op
        _C.row_op = _op._C.row_op = lambda name, *a, **k: (launched.append(name), real(name, *a, **k))[1]
        try:
            got_l, got_p = _run(pp, case, DEV)
        finally:
            _C.row_op = _op._C.row_op = real
    finally:
        activate.deactivate()
    need = {"se3_log_fwd", "se3_log_bwd", "se3_mul_fwd", "se3_mul_bwd"} if case == "chain" else {"se3_act_fwd", "se3_act_bwd"}
    assert need <= set(launched), sorted(set(launched))
    for g, w in zip(got_l, want_l):
        assert g == pytest.approx(w, rel=1e-6, abs=1e-16)
    for g, w in zip(got_p, want_p):
        torch.testing.assert_close(g, w, rtol=1e-7, atol=1e-9)


def test_reference_sparse_lm_cases_on_the_device():
    """tests/optim/test_sparse_lm.py:43-150 as written there (cuda, float64, model.to(device)), un-activated"""
    import pypose.optim.solver as ppos
    pp = load_reference()
    Identity, Chain, _ = models(pp)
    torch.manual_seed(0)
    dt = torch.float64
    xt = torch.randn(8, 1, device=DEV, dtype=dt)
    x0 = xt + 0.1 * torch.randn_like(xt)
    m = Identity(x0).to(DEV)
    opt = pp.optim.LM(m, solver=ppos.PCG(), strategy=pp.optim.strategy.Constant(damping=1e-6), sparse=True)
    for _ in range(6):
        loss = opt.step(input=(), target=xt).item()
    torch.testing.assert_close(m.x.tensor(), xt, rtol=1e-4, atol=1e-4)
    gt = pp.SE3(torch.tensor([[0., 0, 0, 0, 0, 0, 1], [1., 0, 0, 0, 0, 0, 1], [2., 0, 0, 0, 0, 0, 1]], device=DEV, dtype=dt))
    edges = torch.tensor([[0, 1], [1, 2]], device=DEV)
    rel = gt[edges[:, 0]].Inv() @ gt[edges[:, 1]]
    init = gt[1:] * pp.randn_SE3(2, sigma=0.1, device=DEV, dtype=dt)
    m = Chain(gt[:1], init).to(DEV)
    opt = pp.optim.LM(m, solver=ppos.PCG(), strategy=pp.optim.strategy.Constant(damping=1e-4), sparse=True)
    for _ in range(5):
        loss = opt.step(input=(edges, rel)).item()
        if loss < 1e-5:
            break
    assert loss < 1e-5
    torch.testing.assert_close(pp.SE3(m.nodes).translation(), gt[1:].translation(), rtol=1e-3, atol=1e-3)


def test_reference_sparse_call_site_on_the_fused_pose_graph_path():
    """install_bae() + activate(pypose, optim=True) on the GPU: Parameter(sjac=True) / @psjac / solver.PCG / LM(sparse=True)
    written against the reference run pypose_amd's fused pose-graph step (HIP assembly + PCG kernels)"""
    import numpy as np
    from pypose_amd import activate
    from tests.optim_models import load_lm_golden
    pp = load_reference()
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

    activate.activate(pp, optim=True)
    try:
        edges = torch.from_numpy(G["pgo40/edges"]).to(DEV)
        poses = pp.SE3(torch.from_numpy(G["pgo40/poses"])).to(DEV)
        graph = PoseGraph(pp.SE3(torch.from_numpy(G["pgo40/init"]))).to(DEV)
        opt = pp.optim.LM(graph, solver=pp.optim.solver.PCG(tol=1e-12), strategy=pp.optim.strategy.TrustRegion(radius=1e4),
                          min=1e-6, sparse=True)
        losses = [float(opt.step((edges, poses))) for _ in range(4)]
        assert opt.linearization.startswith("fused") or opt.linearization == "graph", opt.linearization
        np.testing.assert_allclose(losses, G["pgo40/noweight/loss"][:4], rtol=1e-7)
    finally:
        activate.deactivate()


def test_reference_ba_example_model_on_the_device():
    """examples/module/ba/bundle_adjustment.py:16-43, 71-73 on the MI355X: the reference's sparse LM over the stand-in with
    HIP Lie kernels (activate), and the same call site on pypose_amd's optimizer (activate(optim=True)); both equal the
    reference's dense LM on the CPU"""
    from pypose_amd import activate
    pp = load_reference()
    from pypose.optim.solver import PCG, Cholesky
    strat = lambda: pp.optim.strategy.TrustRegion(up=2.0, down=0.5 ** 4)
    md, inp = ba_example(pp, False)
    od = pp.optim.LM(md, solver=Cholesky(), strategy=strat(), reject=30, vectorize=True)
    want = [od.step(inp).item() for _ in range(3)]
    for optim in (False, True):
        activate.activate(pp, optim=optim)
        try:
            m, inp = ba_example(pp, True, device=DEV)
            opt = pp.optim.LM(m, solver=pp.optim.solver.PCG(tol=1e-14, maxiter=5000), strategy=strat(), reject=30, sparse=True)
            assert type(opt).__module__.startswith("pypose_amd" if optim else "pypose.")
            got = [opt.step(inp).item() for _ in range(3)]
        finally:
            activate.deactivate()
        for g, w in zip(got, want):
            assert g == pytest.approx(w, rel=1e-6, abs=1e-15), (optim, got, want)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-4)])
def test_jacobian_of_the_relative_pose_residual_takes_the_kernel_route(dtype, tol, monkeypatch):
    """bae.autograd.graph.jacobian on the reference's OWN (un-activated) ops: the history of
    (rel.Inv() @ n1.Inv() @ n2).Log() is recognised and the blocks come from pplie_pgo_linearize -- equal to the six autograd
    sweeps it replaces, fixed root rows (index -1) dropped alike"""
    pp = load_reference()
    import bae.autograd.graph as G
    _, Chain, _ = models(pp)
    torch.manual_seed(3)
    N = 400
    gt = pp.cumprod(pp.randn_SE3(N, sigma=0.5, dtype=dtype, device=DEV), dim=0, left=False)
    e0 = torch.cat([torch.arange(N - 1), torch.randint(0, N, (300,))]).to(DEV)
    e1 = torch.cat([torch.arange(1, N), torch.randint(0, N, (300,))]).to(DEV)
    keep = e0 != e1
    edges = torch.stack([e0[keep], e1[keep]], 1)
    rel = gt[edges[:, 0]].Inv() @ gt[edges[:, 1]] @ pp.randn_SE3(edges.shape[0], sigma=0.05, dtype=dtype, device=DEV)
    model = Chain(gt[:1], (gt[1:] @ pp.randn_SE3(N - 1, sigma=0.1, dtype=dtype, device=DEV))).to(DEV)
    with torch.no_grad():
        J_k = [j.to_dense() for j in G.jacobian(model(edges, rel), [model.nodes])]
    assert G.route_taken["last"] == "kernel:pgo", G.route_taken
    monkeypatch.setattr(G, "_match_pgo", lambda *a, **k: None)
    with torch.no_grad():
        J_a = [j.to_dense() for j in G.jacobian(model(edges, rel), [model.nodes])]
    assert G.route_taken["last"] == "autograd"
    for a, b in zip(J_k, J_a):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= tol * float(b.abs().max())


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 2e-3)])
def test_csr_pcg_and_diagonal_op_run_on_hip_kernels(dtype, tol):
    """bae.utils.pysolvers.PCG / bae.sparse.py_ops.diagonal_op_ on a CSR normal-equation matrix: csrc/csr_pcg.hip against the torch
    formulation of the same iteration (same iteration count +- 1, same solution), the diagonal rewritten in place"""
    import functools
    load_reference()
    from bae.sparse.py_ops import diagonal_op_
    from bae.utils import pysolvers
    torch.manual_seed(5)
    n, m = 6000, 9000
    rows = torch.randint(0, m, (m * 6,))
    cols = torch.randint(0, n, (m * 6,))
    J = torch.sparse_coo_tensor(torch.stack([rows, cols]), torch.randn(m * 6, dtype=dtype), (m, n)).coalesce().to(DEV)
    A = (J.mT.to_sparse_csr() @ J.to_sparse_csr())
    A = (A.to_sparse_coo() + torch.sparse_coo_tensor(torch.stack([torch.arange(n)] * 2).to(DEV), torch.full((n,), 0.5, dtype=dtype, device=DEV),
                                                   (n, n))).coalesce().to_sparse_csr()
    A2 = torch.sparse_csr_tensor(A.crow_indices().clone(), A.col_indices().clone(), A.values().clone(), A.shape)
    for op in (functools.partial(torch.clamp_, min=1.0, max=30.0), functools.partial(torch.mul, other=1.25)):
        diagonal_op_(A, op)                                   # HIP
        crow, col, val = A2.crow_indices(), A2.col_indices(), A2.values()
        row = torch.repeat_interleave(torch.arange(n, device=DEV), crow[1:] - crow[:-1])
        on = (row == col).nonzero().reshape(-1)
        res = op(val[on])
        val[on] = res if res is not None else val[on]        # the torch formulation
        assert torch.equal(A.values(), A2.values())
    b = torch.randn(n, 1, dtype=dtype, device=DEV)
    hip = pysolvers.PCG(maxiter=2000, tol=1e-10 if dtype == torch.float64 else 1e-5)
    x_h = hip(A, b)
    assert hip.route == "hip"
    ref = pysolvers.PCG(maxiter=2000, tol=hip.tol)
    x_t = ref(A.to_dense(), b)                                # (dense input: the torch route)
    assert ref.route == "torch"
    assert abs(hip.iterations - ref.iterations) <= 2, (hip.iterations, ref.iterations)
    assert float((x_h - x_t).abs().max()) <= tol * float(x_t.abs().max())
    assert float((A.to_dense() @ x_h - b).norm() / b.norm()) <= 10 * hip.tol
