"""pplie_pgo_linearize_lap (round 6): the pose-graph linearisation that also leaves the edges' shares of the normal equations, against
the two-step route it replaces (pplie_pgo_linearize, then pplie_graph_assemble_lap reading the J blocks back) -- same products in the
same order (equal up to FMA contraction); both block layouts (full [6, 6] per incidence up to the persistent solve's size, packed triangles beyond),
with and without a built-in robust kernel.  Reference semantics: examples/module/pgo/pgo.py:15-25, pypose/optim/optimizer.py:655-668."""
import pytest
import torch

import pypose_amd as pp
from pypose_amd.optim import fused as F
from tests.optim_models import PoseGraph
from tests.test_optim_gpu import _synthetic_graph

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("robust", [False, True], ids=["trivial", "huber"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("N,E", [(3000, 12_001), (40_000, 130_000)], ids=["full_blocks", "packed_blocks"])
def test_fused_assembly_equals_linearize_then_assemble(N, E, dtype, robust, monkeypatch):
    edges, rel, init = _synthetic_graph(N, E, dtype)
    graph = PoseGraph(init.clone())
    kw = {}
    if robust:
        kw = dict(kernel=pp.optim.kernel.Huber(delta=0.05), corrector=pp.optim.corrector.FastTriggs(pp.optim.kernel.Huber(delta=0.05)))
    opt = pp.optim.LM(graph, solver=pp.optim.solver.PCG(tol=1e-4, maxiter=250), strategy=pp.optim.strategy.TrustRegion(radius=1e4), **kw)
    opt.step((edges, rel))
    assert opt.linearization == "fused:pgo"
    prog = opt._structure_cache["program"][3]
    out = {}
    for fuse in (True, False):
        monkeypatch.setattr(F, "FUSE_PGO_ASSEMBLY", fuse)
        with torch.no_grad():
            lin = F._pgo_linearization(opt, prog, None, graph.nodes, not robust)
            lin.build_normal_equations(1e-6, 1e32)
        plan = lin.plan_blocks()
        assert plan is not None and bool(plan.get('blocks_done')) == fuse
        assert lin.HB_pack == (N > 32768)
        out[fuse] = (lin.R.clone(), lin.J.clone(), lin.HB.clone(), plan['gg'].clone(), lin.B.clone(), lin.g.clone())
    torch.cuda.synchronize()
    # R and J come out of the same instructions: equal bits.  The blocks are the same products summed in the same order, but the compiler
    # contracts them into FMAs differently in the two kernels: equal to a few ulp of the largest entry of a block
    tol = 2e-6 if dtype == torch.float32 else 1e-14
    for a, b, name in zip(out[True], out[False], ("R", "J", "HB", "gg", "B", "g")):
        assert a.shape == b.shape and torch.isfinite(a).all(), name
        if name in ("R", "J"):
            assert torch.equal(a, b), (name, float((a - b).abs().max()))
        else:
            a2, b2 = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
            err = ((a2 - b2).abs().amax(-1) / b2.abs().amax(-1).clamp_min(1e-30)).max()
            assert float(err) <= tol, (name, float(err))
    # and the blocks are what they claim to be: -J_1^T J_1 of an arbitrary edge at both of its incidence slots
    lin_R, lin_J, HB = out[True][0], out[True][1], out[True][2]
    inc = lin.incidence_slots()
    e = E // 3
    S = lin_J[e, 1].mT @ lin_J[e, 1]
    for side in (0, 1):
        blk = HB[int(inc[e, side])]
        if N > 32768:
            iu = torch.triu_indices(6, 6)
            full = torch.zeros(6, 6, dtype=dtype, device=blk.device)
            full[iu[0], iu[1]] = blk
            blk = full + full.triu(1).mT
        torch.testing.assert_close(blk, -S, rtol=1e-5 if dtype == torch.float32 else 1e-12, atol=1e-6 if dtype == torch.float32 else 1e-13)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("N,E", [(3000, 12_001), (40_000, 130_000)], ids=["persistent_solve", "two_launch_solve"])
def test_set_up_launch_that_finishes_the_assembly(N, E, dtype, monkeypatch):
    """pplie_pcg_prepare_lap (the per-node sums of the assembly inside the solve's set-up launch, the control block cleared by the
    linearisation's launch) against the separate launches: the same damped blocks, inverses, right-hand side, and the same solve."""
    from pypose_amd.optim import posegraph as G
    edges, rel, init = _synthetic_graph(N, E, dtype)
    graph = PoseGraph(init.clone())
    solver = pp.optim.solver.PCG(tol=1e-6 if dtype == torch.float32 else 1e-10, maxiter=400)
    opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4))
    opt.step((edges, rel))
    graph.nodes.data.copy_(init.tensor())
    prog = opt._structure_cache["program"][3]
    wsp = next(iter(opt._pcg_workspaces.values()))
    out = {}
    for fuse in (True, False):
        monkeypatch.setattr(F, "FUSE_PGO_ASSEMBLY", fuse)
        monkeypatch.setattr(G.FusedPCG, "fuse_prepare", fuse)
        with torch.no_grad():
            lin = F._pgo_linearization(opt, prog, None, graph.nodes, True)
            lin.build_normal_equations(1e-6, 1e32)
            assert (lin.__dict__.get('_diag_pending') is not None) == fuse and (lin.__dict__.get('_begun') is not None) == fuse
            lin.damp(1e-4)
            x, its = wsp.solve(lin, lin.s, lin.dmin, lin.dmax, solver.tol, 400, None)
            assert lin.__dict__.get('_diag_pending') is None and lin.__dict__.get('_begun') is None
            # a second solve of the same linearisation (a retry after a rejected trial) takes the plain set-up on lin.B / lin.g
            lin.damp(1e-4)
            x2, _ = wsp.solve(lin, lin.s, lin.dmin, lin.dmax, solver.tol, 400, None)
        out[fuse] = (lin.B.clone(), lin.g.clone(), wsp.D.clone(), wsp.Binv.clone(), x.clone(), x2.clone(), its)
    torch.cuda.synchronize()
    tol = 2e-6 if dtype == torch.float32 else 1e-13
    for a, b, name in zip(out[True][:3], out[False][:3], ("B", "g", "D")):
        a2, b2 = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
        err = ((a2 - b2).abs().amax(-1) / b2.abs().amax(-1).clamp_min(1e-30)).max()
        assert float(err) <= tol, (name, float(err))
    # (the inverses of blocks that differ in their last bits differ by the blocks' condition number times that: each is checked
    #  against its own D)
    eye = torch.eye(6, dtype=dtype, device=out[True][2].device)
    if dtype == torch.float64:                # (fp32: blocks with entries from 1e-2 to 1e4 -- the residual says nothing there)
        for fuse in (True, False):
            res = (out[fuse][3] @ out[fuse][2] - eye).abs().amax((-1, -2)).max()
            assert float(res) <= 1e-9, (fuse, float(res))
    for k in (4, 5):
        a, b = out[True][k], out[False][k]
        assert float((a - b).abs().max()) <= (2e-3 if dtype == torch.float32 else 1e-8) * float(b.abs().max()), \
            (k, float((a - b).abs().max()), float(b.abs().max()), out[True][6], out[False][6])
    assert abs(out[True][6] - out[False][6]) <= 1


@pytest.mark.parametrize("robust", [False, True], ids=["trivial", "huber"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.float64, 1e-12)], ids=["fp32", "fp64"])
@pytest.mark.parametrize("N,E", [(40, 41), (3000, 12_001), (40_000, 300_000)])
def test_one_launch_loss_equals_the_models_loss(N, E, dtype, tol, robust):
    """pplie_pgo_loss (the residual kernel whose last workgroup adds the partial sums: one launch, one device scalar) against the
    model's own loss, sum_e rho(|r_e|^2) of the forward's residuals in fp64 (optimizer.py:118-125); twice, for the ticket's rest state."""
    edges, rel, init = _synthetic_graph(N, E, dtype)
    graph = PoseGraph(init.clone())
    kw = dict(kernel=pp.optim.kernel.Huber(delta=0.05)) if robust else {}
    opt = pp.optim.LM(graph, solver=pp.optim.solver.PCG(tol=1e-4, maxiter=250), strategy=pp.optim.strategy.TrustRegion(radius=1e4), **kw)
    opt.step((edges, rel))
    assert opt.linearization == "fused:pgo"
    prog = opt._structure_cache["program"][3]
    from pypose_amd.optim.kernel import robust_code
    code = robust_code(opt.model.kernel[0]) if robust else None
    for _ in range(2):
        with torch.no_grad():
            x = graph(edges, rel).double().square().sum(-1)
            ref = (pp.optim.kernel.Huber(delta=0.05)(x) if robust else x).sum()
            got = prog.loss(None, code)
        assert got.shape == () and got.dtype == dtype
        assert abs(float(got) - float(ref)) <= tol * abs(float(ref)), (float(got), float(ref))
        with torch.no_grad():
            graph.nodes.data.copy_((pp.randn_SE3(N, sigma=1e-3, dtype=dtype, device=init.device) @ init).tensor())
