"""The two-level preconditioner of the pose-graph PCG (block-Jacobi + the gauge modes; csrc/pcg_persist.hip "CZ", the coarse
variants of csrc/graph.hip's two-launch iteration, PCG(gauge=)).  The linear system, the stop test and everything around the solve
are the reference's (optimizer.py:655-668, solver.py:319); what changes is how many iterations the same tolerance takes."""
import numpy as np
import pytest
import torch

import pypose_amd as pp
from pypose_amd.optim import posegraph
from tests.optim_models import PoseGraph, run_steps
from tests.test_optim_gpu import _synthetic_graph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _lm(graph, gauge, tol, maxiter=4000):
    return pp.optim.LM(graph, solver=pp.optim.solver.PCG(tol=tol, maxiter=maxiter, gauge=gauge),
                       strategy=pp.optim.strategy.TrustRegion(radius=1e4))


@pytest.mark.parametrize("N,E", [(3000, 12_000), (40_000, 160_000)])          # persistent ghost-zone solve / two-launch iteration
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_same_trajectory_fewer_iterations(N, E, dtype, monkeypatch):
    """tight solves: the LM trajectory (loss, damping, rejects) and the edge-relative poses do not depend on the preconditioner;
    at the solver settings of bench.py the later LM steps need a fraction of the iterations"""
    if N > posegraph.PERSIST_NODES:
        pass
    edges, rel, init = _synthetic_graph(N, E, dtype)
    tight = 1e-11 if dtype == torch.float64 else 1e-6
    runs = {}
    for gauge in (True, False):
        graph = PoseGraph(pp.SE3(init.tensor().clone()))
        opt = _lm(graph, gauge, tight)
        rec = run_steps(opt, ((edges, rel),), {}, 3)
        assert set(rec["kind"]) == {"fused:pgo"}
        assert {w.cz for w in opt._pcg_workspaces.values()} == {gauge}, "the solve did not take the route the flag asks for"
        runs[gauge] = (rec, graph.nodes.detach().tensor().double().clone())
    a, b = runs[True][0], runs[False][0]
    # (fp32 at tol 1e-6: the two solves stop at different points of an error that the conditioning of the system -- 1e5 along the
    #  gauge, ~30 elsewhere -- turns into 1e-4 of the first step's loss; from the second step on both sit at the same minimum)
    np.testing.assert_allclose(a["loss"], b["loss"], rtol=1e-9 if dtype == torch.float64 else 2e-4)
    assert a["damping"] == b["damping"] and a["reject"] == b["reject"]
    rel_of = lambda nodes: pp.SE3(nodes[edges[:, 0]]).Inv() @ pp.SE3(nodes[edges[:, 1]])
    err = (rel_of(runs[True][1]).Inv() @ rel_of(runs[False][1])).Log().tensor().abs().max().item()
    assert err <= (1e-8 if dtype == torch.float64 else 2e-4), err
    its = {}
    for gauge in (True, False):
        graph = PoseGraph(pp.SE3(init.tensor().clone()))
        opt = _lm(graph, gauge, 1e-4, maxiter=250)
        counts = []
        for _ in range(3):
            opt.step((edges, rel))
            counts.append(opt.solver.iterations)
        its[gauge] = counts
    print(f"\nPCG iterations per LM step at tol 1e-4, {N} nodes, {dtype}: gauge {its[True]}, block-Jacobi {its[False]}")
    assert sum(its[True]) <= 0.85 * sum(its[False]), its
    assert its[True][-1] <= 0.65 * its[False][-1], its


@pytest.mark.parametrize("N,E", [(3000, 12_000), (40_000, 160_000)])
def test_solution_of_one_system_equals_the_block_jacobi_solution(N, E):
    """one damped system solved to 1e-12 by both iterations (fp64): the same vector, gauge component included"""
    edges, rel, init = _synthetic_graph(N, E, torch.float64)
    sol = {}
    for gauge in (True, False):
        graph = PoseGraph(pp.SE3(init.tensor().clone()))
        opt = _lm(graph, gauge, 1e-13, maxiter=20000)
        opt.step((edges, rel))
        sol[gauge] = (graph.nodes.detach().tensor().clone(), opt.solver.iterations)
    # (nodes after ONE step from the same start: Exp(x) n with the two solutions x)
    d = (pp.SE3(sol[True][0]).Inv() @ pp.SE3(sol[False][0])).Log().tensor().abs().max().item()
    assert d <= 1e-7, (d, sol[True][1], sol[False][1])
    assert sol[True][1] < sol[False][1]


def test_non_antisymmetric_linearisations_keep_block_jacobi():
    """a prior on one node (a unary residual) makes the problem gauge-free: J Z != 0, the coarse space does not apply and must not
    be used -- the multi-residual route never sets `antisym`"""
    edges, rel, init = _synthetic_graph(500, 1500, torch.float64)
    graph = PoseGraph(pp.SE3(init.tensor().clone()))
    opt = _lm(graph, True, 1e-10)
    opt.structured, opt.fused = True, False              # the autograd-derived graph linearisation: no antisym declaration
    opt.step((edges, rel))
    assert opt.linearization == "graph"
    assert all(not w.cz for w in (opt.__dict__.get('_pcg_workspaces') or {}).values())


@pytest.mark.parametrize("dtype,tol,eps", [(torch.float64, 1e-13, 1e-8), (torch.float32, 1e-5, 5e-4)])
def test_packed_diagonal_blocks_give_the_same_solve(dtype, tol, eps, monkeypatch):
    """The two-level two-launch iteration reading D / Binv as packed upper triangles (pplie_pcg_prepare_coarse_dp, pplie_pcg2_*_dp,
    csrc/graph.hip DPK) against the same iteration on the full blocks: one damped system, the same step (Binv enters through its
    upper triangle, i.e. exactly symmetrised -- a rounding-level change of the preconditioner), and the triangles themselves."""
    edges, rel, init = _synthetic_graph(40_000, 160_000, dtype)
    sol = {}
    for dp in (True, False):
        monkeypatch.setattr(posegraph.FusedPCG, "pack_diag", dp, raising=False)
        graph = PoseGraph(pp.SE3(init.tensor().clone()))
        opt = _lm(graph, True, tol, maxiter=20000)
        opt.step((edges, rel))
        wsp = [w for w in opt._pcg_workspaces.values()]
        assert len(wsp) == 1 and wsp[0].cz and wsp[0].dp == dp and wsp[0].sym == 'pack'
        sol[dp] = (graph.nodes.detach().tensor().clone(), opt.solver.iterations, wsp[0])
    d = (pp.SE3(sol[True][0]).Inv() @ pp.SE3(sol[False][0])).Log().tensor().abs().max().item()
    assert d <= eps, (d, sol[True][1], sol[False][1])
    assert abs(sol[True][1] - sol[False][1]) <= max(2, sol[False][1] // 10), (sol[True][1], sol[False][1])
    w = sol[True][2]
    iu = torch.triu_indices(6, 6)
    assert torch.equal(w.Dp, w.D[:, iu[0], iu[1]]) and torch.equal(w.Bp, w.Binv[:, iu[0], iu[1]])
