"""Bundle adjustment (several parameters of different widths gathered per observation) on the CPU through the
oracle backend: the reference example's model takes the multi-parameter graph path and reproduces the
trajectories recorded from the reference's dense LM."""
import numpy as np
import pytest
import torch

import pypose_amd as pp
from tests.optim_models import ba_case, compare_trajectory, load_ba_golden, run_steps
from tests.oracle_backend import oracle_backend


@pytest.fixture(scope="module")
def G():
    return load_ba_golden()


@pytest.mark.parametrize("structured", [False, True])
@pytest.mark.parametrize("tag", ["ba_small", "ba_huber"])
def test_ba_trajectory_matches_reference(G, tag, structured):
    with oracle_backend():
        model, opt, args = ba_case(G, tag)
        opt.structured = structured
        rec = run_steps(opt, (args,), {}, 6)
        assert set(rec["kind"]) == ({"multigraph"} if structured else {"dense"}), rec["kind"]
        compare_trajectory(rec, G, tag, floor=1e-12, rtol=1e-7)
        np.testing.assert_allclose(model.P.detach().numpy(), G[f"{tag}/P"], atol=1e-6)
        np.testing.assert_allclose(model.C.detach().tensor().numpy(), G[f"{tag}/C"], atol=1e-6)
        np.testing.assert_allclose(model.K.detach().numpy(), G[f"{tag}/K"], rtol=1e-6, atol=1e-6)


def test_ba_matrix_free_pcg_and_operator(G):
    """PCG on the matrix-free operator == the dense solve; J @ D of the strategy operator == dense J @ D."""
    from pypose_amd.optim import multigraph
    from pypose_amd.optim.optimizer import _linearize, DenseLinearization
    with oracle_backend():
        model, opt, args = ba_case(G, "ba_small")
        pg = opt.param_groups[0]
        with torch.no_grad():
            lin = _linearize(opt, pg, args, None, None)
            assert isinstance(lin, multigraph.MultiGraphLinearization)
            lin.build_normal_equations(pg['min'], pg['max'])
            lin.damp(1e-3)
            D1 = lin.solve(pp.optim.solver.Cholesky())
            D2 = lin.solve(pp.optim.solver.PCG(tol=1e-14, maxiter=5000, check_every=1))
            torch.testing.assert_close(D1, D2, rtol=1e-7, atol=1e-9)
            dense = DenseLinearization(opt, pg, args, None, None)
            dense.build_normal_equations(pg['min'], pg['max'])
            dense.damp(1e-3)
            D3 = dense.solve(pp.optim.solver.Cholesky())
            torch.testing.assert_close(D1, D3, rtol=1e-7, atol=1e-9)
            J, R = lin.strategy_args()
            torch.testing.assert_close(J @ D1, dense.J @ D3, rtol=1e-7, atol=1e-9)
            torch.testing.assert_close(R, dense.R.view(-1, 1))


def test_self_loop_slots_keep_the_exact_diagonal():
    """Two gathers of the same parameter hitting the same row: the cross term lands on the diagonal block."""
    class Pair(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.x = pp.Parameter(torch.randn(5, 2, dtype=torch.float64))
            self.y = pp.Parameter(torch.randn(3, 2, dtype=torch.float64))

        def forward(self, i, j, k):
            return self.x[i] * 2 + self.x[j] ** 2 + self.y[k]

    torch.manual_seed(0)
    i = torch.tensor([0, 1, 2, 3, 4, 1]); j = torch.tensor([1, 1, 3, 0, 4, 2]); k = torch.tensor([0, 1, 2, 0, 1, 2])
    from pypose_amd.optim.optimizer import _linearize, DenseLinearization
    net = Pair()
    opt = pp.optim.LM(net, solver=pp.optim.solver.Cholesky())
    pg = opt.param_groups[0]
    with torch.no_grad():
        lin = _linearize(opt, pg, (i, j, k), None, None)
        assert lin.kind == "multigraph"
        lin.build_normal_equations(1e-6, 1e32)
        dense = DenseLinearization(opt, pg, (i, j, k), None, None)
        dense.build_normal_equations(1e-6, 1e32)
        torch.testing.assert_close(torch.cat([d.reshape(-1) for d in lin.diag_clamped]), dense.A.diagonal())
        lin.damp(0.5), dense.damp(0.5)              # (x[i]*2 + x[j]^2 + y[k] is rank deficient without damping)
        torch.testing.assert_close(lin.solve(pp.optim.solver.Cholesky()), dense.solve(pp.optim.solver.Cholesky()))


def test_gauss_newton_on_a_pose_graph_takes_the_reference_path(G):
    """GN solves the rectangular system with the pseudo-inverse (minimum-norm step on a gauge-free graph): the
    graph / fused linearisations do not apply, the dense reference algorithm must (and did crash before)."""
    from tests.optim_models import PoseGraph, T, load_lm_golden
    L = load_lm_golden()
    with oracle_backend():
        edges, poses = T(L["pgo12/edges"]), pp.SE3(T(L["pgo12/poses"]))
        graph = PoseGraph(pp.SE3(T(L["pgo12/init"])))
        opt = pp.optim.GN(graph)
        losses = [float(opt.step((edges, poses))) for _ in range(3)]
        np.testing.assert_allclose(losses, G["gn_pgo12/loss"], rtol=1e-8)
        np.testing.assert_allclose(graph.nodes.detach().tensor().numpy(), G["gn_pgo12/final"], atol=1e-8)


@pytest.mark.parametrize("tag", ["ba_small", "ba_huber"])
def test_schur_complement_path_matches_reference(G, tag, monkeypatch):
    """With a direct solver and more unknowns than the dense limit, bipartite problems eliminate the point rows exactly
    (Schur complement) and hand the reduced camera system to the user's solver: same trajectory as the reference's
    dense Cholesky."""
    from pypose_amd.optim import multigraph
    monkeypatch.setattr(multigraph, "DENSE_LIMIT", 0)
    with oracle_backend():
        model, opt, args = ba_case(G, tag)
        rec = run_steps(opt, (args,), {}, 6)
        assert set(rec["kind"]) == {"multigraph"}
        lin_plans = opt.__dict__.get("_schur_plans")
        assert lin_plans, "the Schur plan was not built"
        compare_trajectory(rec, G, tag, floor=1e-12, rtol=1e-7)
        np.testing.assert_allclose(model.P.detach().numpy(), G[f"{tag}/P"], atol=1e-6)
        np.testing.assert_allclose(model.C.detach().tensor().numpy(), G[f"{tag}/C"], atol=1e-6)


def test_gauss_newton_graph_path_reproduces_the_pseudo_inverse_steps(G):
    """With a PCG solver (or a graph too large for a dense J) Gauss-Newton runs on the graph linearisation: plain CG
    from zero on the singular, gauge-free normal equations gives the reference's minimum-norm (pinv) steps."""
    from tests.optim_models import PoseGraph, T, load_lm_golden
    from pypose_amd.optim.optimizer import _linearize
    L = load_lm_golden()
    with oracle_backend():
        edges, poses = T(L["pgo12/edges"]), pp.SE3(T(L["pgo12/poses"]))
        graph = PoseGraph(pp.SE3(T(L["pgo12/init"])))
        opt = pp.optim.GN(graph, solver=pp.optim.solver.PCG(tol=1e-13, maxiter=500, check_every=1))
        with torch.no_grad():
            assert _linearize(opt, opt.param_groups[0], (edges, poses), None, None, gauss_newton=True).kind == "graph"
        losses = [float(opt.step((edges, poses))) for _ in range(3)]
        np.testing.assert_allclose(losses, G["gn_pgo12/loss"], rtol=1e-7)
        np.testing.assert_allclose(graph.nodes.detach().tensor().numpy(), G["gn_pgo12/final"], atol=1e-7)


def test_gauss_newton_graph_path_weights_both_sides():
    """GN's rectangular system is W J d = -W R: its normal equations carry W^T W (the dense path is the reference)."""
    from tests.optim_models import PoseGraph, T, load_lm_golden
    L = load_lm_golden()
    torch.manual_seed(4)
    M = torch.randn(6, 6, dtype=torch.float64)
    W = M @ M.T / 6 + torch.eye(6, dtype=torch.float64)
    with oracle_backend():
        edges, poses = T(L["pgo12/edges"]), pp.SE3(T(L["pgo12/poses"]))
        out = []
        for solver in (None, pp.optim.solver.PCG(tol=1e-13, maxiter=500, check_every=1)):
            graph = PoseGraph(pp.SE3(T(L["pgo12/init"])))
            opt = pp.optim.GN(graph, solver=solver, weight=W)
            losses = [float(opt.step((edges, poses))) for _ in range(2)]
            out.append((losses, graph.nodes.detach().tensor().clone()))
        np.testing.assert_allclose(out[1][0], out[0][0], rtol=1e-7)
        torch.testing.assert_close(out[1][1], out[0][1], rtol=0, atol=1e-7)


@pytest.mark.parametrize("tag", ["plain", "kernels", "kernels_weights"])
def test_multi_residual_pose_graph_matches_reference(tag):
    """A model returning three residuals of different width / gather count (odometry, loop closures with their own
    kernel and information matrix, 3-row priors on single nodes) takes the graph path and reproduces the reference's
    dense trajectory -- including its 16-fold rejections when weights and kernels pull apart."""
    from tests.optim_models import load_multires_golden, multires_case
    M = load_multires_golden()
    with oracle_backend():
        model, opt, args, weight = multires_case(M, tag)
        rec = run_steps(opt, (args,), {"weight": weight}, 6)
        assert set(rec["kind"]) == {"graph"}
        compare_trajectory(rec, M, tag, floor=1e-12, rtol=1e-6)
        np.testing.assert_allclose(model.nodes.detach().tensor().numpy(), M[f"{tag}/nodes"], atol=1e-6)
        # and through the matrix-free PCG instead of the dense assembly
        model, opt, args, weight = multires_case(M, tag, solver=pp.optim.solver.PCG(tol=1e-13, maxiter=3000, check_every=1))
        rec = run_steps(opt, (args,), {"weight": weight}, 3)
        np.testing.assert_allclose(rec["loss"][:2], M[f"{tag}/loss"][:2], rtol=1e-6)


@pytest.mark.parametrize("tag", ["plain", "kernel_weights"])
def test_bundle_adjustment_with_prior_residuals_matches_reference(tag):
    """Three residuals (reprojection 2 rows, camera-position prior 3 rows, point prior 3 rows) over three parameters:
    the multi-parameter graph path stacks them (priors share the camera / point slots) and reproduces the reference's
    dense trajectory, rejected steps included; also through the matrix-free PCG."""
    from tests.optim_models import ba_prior_case
    with oracle_backend():
        G, model, opt, args, weight = ba_prior_case(tag)
        rec = run_steps(opt, (args,), {"weight": weight}, 6)
        assert set(rec["kind"]) == {"multigraph"}
        compare_trajectory(rec, G, tag, floor=1e-12, rtol=1e-6)
        np.testing.assert_allclose(model.P.detach().numpy(), G[f"{tag}/P"], atol=1e-6)
        np.testing.assert_allclose(model.C.detach().tensor().numpy(), G[f"{tag}/C"], atol=1e-6)
        G, model, opt, args, weight = ba_prior_case(tag, solver=pp.optim.solver.PCG(tol=1e-14, maxiter=5000, check_every=1))
        rec = run_steps(opt, (args,), {"weight": weight}, 3)
        np.testing.assert_allclose(rec["loss"][:3], G[f"{tag}/loss"][:3], rtol=1e-6)



@pytest.mark.parametrize("tag", ["gn_plain", "gn_priors"])
def test_gauss_newton_on_bundle_adjustment_keeps_the_reference_algorithm(tag):
    """GN on several parameters stays on the dense linearisation (the pseudo-inverse of an ill-conditioned J is not
    something CG reproduces): the reference's recorded, non-monotone GN trajectory, with and without weighted priors."""
    from tests.optim_models import ba_prior_case, Reproj
    with oracle_backend():
        G, model, _, args, _ = ba_prior_case("kernel_weights")
        D = torch.float64
        weight = [torch.eye(2, dtype=D), torch.eye(3, dtype=D) * 25.0, torch.eye(3, dtype=D) * 4.0]
        if tag == "gn_plain":
            model, args, weight = Reproj(model.K.detach().clone(), pp.SE3(model.C.detach().tensor().clone()), model.P.detach().clone()), args[:3], None
        opt = pp.optim.GN(model)
        losses = [float(opt.step(args, weight=weight)) for _ in range(3)]
        np.testing.assert_allclose(losses, G[f"{tag}/loss"], rtol=1e-4)


def test_negative_gather_indices_are_accepted_like_the_dense_path(G):
    """`self.C[cidx]` with -1 for the last camera (valid torch indexing; the reference's dense path accepts it): the
    structured path must take the same steps as with the equivalent non-negative indices."""
    with oracle_backend():
        recs = []
        for negative in (False, True):
            model, opt, (obs, cidx, pidx) = ba_case(G, "ba_small")
            if negative:
                cidx = torch.where(cidx == model.C.shape[0] - 1, torch.full_like(cidx, -1), cidx)
                pidx = torch.where(pidx == model.P.shape[0] - 1, torch.full_like(pidx, -1), pidx)
                assert (cidx < 0).any() and (pidx < 0).any()
            rec = run_steps(opt, ((obs, cidx, pidx),), {}, 4)
            assert set(rec["kind"]) == {"multigraph"}
            recs.append((rec, model.P.detach().clone()))
        assert recs[0][0]["loss"] == pytest.approx(recs[1][0]["loss"], rel=1e-12)
        torch.testing.assert_close(recs[0][1], recs[1][1], rtol=0, atol=1e-12)
