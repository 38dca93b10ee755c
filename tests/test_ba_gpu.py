"""Bundle adjustment on the MI355X: reference trajectories through the HIP Lie kernels, and a BAL-scale synthetic
problem (257 cameras, 65 k points, 225 k observations -- the size of the reference example's default
`problem-257-65132-pre`) through the matrix-free multi-parameter path."""
import numpy as np
import pytest
import torch

import pypose_amd as pp
from tests.optim_models import Reproj, ba_case, compare_trajectory, load_ba_golden, run_steps

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def G():
    return load_ba_golden()


@pytest.mark.parametrize("structured", [False, True, "schur"])
@pytest.mark.parametrize("tag", ["ba_small", "ba_huber"])
def test_ba_trajectory_matches_reference(G, tag, structured, monkeypatch):
    from pypose_amd import _C
    from pypose_amd.optim import multigraph
    assert _C._test_backend is None
    model, opt, args = ba_case(G, tag, DEV)
    if structured == "schur":
        monkeypatch.setattr(multigraph, "DENSE_LIMIT", 0)        # force the Schur-complement solve on the small problem
    opt.structured = bool(structured)
    rec = run_steps(opt, (args,), {}, 6)
    assert set(rec["kind"]) == ({"multigraph"} if structured else {"dense"}), rec["kind"]
    assert (structured == "schur") == bool(opt.__dict__.get("_schur_plans"))
    compare_trajectory(rec, G, tag, floor=1e-12, rtol=1e-7)
    np.testing.assert_allclose(model.P.detach().cpu().numpy(), G[f"{tag}/P"], atol=1e-6)
    np.testing.assert_allclose(model.C.detach().tensor().cpu().numpy(), G[f"{tag}/C"], atol=1e-6)


def synthetic_ba(Nc, Np, per_point, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    P = (torch.randn(Np, 3, generator=g, dtype=dtype) * 0.5).to(DEV)
    base = torch.cat([torch.tensor([[0., 0., -4.]], dtype=dtype).repeat(Nc, 1), pp.identity_SO3(Nc, dtype=dtype).tensor()], -1).to(DEV)
    C = pp.randn_SE3(Nc, sigma=0.15, dtype=dtype, device=DEV) @ pp.SE3(base)
    K = torch.stack([torch.full((Nc,), 500.), torch.full((Nc,), -0.05), torch.full((Nc,), 0.01)], -1).to(dtype).to(DEV)
    cidx = torch.randint(0, Nc, (Np * per_point,), generator=g).to(DEV)
    pidx = torch.arange(Np).repeat_interleave(per_point).to(DEV)
    with torch.no_grad():
        obs = Reproj.project(K[cidx], C[cidx], P[pidx]) + 0.2 * torch.randn(len(cidx), 2, generator=g, dtype=dtype).to(DEV)
    K0 = K * (1 + 0.002 * torch.randn(Nc, 3, generator=g, dtype=dtype).to(DEV))
    C0 = pp.randn_SE3(Nc, sigma=0.005, dtype=dtype, device=DEV) @ C
    P0 = P + 0.02 * torch.randn(Np, 3, generator=g, dtype=dtype).to(DEV)
    return (obs, cidx, pidx), (K0, C0, P0)


@pytest.mark.parametrize("solver", ["pcg", "cholesky"])
def test_ba_bal_scale(solver):
    """PCG on the full matrix-free system, or -- when a direct solver is asked for -- exact elimination of the points
    (Schur complement) and a dense Cholesky of the 2313 x 2313 camera system."""
    Nc, Np = 257, 65_132
    args, (K0, C0, P0) = synthetic_ba(Nc, Np, 4, torch.float64)          # ~260 k observations
    model = Reproj(K0, C0, P0)
    sol = pp.optim.solver.PCG(tol=1e-4, maxiter=250) if solver == "pcg" else pp.optim.solver.Cholesky()
    opt = pp.optim.LM(model, solver=sol, strategy=pp.optim.strategy.TrustRegion(radius=1e4), reject=30)
    l0 = float(opt.model.loss(args, None).detach())
    losses = [float(opt.step(args)) for _ in range(4)]
    assert opt.linearization == "multigraph"
    assert all(b <= a for a, b in zip([l0] + losses, losses)), (l0, losses)
    # noise floor: 0.2 px per coordinate -> sum of squares ~ 0.04 * 2 E; the start is far above it
    E = args[0].shape[0]
    assert losses[-1] < 0.2 * l0 and losses[-1] < 4 * 0.04 * 2 * E, (l0, losses)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 2e-5)])
@pytest.mark.parametrize("weighted", [False, True])
def test_multigraph_hip_products_match_tensor_formulation(dtype, tol, weighted):
    """pplie_mg_jtimes / pplie_mg_jt_segsum / pplie_block_matvec against the device-agnostic formulation of the same
    matrix-free products (which the CPU suite pins to the dense reference algebra)."""
    from pypose_amd.optim import multigraph
    torch.manual_seed(1)
    E, dr = 5003, 2
    N, m = [37, 37, 901], [3, 6, 3]
    params = [torch.zeros(n, w, dtype=dtype, device=DEV) for n, w in zip(N, (3, 6, 3))]
    slots = [(k, torch.randint(0, N[k], (E,), device=DEV), torch.randn(E, dr, m[k], dtype=dtype, device=DEV)) for k in range(3)]
    slots.append((2, torch.randint(0, N[2], (E,), device=DEV), torch.randn(E, dr, 3, dtype=dtype, device=DEV)))   # two slots, one parameter
    W = None
    if weighted:
        A = torch.randn(E, dr, dr, dtype=dtype, device=DEV)
        W = A @ A.mT + torch.eye(dr, dtype=dtype, device=DEV)

    class Opt:
        group = None
    lin = multigraph.MultiGraphLinearization(Opt(), W, torch.randn(E, dr, dtype=dtype, device=DEV), params, slots)
    assert lin.hip_ok()
    tot = sum(n * k for n, k in zip(N, m))
    v, shift = torch.randn(tot, dtype=dtype, device=DEV), torch.rand(tot, dtype=dtype, device=DEV)
    got = lin.matvec_flat(v, shift)
    xs = lin._split(v)
    want = lin._cat([y + sh * x for y, sh, x in zip(lin._Hp(xs), lin._split(shift), xs)])
    assert (got - want).abs().max().item() <= tol * want.abs().max().item()
    Binv = [torch.randn(n, k, k, dtype=dtype, device=DEV) for n, k in zip(N, m)]
    got = lin.precond_flat(v, Binv)
    want = lin._cat([(Bi * x.unsqueeze(-2)).sum(-1) for Bi, x in zip(Binv, xs)])
    assert (got - want).abs().max().item() <= tol * want.abs().max().item()


def test_gauss_newton_graph_path_on_device(G):
    """GN with a PCG solver on the HIP graph path (plain CG through the fused PCG launches): the reference's
    pseudo-inverse steps on the gauge-free 12-node graph (golden recorded from the real reference), then the same
    algorithm at 10^4 nodes, where no dense J exists."""
    from tests.optim_models import PoseGraph, T, load_lm_golden
    from tests.test_optim_gpu import _synthetic_graph
    L = load_lm_golden()
    edges, poses = T(L["pgo12/edges"], DEV), pp.SE3(T(L["pgo12/poses"], DEV))
    graph = PoseGraph(pp.SE3(T(L["pgo12/init"], DEV)))
    opt = pp.optim.GN(graph, solver=pp.optim.solver.PCG(tol=1e-13, maxiter=512, check_every=4))
    losses = [float(opt.step((edges, poses))) for _ in range(3)]
    np.testing.assert_allclose(losses, G["gn_pgo12/loss"], rtol=1e-7)
    np.testing.assert_allclose(graph.nodes.detach().tensor().cpu().numpy(), G["gn_pgo12/final"], atol=1e-7)
    assert opt.__dict__.get("_pcg_workspaces"), "the fused PCG workspace was not used"

    edges, rel, init = _synthetic_graph(10_000, 40_000, torch.float64)
    big = PoseGraph(init)
    opt = pp.optim.GN(big)                      # default PINV solver: too large for a dense J -> graph path, warns once
    with pytest.warns(UserWarning, match="conjugate gradient"):
        first = float(opt.step((edges, rel)))
    start = float(opt.last)
    second = float(opt.step((edges, rel)))
    assert first < 0.05 * start and second <= first * 1.0001


@pytest.mark.parametrize("tag", ["plain", "kernels", "kernels_weights"])
def test_multi_residual_pose_graph_on_device(tag):
    """three residuals (6-row binary edges x 2, 3-row unary priors) stacked with zero padding onto the (6, 6, 2) HIP
    graph kernels: the reference's recorded dense trajectory"""
    from tests.optim_models import load_multires_golden, multires_case
    M = load_multires_golden()
    for solver in (pp.optim.solver.Cholesky(), pp.optim.solver.PCG(tol=1e-13, maxiter=4096, check_every=8)):
        model, opt, args, weight = multires_case(M, tag, DEV, solver=solver)
        rec = run_steps(opt, (args,), {"weight": weight}, 6 if isinstance(solver, pp.optim.solver.Cholesky) else 2)
        assert set(rec["kind"]) == {"graph"}
        if isinstance(solver, pp.optim.solver.Cholesky):
            compare_trajectory(rec, M, tag, floor=1e-12, rtol=1e-6)
            np.testing.assert_allclose(model.nodes.detach().tensor().cpu().numpy(), M[f"{tag}/nodes"], atol=1e-6)
        else:
            np.testing.assert_allclose(rec["loss"][:2], M[f"{tag}/loss"][:2], rtol=1e-6)
            assert opt.__dict__.get("_pcg_workspaces")


@pytest.mark.parametrize("mode", ["dense", "pcg", "schur"])
@pytest.mark.parametrize("tag", ["plain", "kernel_weights"])
def test_bundle_adjustment_with_prior_residuals_on_device(tag, mode, monkeypatch):
    """three residuals over three parameters (priors share the camera / point slots of the reprojection rows) through
    the HIP multi-parameter kernels: dense assembly, matrix-free PCG, and the Schur complement"""
    from pypose_amd.optim import multigraph
    from tests.optim_models import ba_prior_case
    if mode == "schur":
        monkeypatch.setattr(multigraph, "DENSE_LIMIT", 0)
    solver = pp.optim.solver.PCG(tol=1e-14, maxiter=5000, check_every=8) if mode == "pcg" else pp.optim.solver.Cholesky()
    G, model, opt, args, weight = ba_prior_case(tag, DEV, solver=solver)
    rec = run_steps(opt, (args,), {"weight": weight}, 6 if mode != "pcg" else 3)
    assert set(rec["kind"]) == {"multigraph"}
    if mode == "pcg":
        np.testing.assert_allclose(rec["loss"][:3], G[f"{tag}/loss"][:3], rtol=1e-6)
    else:
        compare_trajectory(rec, G, tag, floor=1e-12, rtol=1e-6)
        np.testing.assert_allclose(model.P.detach().cpu().numpy(), G[f"{tag}/P"], atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_three_launch_iteration_equals_the_eleven_launch_one(dtype, monkeypatch):
    """csrc/graph.hip pplie_mg3_*: the fused PCG iteration of the multi-parameter path (work items with the last-arriver
    reduction for camera rows of ~10^3 incidences) against the launch-per-stage formulation: same iteration counts (+- one
    check interval), same LM trajectory (the dot products of both end in per-workgroup atomics: last bits vary from run to run)"""
    from pypose_amd.optim import multigraph
    args, (K0, C0, P0) = synthetic_ba(40, 6000, 5, dtype)
    runs = {}
    for mg3 in (True, False):
        monkeypatch.setattr(multigraph._GraphedPCG, "mg3", mg3, raising=False)
        model = Reproj(K0.clone(), C0.clone(), P0.clone())
        solver = pp.optim.solver.PCG(tol=1e-6 if dtype == torch.float32 else 1e-10, maxiter=2000)
        opt = pp.optim.LM(model, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4), reject=30)
        losses, its = [], []
        for _ in range(3):
            losses.append(float(opt.step(args)))
            its.append(solver.iterations)
        assert opt.linearization == "multigraph"
        runs.setdefault(mg3, []).append((losses, its, model.P.detach().clone()))
    la, ia, pa = runs[True][0]
    lb, ib, pb = runs[False][0]
    np.testing.assert_allclose(la, lb, rtol=1e-4 if dtype == torch.float32 else 1e-9)
    # (beta comes from the recurrence rho - 2 alpha y.z + alpha^2 y.Binv y here, from the reduced r.z there: the counts of these
    #  several-hundred-iteration solves agree to a few check intervals)
    assert all(abs(a - b) <= max(48, 0.35 * b) for a, b in zip(ia, ib)), (ia, ib)
    assert float((pa - pb).abs().max()) <= (1e-3 if dtype == torch.float32 else 1e-7)
