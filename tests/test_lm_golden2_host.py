"""Round-2 parity sizes and the robust kernels / correctors / LSTSQ the first golden file left out, against
tests/golden/lm_golden2.npz (recorded from the real reference by tests/golden/make_lm_golden2.py) -- CPU, oracle
stand-in backend, structured and dense paths."""
import numpy as np
import pytest
import torch

import pypose_amd as pp
from tests.lm_golden2_util import G2, invnet_problem, robust_case, compare2
from tests.optim_models import InvNet, PoseGraph, T, run_steps
from tests.oracle_backend import oracle_backend


@pytest.mark.parametrize("B,strategy", [(64, "constant"), (64, "trustregion"), (1024, "trustregion")])
def test_invnet_block_path_at_survey_sizes(B, strategy):
    G = G2()
    with oracle_backend():
        inp, init = invnet_problem(G, B)
        net = InvNet(init)
        mk = {"constant": lambda: pp.optim.strategy.Constant(damping=1e-4), "trustregion": lambda: pp.optim.strategy.TrustRegion(radius=10.0)}
        opt = pp.optim.LM(net, strategy=mk[strategy]())
        rec = run_steps(opt, (inp,), {}, 5)
        assert set(rec["kind"]) == {"block"}
        compare2(rec, G, f"invnet{B}/{strategy}")
        final = net.pose.detach().tensor().numpy()[::max(1, B // 64)]
        np.testing.assert_allclose(final, G[f"invnet{B}/{strategy}/final"], atol=1e-9)


@pytest.mark.parametrize("tag", ["pgo50", "pgo200"])
def test_pose_graph_at_survey_sizes(tag):
    G = G2()
    with oracle_backend():
        graph = PoseGraph(pp.SE3(T(G[f"{tag}/init"])))
        opt = pp.optim.LM(graph, solver=pp.optim.solver.Cholesky(), strategy=pp.optim.strategy.TrustRegion(radius=1e4), min=1e-6)
        rec = run_steps(opt, ((T(G[f"{tag}/edges"]), pp.SE3(T(G[f"{tag}/poses"]))),), {}, 4)
        assert set(rec["kind"]) == {"graph"}
        compare2(rec, G, f"{tag}/noweight", rtol=1e-7)
        np.testing.assert_allclose(graph.nodes.detach().tensor().numpy(), G[f"{tag}/noweight/final"], atol=1e-7)


@pytest.mark.parametrize("structured", [False, True])
@pytest.mark.parametrize("name", ["pseudohuber", "softlone", "arctan", "tolerant", "triggs_huber", "triggs_cauchy", "lstsq_lm"])
def test_robust_kernels_correctors_and_lstsq(name, structured):
    """PseudoHuber / SoftLOne / Arctan / Tolerant (kernel.py:83-259) under FastTriggs, the Triggs corrector
    (corrector.py:132-167) and the LSTSQ solver (solver.py:71-152): same trajectories as the reference.
    (GN + LSTSQ was recorded too but is not a golden: torch.linalg.lstsq's default 'gelsy' driver returns a different
    vector on every call for the wide rank-deficient [24, 28] Jacobian of this model -- |x| 1.94 / 2.80 / ..., residual 6.6
    where the pseudo-inverse reaches 1e-14 -- so the reference's own trajectory is not reproducible; see test below.)"""
    G = G2()
    with oracle_backend():
        net, opt, inp, n = robust_case(G, name)
        opt.structured = structured
        rec = run_steps(opt, (inp,), {}, n)
        compare2(rec, G, f"robust/{name}", floor=1e-18)
        np.testing.assert_allclose(net.pose.detach().tensor().numpy(), G[f"robust/{name}/final"], atol=1e-8)


def test_lstsq_solver_matches_lapack_on_well_posed_systems():
    """LSTSQ (solver.py:71-152) on full-column-rank systems, where the least-squares solution is unique."""
    g = torch.Generator().manual_seed(0)
    A = torch.randn(3, 12, 5, dtype=torch.float64, generator=g)
    b = torch.randn(3, 12, 2, dtype=torch.float64, generator=g)
    x = pp.optim.solver.LSTSQ()(A, b)
    for k in range(3):
        np.testing.assert_allclose(x[k].numpy(), np.linalg.lstsq(A[k].numpy(), b[k].numpy(), rcond=None)[0], atol=1e-12)
