"""Round-2 parity sizes on the MI355X (tests/golden/lm_golden2.npz, recorded from the real reference):
InvNet LM at B = 64 / 1024 through the device-resident step and the block path, pose graphs at N = 50 / 200 through the
fused program (direct and PCG solve), robust kernels / correctors / LSTSQ; and the fp32 runs of the B = 1024 / N = 200
problems against the reference's fp64 trajectories ("LM-step numerics within 1e-5 of reference")."""
import numpy as np
import pytest
import torch

import pypose_amd as pp
from tests.lm_golden2_util import G2, invnet_problem, robust_case, compare2
from tests.optim_models import InvNet, PoseGraph, T, run_steps

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MK = {"constant": lambda: pp.optim.strategy.Constant(damping=1e-4), "trustregion": lambda: pp.optim.strategy.TrustRegion(radius=10.0)}


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("B,strategy", [(64, "constant"), (64, "trustregion"), (1024, "constant"), (1024, "trustregion")])
def test_invnet_fp64_trajectories(B, strategy, fused):
    G = G2()
    inp, init = invnet_problem(G, B, DEV)
    net = InvNet(init)
    opt = pp.optim.LM(net, strategy=MK[strategy]())
    opt.fused = fused
    rec = run_steps(opt, (inp,), {}, 5)
    assert set(rec["kind"]) == {"fused:se3inv" if fused else "block"}
    compare2(rec, G, f"invnet{B}/{strategy}")
    final = net.pose.detach().tensor().cpu().numpy()[::max(1, B // 64)]
    np.testing.assert_allclose(final, G[f"invnet{B}/{strategy}/final"], atol=1e-9)


@pytest.mark.parametrize("B", [64, 1024])
def test_invnet_fp32_against_the_reference_fp64_trajectory(B):
    """fp32 device-resident LM against the REFERENCE's recorded fp64 trajectory: every loss within 1e-5 relative plus the
    rounding of the residuals themselves (each of the 6 B residual components carries a few fp32 ulps of the O(1) poses,
    delta = 4e-7, so |d loss| <= 2 sqrt(loss) sqrt(6 B) delta by Cauchy-Schwarz), the same damping sequence, and final
    poses within 2e-5 of the reference's."""
    G = G2()
    inp, init = invnet_problem(G, B, DEV, torch.float32)
    net = InvNet(init)
    opt = pp.optim.LM(net, strategy=MK["trustregion"]())
    rec = run_steps(opt, (inp,), {}, 5)
    assert set(rec["kind"]) == {"fused:se3inv"}
    ref = G[f"invnet{B}/trustregion/loss"]
    floor = B * 1e-11
    checked = 0
    for k, (a, b) in enumerate(zip(rec["loss"], ref)):
        if b > floor:
            assert abs(a - b) <= 1e-5 * b + 2 * (b * 6 * B) ** 0.5 * 4e-7, (k, a, b)
            assert rec["damping"][k] == pytest.approx(G[f"invnet{B}/trustregion/damping"][k], rel=1e-6), k
            checked += 1
    assert checked >= 3
    final = net.pose.detach().tensor().double().cpu().numpy()[::max(1, B // 64)]
    np.testing.assert_allclose(final, G[f"invnet{B}/trustregion/final"], atol=2e-5)


@pytest.mark.parametrize("solver", ["cholesky", "pcg"])
@pytest.mark.parametrize("tag", ["pgo50", "pgo200"])
def test_pose_graph_fp64_trajectories(tag, solver):
    G = G2()
    graph = PoseGraph(pp.SE3(T(G[f"{tag}/init"], DEV)))
    sv = pp.optim.solver.Cholesky() if solver == "cholesky" else pp.optim.solver.PCG(tol=1e-13, maxiter=4000)
    opt = pp.optim.LM(graph, solver=sv, strategy=pp.optim.strategy.TrustRegion(radius=1e4), min=1e-6)
    rec = run_steps(opt, ((T(G[f"{tag}/edges"], DEV), pp.SE3(T(G[f"{tag}/poses"], DEV))),), {}, 4)
    assert set(rec["kind"]) == {"fused:pgo"}
    compare2(rec, G, f"{tag}/noweight", rtol=1e-7)
    np.testing.assert_allclose(graph.nodes.detach().tensor().cpu().numpy(), G[f"{tag}/noweight/final"], atol=1e-6)


def test_pose_graph_fp32_against_the_reference_fp64_trajectory():
    """N = 200 / E = 560 in fp32 (fused program, persistent PCG) against the reference's fp64 losses: 1e-5 relative."""
    G = G2()
    tag = "pgo200"
    graph = PoseGraph(pp.SE3(T(G[f"{tag}/init"], DEV).float()))
    opt = pp.optim.LM(graph, solver=pp.optim.solver.PCG(tol=1e-7, maxiter=4000), strategy=pp.optim.strategy.TrustRegion(radius=1e4), min=1e-6)
    rec = run_steps(opt, ((T(G[f"{tag}/edges"], DEV), pp.SE3(T(G[f"{tag}/poses"], DEV).float())),), {}, 3)
    assert set(rec["kind"]) == {"fused:pgo"}
    ref = G[f"{tag}/noweight/loss"]
    for k, (a, b) in enumerate(zip(rec["loss"], ref)):
        assert abs(a - b) <= 1e-5 * b, (rec["loss"], ref)
        # the accept / reject decision of a step compares two losses: it is the reference's wherever the reference's own
        # step moved the loss by more than fp32 can resolve (the third step here changes it by 2e-7 of its value)
        if k == 0 or abs(ref[k - 1] - b) > 1e-5 * b:
            assert rec["damping"][k] == pytest.approx(G[f"{tag}/noweight/damping"][k], rel=1e-6), (k, rec["damping"])


@pytest.mark.parametrize("structured", [False, True])
@pytest.mark.parametrize("name", ["pseudohuber", "softlone", "arctan", "tolerant", "triggs_huber", "triggs_cauchy", "lstsq_lm"])
def test_robust_kernels_correctors_and_lstsq_on_device(name, structured):
    G = G2()
    net, opt, inp, n = robust_case(G, name, DEV)
    opt.structured = structured
    rec = run_steps(opt, (inp,), {}, n)
    # SoftLOne / PseudoHuber are 2 (d sqrt(1/d^2 + x) - 1)-shaped: at a loss of 1e-10 the reference's own fp64 value carries ~1e-16
    # of cancellation noise per row (2e-6 of the loss), and so does anything that follows the same formula in another op order
    compare2(rec, G, f"robust/{name}", floor=1e-18, atol=2e-15 if name in ("softlone", "pseudohuber") else 0.0)
    np.testing.assert_allclose(net.pose.detach().tensor().cpu().numpy(), G[f"robust/{name}/final"], atol=1e-8)
