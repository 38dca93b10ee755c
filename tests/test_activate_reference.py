"""The activation shim (pypose_amd.activate) on top of the REAL reference, when it is present
(build container only: /root/reference does not exist on the GPU box).  With the oracle stand-in
backend and force=True every Lie op of ``pypose`` runs through pypose_amd's Functions; results
and gradients must coincide with the reference's own, and the reference's LM must walk the same
trajectory."""
import os
import sys

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "pypose")), reason="reference not present")


@pytest.fixture(scope="module")
def ref_pp():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    try:
        import pypose
        yield pypose
    finally:
        sys.path.remove(REF)


def test_activated_reference_matches_itself(ref_pp):
    from pypose_amd import activate
    from tests.oracle_backend import oracle_backend
    pp = ref_pp
    torch.manual_seed(0)
    D = torch.float64

    def workload():
        torch.manual_seed(1)
        x = pp.randn_se3(5, dtype=D, requires_grad=True)
        X = pp.randn_SE3(5, dtype=D)
        p = torch.randn(5, 3, dtype=D, requires_grad=True)
        a = pp.randn_se3(5, dtype=D)
        Y = x.Exp() * X
        out = (Y.Inv() @ X).Log().tensor().sum() + Y.Act(p).sum() + Y.Adj(a).tensor().sum() + Y.AdjT(a).tensor().sum()
        out.backward()
        S = pp.randn_Sim3(4, dtype=D)
        return [out.detach(), x.grad.clone(), p.grad.clone(), S.Log().Exp().tensor(), (S * S.Inv()).tensor()]

    want = workload()
    with oracle_backend():
        activate.activate(pp, force=True)
        try:
            got = workload()
            assert type(pp.lietensor.lietensor.SE3_Log).__name__ == "_Dispatch"
        finally:
            activate.deactivate()
    assert pp.lietensor.lietensor.SE3_Log.__name__ == "SE3_Log"
    for g, w in zip(got, want):
        torch.testing.assert_close(g, w, rtol=1e-9, atol=1e-11)


def test_reference_lm_runs_on_activated_ops(ref_pp):
    from pypose_amd import activate
    from tests.optim_models import load_lm_golden
    from tests.oracle_backend import oracle_backend
    pp = ref_pp
    G = load_lm_golden()

    class InvNet(torch.nn.Module):
        def __init__(self, init):
            super().__init__()
            self.pose = pp.Parameter(init)

        def forward(self, input):
            return (self.pose @ input).Log().tensor()

    with oracle_backend():
        activate.activate(pp, force=True)
        try:
            net = InvNet(pp.SE3(torch.from_numpy(G["invnet/init"].copy())))
            opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(damping=1e-4))     # the reference's own LM
            inp = pp.SE3(torch.from_numpy(G["invnet/input"].copy()))
            losses = [float(opt.step(inp)) for _ in range(2)]
        finally:
            activate.deactivate()
    np.testing.assert_allclose(losses, G["invnet/constant/loss"][:2], rtol=1e-6)


def test_rank4_callers_run_unmodified_on_activated_ops(ref_pp):
    """the reference's own bspline / geodesic_loss / ape / rpe are pure
    compositions of hot-path ops -- with the shim active they run on pypose_amd's Functions unmodified."""
    from pypose_amd import activate
    from tests.oracle_backend import oracle_backend
    pp = ref_pp
    D = torch.float64
    torch.manual_seed(4)
    poses = pp.randn_SE3(2, 6, dtype=D, requires_grad=True)
    est = pp.randn_SE3(12, dtype=D)
    gt = est @ pp.randn_SE3(12, sigma=0.05, dtype=D)
    stamps = torch.arange(12, dtype=D)

    def workload():
        wpts = pp.bspline(poses, interval=0.25)
        loss = pp.geodesic_loss(wpts, pp.identity_SE3(*wpts.lshape, dtype=D), reduction='sum')
        (g,) = torch.autograd.grad(loss, poses)
        ape = pp.metric.ape(stamps, gt, stamps, est)
        rpe = pp.metric.rpe(stamps, gt, stamps, est)
        flat = lambda d: torch.stack([torch.as_tensor(v, dtype=D).reshape(-1)[0] for v in d.values()])
        return [wpts.tensor().detach(), loss.detach(), g, flat(ape), flat(rpe)]

    want = workload()
    with oracle_backend():
        activate.activate(pp, force=True)
        try:
            got = workload()
        finally:
            activate.deactivate()
    for g, w in zip(got, want):
        torch.testing.assert_close(g, w, rtol=1e-8, atol=1e-10)


def test_reference_code_gets_the_structured_optimizers(ref_pp):
    """activate(pypose, optim=True): the reference's own `pp.optim.LM(...)` call sites -- its PoseGraph example model
    built from `pypose.LieTensor` / `pypose.Parameter`, its solver and strategy objects -- run on pypose_amd's optimizer,
    take the pose-graph / block linearisations and walk the trajectory recorded from the un-activated reference."""
    from pypose_amd import activate
    from tests.optim_models import load_lm_golden
    from tests.oracle_backend import oracle_backend
    pp = ref_pp
    G = load_lm_golden()
    D = torch.float64

    class PoseGraph(torch.nn.Module):                      # examples/module/pgo/pgo.py:15-25, with the REFERENCE's types
        def __init__(self, nodes):
            super().__init__()
            self.nodes = pp.Parameter(nodes)

        def forward(self, edges, poses):
            node1 = self.nodes[edges[..., 0]]
            node2 = self.nodes[edges[..., 1]]
            error = poses.Inv() @ node1.Inv() @ node2
            return error.Log().tensor()

    class InvNet(torch.nn.Module):
        def __init__(self, init):
            super().__init__()
            self.pose = pp.Parameter(init)

        def forward(self, input):
            return (self.pose @ input).Log().tensor()

    with oracle_backend():
        activate.activate(pp, force=True, optim=True)
        try:
            assert pp.optim.LM.__module__.startswith("pypose_amd")
            edges = torch.from_numpy(G["pgo40/edges"])
            poses = pp.SE3(torch.from_numpy(G["pgo40/poses"]))
            graph = PoseGraph(pp.SE3(torch.from_numpy(G["pgo40/init"])))
            opt = pp.optim.LM(graph, solver=pp.optim.solver.Cholesky(), strategy=pp.optim.strategy.TrustRegion(radius=1e4), min=1e-6)
            losses = [float(opt.step((edges, poses), weight=torch.from_numpy(G["pgo40/infos"]))) for _ in range(4)]
            assert opt.linearization == "graph"
            np.testing.assert_allclose(losses, G["pgo40/infos/loss"][:4], rtol=1e-7)
            net = InvNet(pp.SE3(torch.from_numpy(G["invnet/init"])))
            opt = pp.optim.LM(net, strategy=pp.optim.strategy.Adaptive(damping=1e-6))
            inp = pp.SE3(torch.from_numpy(G["invnet/input"]))
            losses = [float(opt.step(inp)) for _ in range(3)]
            assert opt.linearization == "block"
            np.testing.assert_allclose(losses, G["invnet/adaptive/loss"][:3], rtol=1e-6, atol=1e-16)
        finally:
            activate.deactivate()
    assert pp.optim.LM.__module__.startswith("pypose.")
