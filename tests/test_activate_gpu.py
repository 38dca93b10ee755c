"""activate(pypose) on REAL kernels: the reference package itself (shipped to the GPU box as oracle/_ref, see
oracle/Makefile) with its Lie-op Functions rebound to pypose_amd's HIP-backed ones -- device tensors take the kernels,
host tensors keep the reference's own code -- against the un-activated reference on the CPU."""
import numpy as np
import pytest
import torch

from oracle import ref_loader

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_loader.available(), reason="oracle/_ref not shipped")]
DEV = "cuda:0"


@pytest.fixture(scope="module")
def rpp():
    return ref_loader.load()


def _workload(pp, dev, D):
    torch.manual_seed(1)
    x = pp.randn_se3(257, dtype=D).to(dev).requires_grad_(True)
    X = pp.randn_SE3(257, dtype=D).to(dev)
    p = torch.randn(257, 3, dtype=D).to(dev).requires_grad_(True)
    a = pp.randn_se3(257, dtype=D).to(dev)
    Y = x.Exp() * X
    out = (Y.Inv() @ X).Log().tensor().sum() + Y.Act(p).sum() + Y.Adj(a).tensor().sum() + Y.AdjT(a).tensor().sum()
    out.backward()
    S = pp.randn_Sim3(64, dtype=D).to(dev)
    R = pp.randn_so3(64, dtype=D).to(dev)
    return [out.detach(), x.grad.clone(), p.grad.clone(), S.Log().Exp().tensor(), (S * S.Inv()).tensor(), R.Exp().Log().tensor(),
            X.Jinvp(a).tensor()]


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 2e-5)])
def test_activated_reference_on_the_gpu_matches_the_reference_on_the_cpu(rpp, dtype, tol):
    from pypose_amd import _C, activate
    assert _C._test_backend is None
    want = _workload(rpp, "cpu", dtype)                       # the reference, untouched, on the host
    activate.activate(rpp)                                     # device dispatch: cuda + fp32/fp64 -> HIP kernels
    try:
        assert type(rpp.lietensor.lietensor.SE3_Log).__name__ == "_Dispatch"
        launched = []
        real = _C.row_op
        _C.row_op = lambda name, *a, **k: (launched.append(name), real(name, *a, **k))[1]
        from pypose_amd.lietensor import operation as _op
        _op._C.row_op = _C.row_op
        try:
            got = _workload(rpp, DEV, dtype)
        finally:
            _C.row_op = real
            _op._C.row_op = real
        host_again = _workload(rpp, "cpu", dtype)              # host tensors still take the reference's own code
    finally:
        activate.deactivate()
    assert {"se3_exp_fwd", "se3_log_fwd", "se3_mul_fwd", "se3_act_fwd", "sim3_log_fwd", "se3_log_bwd", "se3_exp_bwd"} <= set(launched)
    for g, w in zip(got, want):
        scale = max(1.0, float(w.abs().max()))
        assert float((g.cpu() - w).abs().max()) <= tol * scale, (float((g.cpu() - w).abs().max()), scale)
    for h, w in zip(host_again, want):
        assert torch.equal(h, w)


def test_reference_lm_on_activated_kernels_walks_the_recorded_trajectory(rpp):
    """the reference's OWN LevenbergMarquardt (dense Jacobian through torch.func on its own modjac) with the model's Lie
    ops on HIP kernels: same losses as recorded from the un-activated reference"""
    from pypose_amd import activate
    from tests.optim_models import load_lm_golden
    pp = rpp
    G = load_lm_golden()

    class InvNet(torch.nn.Module):
        def __init__(self, init):
            super().__init__()
            self.pose = pp.Parameter(init)

        def forward(self, input):
            return (self.pose @ input).Log().tensor()

    activate.activate(pp)
    try:
        net = InvNet(pp.SE3(torch.from_numpy(G["invnet/init"].copy()).to(DEV)))
        opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(damping=1e-4))
        assert type(opt).__module__.startswith("pypose.")
        inp = pp.SE3(torch.from_numpy(G["invnet/input"].copy()).to(DEV))
        losses = [float(opt.step(inp)) for _ in range(2)]
    finally:
        activate.deactivate()
    np.testing.assert_allclose(losses, G["invnet/constant/loss"][:2], rtol=1e-6)


def test_reference_call_sites_get_the_device_resident_optimizer(rpp):
    """activate(pypose, optim=True) on the GPU: the reference's `pp.optim.LM(...)` call site runs pypose_amd's optimizer,
    which recognises the program through the REFERENCE's LieTensor types and takes the fused device-resident step."""
    from pypose_amd import activate
    from tests.lm_golden2_util import G2, invnet_problem, compare2
    from tests.optim_models import run_steps
    pp = rpp
    G = G2()
    inp0, init0 = invnet_problem(G, 64, DEV)

    class InvNet(torch.nn.Module):
        def __init__(self, init):
            super().__init__()
            self.pose = pp.Parameter(init)

        def forward(self, input):
            return (self.pose @ input).Log().tensor()

    activate.activate(pp, optim=True)
    try:
        net = InvNet(pp.SE3(init0.tensor().clone()))
        opt = pp.optim.LM(net, strategy=pp.optim.strategy.TrustRegion(radius=10.0))
        rec = run_steps(opt, (pp.SE3(inp0.tensor().clone()),), {}, 5)
    finally:
        activate.deactivate()
    assert set(rec["kind"]) == {"fused:se3inv"}, rec["kind"]
    compare2(rec, G, "invnet64/trustregion")
