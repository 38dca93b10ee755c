"""The reference's LM convergence scenarios (tests/optim/test_optimizer.py:48-359), re-stated against
pypose_amd: every configuration must reach loss < 1e-5 in fewer than 9 iterations.  Run on the CPU with
the oracle stand-in backend (host logic) and on the MI355X through the HIP kernels."""
import pytest
import torch
from torch import nn

import pypose_amd as pp
from tests.oracle_backend import oracle_backend

ppok, ppos, ppost = pp.optim.kernel, pp.optim.solver, pp.optim.strategy


class PoseInv(nn.Module):
    def __init__(self, *dim, algebra=False):
        super().__init__()
        self.algebra = algebra
        self.pose = pp.Parameter(pp.randn_se3(*dim) if algebra else pp.randn_SE3(*dim))

    def forward(self, inputs):
        pose = self.pose.Exp() if self.algebra else self.pose
        return (pose @ inputs).Log().tensor()


class TwoPoses(nn.Module):
    def __init__(self, *dim):
        super().__init__()
        self.pose1 = pp.Parameter(pp.randn_SE3(*dim))
        self.pose2 = pp.Parameter(pp.randn_SE3(*dim))

    def forward(self, inputs):
        return (self.pose1 @ inputs).Log().tensor() + (self.pose2 @ inputs).Log().tensor()


class TwoResiduals(nn.Module):
    def __init__(self, *dim):
        super().__init__()
        self.pose = pp.Parameter(pp.randn_SE3(*dim))

    def forward(self, poses):
        return (self.pose @ poses).Log().tensor(), self.pose.Log().tensor().sum(-1, keepdim=True)


def converge(opt, step_kwargs, n=10):
    for idx in range(n):
        loss = opt.step(**step_kwargs)
        if loss < 1e-5:
            break
    assert idx < 9, f"Optimization requires too many steps (loss {float(loss)})"
    return idx


def scenarios(device):
    torch.manual_seed(0)
    S = {}
    inputs = pp.randn_SE3(2, 2).to(device)
    S["lie_algebra"] = lambda: (pp.optim.LM(PoseInv(2, 2, algebra=True).to(device), solver=ppos.Cholesky(),
                                            strategy=ppost.Adaptive(damping=1e-6)), dict(input=inputs), "block")
    S["group_weight"] = lambda: (pp.optim.LM(PoseInv(2, 2).to(device), strategy=ppost.TrustRegion(radius=1e6)),
                                 dict(input=inputs, weight=torch.eye(6, device=device)), "block")
    S["cauchy_fasttriggs"] = lambda: (pp.optim.LM(PoseInv(2, 2).to(device), solver=ppos.PINV(), strategy=ppost.Adaptive(damping=1e-6),
                                                  kernel=ppok.Cauchy(), corrector=pp.optim.corrector.FastTriggs(ppok.Cauchy())),
                                      dict(input=inputs), "block")
    S["constant"] = lambda: (pp.optim.LM(PoseInv(2, 2).to(device), strategy=ppost.Constant(damping=1e-6)), dict(input=inputs), "block")
    S["gauss_newton"] = lambda: (pp.optim.GN(PoseInv(2, 2).to(device), solver=ppos.LSTSQ()), dict(input=inputs), None)
    S["multiparameter"] = lambda: (pp.optim.LM(TwoPoses(2, 2).to(device), strategy=ppost.TrustRegion(radius=1e6)), dict(input=inputs), "block")
    wide = pp.randn_SE3(3, 2, 2, 2, sigma=0.0001).to(device)
    S["anybatch_broadcast"] = lambda: (pp.optim.LM(TwoPoses(2, 2).to(device), strategy=ppost.TrustRegion(radius=1e6)), dict(input=wide), "dense")
    S["multi_residual_list_kernels"] = lambda: (
        pp.optim.LM(TwoResiduals(2, 2).to(device), strategy=ppost.TrustRegion(radius=1e6), kernel=[ppok.Huber().to(device), ppok.Scale().to(device)]),
        dict(input={'poses': wide}, weight=[torch.eye(6, device=device), torch.ones(1, device=device)]), "dense")
    S["batch_weight"] = lambda: (
        pp.optim.LM(TwoResiduals(2, 2).to(device), strategy=ppost.TrustRegion(radius=1e6), kernel=[ppok.Huber().to(device), ppok.Scale().to(device)]),
        dict(input={'poses': wide}, weight=[torch.eye(6, device=device).unsqueeze(0).repeat(2, 1, 1), torch.ones(1, device=device).unsqueeze(0)]), "dense")
    return S


NAMES = ["lie_algebra", "group_weight", "cauchy_fasttriggs", "constant", "gauss_newton", "multiparameter", "anybatch_broadcast",
         "multi_residual_list_kernels", "batch_weight"]


@pytest.mark.parametrize("name", NAMES)
def test_scenario_cpu_host_logic(name):
    with oracle_backend():
        opt, kw, kind = scenarios("cpu")[name]()
        converge(opt, kw)
        if kind is not None:
            assert opt.linearization == kind


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_scenario_gpu(name):
    opt, kw, kind = scenarios("cuda:0")[name]()
    converge(opt, kw)
    if kind is not None:
        # plain Log(P @ X) with a Trivial kernel is a recognised program on the GPU (optim/fused.py)
        assert opt.linearization == ("fused:se3inv" if name == "constant" else kind)


@pytest.mark.gpu
def test_scheduler_on_gpu():
    torch.manual_seed(0)
    net = PoseInv(2, 2).to("cuda:0")
    opt = pp.optim.LM(net, strategy=ppost.Constant(damping=1e-4))
    sch = pp.optim.scheduler.StopOnPlateau(opt, steps=10, patience=3, decreasing=1e-3, verbose=False)
    sch.optimize(input=pp.randn_SE3(2, 2).to("cuda:0"))
    assert float(opt.loss) < 1e-6 and sch.steps <= 10
