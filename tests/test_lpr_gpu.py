"""The normal-form LM kernel (csrc/lm_generic.hip, optim/fused.py LprProgram): any  r = Log(L P^s R) - b  /  (L P^s R).a - b  runs
as a device-resident step.  Trajectories against the generic block path (whose trajectories the other tests pin to the
reference's) and, at small n, against the reference optimizer itself on the CPU."""
import numpy as np
import pytest
import torch
from torch import nn

import pypose_amd as pp
from oracle import ref_loader
from tests.optim_models import run_steps

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D = torch.float64
RAND = {"SE3": "randn_SE3", "SO3": "randn_SO3", "Sim3": "randn_Sim3", "RxSO3": "randn_RxSO3"}


class Prog(nn.Module):
    def __init__(self, P, fn, **consts):
        super().__init__()
        self.pose = P.Parameter(consts.pop("init"))
        self.fn = fn
        self.c = consts

    def forward(self, *args):
        return self.fn(self.pose, self.c, *args)


PROGRAMS = {
    "log_pinv_x": lambda p, c: (p.Inv() @ c["X"]).Log().tensor(),
    "log_a_p_b": lambda p, c: (c["A"] @ p @ c["B"]).Log().tensor(),
    "log_chain": lambda p, c: (c["A"].Inv() @ (p.Inv() @ c["B"]) @ c["C"].Inv()).Log().tensor(),
    "act": lambda p, c, pts: p.Act(pts),
    "act_inv_left": lambda p, c, pts: (c["A"] @ p.Inv()).Act(pts),
}


def _problem(P, group, name, n, dtype, dev, seed=0):
    torch.manual_seed(seed)
    rnd = getattr(P, RAND[group])
    mk = lambda rows, s=0.4: rnd(rows, sigma=s, dtype=dtype).to(dev)
    consts = {"init": mk(n), "X": mk(n), "A": mk(n), "B": mk(n), "C": mk(1)}
    args, target = (), None
    if name.startswith("act"):
        # a well-posed fit: the targets are the points moved by a nearby transform (free targets let the scale of Sim3 / RxSO3
        # run away to 0 or infinity on either path)
        pts = torch.randn(n, 3, dtype=dtype).to(dev)
        with torch.no_grad():
            truth = mk(n)
            moved = PROGRAMS[name](truth, consts, pts)
        args, target = (pts,), (moved + 0.01 * torch.randn(n, 3, dtype=dtype).to(dev)).detach()
        consts["init"] = getattr(P, "randn_" + group.lower())(n, sigma=0.05, dtype=dtype).to(dev).Exp() @ truth
    return consts, args, target


@pytest.mark.parametrize("strategy", ["constant", "trustregion"])
@pytest.mark.parametrize("group", ["SE3", "SO3", "Sim3", "RxSO3"])
@pytest.mark.parametrize("name", list(PROGRAMS))
def test_normal_form_steps_equal_the_block_path(name, group, strategy):
    n = 700
    rec = {}
    for fused in (True, False):
        consts, args, target = _problem(pp, group, name, n, D, DEV)
        net = Prog(pp, PROGRAMS[name], **consts)
        st = pp.optim.strategy.Constant(damping=1e-3) if strategy == "constant" else pp.optim.strategy.TrustRegion(radius=1e3)
        opt = pp.optim.LM(net, strategy=st)
        opt.fused = fused
        inp = args[0] if args else None
        model_in = inp if inp is not None else ()
        r = run_steps(opt, (model_in,), {"target": target}, 4)
        rec[fused] = (r, net.pose.detach().tensor().clone())
    assert set(rec[True][0]["kind"]) == {"fused:lpr"}, rec[True][0]["kind"]
    assert set(rec[False][0]["kind"]) == {"block"}
    a, b = rec[True][0], rec[False][0]
    for k in range(4):
        if b["loss"][k] > 1e-12 * b["loss"][0]:
            assert abs(a["loss"][k] - b["loss"][k]) <= 1e-8 * b["loss"][k], (k, a["loss"], b["loss"])
            assert a["reject"][k] == b["reject"][k] and a["damping"][k] == pytest.approx(b["damping"][k], rel=1e-12)
    assert float((rec[True][1] - rec[False][1]).abs().max()) <= 1e-7


@pytest.mark.skipif(not ref_loader.available(), reason="oracle/_ref not shipped")
@pytest.mark.parametrize("name", ["log_pinv_x", "act_inv_left"])
def test_normal_form_steps_equal_the_reference_optimizer(name):
    rpp = ref_loader.load()
    n = 6
    consts, args, target = _problem(rpp, "SE3", name, n, D, "cpu", seed=3)
    ref_net = Prog(rpp, PROGRAMS[name], **{k: v.clone() for k, v in consts.items()})
    ref_opt = rpp.optim.LM(ref_net, strategy=rpp.optim.strategy.TrustRegion(radius=1e3))
    want = [float(ref_opt.step(args[0] if args else (), target=target)) for _ in range(4)]
    mine = {k: pp.SE3(v.tensor().to(DEV)) for k, v in consts.items()}
    net = Prog(pp, PROGRAMS[name], **mine)
    opt = pp.optim.LM(net, strategy=pp.optim.strategy.TrustRegion(radius=1e3))
    got = [float(opt.step(args[0].to(DEV) if args else (), target=None if target is None else target.to(DEV))) for _ in range(4)]
    assert opt.linearization == "fused:lpr"
    for a, b in zip(got, want):
        assert abs(a - b) <= 1e-7 * b + 1e-20, (got, want)
    np.testing.assert_allclose(net.pose.detach().tensor().cpu().numpy(), ref_net.pose.detach().tensor().numpy(), atol=1e-8)


def test_normal_form_fp32_million_problems_and_constant_edits():
    """fp32 at 10^6 problems: converges; editing a constant in place is seen by the next step (the folds are redone)"""
    torch.manual_seed(1)
    n = 1_000_000
    X = pp.randn_SE3(n, device=DEV)
    net = Prog(pp, PROGRAMS["log_pinv_x"], init=pp.randn_SE3(n, device=DEV), X=X)
    opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(damping=1e-4))
    l0 = float(net().square().sum())
    losses = [float(opt.step(())) for _ in range(3)]
    assert opt.linearization == "fused:lpr" and losses[-1] < 1e-6 * l0, (l0, losses)
    resid = (net.pose.detach().Inv() @ X).Log().tensor().abs().max().item()
    assert resid < 1e-3
    X.tensor().copy_(pp.randn_SE3(n, device=DEV).tensor())                   # same storage, new values
    del opt.loss
    l1 = float(opt.step(()))
    assert l1 < 1e-2 * float((net.pose.detach().Inv() @ X).Log().tensor().square().sum()) + 1.0
    for _ in range(3):
        opt.step(())
    assert (net.pose.detach().Inv() @ X).Log().tensor().abs().max().item() < 1e-3
