"""StopOnPlateau.optimize() with the stop rules evaluated on the device (optim/scheduler.py, csrc/lm_common.h ST_PL_*): the same number
of steps, the same counters and the same final loss as the reference's host loop ``while continual(): step`` (scheduler.py:162-203),
which reads a loss back after every step."""
import pytest
import torch

import pypose_amd as pp
from tests.optim_models import InvNet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(device_route, strategy, steps, patience, decreasing, n=4096, sigma=0.3, monkeypatch=None):
    torch.manual_seed(7)
    net = InvNet(pp.randn_SE3(n, sigma=sigma, device=DEV))
    inp = pp.randn_SE3(n, sigma=sigma, device=DEV)
    opt = pp.optim.LM(net, strategy=strategy())
    sch = pp.optim.scheduler.StopOnPlateau(opt, steps=steps, patience=patience, decreasing=decreasing, verbose=False)
    if not device_route:
        sch._optimize_on_device = lambda *a, **k: False
    rets = []
    real = sch._optimize_on_device
    sch._optimize_on_device = lambda *a, **k: (rets.append(real(*a, **k)), rets[-1])[1]
    sch.optimize(inp)
    torch.cuda.synchronize()
    assert any(rets) == device_route, rets          # (the route under test is the one that ran)
    return {"steps": sch.steps, "count": sch.patience_count, "continual": sch.continual(), "loss": float(opt.loss),
            "pose": net.pose.detach().tensor().clone(), "linearization": opt.linearization, "reject": int(opt.reject_count)}


@pytest.mark.parametrize("name,strategy,steps,patience,decreasing", [
    ("patience", lambda: pp.optim.strategy.Constant(damping=1e-4), 20, 2, 1e-3),
    ("max_steps", lambda: pp.optim.strategy.Constant(damping=1e-4), 3, 5, 1e-30),
    ("max_steps_long", lambda: pp.optim.strategy.Adaptive(damping=1e-4), 9, 50, -1.0),
    ("trust_region", lambda: pp.optim.strategy.TrustRegion(radius=1e4), 20, 3, 1e-6),
])
def test_device_stop_rules_equal_the_host_loop(name, strategy, steps, patience, decreasing):
    host = _run(False, strategy, steps, patience, decreasing)
    dev = _run(True, strategy, steps, patience, decreasing)
    assert host["linearization"] == dev["linearization"] == "fused:se3inv"
    assert dev["steps"] == host["steps"] and dev["count"] == host["count"], (name, dev["steps"], host["steps"], dev["count"], host["count"])
    assert dev["continual"] is False and host["continual"] is False
    assert abs(dev["loss"] - host["loss"]) <= 1e-6 * max(abs(host["loss"]), 1e-30) + 1e-30, (dev["loss"], host["loss"])
    torch.testing.assert_close(dev["pose"], host["pose"], rtol=0, atol=1e-6)


def test_steps_behind_the_stop_do_nothing():
    """the device route enqueues every remaining step; those behind the stopping one must not move the parameters"""
    torch.manual_seed(3)
    net = InvNet(pp.randn_SE3(2048, sigma=0.3, device=DEV))
    inp = pp.randn_SE3(2048, sigma=0.3, device=DEV)
    opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(damping=1e-4))
    sch = pp.optim.scheduler.StopOnPlateau(opt, steps=40, patience=2, decreasing=1e-3)
    sch.optimize(inp)
    assert sch.steps < 12 and not sch.continual()
    pose = net.pose.detach().tensor().clone()
    # the plain loop from the same start takes exactly that many steps and lands on the same parameters
    torch.manual_seed(3)
    net2 = InvNet(pp.randn_SE3(2048, sigma=0.3, device=DEV))
    opt2 = pp.optim.LM(net2, strategy=pp.optim.strategy.Constant(damping=1e-4))
    for _ in range(sch.steps):
        opt2.step(inp)
    torch.testing.assert_close(net2.pose.detach().tensor(), pose, rtol=0, atol=1e-6)
    # and the optimizer is usable afterwards: a manual step is a real step again (the stop flag was the scheduler's, not the optimizer's)
    before = float(opt.loss)
    opt.step(inp)
    assert torch.isfinite(opt.loss) and float(opt.loss) <= before * (1 + 1e-3) + 1e-12
