"""StopOnPlateau.optimize() on 10^6 InvNet problems: the reference's host loop (a loss read back per step) against the stop rules on the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from tests.optim_models import InvNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
torch.manual_seed(0)
init, inp = pp.randn_SE3(B, device="cuda"), pp.randn_SE3(B, device="cuda")
net = InvNet(init.clone())
opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(damping=1e-4))
pp.optim.freeze_gc()
for route in ("host", "device", "host", "device"):
    ts, steps = [], None
    for rep in range(6):
        net.pose.data.copy_(init.tensor())
        if hasattr(opt, "loss"):
            del opt.loss
        sch = pp.optim.scheduler.StopOnPlateau(opt, steps=12, patience=2, decreasing=1e-3)
        if route == "host":
            sch._optimize_on_device = lambda *a, **k: False
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sch.optimize(inp)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
        steps = sch.steps
    print(f"B={B} {route:6s}: {sorted(ts[1:])[len(ts[1:]) // 2]:8.1f} us per optimize() of {steps} steps")
