"""Why is bench.py's lm_invnet default leg slower than tools/time_lpr.py's?  Runs the leg, then profiles its default steps."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
r = bench.invnet_lm_rate(dev, reps=40)
print({k: r[k] for k in ("value", "static_model_value", "path")})
import pypose_amd as pp
from tests.optim_models import InvNet
torch.manual_seed(0)
B = 1_000_000
init, inp = pp.randn_SE3(B, device=dev), pp.randn_SE3(B, device=dev)
net = InvNet(init.clone())
opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(damping=1e-4))
for _ in range(6):
    opt.step(inp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    opt.step(inp)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("enqueue us/step", (t1 - t0) / 300 * 1e6, "with drain", (time.perf_counter() - t0) / 300 * 1e6, opt.linearization)
pr = cProfile.Profile(); pr.enable()
for _ in range(300):
    net.pose.data.copy_(init.tensor())
    if hasattr(opt, "loss"):
        del opt.loss
    for _ in range(3):
        opt.step(inp)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumtime").print_stats(25)
