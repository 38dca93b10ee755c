"""The GENERIC block path (any row-independent residual; here the InvNet model with the fused program switched off) at
1M problems: `python tools/prof_block_generic.py` under rocprofv3 --kernel-trace --stats shows where a step's time goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from tests.optim_models import InvNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = "cuda:0"
torch.manual_seed(0); net = InvNet(pp.randn_SE3(B, device=dev))
torch.manual_seed(1); inp = pp.randn_SE3(B, device=dev)
strat = pp.optim.strategy.TrustRegion(radius=1e4) if os.environ.get('STRAT') == 'tr' else pp.optim.strategy.Constant(damping=1e-4)
opt = pp.optim.LM(net, strategy=strat)
opt.fused = False
init = net.pose.detach().tensor().clone()
opt.step(inp); opt.step(inp)
torch.cuda.synchronize()
ts = []
for rep in range(5):
    with torch.no_grad():
        net.pose.copy_(pp.SE3(init))
    if hasattr(opt, "loss"):
        del opt.loss
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        opt.step(inp)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 3)
print("path", opt.linearization, "ms/step", [round(t * 1e3, 3) for t in ts])
