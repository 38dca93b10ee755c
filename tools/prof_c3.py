"""C3 (InvNet SE3, B = 1M) LM steps for rocprofv3 --kernel-trace --stats."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from tests.optim_models import InvNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = "cuda:0"
torch.manual_seed(0); net = InvNet(pp.randn_SE3(B, device=dev))
torch.manual_seed(1); inp = pp.randn_SE3(B, device=dev)
opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(damping=1e-4))
opt.step(inp)
with torch.no_grad():
    net.pose.copy_(pp.randn_SE3(B, device=dev)); del opt.loss
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    opt.step(inp)
torch.cuda.synchronize(); print("s/step", (time.perf_counter() - t0) / 5)
