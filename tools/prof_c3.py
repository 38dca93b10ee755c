"""C3 (B independent SE3 inversion problems, BASELINE configs[2]) LM steps: `python tools/prof_c3.py B [host]`.
Run under rocprofv3 --kernel-trace --stats for the kernel breakdown; `host` prints a cProfile of the step."""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from tests.optim_models import InvNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
torch.manual_seed(0)
net = InvNet(pp.randn_SE3(B, device="cuda"))
inp = pp.randn_SE3(B, device="cuda")
init = net.pose.detach().clone()
opt = pp.optim.LM(net, strategy=pp.optim.strategy.Adaptive(damping=1e-6))
if "block" in sys.argv:
    opt.fused = False
opt.step(inp)
def run(k):
    net.pose.data.copy_(init)
    if hasattr(opt, "loss"):
        del opt.loss
    for _ in range(k):
        opt.step(inp)
run(3)
torch.cuda.synchronize()
pr = cProfile.Profile() if "host" in sys.argv else None
t0 = time.perf_counter()
if pr:
    pr.enable()
for _ in range(10):
    run(3)
torch.cuda.synchronize()
if pr:
    pr.disable()
print(opt.linearization, B, "ms/step", (time.perf_counter() - t0) / 30 * 1e3, "loss", float(opt.loss))
if pr:
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40)
    print(s.getvalue()[:8000])
