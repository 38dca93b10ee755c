"""host cost of one device-resident LM step's launch: the C entry (two kernel launches) against a hipGraph replay of it"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from pypose_amd import _C
from tests.optim_models import InvNet
B = 1000
torch.manual_seed(0)
net = InvNet(pp.randn_SE3(B, device="cuda"))
inp = pp.randn_SE3(B, device="cuda")
opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(damping=1e-4), static=True)
for _ in range(5):
    opt.step(inp)
torch.cuda.synchronize()
dev = opt._device_lm
save, partials, state, sync = dev.ptrs
k, lp, la = dev._slot()
st = _C.stream_ptr(dev.device)
N = 3000


def direct():
    dev._launch(save, partials, state[0], state[1], sync, lp, la, st)


for _ in range(100): direct()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): direct()
t1 = time.perf_counter(); torch.cuda.synchronize()
print("direct C entry (2 launches): %.2f us host, %.2f us with drain" % ((t1 - t0) / N * 1e6, (time.perf_counter() - t0) / N * 1e6))
g = torch.cuda.CUDAGraph()
with _C.graph_capture(g):
    dev._launch(save, partials, state[0], state[1], sync, lp, la, _C.stream_ptr(dev.device))
for _ in range(100): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): g.replay()
t1 = time.perf_counter(); torch.cuda.synchronize()
print("hipGraph replay of the same:  %.2f us host, %.2f us with drain" % ((t1 - t0) / N * 1e6, (time.perf_counter() - t0) / N * 1e6))
t0 = time.perf_counter()
for _ in range(N): opt.step(inp)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("opt.step static:              %.2f us host" % ((t1 - t0) / N * 1e6))
