"""Host-side cost of one device-resident LM step (C3): cProfile over the fast path + a split of the enqueue time."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from tests.optim_models import InvNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
torch.manual_seed(0)
init = pp.randn_SE3(B, device="cuda"); inp = pp.randn_SE3(B, device="cuda")
net = InvNet(init.clone())
opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(damping=1e-4), static=("static" in sys.argv))
for _ in range(5):
    opt.step(inp)
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze()
N = 3000
t0 = time.perf_counter()
for _ in range(N):
    opt.step(inp)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"B={B}: host enqueue {(t1 - t0) / N * 1e6:.2f} us/step, with drain {(time.perf_counter() - t0) / N * 1e6:.2f} us/step")
dev = opt._device_lm
t0 = time.perf_counter()
for _ in range(N):
    dev.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"DeviceLM.step() alone: {(t1 - t0) / N * 1e6:.2f} us")
t0 = time.perf_counter()
for _ in range(N):
    dev.try_step(inp, None, None)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"DeviceLM.try_step(): {(t1 - t0) / N * 1e6:.2f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    opt.step(inp)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)

# ---- the default (static=False): the model's Python runs (dry) at every step
net2 = InvNet(init.clone())
opt2 = pp.optim.LM(net2, strategy=pp.optim.strategy.Constant(damping=1e-4))
for _ in range(5):
    opt2.step(inp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    opt2.step(inp)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"DEFAULT (dry trace every step): host enqueue {(t1 - t0) / N * 1e6:.2f} us/step")
from pypose_amd.optim import fused as F
params = [net2.pose]
t0 = time.perf_counter()
for _ in range(N):
    F.dry_program(opt2, params, inp, None)
print(f"  dry_program alone: {(time.perf_counter() - t0) / N * 1e6:.2f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    opt2.step(inp)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
