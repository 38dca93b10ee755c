"""cProfile of IMUPreintegrator.forward's host side (states only, 4096 x 1024): is the module call itself the bottleneck?"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
B, F, dev = 4096, 1024, "cuda"
dt = torch.full((B, F, 1), 0.005, device=dev); gyro = 0.1 * torch.randn(B, F, 3, device=dev)
acc = torch.randn(B, F, 3, device=dev) + torch.tensor([0., 0., 9.81], device=dev)
integ = pp.module.IMUPreintegrator(prop_cov=False, reset=True).to(dev)
for _ in range(10): integ(dt=dt, gyro=gyro, acc=acc)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200): integ(dt=dt, gyro=gyro, acc=acc)
host = (time.perf_counter() - t) / 200 * 1e6
torch.cuda.synchronize()
print("host enqueue us per forward:", host)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): integ(dt=dt, gyro=gyro, acc=acc)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(500): integ(dt=dt, gyro=gyro, acc=acc)
torch.cuda.synchronize()
print("wall us per forward, 500 back to back:", (time.perf_counter() - t) / 500 * 1e6)
