"""cProfile of the host side of one training pair through IMUPreintegrator (forward + torch.autograd.grad, 4096 x 1024): the GPU
side is two kernels, 172 us; is the Python around them longer than that?  Uses a SMALL batch so the GPU never back-pressures."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
B, F, dev = 64, 256, "cuda"
dt = torch.full((B, F, 1), 0.005, device=dev)
gyro = (0.1 * torch.randn(B, F, 3, device=dev)).requires_grad_(True)
acc = (torch.randn(B, F, 3, device=dev) + torch.tensor([0., 0., 9.81], device=dev)).requires_grad_(True)
Wr, Wv, Wp = torch.randn(B, F, 4, device=dev), torch.randn(B, F, 3, device=dev), torch.randn(B, F, 3, device=dev)
integ = pp.module.IMUPreintegrator(prop_cov=False, reset=True).to(dev)


def pair():
    o = integ(dt=dt, gyro=gyro, acc=acc)
    return torch.autograd.grad([o["rot"].tensor(), o["vel"], o["pos"]], [gyro, acc], [Wr, Wv, Wp])


def fwd():
    return integ(dt=dt, gyro=gyro, acc=acc)


for f, name in ((fwd, "forward with grad"), (pair, "forward + autograd.grad")):
    for _ in range(20): f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(300): f()
    torch.cuda.synchronize()
    print(f"{name}: wall us per call (tiny batch: host time) {(time.perf_counter() - t) / 300 * 1e6:.1f}")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): pair()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
