"""configs[0] fwd + bwd latency is bimodal (131 us in bench.py, 76 us later in tools/time_c1_parts.py, same code): which history flips it?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
dev = "cuda:0"
x = pp.randn_se3(1024, device=dev, requires_grad=True)


def timeit(f, n=300):
    for _ in range(30): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6


def fwd_bwd():
    x.grad = None
    x.Exp().Log().tensor().sum().backward()


def fwd():
    return x.Exp().Log().tensor().sum()


def grad_api():
    return torch.autograd.grad(x.Exp().Log().tensor().sum(), x)


g = torch.ones(1024, 6, device=dev)
print("fresh process      ", [round(timeit(fwd_bwd), 1) for _ in range(4)])
print("after fwd only     ", round(timeit(fwd), 1), [round(timeit(fwd_bwd), 1) for _ in range(2)])
print("after autograd.grad", round(timeit(grad_api), 1), [round(timeit(fwd_bwd), 1) for _ in range(2)])
y = x.Exp().Log().tensor()
print("after retain bwd   ", round(timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True)), 1), [round(timeit(fwd_bwd), 1) for _ in range(2)])
time.sleep(1.0)
print("after 1 s idle     ", [round(timeit(fwd_bwd), 1) for _ in range(3)])
torch.cuda.synchronize()
# per-iteration host time without waiting for the GPU (is the loop host- or device-bound?)
t0 = time.perf_counter()
for _ in range(300): fwd_bwd()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.1f us / iteration, drain after the loop %.1f us total" % ((t1 - t0) / 300 * 1e6, (t2 - t1) * 1e6))
import threading
print("threads:", threading.active_count(), "torch threads", torch.get_num_threads())
torch.set_num_threads(1)
print("1 intra-op thread  ", [round(timeit(fwd_bwd), 1) for _ in range(2)])
