"""configs[0] latency split: forward with grad, backward alone, and the same chain in plain torch ops (the floor of torch's own
autograd machinery for four elementwise kernels + sum)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
dev = "cuda:0"
x = pp.randn_se3(1024, device=dev, requires_grad=True)
t = torch.randn(1024, 6, device=dev, requires_grad=True)


def timeit(f, n=300):
    for _ in range(30): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6


def fwd():
    return x.Exp().Log().tensor().sum()


def fwd_bwd():
    x.grad = None
    x.Exp().Log().tensor().sum().backward()


def torch_chain():
    t.grad = None
    (t.sin().cos()).sum().backward()


def torch_fwd():
    return (t.sin().cos()).sum()


print("ours  fwd (grad recorded)   %.1f us" % timeit(fwd))
print("ours  fwd + bwd             %.1f us" % timeit(fwd_bwd))
print("torch fwd (sin.cos.sum)     %.1f us" % timeit(torch_fwd))
print("torch fwd + bwd             %.1f us" % timeit(torch_chain))
with torch.no_grad():
    print("ours  fwd (no_grad)         %.1f us" % timeit(lambda: x.Exp().Log()))
g = torch.ones(1024, 6, device=dev)
y = x.Exp().Log().tensor()
print("ours  bwd only (retain)     %.1f us" % timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True)))
ty = t.sin().cos()
print("torch bwd only (retain)     %.1f us" % timeit(lambda: torch.autograd.grad(ty, t, g, retain_graph=True)))


# ---- where do fwd + bwd's extra microseconds go? (79.9 us in r03 -> 115-136 us in r05)
def only_grad_none():
    x.grad = None


def bwd_contig():
    x.grad = None
    x.Exp().Log().tensor().backward(g)


def bwd_sum_plain_leaf():
    xt.grad = None
    pp.se3(xt).Exp().Log().tensor().sum().backward()


def fwd_bwd_autograd_grad():
    return torch.autograd.grad(x.Exp().Log().tensor().sum(), x)


xt = x.detach().tensor().clone().requires_grad_(True)
print("x.grad = None alone         %.1f us" % timeit(only_grad_none))
print("fwd + backward(ones)        %.1f us" % timeit(bwd_contig))
print("fwd + sum + bwd, plain leaf %.1f us" % timeit(bwd_sum_plain_leaf))
print("fwd + sum + autograd.grad   %.1f us" % timeit(fwd_bwd_autograd_grad))
print("ours  fwd + bwd (again)     %.1f us" % timeit(fwd_bwd))
import cProfile, pstats, io
pr = cProfile.Profile()
for _ in range(50): fwd_bwd()
torch.cuda.synchronize()
pr.enable()
for _ in range(300): fwd_bwd()
torch.cuda.synchronize()
pr.disable()
sio = io.StringIO()
pstats.Stats(pr, stream=sio).sort_stats("tottime").print_stats(18)
print("\n".join(l[:150] for l in sio.getvalue().splitlines()[:40]))
try:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        for _ in range(100): fwd_bwd()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=60)[:6000])
except Exception as e:
    print("profiler:", repr(e))
