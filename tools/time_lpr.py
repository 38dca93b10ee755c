"""LM steps/s of the normal-form kernel (csrc/lm_generic.hip) at 10^6 problems beside the hand-derived InvNet program and the
generic block path:   python tools/time_lpr.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from torch import nn

dev = "cuda:0"
n = 1_000_000


class Net(nn.Module):
    def __init__(self, init, fn, **c):
        super().__init__()
        self.pose = pp.Parameter(init)
        self.fn, self.c = fn, c

    def forward(self, *a):
        return self.fn(self.pose, self.c, *a)


def rate(make, args, target, fused=True, static=False, steps=3, reps=30):
    net, init = make()
    opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(damping=1e-4), static=static)
    opt.fused = fused

    def run(k):
        for _ in range(k):
            net.pose.data.copy_(init.tensor())
            if hasattr(opt, "loss"):
                del opt.loss
            for _ in range(steps):
                loss = opt.step(*args, target=target)
        return loss
    run(2)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        loss = run(reps)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (reps * steps))
    return {"steps_per_s": round(1 / best, 1), "ms_per_step": round(best * 1e3, 4), "path": opt.linearization, "final_loss": float(loss)}


out = {}
for group, rnd in (("SE3", pp.randn_SE3), ("SO3", pp.randn_SO3), ("Sim3", pp.randn_Sim3)):
    torch.manual_seed(0)
    init, X, A = rnd(n, device=dev), rnd(n, device=dev), rnd(n, device=dev)
    pts = torch.randn(n, 3, device=dev)
    with torch.no_grad():                          # targets = the points under a transform NEAR the start (a solvable problem)
        tgt = (rnd(n, sigma=0.05, device=dev) @ init).Act(pts).contiguous()
    progs = {"Log(P X)": (lambda p, c: (p @ c["X"]).Log().tensor(), (X,), None, ()),
             "Log(P^-1 X)": (lambda p, c: (p.Inv() @ c["X"]).Log().tensor(), (X,), None, ()),
             "Log(A P^-1 X)": (lambda p, c: (c["A"] @ p.Inv() @ c["X"]).Log().tensor(), (X,), None, ()),
             "P.Act(a) - b": (lambda p, c, a: p.Act(a), (), tgt, (pts,))}
    for name, (fn, _, target, args) in progs.items():
        make = lambda fn=fn: (Net(init.clone(), fn, X=X, A=A), init)
        key = f"{group} {name}"
        out[key] = {"default": rate(make, args if args else ((),), target), "static": rate(make, args if args else ((),), target, static=True)}
        if group == "SE3":
            out[key]["block_path"] = rate(make, args if args else ((),), target, fused=False, reps=4)
        print(key, json.dumps(out[key]), flush=True)
print(json.dumps(out))
