"""Host-side segments of the captured pose-graph LM step (10 k / 40 k): where the ~45 us between one trial's last kernel and the next
trial's first go.  Wraps the step's stages with perf_counter accumulators:  python tools/time_pgo_host.py [static]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pypose_amd as pp
from pypose_amd.optim import pgograph, fused

static = len(sys.argv) > 1 and sys.argv[1] == "1"
dev = torch.device("cuda:0")
e, rel, init = bench._pose_graph_problem(dev, 10_000, 40_000)
graph = bench._pose_graph_model(init.clone())
solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250)
opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4), static=static)
pp.optim.freeze_gc()
acc, marks = {}, {}


def wrap(owner, name, label):
    real = getattr(owner, name)

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return real(*a, **k)
        finally:
            t1 = time.perf_counter()
            s = acc.setdefault(label, [0.0, 0])
            s[0] += t1 - t0
            s[1] += 1
            marks[label] = (t0, t1)
    setattr(owner, name, timed)


wrap(pgograph.PgoGraphStep, "usable", "usable")
wrap(pgograph.PgoGraphStep, "launch", "launch")
wrap(pgograph.PgoGraphStep, "finish", "finish")
wrap(pgograph.TrialTail, "wait", "wait (inside finish)")
wrap(fused, "dry_program", "dry_program")
wrap(torch.cuda.CUDAGraph, "replay", "replay (inside launch)")
gaps = {"wait-return -> step return": [0.0, 0], "step entry -> replay entered": [0.0, 0], "replay returned -> wait entered": [0.0, 0],
        "wait-return -> next replay returned": [0.0, 0]}
for rep in range(12):
    graph.nodes.data.copy_(init.tensor())
    if hasattr(opt, "loss"):
        del opt.loss
    opt.param_groups[0].update(opt.strategy.defaults)
    torch.cuda.synchronize()
    if rep == 3:
        acc.clear()
        for g in gaps.values():
            g[0] = g[1] = 0
    prev_wait_end = None
    t_rep = time.perf_counter()
    for _ in range(3):
        marks.clear()
        t0 = time.perf_counter()
        opt.step((e, rel))
        t1 = time.perf_counter()
        if "wait (inside finish)" in marks and "replay (inside launch)" in marks:
            for key, val in (("wait-return -> step return", t1 - marks["wait (inside finish)"][1]),
                             ("step entry -> replay entered", marks["replay (inside launch)"][0] - t0),
                             ("replay returned -> wait entered", marks["wait (inside finish)"][0] - marks["replay (inside launch)"][1])):
                gaps[key][0] += val
                gaps[key][1] += 1
            if prev_wait_end is not None:
                gaps["wait-return -> next replay returned"][0] += marks["replay (inside launch)"][1] - prev_wait_end
                gaps["wait-return -> next replay returned"][1] += 1
            prev_wait_end = marks["wait (inside finish)"][1]
    torch.cuda.synchronize()
    print("rep", rep, "us/step", round((time.perf_counter() - t_rep) / 3 * 1e6, 1), flush=True)
for k, (s, n) in sorted(acc.items()):
    print(f"{k:40s} {s / max(n, 1) * 1e6:8.1f} us x {n}")
for k, (s, n) in gaps.items():
    print(f"{k:40s} {s / max(n, 1) * 1e6:8.1f} us x {n}")
