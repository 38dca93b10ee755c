"""Where the time of a captured pose-graph LM step goes on the HOST (no profiler attached): wall-clock stamps around the wait for
the trial's verdict, the rest of finish(), the way back to launch(), and the replay call.   python tools/time_pgo_host.py [static]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pypose_amd as pp
from pypose_amd.optim import pgograph

static = "static" in sys.argv
dev = torch.device("cuda:0")
e, rel, init = bench._pose_graph_problem(dev, 10_000, 40_000)
graph = bench._pose_graph_model(init.clone())
solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250)
opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4), static=static)
pp.optim.freeze_gc()
T = {"wait": 0.0, "finish_rest": 0.0, "launch": 0.0, "replay": 0.0, "n": 0, "_w": 0.0}
pc = time.perf_counter
G = pgograph.PgoGraphStep
TT = pgograph.TrialTail
_wait, _finish, _launch = TT.wait, G.finish, G.launch


def wait(self):
    t0 = pc(); r = _wait(self); T['_w'] = pc() - t0; return r


def finish(self, pg):
    t0 = pc(); r = _finish(self, pg); T["wait"] += T['_w']; T["finish_rest"] += pc() - t0 - T['_w']; T["n"] += 1; return r


def launch(self, pg):
    g = self.graph
    t0 = pc()

    class R:
        @staticmethod
        def replay():
            t1 = pc(); g.replay(); T["replay"] += pc() - t1
    self.graph = R
    try:
        _launch(self, pg)
    finally:
        self.graph = g
    T["launch"] += pc() - t0


TT.wait, G.finish, G.launch = wait, finish, launch
for rep in range(13):
    graph.nodes.data.copy_(init.tensor())
    if hasattr(opt, "loss"):
        del opt.loss
    opt.param_groups[0].update(opt.strategy.defaults)
    torch.cuda.synchronize()
    if rep == 3:
        for k in T: T[k] = 0
        t_all = pc()
    for _ in range(3):
        opt.step((e, rel))
    torch.cuda.synchronize()
n = T["n"]
tot = (pc() - t_all) / n * 1e6
print(f"static={static} steps {n}: per step {tot:.1f} us = wait {T['wait']/n*1e6:.1f} + finish-after-wait {T['finish_rest']/n*1e6:.1f} + launch() {T['launch']/n*1e6:.1f} "
      f"(of which graph.replay {T['replay']/n*1e6:.1f}) + everything else {tot - (T['wait']+T['finish_rest']+T['launch'])/n*1e6:.1f}")
