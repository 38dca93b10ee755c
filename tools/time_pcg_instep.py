"""The persistent solve INSIDE real LM steps (10 k / 40 k, tol 1e-4: 17 / 19 / 25 iterations), clocked by its own workgroups
(profiling instantiation): prologue, when each pass starts, the total -- to compare with 12.9 + 7.35 x passes of the isolated solve
(tools/time_pcg_iter.py) and with the kernel's duration in a rocprofv3 timeline of the step."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pypose_amd as pp
from pypose_amd.optim import posegraph as G

dev = torch.device("cuda:0")
e, rel, init = bench._pose_graph_problem(dev, 10_000, 40_000)
graph = bench._pose_graph_model(init.clone())
solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250)
opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4))
opt.graph_step = False                                   # (un-captured steps: the profiling launch is not part of a captured graph)
G.FusedPCG.profile = True
for rep in range(3):
    graph.nodes.data.copy_(init.tensor())
    if hasattr(opt, "loss"):
        del opt.loss
    opt.param_groups[0].update(opt.strategy.defaults)
    for step in range(3):
        opt.step((e, rel))
        torch.cuda.synchronize()
        w = next(iter(opt._pcg_workspaces.values()))
        tk = w.rr_hist[w.cap - 48:w.cap - 48 + 14].tolist()
        starts = w.rr_hist[w.cap - 128:w.cap - 64].tolist()
        its = solver.iterations
        if rep == 2:
            print(json.dumps({"step": step, "iterations": its, "passes": its + 1, "prologue_us": round((tk[9] + tk[10] + tk[8]) * 0.01, 1),
                              "all_slots_sum_us": round(sum(tk) * 0.01, 1), "us_per_pass": round((sum(tk[:8])) * 0.01 / (its + 1), 2),
                              "pass_start_us": [round(t * 0.01, 1) for t in starts[:its + 1]]}))
