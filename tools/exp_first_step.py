"""Why is the persistent solve of the FIRST LM step of a run ~55 us slower than 7.4 us x iterations says (202 us at 18 iterations against
152 at 21, 172 at 25)?  Runs of 6 steps; variants: plain | idle (sync + 1 ms sleep before step 4) | hot (no sync / reset gap: the next run's
reset is enqueued without waiting).  Under rocprofv3 --kernel-trace; tools/exp_first_step_report.py lists the solve's durations per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pypose_amd as pp

variant = sys.argv[1] if len(sys.argv) > 1 else "plain"
dev = torch.device("cuda:0")
e, rel, init = bench._pose_graph_problem(dev, 10_000, 40_000)
graph = bench._pose_graph_model(init.clone())
solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250)
opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4))
pp.optim.freeze_gc()
for rep in range(6):
    graph.nodes.data.copy_(init.tensor())
    if hasattr(opt, "loss"):
        del opt.loss
    opt.param_groups[0].update(opt.strategy.defaults)
    if variant != "hot":
        torch.cuda.synchronize()
    its = []
    for k in range(6):
        if variant == "idle" and k == 3:
            torch.cuda.synchronize()
            time.sleep(0.001)
        opt.step((e, rel))
        its.append(solver.iterations)
    torch.cuda.synchronize()
    print("rep", rep, variant, "its", its, flush=True)
