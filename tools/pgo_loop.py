"""The lm_pgo loop of bench.py alone (for rocprofv3 --kernel-trace):  python tools/pgo_loop.py [nodes edges reps static]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pypose_amd as pp

nodes, edges, reps, static = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), bool(int(sys.argv[4]))) if len(sys.argv) > 4 else (10_000, 40_000, 8, False)
dev = torch.device("cuda:0")
e, rel, init = bench._pose_graph_problem(dev, nodes, edges)
graph = bench._pose_graph_model(init.clone())
solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250)
opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4), static=static)
pp.optim.freeze_gc()
for rep in range(reps + 1):
    graph.nodes.data.copy_(init.tensor())
    if hasattr(opt, "loss"):
        del opt.loss
    opt.param_groups[0].update(opt.strategy.defaults)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    its = []
    for _ in range(3):
        opt.step((e, rel))
        its.append(solver.iterations)
    torch.cuda.synchronize()
    print("rep", rep, "ms/step", round((time.perf_counter() - t0) / 3 * 1e3, 3), "its", its, flush=True)
