"""bisect the captured pose-graph trial on a small fp64 weighted graph"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pypose_amd as pp
from tests.optim_models import PoseGraph, run_steps

DEV = torch.device("cuda:0")
G = np.load("tests/golden/lm_golden.npz")

T = lambda a: torch.as_tensor(a).to(DEV)
which = sys.argv[1]
dtype = torch.float64 if "f64" in which else torch.float32
edges, poses = T(G["pgo40/edges"]), pp.SE3(T(G["pgo40/poses"]).to(dtype))
graph = PoseGraph(pp.SE3(T(G["pgo40/init"]).to(dtype)))
tol = 1e-13 if "tight" in which else 1e-6
opt = pp.optim.LM(graph, solver=pp.optim.solver.PCG(tol=tol, maxiter=2000, check_every=1), strategy=pp.optim.strategy.TrustRegion(radius=1e4), min=1e-6)
w = T(G["pgo40/infos"]).to(dtype) if "w" in which.split("-") else None
for k in range(6):
    loss = opt.step((edges, poses), weight=w)
    torch.cuda.synchronize()
    print(which, "step", k, float(loss), opt.linearization, "graph step" if opt.__dict__.get('_pgo_graph_step') is not None else "-", "its", opt.solver.iterations, flush=True)
