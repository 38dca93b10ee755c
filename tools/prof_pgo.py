"""C4 (100k nodes / 400k edges) LM steps for rocprofv3 --kernel-trace --stats."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from tests.optim_models import PoseGraph
from tests.test_optim_gpu import _synthetic_graph
N, E = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100_000, 400_000)
edges, rel, init = _synthetic_graph(N, E, torch.float32)
graph = PoseGraph(init)
solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250)
opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4))
opt.step((edges, rel))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    opt.step((edges, rel)); print("pcg its", solver.iterations)
torch.cuda.synchronize(); print("s/step", (time.perf_counter() - t0) / 3)
