"""Host-side time breakdown of one pose-graph LM step (cProfile, top cumulative entries)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from tests.optim_models import PoseGraph
from tests.test_optim_gpu import _synthetic_graph
N, E = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10_000, 40_000)
edges, rel, init = _synthetic_graph(N, E, torch.float32)
graph = PoseGraph(init)
solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250)
opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4))
opt.step((edges, rel)); opt.step((edges, rel))
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    opt.step((edges, rel))
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
