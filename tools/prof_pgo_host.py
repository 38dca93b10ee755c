"""cProfile of the host side of the LM step on the 10k / 40k pose graph (where does the GPU wait for Python?)."""
import cProfile, gc, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from tests.optim_models import PoseGraph
from tests.test_optim_gpu import _synthetic_graph

N, E = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10_000, 40_000)
edges, rel, init = _synthetic_graph(N, E, torch.float32)
graph = PoseGraph(init.clone())
solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250)
opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4), static="static" in sys.argv)


def run(k):
    graph.nodes.data.copy_(init.tensor())
    if hasattr(opt, "loss"):
        del opt.loss
    opt.param_groups[0].update(opt.strategy.defaults)
    for _ in range(k):
        opt.step((edges, rel))


run(3); run(3); torch.cuda.synchronize()
gc.collect(); gc.freeze()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    run(3)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumtime").print_stats(45)
