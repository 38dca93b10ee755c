"""Wall-clock breakdown of the default LM step on the 10k / 40k pose graph (timers around the step's stages)."""
import gc, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from pypose_amd.optim import optimizer as O, fused as F, posegraph as G
from tests.optim_models import PoseGraph
from tests.test_optim_gpu import _synthetic_graph

N, E = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10_000, 40_000)
edges, rel, init = _synthetic_graph(N, E, torch.float32)
graph = PoseGraph(init.clone())
solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250)
opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4), static="static" in sys.argv)
T = {}


def timed(owner, name, label):
    f = getattr(owner, name)
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            T[label] = T.get(label, 0.0) + time.perf_counter() - t
    setattr(owner, name, w)


timed(O, "_linearize", "linearize (trace + kernel)")
timed(F, "try_fused", "  try_fused")
timed(O.RobustModel, "forward", "    model.forward (trace)")
timed(F.PgoProgram, "linearize", "    pgo linearize kernel call")
timed(G.GraphLinearization, "build_normal_equations", "assemble")
timed(G.GraphLinearization, "solve", "solve (PCG)")
timed(G.FusedPCG, "solve", "  FusedPCG.solve")
timed(O._Optimizer, "update_parameter", "update_parameter")
timed(F.PgoProgram, "loss", "loss kernel + read-back")
timed(O.LevenbergMarquardt, "_strategy_update", "strategy update")
timed(G.GraphLinearization, "csr", "  csr()")


def run(k):
    graph.nodes.data.copy_(init.tensor())
    if hasattr(opt, "loss"):
        del opt.loss
    opt.param_groups[0].update(opt.strategy.defaults)
    for _ in range(k):
        opt.step((edges, rel))


run(3); run(3); torch.cuda.synchronize(); T.clear()
gc.collect(); gc.freeze()
t0 = time.perf_counter()
for _ in range(20):
    run(3)
torch.cuda.synchronize()
print(f"ms/step {(time.perf_counter() - t0) / 60 * 1e3:.3f}  (PCG iterations of the last step: {solver.iterations})")
for k, v in T.items():
    print(f"  {k:34s} {v / 60 * 1e3:.3f} ms/step")
