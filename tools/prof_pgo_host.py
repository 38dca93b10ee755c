"""cProfile of LM steps on the 10k / 40k pose graph: where the host time of a step goes.
    python tools/prof_pgo_host.py [nodes edges]"""
import cProfile
import pstats
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import pypose_amd as pp  # noqa: E402
from tests.test_optim_gpu import _synthetic_graph  # noqa: E402
from tests.optim_models import PoseGraph  # noqa: E402

N, E = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10_000, 40_000)
edges, rel, init = _synthetic_graph(N, E, torch.float32)
graph = PoseGraph(init.clone())
solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250)
opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4))


def run(reps):
    for _ in range(reps):
        graph.nodes.data.copy_(init.tensor())
        if hasattr(opt, "loss"):
            del opt.loss
        opt.param_groups[0].update(opt.strategy.defaults)
        for _ in range(3):
            opt.step((edges, rel))
    torch.cuda.synchronize()


run(2)
import time
t0 = time.perf_counter(); run(10); dt = (time.perf_counter() - t0) / 30
print(f"ms/step {dt * 1e3:.3f}")
pr = cProfile.Profile()
pr.enable()
run(10)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
