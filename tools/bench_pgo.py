"""LM steps/s on synthetic pose graphs (BASELINE metric second half):
`python tools/bench_pgo.py [N E] [steps]`; every repetition restarts from the same initial estimate."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from tests.optim_models import PoseGraph
from tests.test_optim_gpu import _synthetic_graph
cases = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(10_000, 40_000), (100_000, 400_000)]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for N, E in cases:
    edges, rel, init = _synthetic_graph(N, E, torch.float32)
    for fused in (True, False):
        graph = PoseGraph(init.clone())
        solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250)
        opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4))
        opt.fused = fused
        times, its = [], []
        for rep in range(4):
            graph.nodes.data.copy_(init.tensor())
            if hasattr(opt, "loss"):
                del opt.loss
            opt.param_groups[0].update(opt.strategy.defaults)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            it = []
            for _ in range(steps):
                loss = opt.step((edges, rel)); it.append(solver.iterations)
            torch.cuda.synchronize(); times.append((time.perf_counter() - t0) / steps)
            its = it
        print(json.dumps({"nodes": N, "edges": E, "path": opt.linearization, "ms_per_step": [round(t * 1e3, 3) for t in times],
                          "steps_per_s": round(1 / min(times[1:]), 1), "pcg_iterations": its, "final_loss": float(loss)}))
