"""Per-iteration time of the two-launch PCG iteration (pplie_pcg2_spmv + pplie_pcg2_step, csrc/graph.hip) on a pose graph beyond
the persistent solve -- BASELINE configs[3], 100k nodes / 400k edges by default:
    python tools/time_pcg2.py [N E]
The first LM step's system is solved to a tolerance it cannot reach (exactly `maxiter` iterations); HIP events around the solve."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from pypose_amd.optim import fused as F
from tests.optim_models import PoseGraph
from tests.test_optim_gpu import _synthetic_graph

N, E = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100_000, 400_000)
edges, rel, init = _synthetic_graph(N, E, torch.float32)
graph = PoseGraph(init.clone())
solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250)
opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4))
opt.step((edges, rel))
prog = opt._structure_cache["program"][3]
out = {"nodes": N, "edges": E}
with torch.no_grad():
    lin = F._pgo_linearization(opt, prog, None, graph.nodes, True)
    lin.build_normal_equations(1e-6, 1e32)
    lin.damp(1e-4)
    wsp = next(iter(opt._pcg_workspaces.values()))
    res = {}
    for iters in (64, 192):
        ts = []
        for rep in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            x, its = wsp.solve(lin, lin.s, lin.dmin, lin.dmax, 1e-30, iters, None)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        res[iters] = (sorted(ts[1:])[len(ts[1:]) // 2], its)
    out["us_per_solve"] = {str(k): round(v[0], 1) for k, v in res.items()}
    out["marginal_us_per_iteration"] = round((res[192][0] - res[64][0]) / (res[192][1] - res[64][1]), 2)
    x, its = wsp.solve(lin, lin.s, lin.dmin, lin.dmax, 1e-4, 250, None)
    out["iterations_to_tol_1e-4"] = its
    out["solution_checksum"] = float(x.double().abs().sum())
print(json.dumps(out))
