"""Experiment (wrong results by construction): how much of the 100k / 400k SpMV's time are L2 MISSES of the p gathers?
The neighbour indices of the two-launch iteration are folded into the first K nodes (other % K): same instruction stream, same
number of gathers, but the gathered rows fit any L2 for small K.  HIP events around 64 / 192 forced iterations, as tools/time_pcg2.py."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from pypose_amd.optim import fused as F
from tests.optim_models import PoseGraph
from tests.test_optim_gpu import _synthetic_graph

N, E = 100_000, 400_000
edges, rel, init = _synthetic_graph(N, E, torch.float32)
graph = PoseGraph(init.clone())
solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250)
opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4))
opt.step((edges, rel))
prog = opt._structure_cache["program"][3]
out = {}
with torch.no_grad():
    lin = F._pgo_linearization(opt, prog, None, graph.nodes, True)
    lin.build_normal_equations(1e-6, 1e32)
    lin.damp(1e-4)
    wsp = next(iter(opt._pcg_workspaces.values()))
    wsp.solve(lin, lin.s, lin.dmin, lin.dmax, 1e-30, 8, None)
    keep = wsp.other.clone()
    wsp.HB.mul_(1e-3)            # (diagonally dominant whatever the indices are: the recurrences stay finite; HB is not re-assembled by solve())
    for K in (N, 32768, 4096, 256):
        wsp.other.copy_(keep % K)
        wsp.graph = None
        res = {}
        for iters in (64, 192):
            ts = []
            for rep in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                try:
                    x, its = wsp.solve(lin, lin.s, lin.dmin, lin.dmax, 1e-30, iters, None)
                except AssertionError as e:
                    its = -1
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            res[iters] = (sorted(ts[1:])[len(ts[1:]) // 2], its)
        out[str(K)] = {"us": {str(k): round(v[0], 1) for k, v in res.items()}, "its": [v[1] for v in res.values()],
                       "marginal_us_per_iteration": round((res[192][0] - res[64][0]) / 128, 2)}
print(json.dumps(out))
