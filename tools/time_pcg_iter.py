"""Per-iteration time of the persistent PCG solve (csrc/pcg_persist.hip) on the metric's 10k / 40k pose graph:
    python tools/time_pcg_iter.py [N E]
The linearisation of the first LM step is solved to a tolerance it cannot reach (so every run does exactly `maxiter`
iterations), for several workgroup counts; HIP events around the solve's launches; the two-launch iteration beside it."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from pypose_amd.optim import fused as F, posegraph as G
from tests.optim_models import PoseGraph
from tests.test_optim_gpu import _synthetic_graph

N, E = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10_000, 40_000)
edges, rel, init = _synthetic_graph(N, E, torch.float32)
graph = PoseGraph(init.clone())
solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250)
opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4))
opt.step((edges, rel))                                           # establishes the fused program
cache = opt._structure_cache
prog = cache["program"][3]
with torch.no_grad():
    lin = F._pgo_linearization(opt, prog, None, graph.nodes, True)
    lin.build_normal_equations(1e-6, 1e32)
    lin.damp(1e-4)
    wsp = next(iter(opt._pcg_workspaces.values()))
    out = {"nodes": N, "edges": E}
    # the ghost-zone form (default) first: marginal cost per iteration and agreement with the two-dependency kernel
    G.GHOST_GRIDS = ()                           # (exactly the grid asked for)
    for grid in (128, 160, 176, 192, 208, 224, 256):
        G.PERSIST_GRID = grid
        wsp.__dict__.pop('_no_ghost', None)
        res = {}
        for iters in (40, 200):
            ts = []
            for rep in range(6):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                x, its = wsp.solve(lin, lin.s, lin.dmin, lin.dmax, 1e-30, iters, None)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            res[iters] = (sorted(ts[1:])[len(ts[1:]) // 2], its)
        out[f"ghost_grid{grid}"] = {"us_per_solve_40": round(res[40][0], 1), "us_per_solve_200": round(res[200][0], 1),
                                    "marginal_us_per_iteration": round((res[200][0] - res[40][0]) / (res[200][1] - res[40][1]), 2),
                                    "used": not wsp.__dict__.get("_no_ghost", False)}
    for pg in (160, 256):
        G.PERSIST_GRID = pg
        wsp.__dict__.pop('_no_ghost', None)
        G.FusedPCG.profile = True                    # phases of the ghost-zone iteration (ticks of 10 ns, thread 0 of a clocked workgroup)
        try:
            x, its = wsp.solve(lin, lin.s, lin.dmin, lin.dmax, 1e-30, 200, None)
        finally:
            G.FusedPCG.profile = False
        torch.cuda.synchronize()
        # all tick slots of the two clocked workgroups (middle: a plain member of the exchange; 0: a group leader at grids >= 224).
        # Slots: 0 spmv+put_q | 1 wave sums+barrier1 | 5 own row summed+published | 6 leader's group gather+publish | 7 final gather |
        # 2 rest of the exchange call | 3 ghost q wait+barrier2 | 4 update+barrier3
        for nm, off in (("mid", 48), ("wg0", 32)):
            tk = wsp.rr_hist[wsp.cap - off:wsp.cap - off + 14].tolist()
            out["ghost_grid%d_ticks_us_per_iteration_%s" % (pg, nm)] = [round(t * 0.01 / max(its, 1), 3) for t in tk]
            # once per solve: slot 9 = registers + block slice staged, 10 = the set-up exchange, 8 = the rest before iteration 0
            out["ghost_grid%d_prologue_us_%s" % (pg, nm)] = {"stage": round(tk[9] * 0.01, 2), "setup_exchange": round(tk[10] * 0.01, 2),
                                                             "rest": round(tk[8] * 0.01, 2), "all_slots_sum": round(sum(tk) * 0.01, 1)}
        starts = wsp.rr_hist[wsp.cap - 128:wsp.cap - 64].tolist()
        out["ghost_grid%d_pass_start_us" % pg] = [round(t * 0.01, 1) for t in starts[:24]]
        # what a solve costs beside its iterations: HIP events around solves of 1 .. 40 iterations (fill + set-up launch + the kernel)
        res = {}
        for iters in (1, 5, 10, 20, 40):
            ts = []
            for rep in range(6):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                wsp.solve(lin, lin.s, lin.dmin, lin.dmax, 1e-30, iters, None)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            res[iters] = round(sorted(ts[1:])[len(ts[1:]) // 2], 1)
        out["ghost_grid%d_us_per_solve_by_iterations" % pg] = res
    G.PERSIST_GRID = 256
    xg, itg = wsp.solve(lin, lin.s, lin.dmin, lin.dmax, 1e-6, 2000, None)
    G.FusedPCG.ghost = False
    xp, itp = wsp.solve(lin, lin.s, lin.dmin, lin.dmax, 1e-6, 2000, None)
    out["ghost_vs_persist"] = {"max_abs_diff": float((xg - xp).abs().max()), "iterations": [itg, itp]}
    for iters in (40, 200):
        for grid in (64, 128, 192, 256):
            G.PERSIST_GRID = grid
            ts = []
            for rep in range(6):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                x, its = wsp.solve(lin, lin.s, lin.dmin, lin.dmax, 1e-30, iters, None)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            t = sorted(ts[1:])[len(ts[1:]) // 2]
            out[f"persist_grid{grid}_it{iters}"] = {"us_per_solve": round(t, 1), "iterations": its, "us_per_iteration": round(t / max(its, 1), 2)}
    # per-iteration cost net of the solve's fixed part (prepare launch, fill, prologue, read-back)
    for grid in (64, 128, 192, 256):
        a, b = out[f"persist_grid{grid}_it40"], out[f"persist_grid{grid}_it200"]
        out[f"persist_grid{grid}_marginal_us_per_iteration"] = round((b["us_per_solve"] - a["us_per_solve"]) / (b["iterations"] - a["iterations"]), 2)
    # phase breakdown of the persistent iteration (wall-clock ticks of 10 ns, thread 0 of the middle workgroup)
    for grid in (64, 128, 256):
        G.PERSIST_GRID = grid
        cap = wsp.cap
        G.FusedPCG.profile = True
        try:
            x, its = wsp.solve(lin, lin.s, lin.dmin, lin.dmax, 1e-30, 200, None)
        finally:
            G.FusedPCG.profile = False
        torch.cuda.synchronize()
        tk = wsp.rr_hist[cap - 8:cap - 2].tolist()
        names = ("spmv", "transpose+wave_sums", "barrier1", "publish+allgather", "barrier2", "update+handoff")
        out[f"persist_grid{grid}_phase_us_per_iteration"] = {n: round(t * 0.01 / max(its, 1), 3) for n, t in zip(names, tk)}
    G.PERSIST_GRID = 256
    x1, _ = wsp.solve(lin, lin.s, lin.dmin, lin.dmax, 1e-6, 2000, None)
    G.FusedPCG.persist = False
    x2, its2 = wsp.solve(lin, lin.s, lin.dmin, lin.dmax, 1e-6, 2000, None)
    out["persist_vs_two_launch_max_abs_diff"] = float((x1 - x2).abs().max())
    out["solution_max_abs"] = float(x2.abs().max())
    ts = []
    for rep in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        x, its = wsp.solve(lin, lin.s, lin.dmin, lin.dmax, 1e-30, 200, None)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    out["two_launch_it200"] = {"us_per_solve": round(sorted(ts[1:])[1], 1), "iterations": its, "us_per_iteration": round(sorted(ts[1:])[1] / max(its, 1), 2)}
for k, v in out.items():
    print(json.dumps({k: v}))
