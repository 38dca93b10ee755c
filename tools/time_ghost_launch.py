"""What does ONE launch of the persistent ghost-zone solve cost beside the work its workgroups clock themselves?
The kernel alone, back to back, for 0 and 8 iterations, alone and alternating with a tiny fill kernel, eagerly and as ONE captured
hipGraph; HIP events over 40 launches.  (The tag tables are not cleared between launches: every poll finds complete rows of an earlier launch at
once -- a LOWER bound of the exchange, which is what isolates the launch cost.)"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from pypose_amd import _C
from pypose_amd.optim import fused as F, posegraph as G
from tests.optim_models import PoseGraph
from tests.test_optim_gpu import _synthetic_graph

edges, rel, init = _synthetic_graph(10_000, 40_000, torch.float32)
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
    w = next(iter(opt._pcg_workspaces.values()))
    G.GHOST_GRIDS = ()
    for grid in (160, 256):
        G.PERSIST_GRID = grid
        w.__dict__.pop('_no_ghost', None)
        w.solve(lin, lin.s, lin.dmin, lin.dmax, 1e-30, 5, None)              # buffers, ghost map, D / Binv / shift in place
        slot, gptr, gids, max_cnt, max_ghost = w._ghost_map(lin, grid)
        fn = _C.library().symbol("pplie_pcg_ghost_coarse" + w.sfx, G._GHOST_CZ_SIG)
        st = _C.stream_ptr(w.device)

        def launch(iters):
            return fn(w.ptr.data_ptr(), slot.data_ptr(), w.HB.data_ptr(), w.D.data_ptr(), w.Binv.data_ptr(), w.shift.data_ptr(), w.x.data_ptr(),
                      w.r.data_ptr(), w.z.data_ptr(), gptr.data_ptr(), gids.data_ptr(), w.part.data_ptr(), w.ptag.data_ptr(), w.rr_hist.data_ptr(),
                      w.info.data_ptr(), w.it.data_ptr(), 1e-30, iters, w.cap, grid, max_cnt, max_ghost, w.N, w.m, st)
        # alternating with another (tiny) kernel: does the persistent kernel pay for following a different kernel?
        for iters in (0, 8):
            for other in ("fill", "none"):
                n = 40
                ts = []
                for rep in range(5):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(n):
                        if other == "fill":
                            w.info.zero_()
                        assert launch(iters) == 0
                    b.record(); torch.cuda.synchronize()
                    ts.append(a.elapsed_time(b) * 1e3 / n)
                out[f"grid{grid}_{iters}it_after_{other}_us_per_launch"] = round(sorted(ts)[2], 2)
        ts = []
        for rep in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(40):
                w.info.zero_()
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / 40)
        out[f"grid{grid}_fill_alone_us"] = round(sorted(ts)[2], 2)
        # the same alternation as ONE captured hipGraph (kernel nodes chained by the stream's order): what does a graph edge cost?
        for iters in (0, 8):
            n = 40
            g = torch.cuda.CUDAGraph()
            st_keep = st
            with _C.graph_capture(g):
                st = _C.stream_ptr(w.device)                       # (the capture's stream)
                for _ in range(n):
                    w.info.zero_()
                    assert launch(iters) == 0
            st = st_keep
            ts = []
            for rep in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); g.replay(); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3 / n)
            out[f"grid{grid}_{iters}it_after_fill_GRAPH_us_per_launch"] = round(sorted(ts)[2], 2)
for k, v in out.items():
    print(json.dumps({k: v}))
