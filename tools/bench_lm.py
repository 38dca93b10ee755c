"""LM step rates on one MI355X for BASELINE configs[2] (InvNet, B independent problems) and
configs[3] / the metric's PGO sizes.  Writes gpurun_out/bench_lm.json."""
import json, sys, time
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pypose_amd as pp
from tests.optim_models import InvNet, PoseGraph
from tests.test_optim_gpu import _synthetic_graph

dev = "cuda:0"
out = {"device": torch.cuda.get_device_name(0)}


def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = [fn() for _ in range(n)]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, r

# ---- C3: InvNet SE3, B = 1M, fp32, Constant(1e-4) (README.md:120-129 of the reference)
for B in (1024, 1_000_000):
    torch.manual_seed(0); net = InvNet(pp.randn_SE3(B, device=dev))
    torch.manual_seed(1); inp = pp.randn_SE3(B, device=dev)
    opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(damping=1e-4))
    opt.step(inp)                                # structure probe + first step (not timed)
    with torch.no_grad():
        net.pose.copy_(pp.randn_SE3(B, device=dev)); del opt.loss
    dt, losses = timed(lambda: float(opt.step(inp)), 5)
    out[f"c3_invnet_B{B}"] = {"s_per_step": dt, "lm_steps_per_s": 1 / dt, "problem_steps_per_s": B / dt,
                              "losses": losses, "path": opt.linearization}
    print(out[f"c3_invnet_B{B}"], flush=True)

# ---- C4: synthetic pose graphs, PCG tol 1e-4 maxiter 250, TrustRegion(1e4)
for N, E in ((10_000, 40_000), (100_000, 400_000)):
    edges, rel, init = _synthetic_graph(N, E, torch.float32)
    graph = PoseGraph(init)
    solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250)
    opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4))
    l0 = float(graph(edges, rel).square().sum())
    opt.step((edges, rel))
    its = []
    def one():
        l = float(opt.step((edges, rel))); its.append(getattr(solver, "iterations", -1)); return l
    dt, losses = timed(one, 5)
    out[f"c4_pgo_N{N}_E{E}"] = {"s_per_step": dt, "lm_steps_per_s": 1 / dt, "edge_steps_per_s": E / dt, "initial_loss": l0,
                                "losses": losses, "pcg_iterations": its, "path": opt.linearization}
    print(out[f"c4_pgo_N{N}_E{E}"], flush=True)
json.dump(out, open("gpurun_out/bench_lm.json", "w"), indent=1)
