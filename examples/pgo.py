"""Pose-graph optimisation from a g2o file on the MI355X -- the workflow of the reference's
examples/module/pgo/pgo.py (model, Cholesky solver, TrustRegion, StopOnPlateau) with `import pypose_amd as pp`.

    python examples/pgo.py --g2o tests/golden/sample.g2o            # any VERTEX_SE3:QUAT / EDGE_SE3:QUAT file
    python examples/pgo.py --synthetic 10000 40000                  # chain + random loop closures

The model is the reference's own `PoseGraph`; the optimizer recognises it and runs the per-edge linearisation,
node-parallel assembly and device-resident PCG kernels.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn

import pypose_amd as pp


class PoseGraph(nn.Module):
    def __init__(self, nodes):
        super().__init__()
        self.nodes = pp.Parameter(nodes)

    def forward(self, edges, poses):
        node1 = self.nodes[edges[..., 0]]
        node2 = self.nodes[edges[..., 1]]
        error = poses.Inv() @ node1.Inv() @ node2
        return error.Log().tensor()


def synthetic(N, E, device, dtype=torch.float32, seed=0):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    gt = pp.cumprod(pp.randn_SE3(N, sigma=0.3, device=device, dtype=dtype), dim=0, left=False)
    chain = torch.stack([torch.arange(N - 1), torch.arange(1, N)], -1)
    extra = torch.randint(0, N, (E - (N - 1), 2), generator=g)
    extra[:, 1] = torch.where(extra[:, 0] == extra[:, 1], (extra[:, 1] + 1) % N, extra[:, 1])
    edges = torch.cat([chain, extra], 0).to(device)
    poses = gt[edges[:, 0]].Inv() @ gt[edges[:, 1]] @ pp.randn_SE3(E, sigma=0.01, device=device, dtype=dtype)
    nodes = gt @ pp.randn_SE3(N, sigma=0.05, device=device, dtype=dtype)
    return nodes, edges, poses, None


def main(argv=None):
    ap = argparse.ArgumentParser(description="Pose Graph Optimization")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--g2o", default=None, help="g2o file (VERTEX_SE3:QUAT / EDGE_SE3:QUAT records)")
    ap.add_argument("--synthetic", nargs=2, type=int, default=None, metavar=("NODES", "EDGES"))
    ap.add_argument("--radius", type=float, default=1e4, help="trust region radius")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--save", default=None, help="write the optimised graph as g2o")
    a = ap.parse_args(argv)
    if a.g2o:
        d = pp.io.read_g2o(a.g2o, device=a.device)
        nodes, edges, poses, infos = d["nodes"], d["edges"], d["poses"], d["infos"]    # (edges index vertex rows, as in the reference)
    else:
        nodes, edges, poses, infos = synthetic(*(a.synthetic or (1000, 4000)), a.device)
    graph = PoseGraph(nodes).to(a.device)
    big = graph.nodes.shape[0] * 6 > 4096
    solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250) if big else pp.optim.solver.Cholesky()
    optimizer = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=a.radius), min=1e-6)
    scheduler = pp.optim.scheduler.StopOnPlateau(optimizer, steps=a.steps, patience=3, decreasing=1e-3, verbose=True)
    pp.optim.freeze_gc()          # a full Python GC pass with torch loaded costs tens of LM steps
    t0 = time.perf_counter()
    while scheduler.continual():
        loss = optimizer.step(input=(edges, poses), weight=infos)
        scheduler.step(loss)
    if a.device.startswith("cuda"):
        torch.cuda.synchronize()
    print(f"{scheduler.steps} LM steps in {time.perf_counter() - t0:.3f} s on the '{optimizer.linearization}' path, "
          f"final loss {float(loss):.6g}")
    if a.save:
        pp.io.write_g2o(a.save, graph.nodes.detach().tensor(), edges, poses.tensor(), infos)
    return float(loss)


if __name__ == "__main__":
    main()
