"""Generate tests/golden/lm_golden.npz from the REAL reference (run in the build container):

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_lm_golden.py

Records, for a handful of small fp64 problems, the initial state and the per-step trajectory
(loss, damping, reject count) plus the final parameters produced by pypose.optim.LM / GN.
"""
import os
import sys

import numpy as np
import torch
from torch import nn

sys.dont_write_bytecode = True
import pypose as pp  # noqa: E402  the reference

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lm_golden.npz")
D = torch.float64


class InvNet(nn.Module):                       # reference README.md:120-129
    def __init__(self, init):
        super().__init__()
        self.pose = pp.Parameter(init)

    def forward(self, input):
        return (self.pose @ input).Log().tensor()


class PoseGraph(nn.Module):                    # reference examples/module/pgo/pgo.py:15-25
    def __init__(self, nodes):
        super().__init__()
        self.nodes = pp.Parameter(nodes)

    def forward(self, edges, poses):
        node1 = self.nodes[edges[..., 0]]
        node2 = self.nodes[edges[..., 1]]
        error = poses.Inv() @ node1.Inv() @ node2
        return error.Log().tensor()


def make_graph(N, E, seed):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    steps = pp.randn_SE3(N, sigma=0.3, dtype=D)
    gt = pp.cumprod(steps, dim=0, left=False)
    chain = torch.stack([torch.arange(N - 1), torch.arange(1, N)], -1)
    extra = torch.randint(0, N, (E - (N - 1), 2), generator=g)
    extra[:, 1] = torch.where(extra[:, 0] == extra[:, 1], (extra[:, 1] + 1) % N, extra[:, 1])
    edges = torch.cat([chain, extra], 0)
    rel = gt[edges[:, 0]].Inv() @ gt[edges[:, 1]] @ pp.randn_SE3(E, sigma=0.01, dtype=D)
    init = gt @ pp.randn_SE3(N, sigma=0.05, dtype=D)
    A = torch.randn(E, 6, 6, dtype=D, generator=g) * 0.3
    infos = A @ A.mT + torch.eye(6, dtype=D)
    return edges, rel, init, infos


def run(opt, step_args, nsteps, params):
    rec = {"loss": [], "damping": [], "reject": []}
    for _ in range(nsteps):
        loss = opt.step(*step_args[0], **step_args[1])
        rec["loss"].append(float(loss))
        rec["damping"].append(float(opt.param_groups[0].get("damping", 0.0)))
        rec["reject"].append(int(getattr(opt, "reject_count", 0)))
    rec["final"] = params.detach().clone().numpy()
    return rec


def main():
    S = {}
    torch.manual_seed(0)
    inp = pp.randn_SE3(2, 2, dtype=D)
    init = pp.randn_SE3(2, 2, dtype=D)
    S["invnet/input"], S["invnet/init"] = inp.numpy(), init.numpy()
    strategies = {"constant": lambda: pp.optim.strategy.Constant(damping=1e-4),
                  "adaptive": lambda: pp.optim.strategy.Adaptive(damping=1e-6),
                  "trustregion": lambda: pp.optim.strategy.TrustRegion()}
    for name, mk in strategies.items():
        net = InvNet(init.clone())
        opt = pp.optim.LM(net, strategy=mk())
        rec = run(opt, ((inp,), {}), 6, net.pose)
        for k, v in rec.items():
            S[f"invnet/{name}/{k}"] = np.asarray(v)
    # weight + robust kernel (FastTriggs is the default corrector when a kernel is given)
    g = torch.Generator().manual_seed(3)
    Wm = torch.randn(2, 2, 6, 6, dtype=D, generator=g) * 0.2
    Wm = Wm @ Wm.mT + torch.eye(6, dtype=D)
    S["invnet/weight"] = Wm.numpy()
    net = InvNet(init.clone())
    opt = pp.optim.LM(net, strategy=pp.optim.strategy.Adaptive(damping=1e-6), kernel=pp.optim.kernel.Huber(delta=0.5))
    rec = run(opt, ((inp,), {"weight": Wm}), 6, net.pose)
    for k, v in rec.items():
        S[f"invnet/huber_weight/{k}"] = np.asarray(v)
    # target + Cauchy kernel + Triggs-free default corrector
    net = InvNet(init.clone())
    tgt = 0.01 * torch.randn(2, 2, 6, dtype=D, generator=g)
    S["invnet/target"] = tgt.numpy()
    opt = pp.optim.LM(net, strategy=pp.optim.strategy.TrustRegion(radius=1e4), kernel=pp.optim.kernel.Cauchy())
    rec = run(opt, ((inp, tgt), {}), 6, net.pose)
    for k, v in rec.items():
        S[f"invnet/cauchy_target/{k}"] = np.asarray(v)
    # Gauss-Newton
    net = InvNet(init.clone())
    opt = pp.optim.GN(net)
    rec = run(opt, ((inp,), {}), 4, net.pose)
    for k, v in rec.items():
        S[f"invnet/gn/{k}"] = np.asarray(v)
    # a rejected-step scenario: huge initial perturbation, tiny damping
    far = pp.randn_SE3(2, 2, sigma=3.0, dtype=D)
    S["invnet/far_init"] = far.numpy()
    net = InvNet(far.clone())
    opt = pp.optim.LM(net, strategy=pp.optim.strategy.TrustRegion(radius=1e8), min=1e-12)
    rec = run(opt, ((inp,), {}), 8, net.pose)
    for k, v in rec.items():
        S[f"invnet/far/{k}"] = np.asarray(v)

    # pose graphs (dense reference path, Cholesky, TrustRegion(radius=1e4) as in pgo.py:66-69)
    for tag, (N, E) in {"pgo12": (12, 30), "pgo40": (40, 110)}.items():
        edges, rel, init, infos = make_graph(N, E, seed=11 + N)
        S[f"{tag}/edges"], S[f"{tag}/poses"], S[f"{tag}/init"], S[f"{tag}/infos"] = \
            edges.numpy(), rel.numpy(), init.numpy(), infos.numpy()
        for wname, w in (("noweight", None), ("infos", infos)):
            graph = PoseGraph(init.clone())
            opt = pp.optim.LM(graph, solver=pp.optim.solver.Cholesky(), strategy=pp.optim.strategy.TrustRegion(radius=1e4), min=1e-6)
            rec = run(opt, (((edges, rel),), {"weight": w}), 5, graph.nodes)
            for k, v in rec.items():
                S[f"{tag}/{wname}/{k}"] = np.asarray(v)
    # a hard start (large initial error, almost no damping): exercises rejected steps
    edges, rel, init, infos = make_graph(12, 30, seed=23)
    torch.manual_seed(5)
    far = init @ pp.randn_SE3(12, sigma=1.5, dtype=D)
    S["pgofar/edges"], S["pgofar/poses"], S["pgofar/init"] = edges.numpy(), rel.numpy(), far.numpy()
    graph = PoseGraph(far.clone())
    opt = pp.optim.LM(graph, solver=pp.optim.solver.Cholesky(), strategy=pp.optim.strategy.TrustRegion(radius=1e8), min=1e-9)
    rec = run(opt, (((edges, rel),), {}), 10, graph.nodes)
    for k, v in rec.items():
        S[f"pgofar/{k}"] = np.asarray(v)
    print("pgofar reject", rec["reject"], "damping", rec["damping"])
    np.savez_compressed(OUT, **S)
    print("wrote", OUT, len(S), "arrays", os.path.getsize(OUT), "bytes")
    for k in sorted(S):
        if k.endswith("/loss"):
            print(k, S[k])


if __name__ == "__main__":
    main()
