"""Golden LM trajectories of a small bundle-adjustment problem from the REAL reference (dense LM path):

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_ba_golden.py

Model = the reference example's Reproj (examples/module/ba/bundle_adjustment.py:16-43), synthetic BAL-style
data (BAL convention: p = -P_cam[:2] / P_cam[2]; pixel = f (1 + k1 r^2 + k2 r^4) p)."""
import os
import sys

import numpy as np
import torch
from torch import nn

sys.dont_write_bytecode = True
import pypose as pp  # noqa: E402  the reference

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ba_golden.npz")
D = torch.float64


class Reproj(nn.Module):
    def __init__(self, K, C, P):
        super().__init__()
        self.K = pp.Parameter(K)
        self.C = pp.Parameter(C)
        self.P = pp.Parameter(P)

    def forward(self, observe, cidx, pidx):
        return Reproj.project(self.K[cidx], self.C[cidx], self.P[pidx]) - observe

    @staticmethod
    def project(K, C, P):       # (the example's @psjac needs the un-vendored bae plugin; it does not change values)
        cp = C.Act(P)
        n = - cp[..., :2] / cp[..., [2]]
        radius = n.square().sum(dim=-1, keepdim=True)
        focal, k1, k2 = K[..., :1], K[..., 1:2], K[..., 2:3]
        distortion = 1 + k1 * radius + k2 * radius.square()
        return focal * distortion * n


def make_problem(Nc, Np, per_point, seed):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    P = torch.randn(Np, 3, dtype=D, generator=g) * 0.5
    # cameras on a ring looking at the origin from z = -4 (BAL cameras look down -z)
    C = pp.SE3(torch.cat([torch.tensor([[0., 0., -4.]], dtype=D).repeat(Nc, 1), pp.identity_SO3(Nc, dtype=D).tensor()], -1))
    C = pp.randn_SE3(Nc, sigma=0.15, dtype=D) @ C
    K = torch.stack([torch.full((Nc,), 500., dtype=D), torch.full((Nc,), -0.05, dtype=D), torch.full((Nc,), 0.01, dtype=D)], -1)
    cidx = torch.cat([torch.randperm(Nc, generator=g)[:per_point] for _ in range(Np)])
    pidx = torch.arange(Np).repeat_interleave(per_point)
    with torch.no_grad():
        obs = Reproj.project(K[cidx], C[cidx], P[pidx]) + 0.2 * torch.randn(len(cidx), 2, dtype=D, generator=g)
    K0 = K * (1 + 0.01 * torch.randn(Nc, 3, dtype=D, generator=g))
    C0 = pp.randn_SE3(Nc, sigma=0.02, dtype=D) @ C
    P0 = P + 0.05 * torch.randn(Np, 3, dtype=D, generator=g)
    return obs, cidx, pidx, K0, C0, P0


def main():
    S = {}
    for tag, (Nc, Np, pp_, kw) in {"ba_small": (4, 16, 3, {}), "ba_huber": (5, 24, 3, {"kernel": "huber"})}.items():
        obs, cidx, pidx, K0, C0, P0 = make_problem(Nc, Np, pp_, seed=5 + Nc)
        for k, v in (("obs", obs), ("cidx", cidx), ("pidx", pidx), ("K0", K0), ("C0", C0.tensor()), ("P0", P0)):
            S[f"{tag}/{k}"] = v.numpy()
        model = Reproj(K0.clone(), C0.clone(), P0.clone())
        kernel = pp.optim.kernel.Huber(delta=1.0) if kw.get("kernel") else None
        opt = pp.optim.LM(model, solver=pp.optim.solver.Cholesky(), strategy=pp.optim.strategy.TrustRegion(radius=1e4),
                          kernel=kernel, min=1e-6)
        rec = {"loss": [], "damping": [], "reject": []}
        for _ in range(6):
            loss = opt.step((obs, cidx, pidx))
            rec["loss"].append(float(loss))
            rec["damping"].append(float(opt.param_groups[0]["damping"]))
            rec["reject"].append(int(opt.reject_count))
        for k, v in rec.items():
            S[f"{tag}/{k}"] = np.asarray(v)
        S[f"{tag}/K"], S[f"{tag}/C"], S[f"{tag}/P"] = model.K.detach().numpy(), model.C.detach().tensor().numpy(), model.P.detach().numpy()
        print(tag, rec)
    # Gauss-Newton (pseudo-inverse of the rectangular J) on the gauge-free pose graph pgo12 of lm_golden.npz
    L = np.load(os.path.join(os.path.dirname(OUT), "lm_golden.npz"))

    class PoseGraph(nn.Module):
        def __init__(self, nodes):
            super().__init__()
            self.nodes = pp.Parameter(nodes)

        def forward(self, edges, poses):
            n1, n2 = self.nodes[edges[..., 0]], self.nodes[edges[..., 1]]
            return (poses.Inv() @ n1.Inv() @ n2).Log().tensor()

    edges, poses = torch.from_numpy(L["pgo12/edges"]), pp.SE3(torch.from_numpy(L["pgo12/poses"]))
    graph = PoseGraph(pp.SE3(torch.from_numpy(L["pgo12/init"])))
    opt = pp.optim.GN(graph)
    S["gn_pgo12/loss"] = np.asarray([float(opt.step((edges, poses))) for _ in range(3)])
    S["gn_pgo12/final"] = graph.nodes.detach().tensor().numpy()
    print("gn_pgo12", S["gn_pgo12/loss"])
    np.savez_compressed(OUT, **S)


if __name__ == "__main__":
    main()
