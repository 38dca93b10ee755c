"""Golden LM trajectories of a pose graph with THREE residuals from the REAL reference (dense LM path):
odometry edges, loop closures (own robust kernel, own information matrix) and unary priors.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_multires_golden.py
"""
import os
import sys

import numpy as np
import torch
from torch import nn

sys.dont_write_bytecode = True
import pypose as pp  # noqa: E402  the reference

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multires_golden.npz")
D = torch.float64


class MixedGraph(nn.Module):
    def __init__(self, nodes):
        super().__init__()
        self.nodes = pp.Parameter(nodes)

    def forward(self, odo, zodo, loop, zloop, pidx, prior):
        a = (zodo.Inv() @ self.nodes[odo[:, 0]].Inv() @ self.nodes[odo[:, 1]]).Log().tensor()
        b = (zloop.Inv() @ self.nodes[loop[:, 0]].Inv() @ self.nodes[loop[:, 1]]).Log().tensor()
        c = (prior.Inv() @ self.nodes[pidx]).Log().tensor()[..., :3]          # position prior (3 rows per factor)
        return a, b, c


def main():
    torch.manual_seed(21)
    N = 14
    gt = pp.cumprod(pp.randn_SE3(N, sigma=0.4, dtype=D), dim=0, left=False)
    odo = torch.stack([torch.arange(N - 1), torch.arange(1, N)], -1)
    loop = torch.tensor([[0, 5], [2, 9], [4, 13], [1, 11], [6, 12], [3, 8], [7, 13]])
    pidx = torch.tensor([0, 7, 13])
    zodo = gt[odo[:, 0]].Inv() @ gt[odo[:, 1]] @ pp.randn_SE3(N - 1, sigma=0.02, dtype=D)
    zloop = gt[loop[:, 0]].Inv() @ gt[loop[:, 1]] @ pp.randn_SE3(len(loop), sigma=0.05, dtype=D)
    zloop = pp.SE3(torch.cat([zloop.tensor()[:1] * 0 + pp.randn_SE3(1, sigma=1.0, dtype=D).tensor(), zloop.tensor()[1:]]))   # one outlier
    prior = gt[pidx] @ pp.randn_SE3(3, sigma=0.01, dtype=D)
    init = gt @ pp.randn_SE3(N, sigma=0.08, dtype=D)
    M = torch.randn(6, 6, dtype=D)
    Wloop = M @ M.T / 6 + torch.eye(6, dtype=D)
    S = {"odo": odo, "zodo": zodo.tensor(), "loop": loop, "zloop": zloop.tensor(), "pidx": pidx, "prior": prior.tensor(),
         "init": init.tensor(), "Wloop": Wloop}
    cases = {
        "plain": dict(),
        "kernels": dict(kernel=[None, pp.optim.kernel.Huber(delta=0.3), None]),
        "kernels_weights": dict(kernel=[None, pp.optim.kernel.Cauchy(delta=0.5), None],
                                weight=[torch.eye(6, dtype=D) * 2.0, Wloop, torch.eye(3, dtype=D) * 10.0]),
    }
    for tag, kw in cases.items():
        model = MixedGraph(init.clone())
        weight = kw.pop("weight", None)
        opt = pp.optim.LM(model, solver=pp.optim.solver.Cholesky(), strategy=pp.optim.strategy.TrustRegion(radius=1e4), min=1e-6, **kw)
        rec = {"loss": [], "damping": [], "reject": []}
        for _ in range(6):
            loss = opt.step((odo, zodo, loop, zloop, pidx, prior), weight=weight)
            rec["loss"].append(float(loss))
            rec["damping"].append(float(opt.param_groups[0]["damping"]))
            rec["reject"].append(int(opt.reject_count))
        for k, v in rec.items():
            S[f"{tag}/{k}"] = np.asarray(v)
        S[f"{tag}/nodes"] = model.nodes.detach().tensor()
        print(tag, rec)
    np.savez_compressed(OUT, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in S.items()})


if __name__ == "__main__":
    main()
