"""Golden LM trajectories of bundle adjustment WITH prior residuals from the REAL reference (dense LM path): the model
returns three residuals -- reprojection errors, position priors on some cameras, position priors on some points.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_ba_prior_golden.py
"""
import os
import sys

import numpy as np
import torch
from torch import nn

sys.dont_write_bytecode = True
import pypose as pp  # noqa: E402  the reference

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_ba_golden import Reproj, make_problem, D  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ba_prior_golden.npz")


class ReprojWithPriors(nn.Module):
    def __init__(self, K, C, P):
        super().__init__()
        self.K = pp.Parameter(K)
        self.C = pp.Parameter(C)
        self.P = pp.Parameter(P)

    def forward(self, observe, cidx, pidx, cam_ids, cam_pos, pt_ids, pt_pos):
        reproj = Reproj.project(self.K[cidx], self.C[cidx], self.P[pidx]) - observe
        cam_prior = self.C[cam_ids].translation() - cam_pos
        pt_prior = self.P[pt_ids] - pt_pos
        return reproj, cam_prior, pt_prior


def main():
    S = {}
    obs, cidx, pidx, K0, C0, P0 = make_problem(5, 20, 3, seed=17)
    torch.manual_seed(3)
    cam_ids = torch.tensor([0, 2, 4])
    cam_pos = C0[cam_ids].translation() + 0.01 * torch.randn(3, 3, dtype=D)
    pt_ids = torch.tensor([1, 5, 7, 11, 19])
    pt_pos = P0[pt_ids] + 0.02 * torch.randn(5, 3, dtype=D)
    for k, v in (("obs", obs), ("cidx", cidx), ("pidx", pidx), ("K0", K0), ("C0", C0.tensor()), ("P0", P0), ("cam_ids", cam_ids),
                 ("cam_pos", cam_pos), ("pt_ids", pt_ids), ("pt_pos", pt_pos)):
        S[k] = v.numpy()
    Wc = torch.eye(3, dtype=D) * 25.0
    cases = {"plain": ({}, None),
             "kernel_weights": ({"kernel": [pp.optim.kernel.Huber(delta=1.0), None, None]}, [torch.eye(2, dtype=D), Wc, torch.eye(3, dtype=D) * 4.0])}
    for tag, (kw, weight) in cases.items():
        model = ReprojWithPriors(K0.clone(), C0.clone(), P0.clone())
        opt = pp.optim.LM(model, solver=pp.optim.solver.Cholesky(), strategy=pp.optim.strategy.TrustRegion(radius=1e4), min=1e-6, **kw)
        rec = {"loss": [], "damping": [], "reject": []}
        for _ in range(6):
            loss = opt.step((obs, cidx, pidx, cam_ids, cam_pos, pt_ids, pt_pos), weight=weight)
            rec["loss"].append(float(loss))
            rec["damping"].append(float(opt.param_groups[0]["damping"]))
            rec["reject"].append(int(opt.reject_count))
        for k, v in rec.items():
            S[f"{tag}/{k}"] = np.asarray(v)
        S[f"{tag}/K"], S[f"{tag}/C"], S[f"{tag}/P"] = model.K.detach().numpy(), model.C.detach().tensor().numpy(), model.P.detach().numpy()
        print(tag, rec)
    # Gauss-Newton (pseudo-inverse of the rectangular W J) on the same problem: without priors (gauge-free: minimum-norm
    # steps) and with the weighted priors
    for tag, use_priors in (("gn_plain", False), ("gn_priors", True)):
        if use_priors:
            model = ReprojWithPriors(K0.clone(), C0.clone(), P0.clone())
            args = (obs, cidx, pidx, cam_ids, cam_pos, pt_ids, pt_pos)
            weight = [torch.eye(2, dtype=D), Wc, torch.eye(3, dtype=D) * 4.0]
        else:
            model = Reproj(K0.clone(), C0.clone(), P0.clone())
            args, weight = (obs, cidx, pidx), None
        opt = pp.optim.GN(model)
        S[f"{tag}/loss"] = np.asarray([float(opt.step(args, weight=weight)) for _ in range(3)])
        S[f"{tag}/P"], S[f"{tag}/C"] = model.P.detach().numpy(), model.C.detach().tensor().numpy()
        print(tag, S[f"{tag}/loss"])
    np.savez_compressed(OUT, **S)


if __name__ == "__main__":
    main()
