"""Bundle adjustment on the MI355X -- the model of the reference's examples/module/ba/bundle_adjustment.py (intrinsics K,
camera poses C and points P as three `pp.Parameter`s gathered per observation, `@psjac` projection, LM(sparse=True) with
solver.PCG) on a synthetic BAL-style problem, with `import pypose_amd as pp`.

    python examples/ba.py --cameras 257 --points 65132 --per-point 4
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn

import pypose_amd as pp
from pypose_amd.autograd.function import psjac


class Reproj(nn.Module):
    def __init__(self, K, C, P):
        super().__init__()
        self.K = pp.Parameter(K, sjac=True)
        self.C = pp.Parameter(C, sjac=True)
        self.P = pp.Parameter(P, sjac=True)

    def forward(self, observe, cidx, pidx):
        return Reproj.project(self.K[cidx], self.C[cidx], self.P[pidx]) - observe

    @psjac
    def project(K, C, P):
        cp = C.Act(P)
        n = - cp[..., :2] / cp[..., [2]]
        radius = n.square().sum(dim=-1, keepdim=True)
        focal, k1, k2 = K[..., :1], K[..., 1:2], K[..., 2:3]
        distortion = 1 + k1 * radius + k2 * radius.square()
        return focal * distortion * n


def synthetic(Nc, Np, per_point, device, seed=0):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    P = (torch.randn(Np, 3, generator=g) * 0.5).to(device)
    base = torch.cat([torch.tensor([[0., 0., -4.]]).repeat(Nc, 1), pp.identity_SO3(Nc).tensor()], -1).to(device)
    C = pp.randn_SE3(Nc, sigma=0.15, device=device) @ pp.SE3(base)
    K = torch.stack([torch.full((Nc,), 500.), torch.full((Nc,), -0.05), torch.full((Nc,), 0.01)], -1).to(device)
    cidx = torch.randint(0, Nc, (Np * per_point,), generator=g).to(device)
    pidx = torch.arange(Np).repeat_interleave(per_point).to(device)
    with torch.no_grad():
        obs = Reproj.project(K[cidx], C[cidx], P[pidx]) + 0.2 * torch.randn(len(cidx), 2, generator=g).to(device)
    K0 = K * (1 + 0.002 * torch.randn(Nc, 3, generator=g).to(device))
    C0 = pp.randn_SE3(Nc, sigma=0.005, device=device) @ C
    P0 = P + 0.02 * torch.randn(Np, 3, generator=g).to(device)
    return (obs, cidx, pidx), (K0, C0, P0)


def main(argv=None):
    ap = argparse.ArgumentParser(description="Bundle adjustment")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--cameras", type=int, default=257)
    ap.add_argument("--points", type=int, default=65132)
    ap.add_argument("--per-point", type=int, default=4)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--solver", choices=["pcg", "cholesky"], default="pcg",
                    help="pcg: matrix-free PCG on the full system (the reference example); cholesky: exact Schur-complement solve")
    a = ap.parse_args(argv)
    args, (K0, C0, P0) = synthetic(a.cameras, a.points, a.per_point, a.device)
    model = Reproj(K0, C0, P0)
    solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250) if a.solver == "pcg" else pp.optim.solver.Cholesky()
    optimizer = pp.optim.LM(model, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4), reject=30, sparse=True)
    loss0 = float(optimizer.model.loss(args, None).detach())
    t0 = time.perf_counter()
    for k in range(a.steps):
        loss = optimizer.step(args)
    torch.cuda.synchronize()
    print(f"{a.cameras} cameras, {a.points} points, {args[0].shape[0]} observations: loss {loss0:.6g} -> {float(loss):.6g} in "
          f"{a.steps} LM steps, {(time.perf_counter() - t0) / a.steps * 1e3:.1f} ms/step on the '{optimizer.linearization}' path")
    return loss0, float(loss)


if __name__ == "__main__":
    main()
