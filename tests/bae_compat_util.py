"""Shared cases for the `bae` plugin stand-in (pypose_amd/compat/bae): the reference's OWN sparse-LM code path
(pypose/optim/optimizer.py:629-664) driven through it.  The models mirror the reference's tests/optim/test_sparse_lm.py
(identity, chain PGO) plus a two-parameter reprojection problem (examples/module/ba pattern)."""
import torch
from torch import nn

from oracle import ref_loader
from pypose_amd.compat import install_bae


def load_reference():
    install_bae()
    return ref_loader.load()


def models(pp):
    from pypose.autograd.function import psjac

    @psjac
    def edge_error(n1, n2, rel):
        return (rel.Inv() @ n1.Inv() @ n2).Log().tensor()

    class Identity(nn.Module):
        def __init__(self, x0, sjac=True):
            super().__init__()
            self.x = pp.Parameter(x0, sjac=sjac)

        def forward(self):
            return self.x

    class Chain(nn.Module):
        def __init__(self, root, nodes, sjac=True):
            super().__init__()
            self.register_buffer("root", root)
            self.nodes = pp.Parameter(nodes, sjac=sjac)

        def forward(self, edges, rel):
            nodes = torch.cat((self.root, self.nodes), dim=0)
            return edge_error(nodes[edges[:, 0]], nodes[edges[:, 1]], rel)

    @psjac
    def reproject(pose, point, pixel, f):
        q = pose.Act(point)
        return f * q[..., :2] / q[..., 2:] - pixel

    class Reproj(nn.Module):
        def __init__(self, poses, points, sjac=True):
            super().__init__()
            self.poses = pp.Parameter(poses, sjac=sjac)
            self.points = pp.Parameter(points, sjac=sjac) if sjac else nn.Parameter(points)

        def forward(self, cam, pt, pixel, f):
            return reproject(self.poses[cam], self.points[pt], pixel, f)

    return Identity, Chain, Reproj


def chain_problem(pp, n=12, dtype=torch.float64, device="cpu", seed=0):
    g = torch.Generator().manual_seed(seed)
    t = torch.zeros(n, 7, dtype=dtype)
    t[:, 0] = torch.arange(n, dtype=dtype)
    t[:, 6] = 1
    gt = pp.SE3(t) * pp.SE3(pp.se3(0.3 * torch.randn(n, 6, generator=g, dtype=dtype)).Exp().tensor())
    i = torch.arange(n - 1)
    edges = torch.cat([torch.stack([i, i + 1], 1), torch.stack([i[:-1], i[:-1] + 2], 1)], 0)
    rel = gt[edges[:, 0]].Inv() @ gt[edges[:, 1]]
    init = gt[1:] * pp.se3(0.1 * torch.randn(n - 1, 6, generator=g, dtype=dtype)).Exp()
    mv = lambda x: x.to(device)
    return mv(gt), mv(edges), mv(rel), mv(init)


def reproj_problem(pp, C=4, N=30, dtype=torch.float64, device="cpu", seed=1):
    g = torch.Generator().manual_seed(seed)
    pts = torch.randn(N, 3, generator=g, dtype=dtype)
    pts[:, 2] += 6.0
    poses = pp.se3(0.1 * torch.randn(C, 6, generator=g, dtype=dtype)).Exp()
    cam = torch.arange(C).repeat_interleave(N)
    pt = torch.arange(N).repeat(C)
    f = 500.0
    q = poses[cam].Act(pts[pt])
    pixel = f * q[..., :2] / q[..., 2:]
    poses0 = poses * pp.se3(0.02 * torch.randn(C, 6, generator=g, dtype=dtype)).Exp()
    pts0 = pts + 0.05 * torch.randn(N, 3, generator=g, dtype=dtype)
    mv = lambda x: x.to(device)
    return mv(poses0), mv(pts0), mv(cam), mv(pt), mv(pixel), f


def ba_example(pp, sjac, NC=5, NP=40, dtype=torch.float64, device="cpu"):
    """examples/module/ba/bundle_adjustment.py:16-43 as written there (three tracked parameters: intrinsics, camera poses,
    points; dict input), on a synthetic BAL-shaped problem"""
    from pypose.autograd.function import psjac

    class Reproj(nn.Module):
        def __init__(self, K, C, P):
            super().__init__()
            self.K = pp.Parameter(K, sjac=True) if sjac else nn.Parameter(K)
            self.C = pp.Parameter(C, sjac=sjac)
            self.P = pp.Parameter(P, sjac=True) if sjac else nn.Parameter(P)

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

    g = torch.Generator().manual_seed(0)
    P = torch.randn(NP, 3, generator=g, dtype=dtype)
    P[:, 2] -= 6
    C = pp.se3(0.1 * torch.randn(NC, 6, generator=g, dtype=dtype)).Exp()
    K = torch.tensor([[500., 1e-2, 1e-4]], dtype=dtype).repeat(NC, 1)
    cidx, pidx = torch.arange(NC).repeat_interleave(NP), torch.arange(NP).repeat(NC)
    obs = Reproj.project(K[cidx], C[cidx], P[pidx])
    C0 = C * pp.se3(0.01 * torch.randn(NC, 6, generator=g, dtype=dtype)).Exp()
    P0 = P + 0.02 * torch.randn(NP, 3, generator=g, dtype=dtype)
    K0 = K * (1 + 0.01 * torch.randn(NC, 3, generator=g, dtype=dtype))
    inp = {"observe": obs.to(device), "cidx": cidx.to(device), "pidx": pidx.to(device)}
    return Reproj(K0.to(device), C0.to(device), P0.to(device)).to(device), inp
