"""The reference's example models, re-stated against pypose_amd (tests only)."""
import os
import numpy as np
import torch
from torch import nn

import pypose_amd as pp


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


def load_lm_golden():
    import os
    return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lm_golden.npz")))


def T(a, device="cpu"):
    return torch.from_numpy(np.array(a, copy=True)).to(device)


def run_steps(opt, args, kwargs, nsteps):
    rec = {"loss": [], "damping": [], "reject": [], "kind": []}
    for _ in range(nsteps):
        loss = opt.step(*args, **kwargs)
        rec["loss"].append(float(loss))
        rec["damping"].append(float(opt.param_groups[0].get("damping", 0.0)))
        rec["reject"].append(int(getattr(opt, "reject_count", 0)))
        rec["kind"].append(getattr(opt, "linearization", "?"))
    return rec


def invnet_cases(G, device="cpu"):
    """name -> (make optimizer+model, step args, kwargs, n steps) for every recorded InvNet run."""
    S = pp.optim.strategy
    inp = pp.SE3(T(G["invnet/input"], device))
    init = lambda key="invnet/init": pp.SE3(T(G[key], device))
    W = T(G["invnet/weight"], device)
    tgt = T(G["invnet/target"], device)
    return {
        "constant": (lambda net: pp.optim.LM(net, strategy=S.Constant(damping=1e-4)), init(), (inp,), {}, 6),
        "adaptive": (lambda net: pp.optim.LM(net, strategy=S.Adaptive(damping=1e-6)), init(), (inp,), {}, 6),
        "trustregion": (lambda net: pp.optim.LM(net, strategy=S.TrustRegion()), init(), (inp,), {}, 6),
        "huber_weight": (lambda net: pp.optim.LM(net, strategy=S.Adaptive(damping=1e-6), kernel=pp.optim.kernel.Huber(delta=0.5)),
                         init(), (inp,), {"weight": W}, 6),
        "cauchy_target": (lambda net: pp.optim.LM(net, strategy=S.TrustRegion(radius=1e4), kernel=pp.optim.kernel.Cauchy()),
                          init(), (inp, tgt), {}, 6),
        "gn": (lambda net: pp.optim.GN(net), init(), (inp,), {}, 4),
        "far": (lambda net: pp.optim.LM(net, strategy=S.TrustRegion(radius=1e8), min=1e-12), init("invnet/far_init"), (inp,), {}, 8),
    }


def compare_trajectory(rec, G, prefix, floor=1e-16, rtol=1e-6):
    """Loss sequence equal to the reference's while above the fp64 noise floor; same damping /
    reject sequence over that range."""
    ref = G[prefix + "/loss"]
    for k, (a, b) in enumerate(zip(rec["loss"], ref)):
        if b > floor:
            assert abs(a - b) <= rtol * b, (prefix, k, a, b)
            # accept/reject and the damping update are decided by the sign / size of (last - loss):
            # only meaningful while the step still changes the loss beyond rounding
            prev = ref[k - 1] if k else None
            if prev is None or abs(prev - b) > 1e-7 * b:
                assert np.isclose(rec["damping"][k], G[prefix + "/damping"][k], rtol=1e-12), (prefix, k)
                assert rec["reject"][k] == G[prefix + "/reject"][k], (prefix, k)
        else:
            assert a <= max(floor, 100 * b), (prefix, k, a, b)


class Reproj(nn.Module):                       # reference examples/module/ba/bundle_adjustment.py:16-43, verbatim structure
    def __init__(self, K, C, P):
        super().__init__()
        self.K = pp.Parameter(K, sjac=True)
        self.C = pp.Parameter(C, sjac=True)
        self.P = pp.Parameter(P, sjac=True)

    def forward(self, observe, cidx, pidx):
        return Reproj.project(self.K[cidx], self.C[cidx], self.P[pidx]) - observe

    @pp.autograd.function.psjac
    def project(K, C, P):
        cp = C.Act(P)
        n = - cp[..., :2] / cp[..., [2]]
        radius = n.square().sum(dim=-1, keepdim=True)
        focal, k1, k2 = K[..., :1], K[..., 1:2], K[..., 2:3]
        distortion = 1 + k1 * radius + k2 * radius.square()
        return focal * distortion * n


def load_ba_golden():
    return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ba_golden.npz")))


def ba_case(G, tag, device="cpu"):
    model = Reproj(T(G[f"{tag}/K0"], device), pp.SE3(T(G[f"{tag}/C0"], device)), T(G[f"{tag}/P0"], device))
    args = (T(G[f"{tag}/obs"], device), T(G[f"{tag}/cidx"], device), T(G[f"{tag}/pidx"], device))
    kernel = pp.optim.kernel.Huber(delta=1.0) if tag == "ba_huber" else None
    opt = pp.optim.LM(model, solver=pp.optim.solver.Cholesky(), strategy=pp.optim.strategy.TrustRegion(radius=1e4),
                      kernel=kernel, min=1e-6)
    return model, opt, args


class MixedGraph(nn.Module):
    """Three residuals over one node set: odometry edges, loop closures, position priors
    (tests/golden/make_multires_golden.py recorded the reference's dense LM on it)."""

    def __init__(self, nodes):
        super().__init__()
        self.nodes = pp.Parameter(nodes)

    def forward(self, odo, zodo, loop, zloop, pidx, prior):
        a = (zodo.Inv() @ self.nodes[odo[:, 0]].Inv() @ self.nodes[odo[:, 1]]).Log().tensor()
        b = (zloop.Inv() @ self.nodes[loop[:, 0]].Inv() @ self.nodes[loop[:, 1]]).Log().tensor()
        c = (prior.Inv() @ self.nodes[pidx]).Log().tensor()[..., :3]
        return a, b, c


def load_multires_golden():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "multires_golden.npz"))


def multires_case(G, tag, device="cpu", **lm_kw):
    D = torch.float64
    t = lambda k: T(G[k], device)
    args = (t("odo"), pp.SE3(t("zodo")), t("loop"), pp.SE3(t("zloop")), t("pidx"), pp.SE3(t("prior")))
    kw = {"plain": {}, "kernels": {"kernel": [None, pp.optim.kernel.Huber(delta=0.3), None]},
          "kernels_weights": {"kernel": [None, pp.optim.kernel.Cauchy(delta=0.5), None]}}[tag]
    weight = None
    if tag == "kernels_weights":
        weight = [torch.eye(6, dtype=D, device=device) * 2.0, t("Wloop"), torch.eye(3, dtype=D, device=device) * 10.0]
    model = MixedGraph(pp.SE3(t("init")))
    lm_kw.setdefault("solver", pp.optim.solver.Cholesky())
    opt = pp.optim.LM(model, strategy=pp.optim.strategy.TrustRegion(radius=1e4), min=1e-6, **kw, **lm_kw)
    return model, opt, args, weight


class ReprojWithPriors(nn.Module):
    """Bundle adjustment with position priors on some cameras and some points: three residuals over three parameters
    (tests/golden/make_ba_prior_golden.py recorded the reference's dense LM on it)."""

    def __init__(self, K, C, P):
        super().__init__()
        self.K = pp.Parameter(K)
        self.C = pp.Parameter(C)
        self.P = pp.Parameter(P)

    def forward(self, observe, cidx, pidx, cam_ids, cam_pos, pt_ids, pt_pos):
        reproj = Reproj.project(self.K[cidx], self.C[cidx], self.P[pidx]) - observe
        return reproj, self.C[cam_ids].translation() - cam_pos, self.P[pt_ids] - pt_pos


def ba_prior_case(tag, device="cpu", **lm_kw):
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ba_prior_golden.npz"))
    t = lambda k: T(G[k], device)
    D = torch.float64
    model = ReprojWithPriors(t("K0"), pp.SE3(t("C0")), t("P0"))
    args = (t("obs"), t("cidx"), t("pidx"), t("cam_ids"), t("cam_pos"), t("pt_ids"), t("pt_pos"))
    kw, weight = {}, None
    if tag == "kernel_weights":
        kw = {"kernel": [pp.optim.kernel.Huber(delta=1.0), None, None]}
        weight = [torch.eye(2, dtype=D, device=device), torch.eye(3, dtype=D, device=device) * 25.0, torch.eye(3, dtype=D, device=device) * 4.0]
    lm_kw.setdefault("solver", pp.optim.solver.Cholesky())
    opt = pp.optim.LM(model, strategy=pp.optim.strategy.TrustRegion(radius=1e4), min=1e-6, **kw, **lm_kw)
    return G, model, opt, args, weight
