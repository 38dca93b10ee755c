"""Helpers of the lm_golden2 tests (CPU and GPU)."""
import os

import numpy as np
import torch

import pypose_amd as pp
from tests.optim_models import InvNet, T, load_lm_golden

_G2 = None


def G2():
    global _G2
    if _G2 is None:
        _G2 = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lm_golden2.npz")))
    return _G2


def invnet_problem(G, B, device="cpu", dtype=torch.float64):
    """the generator of tests/golden/make_lm_golden2.py (CPU generator, fp64), checked against the recorded head rows"""
    D = torch.float64
    g = torch.Generator().manual_seed(int(G[f"invnet{B}/seed"]))

    def se3():
        d = torch.randn(B, 3, dtype=D, generator=g)
        d = d / d.norm(dim=-1, keepdim=True)
        return torch.cat([torch.randn(B, 3, dtype=D, generator=g), d * torch.randn(B, 1, dtype=D, generator=g)], -1)
    xi, xp = se3(), se3()
    from oracle import lie_np
    inp = torch.from_numpy(lie_np.se3_exp_fwd(xi.numpy())[0])
    init = torch.from_numpy(lie_np.se3_exp_fwd(xp.numpy())[0])
    np.testing.assert_allclose(inp[:4].numpy(), G[f"invnet{B}/input_head"], atol=1e-12)
    np.testing.assert_allclose(init[:4].numpy(), G[f"invnet{B}/init_head"], atol=1e-12)
    return pp.SE3(inp.to(dtype).to(device)), pp.SE3(init.to(dtype).to(device))


def robust_case(G, name, device="cpu"):
    G1 = load_lm_golden()
    inp, init = pp.SE3(T(G1["invnet/input"], device)), pp.SE3(T(G1["invnet/init"], device))
    K = pp.optim.kernel
    kw = {
        "pseudohuber": dict(kernel=K.PseudoHuber(delta=0.7)),
        "softlone": dict(kernel=K.SoftLOne(delta=0.9)),
        "arctan": dict(kernel=K.Arctan(delta=1.3)),
        "tolerant": dict(kernel=K.Tolerant(a=1.5, b=-0.8)),
        "triggs_huber": dict(kernel=K.Huber(delta=0.5), corrector=pp.optim.corrector.Triggs(K.Huber(delta=0.5))),
        "triggs_cauchy": dict(kernel=K.Cauchy(delta=0.8), corrector=pp.optim.corrector.Triggs(K.Cauchy(delta=0.8))),
        "lstsq_lm": dict(solver=pp.optim.solver.LSTSQ()),
        "lstsq_gn": dict(solver=pp.optim.solver.LSTSQ()),
    }[name]
    net = InvNet(init)
    if name == "lstsq_gn":
        return net, pp.optim.GN(net, **kw), inp, 4
    return net, pp.optim.LM(net, strategy=pp.optim.strategy.Adaptive(damping=1e-6), **kw), inp, 5


def compare2(rec, G, prefix, floor=1e-16, rtol=1e-6, damping=True, atol=0.0):
    """loss sequence equal to the reference's while above the noise floor; damping / reject sequence over that range.
    ``atol``: absolute rounding noise of the loss itself (kernels written as a difference of nearly equal numbers)"""
    ref = G[prefix + "/loss"]
    for k, (a, b) in enumerate(zip(rec["loss"], ref)):
        if b > floor:
            assert abs(a - b) <= rtol * b + atol, (prefix, k, a, b, rec["loss"], ref)
            prev = ref[k - 1] if k else None
            if damping and (prev is None or abs(prev - b) > 1e-7 * b):
                assert np.isclose(rec["damping"][k], G[prefix + "/damping"][k], rtol=1e-12), (prefix, k, rec["damping"], G[prefix + "/damping"])
                assert rec["reject"][k] == G[prefix + "/reject"][k], (prefix, k)
        else:
            assert a <= max(floor, 100 * b), (prefix, k, a, b)
