"""Shared cases of the closed-form reprojection linearisation (csrc/reproj.hip, function/geometry.py, optim/multigraph.py)
against tests/golden/reproj_golden.npz (recorded from the reference by tests/golden/make_reproj_golden.py)."""
import os

import numpy as np
import torch

import pypose_amd as pp

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reproj_golden.npz"))


def t(key, dev="cpu", dtype=None):
    v = torch.from_numpy(G[key])
    return v.to(device=dev, dtype=dtype) if dtype is not None and v.is_floating_point() else v.to(dev)


def cam_rows(K, uv):
    return torch.cat([K.reshape(1, 9).expand(uv.shape[0], 9), uv], -1).contiguous()


def check_ops(dev, dtype, rtol):
    """the three row ops against the reference's residuals and autograd Jacobian blocks"""
    from pypose_amd import _C
    X, p, K, uv = t("X", dev, dtype), t("p", dev, dtype), t("K", dev, dtype), t("uv", dev, dtype)
    cam = cam_rows(K, uv)
    r, J = _C.row_op("se3_reproj_lin", [X.contiguous(), p.contiguous(), cam], (2, 18))
    (r2,) = _C.row_op("se3_reproj_fwd", [X.contiguous(), p.contiguous(), cam], (2,))
    J = J.reshape(-1, 2, 9).cpu().double()
    want_r, want_Jx, want_Jp = t("r"), t("J_pose"), t("J_point")
    sc = lambda w: rtol * float(w.abs().max())
    assert float((r.cpu().double() - want_r).abs().max()) <= sc(want_r) + rtol * float(uv.abs().max() + 700)
    assert float((r - r2).abs().max()) <= 1e-12 * float(r.abs().max())     # (one kernel body on the device: bit-equal there)
    assert float((J[..., :6] - want_Jx).abs().max()) <= sc(want_Jx)
    assert float((J[..., 6:] - want_Jp).abs().max()) <= sc(want_Jp)
    g = torch.randn(X.shape[0], 2, dtype=dtype, generator=torch.Generator().manual_seed(1)).to(dev)
    gX, gp = _C.row_op("reproj_vjp", [J.reshape(-1, 18).to(device=dev, dtype=dtype).contiguous(), g], (7, 3))
    want = torch.einsum("na,nac->nc", g.cpu().double(), J)
    assert float((gX.cpu().double()[:, :6] - want[:, :6]).abs().max()) <= sc(want) and float(gX[:, 6].abs().max()) == 0
    assert float((gp.cpu().double() - want[:, 6:]).abs().max()) <= sc(want)
    # the clamp of homo2cart (depth 0, +-denormal): value and (zero) depth-gradient as the reference's autograd gives them
    if dtype == torch.float64:
        pc, Kc = t("clamp_p", dev, dtype), t("clamp_K", dev, dtype)
        Xi = torch.tensor([[0., 0, 0, 0, 0, 0, 1]], dtype=dtype, device=dev).repeat(3, 1)
        rc, Jc = _C.row_op("se3_reproj_lin", [Xi, pc.contiguous(), cam_rows(Kc, torch.zeros(3, 2, dtype=dtype, device=dev))], (2, 18))
        np.testing.assert_allclose(rc.cpu().numpy(), G["clamp_r"], rtol=1e-12)
        np.testing.assert_allclose(Jc.reshape(3, 2, 9)[..., 6:].cpu().numpy(), G["clamp_J_point"], rtol=1e-12)


class BA(torch.nn.Module):
    def __init__(self, poses, points):
        super().__init__()
        self.poses = pp.Parameter(poses)
        self.points = torch.nn.Parameter(points)

    def forward(self, cidx, pidx, pixels, K):
        return pp.reprojerr(self.points[pidx].unsqueeze(-2), pixels.unsqueeze(-2), K, self.poses[cidx]).squeeze(-2)


def run_ba(dev, dtype, closed_form, kernel=None, steps=4):
    model = BA(pp.SE3(t("ba/poses0", dev, dtype)), t("ba/points0", dev, dtype)).to(dev)
    kw = {"kernel": kernel} if kernel is not None else {}
    opt = pp.optim.LM(model, strategy=pp.optim.strategy.TrustRegion(radius=1e4), min=1e-6, **kw)
    opt.closed_form = closed_form
    inp = (t("ba/cidx", dev), t("ba/pidx", dev), t("ba/pixels", dev, dtype), t("ba/K", dev, dtype))
    losses = [float(opt.step(inp)) for _ in range(steps)]
    return opt, model, losses
