"""SURVEY.md section 8(f) rank 3, the robust-weighting half: closed-form rho / rho' in HIP kernels (csrc/robust.h, robust.hip,
the robust entries of pgo_fused.hip) against the reference's formulation -- autograd through kernel(x).sum() and element-wise
scaling (pypose/optim/corrector.py:69-96, 132-167; pypose/optim/kernel.py) -- and against the reference package itself."""
import numpy as np
import pytest
import torch

import pypose_amd as pp
from pypose_amd.optim import corrector as C
from pypose_amd.optim import kernel as K
from tests.optim_models import PoseGraph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KERNELS = [("Huber", (0.7,)), ("PseudoHuber", (0.9,)), ("Cauchy", (1.3,)), ("SoftLOne", (0.8,)), ("Arctan", (1.1,)), ("Scale", (0.4,)),
           ("Tolerant", (1.5, -0.6))]


def unfused(name, args):
    """the same kernel as a SUBCLASS: robust_code() is None for it, so everything takes the torch / autograd route"""
    return type(name + "Torch", (getattr(K, name),), {})(*args)


@pytest.mark.parametrize("name,args", KERNELS)
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_rho_kernel_matches_the_torch_formula(name, args, dtype):
    torch.manual_seed(0)
    x = torch.cat([torch.zeros(3), torch.rand(4000) * 4, torch.tensor([0.49, 0.4900001, 0.81, 1e-12, 1e4])]).to(dtype).to(DEV)
    got = getattr(K, name)(*args)(x)
    want = unfused(name, args)(x.double()).to(dtype)
    tol = 1e-12 if dtype == torch.float64 else 2e-6
    assert (got - want).abs().max() <= tol * max(1.0, float(want.abs().max()))
    assert K.robust_code(getattr(K, name)(*args)) is not None and K.robust_code(unfused(name, args)) is None


@pytest.mark.parametrize("name,args", KERNELS)
@pytest.mark.parametrize("corr", ["FastTriggs", "Triggs"])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_fused_corrector_matches_the_autograd_corrector(name, args, corr, dtype, monkeypatch):
    if corr == "Triggs" and name == "Scale":
        with pytest.raises(RuntimeError):          # rho' is a constant: the reference's second autograd.grad raises, so do we
            C.Triggs(K.Scale(*args))(R=torch.randn(4, 3, dtype=dtype, device=DEV), J=torch.randn(4, 3, 5, dtype=dtype, device=DEV))
        return
    calls = []
    real = C.fused_scale_rows
    monkeypatch.setattr(C, "fused_scale_rows", lambda *a, **k: (lambda r: (calls.append(r is not None), r)[1])(real(*a, **k)))
    torch.manual_seed(1)
    for shape_J, dr in (((500, 6, 12), 6), ((501, 2, 9), 2), ((64, 2, 6, 6), 6), ((300 * 3, 40), 3)):
        n = shape_J[0] if len(shape_J) != 2 else shape_J[0] // dr
        R = (torch.randn(n, dr, dtype=dtype, device=DEV) * torch.rand(n, 1, dtype=dtype, device=DEV) * 2)
        R[0] = 0                                               # x = 0 row
        J = torch.randn(shape_J, dtype=dtype, device=DEV)
        fused = getattr(C, corr)(getattr(K, name)(*args))
        plain = getattr(C, corr)(unfused(name, args))
        calls.clear()
        Rf, Jf = fused(R=R, J=J.clone())
        assert calls == [True], "the built-in kernel did not take the one-launch route"
        if len(shape_J) == 4:                                  # edge blocks [E, K, dr, m]: the reference's layout is [E, dr, K m]
            Rp, Jp = plain(R=R, J=J.permute(0, 2, 1, 3).reshape(n, dr, -1))
            Jp = Jp.reshape(n, dr, shape_J[1], shape_J[3]).permute(0, 2, 1, 3)
        else:
            Rp, Jp = plain(R=R, J=J.clone())
        tol = 1e-12 if dtype == torch.float64 else 3e-6
        assert (Rf - Rp).abs().max() <= tol * max(1.0, float(Rp.abs().max()))
        assert (Jf - Jp.reshape(Jf.shape)).abs().max() <= tol * max(1.0, float(Jp.abs().max()))
    # in place on a tensor the caller owns; the input of the default call is left alone
    J0 = torch.randn(100, 6, 12, dtype=dtype, device=DEV); keep = J0.clone()
    R0 = torch.randn(100, 6, dtype=dtype, device=DEV)
    _, J1 = fused(R=R0, J=J0)
    assert torch.equal(J0, keep) and J1.data_ptr() != J0.data_ptr()
    _, J2 = fused(R=R0, J=J0, inplace=True)
    assert J2.data_ptr() == J0.data_ptr() and torch.equal(J2, J1)


def _graph(n, e, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    gt = pp.cumprod(pp.randn_SE3(n, sigma=0.3, dtype=dtype, device=DEV), dim=0, left=False)
    chain = torch.stack([torch.arange(n - 1), torch.arange(1, n)], -1)
    extra = torch.randint(0, n, (e - (n - 1), 2), generator=g)
    extra[:, 1] = torch.where(extra[:, 0] == extra[:, 1], (extra[:, 1] + 1) % n, extra[:, 1])
    ed = torch.cat([chain, extra], 0).to(DEV)
    rel = gt[ed[:, 0]].Inv() @ gt[ed[:, 1]] @ pp.randn_SE3(e, sigma=0.02, dtype=dtype, device=DEV)
    # gross outliers on a tenth of the closures: what the kernel is for
    bad = torch.arange(n - 1, e, 10, device=DEV)
    rel = pp.SE3(rel.tensor().index_copy(0, bad, (rel[bad] @ pp.randn_SE3(len(bad), sigma=1.0, dtype=dtype, device=DEV)).tensor()))
    init = gt @ pp.randn_SE3(n, sigma=0.05, dtype=dtype, device=DEV)
    return ed, rel, init


@pytest.mark.parametrize("name,args", [("Huber", (0.3,)), ("Cauchy", (0.5,)), ("Tolerant", (0.5, -0.2))])
def test_pose_graph_lm_with_a_fused_robust_kernel(name, args):
    """LM on a pose graph with outlier closures: the kernel rides inside pplie_pgo_linearize_robust / pplie_pgo_residual_robust
    (kind 'fused:pgo', no corrector pass); same per-step losses and damping as the autograd-corrector route (the same kernel as
    a subclass) to fp64 rounding"""
    ed, rel, init = _graph(400, 1400, torch.float64)
    out = {}
    for tag, kern in (("fused", getattr(K, name)(*args)), ("torch", unfused(name, args))):
        g = PoseGraph(init.clone())
        opt = pp.optim.LM(g, solver=pp.optim.solver.PCG(tol=1e-12, maxiter=2000), strategy=pp.optim.strategy.TrustRegion(radius=1e4),
                          kernel=kern)
        losses, damp = [], []
        for _ in range(5):
            losses.append(float(opt.step((ed, rel))))
            damp.append(float(opt.param_groups[0]["damping"]))
        out[tag] = (losses, damp, opt.linearization, getattr(opt, "_last_lin_robust", None))
    assert out["fused"][2] == "fused:pgo"
    lf, lt = np.array(out["fused"][0]), np.array(out["torch"][0])
    assert np.abs(lf - lt).max() <= 1e-9 * lt.max(), (lf, lt)
    assert out["fused"][1] == out["torch"][1]
    assert lf[-1] < lf[0]


def test_pose_graph_lm_robust_matches_the_reference_package():
    """small graph, fp64: this package (fused robust kernels) against the reference itself (dense LM, FastTriggs by autograd)
    on the same device"""
    from oracle.ref_loader import load as load_reference
    rpp = load_reference()
    ed, rel, init = _graph(40, 100, torch.float64, seed=3)

    class RefGraph(torch.nn.Module):
        def __init__(self, nodes):
            super().__init__()
            self.nodes = rpp.Parameter(nodes)

        def forward(self, e, poses):
            return (poses.Inv() @ self.nodes[e[..., 0]].Inv() @ self.nodes[e[..., 1]]).Log().tensor()
    ro = rpp.optim.LM(RefGraph(rpp.SE3(init.tensor().clone())), strategy=rpp.optim.strategy.TrustRegion(radius=1e4),
                      kernel=rpp.optim.kernel.Huber(0.3))
    oo = pp.optim.LM(PoseGraph(init.clone()), solver=pp.optim.solver.PCG(tol=1e-13, maxiter=4000),
                     strategy=pp.optim.strategy.TrustRegion(radius=1e4), kernel=K.Huber(0.3))
    for _ in range(4):
        lr = float(ro.step((ed, rpp.SE3(rel.tensor()))))
        lo = float(oo.step((ed, rel)))
        assert abs(lr - lo) <= 1e-8 * abs(lr), (lr, lo)
    assert oo.linearization == "fused:pgo"


@pytest.mark.parametrize("corr", ["none", "user", "subclass", "user_kernel"])
def test_robust_kernel_with_foreign_corrector_is_never_captured(corr):
    """a built-in robust kernel gives the fused pose-graph linearisation a `fast_loss` even when the
    corrector is not FastTriggs / Triggs over a built-in kernel.  The captured trial (optim/pgograph.py) evaluates the PLAIN loss,
    so such a problem must stay un-captured past PgoGraphStep.MIN_STREAK and reproduce the path that never considers capture."""
    from pypose_amd.optim.pgograph import PgoGraphStep
    from tests.optim_models import run_steps
    torch.manual_seed(5)
    N, E = 60, 150
    gt = pp.randn_SE3(N, sigma=0.3, device=DEV).cumprod(dim=0)
    idx = torch.cat([torch.stack([torch.arange(N - 1), torch.arange(1, N)], 1), torch.randint(0, N, (E - N + 1, 2))]).to(DEV)
    idx = idx[idx[:, 0] != idx[:, 1]]
    rel = gt[idx[:, 0]].Inv() * gt[idx[:, 1]] * pp.randn_SE3(idx.shape[0], sigma=0.05, device=DEV)
    init = gt * pp.randn_SE3(N, sigma=0.05, device=DEV)

    class UserCorrector(torch.nn.Module):
        def forward(self, R, J):
            return R * 0.5, J * 0.5

    def corrector():
        k = K.Huber(0.3)
        return {"none": None, "user": UserCorrector(), "subclass": type("FT2", (C.FastTriggs,), {})(k),
                "user_kernel": C.FastTriggs(unfused("Huber", (0.3,)))}[corr]

    runs = {}
    for consider in (True, False):
        graph = PoseGraph(pp.SE3(init.tensor().clone()))
        opt = pp.optim.LM(graph, solver=pp.optim.solver.PCG(tol=1e-6, maxiter=500), strategy=pp.optim.strategy.TrustRegion(radius=1e4),
                          kernel=K.Huber(0.3), corrector=corrector(), min=1e-6)
        opt.graph_step = consider
        rec = run_steps(opt, ((idx, rel),), {}, PgoGraphStep.MIN_STREAK + 5)
        assert opt.__dict__.get('_pgo_graph_step') is None, "a robust-loss problem was captured with the plain-loss trial tail"
        runs[consider] = (rec, graph.nodes.detach().tensor().clone())
    a, b = runs[True][0], runs[False][0]
    assert a["loss"] == b["loss"] and a["damping"] == b["damping"] and a["reject"] == b["reject"], (a, b)
    assert torch.equal(runs[True][1], runs[False][1])
