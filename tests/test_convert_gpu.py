"""Conversion kernels (csrc/convert.hip) on the MI355X through the C ABI: goldens recorded from the real
reference (fp64), fp32 against the fp64 anchor, round-trip properties at 10^6 rows, and a g2o file driven
through the pose-graph LM path."""
import os

import numpy as np
import pytest
import torch

import pypose_amd as pp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def G():
    return dict(np.load(os.path.join(HERE, "golden", "convert_golden.npz")))


def T(a, dtype=torch.float64):
    return torch.from_numpy(np.array(a)).to(dtype).to(DEV)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 3e-6)])
def test_kernels_match_reference_goldens(G, dtype, tol):
    from pypose_amd import _C
    assert _C._test_backend is None
    R = T(G["mat2so3/R"], dtype).requires_grad_(True)
    q = pp.mat2SO3(R, check=False)
    np.testing.assert_allclose(q.tensor().detach().cpu().double().numpy(), G["mat2so3/q"], rtol=0, atol=tol)
    (gR,) = torch.autograd.grad(q.tensor(), R, T(G["mat2so3/g"], dtype))
    scale = np.abs(G["mat2so3/gR"]).max(axis=(-1, -2), keepdims=True) + 1
    assert (np.abs(gR.cpu().double().numpy() - G["mat2so3/gR"]) / scale).max() <= 20 * tol
    e = T(G["euler2so3/e"], dtype).requires_grad_(True)
    qe = pp.euler2SO3(e)
    np.testing.assert_allclose(qe.tensor().detach().cpu().double().numpy(), G["euler2so3/q"], rtol=0, atol=tol)
    (ge,) = torch.autograd.grad(qe.tensor(), e, T(G["euler2so3/g"], dtype))
    np.testing.assert_allclose(ge.cpu().double().numpy(), G["euler2so3/ge"], rtol=0, atol=10 * tol)
    Q = T(G["euler/Q"], dtype).requires_grad_(True)
    eu = pp.SO3(Q).euler()
    # (in fp32 the rows inside the singular band eps = 2e-4 of pitch = +-pi/2 are compared in fp64 only)
    rows = np.ones(len(G["euler/e"]), bool) if dtype == torch.float64 else np.abs(np.abs(G["euler/e"][:, 1]) - np.pi / 2) > 5e-2
    np.testing.assert_allclose(eu.detach().cpu().double().numpy()[rows], G["euler/e"][rows], rtol=0, atol=4 * tol)
    (gQ,) = torch.autograd.grad(eu, Q, T(G["euler/g"], dtype))
    ok = np.isfinite(G["euler/gQ"]).all(-1) & rows
    sc = np.abs(G["euler/gQ"][ok]).max(-1, keepdims=True) + 1
    assert (np.abs(gQ.cpu().double().numpy()[ok] - G["euler/gQ"][ok]) / sc).max() <= 50 * tol
    for fn, key, M in ((pp.mat2SE3, "mat2se3/X", "mat2se3/M"), (pp.mat2Sim3, "mat2sim3/X", "mat2sim3/M"),
                       (pp.mat2RxSO3, "mat2rxso3/X", "mat2rxso3/M")):
        np.testing.assert_allclose(fn(T(G[M], dtype), atol=1e-4).tensor().cpu().double().numpy(), G[key], rtol=0, atol=10 * tol)
    np.testing.assert_allclose(pp.euler(pp.se3(T(G["euler/se3"], dtype))).cpu().double().numpy(), G["euler/se3_e"], rtol=0, atol=10 * tol)


def test_edge_sizes_and_vmap():
    assert pp.mat2SO3(torch.zeros(0, 3, 3, device=DEV), check=False).shape == (0, 4)
    assert pp.euler2SO3(torch.zeros(0, 3, device=DEV)).shape == (0, 4)
    for n in (1, 255, 257, 4099):
        X = pp.randn_SO3(n, device=DEV, dtype=torch.float64)
        q = pp.mat2SO3(X.matrix())
        sign = torch.sign((q.tensor() * X.tensor()).sum(-1, keepdim=True))       # q and -q are the same rotation
        assert (q.tensor() * sign - X.tensor()).abs().max().item() < 1e-12
    # jacobian(vectorize=True) drives the backward under vmap
    e = torch.randn(5, 3, device=DEV, dtype=torch.float64)
    J = torch.autograd.functional.jacobian(lambda t: pp.euler2SO3(t).tensor().sum(0), e, vectorize=True)
    J2 = torch.autograd.functional.jacobian(lambda t: pp.euler2SO3(t).tensor().sum(0), e, vectorize=False)
    torch.testing.assert_close(J, J2)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 2e-5)])
def test_round_trips_at_one_million_rows(dtype, tol):
    n = 1_000_000
    torch.manual_seed(3)
    X = pp.randn_SO3(n, device=DEV, dtype=dtype, sigma=2.0)
    R = X.matrix()
    q = pp.mat2SO3(R, check=False)
    assert (q.matrix() - R).abs().max().item() <= tol                  # matrix -> quaternion -> matrix
    assert (q.tensor().norm(dim=-1) - 1).abs().max().item() <= tol
    e = X.euler()
    back = pp.euler2SO3(e)
    # inside the band |sin(pitch)| >= 1 - 2e-4 the reference's euler() sets roll = 0: an O(sqrt(2 eps)) = 0.02 rad error by design
    assert (back.matrix() - R).abs().max().item() <= 0.05
    away = (e[:, 1].abs() < 1.4)
    assert (back.matrix() - R)[away].abs().max().item() <= 100 * tol


def test_g2o_file_through_pose_graph_lm():
    d = pp.io.read_g2o(os.path.join(HERE, "golden", "sample.g2o"), device=DEV, dtype=torch.float64)
    remap = {int(i): k for k, i in enumerate(d["ids"])}                  # vertex ids -> rows
    nodes = d["nodes"]
    edges = torch.stack([torch.arange(20) % 12, (torch.arange(20) * 5 + 1) % 12], -1).to(DEV)
    assert len(remap) == 12 and torch.equal(edges, d["edges"])
    # measurements consistent with the vertices + noise: the optimum reproduces the vertices up to gauge
    rel = nodes[edges[:, 0]].Inv() @ nodes[edges[:, 1]]

    class PoseGraph(torch.nn.Module):
        def __init__(self, init):
            super().__init__()
            self.nodes = pp.Parameter(init)

        def forward(self, edges, poses):
            n1, n2 = self.nodes[edges[..., 0]], self.nodes[edges[..., 1]]
            return (poses.Inv() @ n1.Inv() @ n2).Log().tensor()

    torch.manual_seed(0)
    graph = PoseGraph(nodes @ pp.randn_SE3(12, sigma=0.05, device=DEV, dtype=torch.float64))
    opt = pp.optim.LM(graph, strategy=pp.optim.strategy.TrustRegion(radius=1e4))
    l0 = float(opt.model.loss((edges, rel), None).detach())
    for _ in range(8):
        loss = opt.step((edges, rel), weight=d["infos"])
    assert opt.linearization == "fused:pgo" and float(loss) < 1e-12 * l0
