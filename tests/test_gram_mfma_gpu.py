"""pplie_block_gram_mfma (csrc/gram_mfma.hip): the normal equations of dense Jacobian blocks with a large residual dimension on
the matrix cores -- against torch's batched products, and inside Levenberg-Marquardt on a model with 30 residuals per problem."""
import ctypes

import pytest
import torch
from torch import nn

import pypose_amd as pp
from pypose_amd import _C
from pypose_amd.optim import blocks

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float64, 1e-12)])
@pytest.mark.parametrize("n,dr,dp", [(1, 1, 1), (5, 3, 2), (1000, 8, 6), (257, 64, 7), (4099, 30, 6), (33, 513, 15), (70, 100, 8)])
def test_gram_kernel_equals_batched_matmul(n, dr, dp, dtype, tol):
    torch.manual_seed(n + dr)
    J = torch.randn(n, dr, dp, dtype=dtype, device=DEV)
    R = torch.randn(n, dr, dtype=dtype, device=DEV)
    A = torch.empty(n, dp, dp, dtype=dtype, device=DEV)
    g = torch.empty(n, dp, dtype=dtype, device=DEV)
    rr = torch.empty(n, dtype=dtype, device=DEV)
    fn = _C.library().symbol("pplie_block_gram_mfma" + ("_f32" if dtype == torch.float32 else "_f64"), blocks._GRAM_SIG)
    assert fn(J.data_ptr(), R.data_ptr(), A.data_ptr(), g.data_ptr(), rr.data_ptr(), n, dr, dp, _C.stream_ptr(torch.device(DEV))) == 0
    Jd, Rd = J.double(), R.double()
    scale = float((Jd.mT @ Jd).abs().max())
    assert float((A.double() - Jd.mT @ Jd).abs().max()) <= tol * scale
    assert float((g.double() - (Jd.mT @ Rd.unsqueeze(-1)).squeeze(-1)).abs().max()) <= tol * scale
    assert float((rr.double() - Rd.square().sum(-1)).abs().max()) <= tol * max(1.0, float(Rd.square().sum(-1).max()))
    A2, g2 = blocks.normal_equations(J, R)                               # the block path's entry: MFMA beyond the register-kernel sizes
    if not (dr in blocks._HIP_DR and dp in blocks._HIP_DP):
        assert torch.equal(A2, A) and torch.equal(g2, g)


def test_lm_with_thirty_residuals_per_problem_matches_the_dense_path():
    class Align(nn.Module):                      # every pose aligns its own 10 points: residual [n, 30]
        def __init__(self, init, pts):
            super().__init__()
            self.pose, self.pts = pp.Parameter(init), pts

        def forward(self, target):
            return (self.pose.unsqueeze(-2).Act(self.pts) - target).flatten(-2)
    torch.manual_seed(0)
    n = 40
    truth = pp.randn_SE3(n, dtype=torch.float64, device=DEV)
    pts = torch.randn(n, 10, 3, dtype=torch.float64, device=DEV)
    tgt = truth.unsqueeze(-2).Act(pts) + 0.01 * torch.randn(n, 10, 3, dtype=torch.float64, device=DEV)
    init = pp.randn_se3(n, sigma=0.2, dtype=torch.float64, device=DEV).Exp() @ truth
    runs = {}
    for structured in (True, False):
        net = Align(init.clone(), pts)
        opt = pp.optim.LM(net, strategy=pp.optim.strategy.TrustRegion(radius=1e3))
        opt.structured = structured
        losses = [float(opt.step(tgt)) for _ in range(4)]
        runs[structured] = (losses, opt.linearization, net.pose.detach().tensor().clone())
    assert runs[True][1] == "block" and runs[False][1] == "dense"
    for a, b in zip(runs[True][0], runs[False][0]):
        assert abs(a - b) <= 1e-9 * b
    assert float((runs[True][2] - runs[False][2]).abs().max()) <= 1e-8
