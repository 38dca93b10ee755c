"""The reference's sparse-LM entry points (tests/optim/test_sparse_lm.py:10-149) re-stated against
pypose_amd: ``Parameter(sjac=True)``, ``psjac`` and ``LM(sparse=True, solver=PCG())`` are accepted and the
same two problems converge -- through the automatically detected block and pose-graph paths (the
reference needs its un-vendored ``bae`` plugin and CUDA for these)."""
import pytest
import torch
from torch import nn

import pypose_amd as pp
import pypose_amd.autograd.function as ppaf
import pypose_amd.optim.solver as ppos
from pypose_amd.autograd.function import parallel_for_sparse_jacobian, psjac
from tests.oracle_backend import oracle_backend


@parallel_for_sparse_jacobian
def edge_error(node1, node2, relpose):
    return (relpose.Inv() @ node1.Inv() @ node2).Log().tensor()


class SparseIdentityModel(nn.Module):
    def __init__(self, x0):
        super().__init__()
        self.x = pp.Parameter(x0, sjac=True)

    def forward(self):
        return self.x


class SparseChainPGO(nn.Module):
    def __init__(self, root, nodes):
        super().__init__()
        self.register_buffer("root", root)
        self.nodes = pp.Parameter(nodes, sjac=True)

    def forward(self, edges, relposes):
        nodes = torch.cat((self.root, self.nodes), dim=0)
        return edge_error(nodes[edges[:, 0]], nodes[edges[:, 1]], relposes)


def test_psjac_export():
    assert psjac is parallel_for_sparse_jacobian and psjac is ppaf.psjac


def _identity_model(device):
    torch.manual_seed(0)
    dtype = torch.float64
    x_true = torch.randn(8, 1, device=device, dtype=dtype)
    x0 = x_true + 0.1 * torch.randn_like(x_true)
    model = SparseIdentityModel(x0).to(device)
    opt = pp.optim.LM(model, solver=ppos.PCG(), strategy=pp.optim.strategy.Constant(damping=1e-6), sparse=True)
    with torch.no_grad():
        loss0 = opt.model.loss(input=(), target=x_true).item()
    for _ in range(6):
        loss = opt.step(input=(), target=x_true).item()
    assert loss < loss0 and opt.linearization == "block"
    torch.testing.assert_close(model.x.detach(), x_true, rtol=1e-4, atol=1e-4)


def _chain_pgo(device, n=3):
    torch.manual_seed(0)
    dtype = torch.float64
    gt = torch.zeros(n, 7, device=device, dtype=dtype)
    gt[:, 0] = torch.arange(n, device=device, dtype=dtype)
    gt[:, 6] = 1
    gt_nodes = pp.SE3(gt)
    edges = torch.stack([torch.arange(n - 1), torch.arange(1, n)], -1).to(device)
    relposes = gt_nodes[edges[:, 0]].Inv() @ gt_nodes[edges[:, 1]]
    # (a long open chain with independent 0.1 rad errors per node is far outside LM's basin: scale the noise)
    init = gt_nodes[1:] * pp.randn_SE3(n - 1, sigma=0.1 if n <= 12 else 0.01, device=device, dtype=dtype)
    model = SparseChainPGO(gt_nodes[:1], init).to(device)
    opt = pp.optim.LM(model, solver=ppos.PCG(tol=1e-10), strategy=pp.optim.strategy.Constant(damping=1e-4), sparse=True)
    with torch.no_grad():
        loss0 = opt.model.loss(input=(edges, relposes), target=None).item()
    for _ in range(5 if n == 3 else 12):                # n == 3 is the reference's case (it stops at the first
        loss = opt.step(input=(edges, relposes)).item()  # loss < 1e-5, which leaves ~1e-3 error for some seeds)
        if loss < 1e-14:
            break
    # (the reference's LM converges only linearly on chains -- identical for its dense path -- so the long
    #  chain is asked for a 100x reduction, the reference-sized ones for the reference's threshold)
    assert loss < loss0 and loss < (1e-5 if n <= 12 else 1e-2 * loss0)
    assert opt.linearization == "graph"                 # root + nodes concatenation is seen through
    if n <= 12:
        torch.testing.assert_close(pp.SE3(model.nodes.detach()).translation(), gt_nodes[1:].translation(), rtol=1e-4, atol=2e-4)


def test_sparse_identity_cpu():
    with oracle_backend():
        _identity_model("cpu")


@pytest.mark.parametrize("n", [3, 12])
def test_sparse_chain_pgo_cpu(n):
    with oracle_backend():
        _chain_pgo("cpu", n)


@pytest.mark.gpu
def test_sparse_identity_gpu():
    _identity_model("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("n", [3, 200])     # (open chains are ill-conditioned ~n^2 for Jacobi-preconditioned CG)
def test_sparse_chain_pgo_gpu(n):
    _chain_pgo("cuda:0", n)
