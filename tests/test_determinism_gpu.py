"""Run-to-run reproducibility of the pose-graph path.

The reference assembles J^T J on the CPU with `index_add` / dense matmul: deterministic.  Here

* the single-process path (node-parallel CSR assembly, ordered all-gather of the PCG partial sums in the one-launch solve)
  has no floating-point atomics on anything that reaches the parameters: repeated runs are BIT-identical, asserted below;
* the scatter-add assembly used by edge shards (`pplie_graph_assemble`, hardware fp32 atomics, `-munsafe-fp-atomics`) sums a
  node's incident blocks in arrival order: repeated runs differ by re-association only, bounded here by
  `multiplicity * eps * sum |terms|` (asserted: 4 eps * multiplicity relative to the largest entry).
"""
import ctypes

import numpy as np
import pytest
import torch

import pypose_amd as pp
from pypose_amd import _C
from tests.optim_models import PoseGraph, run_steps

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")


def _chain_graph(N, E, seed, dtype):
    """an odometry chain plus random closures, noisy measurements, perturbed start (built on the device)"""
    g = torch.Generator().manual_seed(seed)
    truth = pp.cumprod(pp.randn_SE3(N, sigma=0.3, dtype=dtype, device=DEV), dim=0)
    i = torch.cat([torch.arange(N - 1), torch.randint(0, N, (E - N + 1,), generator=g)])
    j = torch.cat([torch.arange(1, N), torch.randint(0, N, (E - N + 1,), generator=g)])
    keep = i != j
    i, j = i[keep].to(DEV), j[keep].to(DEV)
    rel = truth[i].Inv() @ truth[j] @ pp.randn_SE3(len(i), sigma=0.01, dtype=dtype, device=DEV)
    init = truth @ pp.randn_SE3(N, sigma=0.05, dtype=dtype, device=DEV)
    return torch.stack([i, j], 1), rel.tensor(), init.tensor()


def _run(edges, rel, init, solver):
    graph = PoseGraph(pp.SE3(init.clone().to(DEV)))
    opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4))
    rec = run_steps(opt, ((edges.to(DEV), pp.SE3(rel.to(DEV))),), {}, 3)
    torch.cuda.synchronize()
    return rec, graph.nodes.detach().tensor().clone()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_single_process_pose_graph_lm_is_bit_reproducible(dtype):
    torch.manual_seed(3)
    edges, rel, init = _chain_graph(1500, 6000, 3, dtype)
    (a, pa), (b, pb) = (_run(edges, rel, init, pp.optim.solver.PCG(tol=1e-6, maxiter=500)) for _ in range(2))
    assert a["kind"][-1] == "fused:pgo"
    assert a["loss"] == b["loss"] and a["damping"] == b["damping"] and a["reject"] == b["reject"]
    assert torch.equal(pa, pb)


def test_scatter_add_assembly_repeats_within_reassociation_bound():
    """`pplie_graph_assemble_f32` (the edge-shard path): 64 edges per node on average, fp32 atomics"""
    torch.manual_seed(0)
    E, N, m = 200_000, 3_000, 6
    J = torch.randn(E, 2, m, m, device=DEV)
    R = torch.randn(E, m, device=DEV)
    idx = torch.randint(0, N, (E, 2), device=DEV)
    sig = [ctypes.c_void_p] * 7 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    fn = _C.library().symbol("pplie_graph_assemble_f32", sig)
    outs = []
    for _ in range(3):
        B = torch.zeros(N, m, m, device=DEV)
        g = torch.zeros(N, m, device=DEV)
        with _C._on_device(DEV):
            _C.check(fn(J.data_ptr(), None, R.data_ptr(), idx.data_ptr(), B.data_ptr(), g.data_ptr(), None, E, m, m, 2,
                        _C.stream_ptr(DEV)), "pplie_graph_assemble")
        torch.cuda.synchronize()
        outs.append((B, g))
    mult = int(torch.bincount(idx.reshape(-1), minlength=N).max())
    eps = np.finfo(np.float32).eps
    want = torch.zeros(N, m, m, device=DEV, dtype=torch.float64)
    for k in range(2):
        want.index_add_(0, idx[:, k], (J[:, k].mT.double() @ J[:, k].double()))
    for B, g in outs:
        assert (B.double() - want).abs().max().item() <= 4 * eps * mult * want.abs().max().item()
    for (B1, g1), (B2, g2) in zip(outs, outs[1:]):
        assert (B1 - B2).abs().max().item() <= 4 * eps * mult * B1.abs().max().item()
        assert (g1 - g2).abs().max().item() <= 4 * eps * mult * g1.abs().max().item() * m
