"""solver.CG against solutions recorded from the reference's CG (tests/golden/make_cg_golden.py; reference solver.py:219-340):
the same bits for every look interval -- the stop test runs on the device and gates the updates, so the iteration that meets
the tolerance is the last one that moves x whether or not the host looks at that iteration."""
import os

import numpy as np
import pytest
import torch

import pypose_amd as pp

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "cg_golden.npz"))
TAGS = sorted({k.split("/")[0] for k in G.files})


def _case(tag, device="cpu"):
    tol, maxiter, withM, withx, csr = G[f"{tag}/cfg"].tolist()
    A, b = torch.from_numpy(G[f"{tag}/A"]).to(device), torch.from_numpy(G[f"{tag}/b"]).to(device)
    M = torch.from_numpy(G[f"{tag}/M"]).to(device) if withM else None
    x0 = torch.from_numpy(G[f"{tag}/x0"]).to(device) if withx else None
    return (A.to_sparse_csr() if csr else A), b, x0, M, tol, (None if maxiter < 0 else int(maxiter))


@pytest.mark.parametrize("every", [1, 3, 8])
@pytest.mark.parametrize("tag", TAGS)
def test_cg_returns_the_reference_bits(tag, every):
    A, b, x0, M, tol, maxiter = _case(tag)
    x = pp.optim.solver.CG(maxiter=maxiter, tol=tol, check_every=every)(A, b, x=x0, M=M)
    assert torch.equal(x, torch.from_numpy(G[f"{tag}/x"])), float((x - torch.from_numpy(G[f"{tag}/x"])).abs().max())


def test_cg_zero_right_hand_side_and_default_look_interval():
    A = torch.eye(4)
    assert torch.equal(pp.optim.solver.CG()(A, torch.zeros(4, 1)), torch.zeros(4, 1))     # solver.py:300-301
    x = pp.optim.solver.CG()(A, torch.ones(4))                                             # a vector b gains a column (:292-293)
    assert x.shape == (4, 1) and torch.equal(x, torch.ones(4, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_cg_on_the_device_is_independent_of_the_look_interval(tag):
    A, b, x0, M, tol, maxiter = _case(tag, "cuda:0")
    xs = [pp.optim.solver.CG(maxiter=maxiter, tol=tol, check_every=e)(A, b, x=x0, M=M) for e in (1, 8, None)]
    assert torch.equal(xs[0], xs[1]) and torch.equal(xs[0], xs[2])
    want = torch.from_numpy(G[f"{tag}/x"])
    if tag != "capped":                                                     # (13 iterations of 600: no solution to compare)
        torch.testing.assert_close(xs[0].cpu(), want, rtol=0, atol=float(tol) * 1000 * float(want.abs().max()))
