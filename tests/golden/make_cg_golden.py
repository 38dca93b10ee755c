"""tests/golden/cg_golden.npz: solutions of the reference's CG (optim/solver.py:219-340) on fixed systems, recorded from the
REAL reference (build container only):
    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_cg_golden.py
The iterates of CG are a deterministic sequence of the same tensor ops, so pypose_amd's CG -- whose stop test runs on the device
and is looked at only every few iterations -- must return the SAME BITS, including when the solve ends by the tolerance in the
middle of a look interval and when it ends by maxiter."""
import os, sys
import numpy as np
import torch
sys.dont_write_bytecode = True
import pypose.optim.solver as ppos  # the reference

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cg_golden.npz")
S = {}
g = torch.Generator().manual_seed(5)


def spd(n, dtype, batch=()):
    Q = torch.randn(*batch, n, n, dtype=dtype, generator=g)
    return Q @ Q.mT / n + 0.05 * torch.eye(n, dtype=dtype)


CASES = {  # tag: (n, batch, dtype, tol, maxiter, with M, with x0, csr)
    "dense32": (40, (), torch.float32, 1e-5, None, False, False, False),
    "dense64": (60, (), torch.float64, 1e-10, None, False, False, False),
    "capped": (60, (), torch.float64, 1e-14, 13, False, False, False),
    "precond": (50, (), torch.float64, 1e-9, None, True, False, False),
    "guess": (30, (), torch.float64, 1e-9, None, False, True, False),
    "batched": (24, (3,), torch.float64, 1e-8, None, False, False, False),
    "csr": (45, (), torch.float64, 1e-9, None, False, False, True),
}
for tag, (n, batch, dtype, tol, maxiter, withM, withx, csr) in CASES.items():
    A = spd(n, dtype, batch)
    if csr:
        A = A * (torch.rand(n, n, generator=g) < 0.2).to(dtype)
        A = (A + A.mT) / 2 + 2.0 * torch.eye(n, dtype=dtype)
    b = torch.randn(*batch, n, 1, dtype=dtype, generator=g)
    M = torch.diag_embed(1.0 / torch.diagonal(A, dim1=-2, dim2=-1)) if withM else None
    x0 = 0.1 * torch.randn(*batch, n, 1, dtype=dtype, generator=g) if withx else None
    S[f"{tag}/A"], S[f"{tag}/b"] = A.numpy(), b.numpy()
    S[f"{tag}/cfg"] = np.array([tol, -1 if maxiter is None else maxiter, withM, withx, csr], dtype=np.float64)
    if withM:
        S[f"{tag}/M"] = M.numpy()
    if withx:
        S[f"{tag}/x0"] = x0.numpy()
    Ain = A.to_sparse_csr() if csr else A
    x = ppos.CG(maxiter=maxiter, tol=tol)(Ain, b.clone(), x=None if x0 is None else x0.clone(), M=M)
    S[f"{tag}/x"] = x.numpy()
    print(tag, float((A @ x - b).norm() / b.norm()))
np.savez_compressed(OUT, **S)
print("wrote", OUT, len(S))
