"""``PCG``: Jacobi-preconditioned conjugate gradient on a sparse (CSR) system -- what ``pypose.optim.solver.PCG`` resolves
to (solver.py:358-364); call shape of the reference's solvers, ``solver(A=, b=) -> x`` (solver.py:204)."""
import torch
from torch import nn


class PCG(nn.Module):
    def __init__(self, maxiter=None, tol=1e-5):
        super().__init__()
        self.maxiter, self.tol = maxiter, tol
        self.iterations = 0

    def forward(self, A, b, x=None, M=None):
        b = b.reshape(-1)
        n = b.numel()
        maxiter = self.maxiter if self.maxiter is not None else 10 * n
        if A.layout == torch.sparse_csr:
            crow, col, val = A.crow_indices(), A.col_indices(), A.values()
            row = torch.repeat_interleave(torch.arange(n, device=col.device), crow[1:] - crow[:-1])
            diag = torch.zeros(n, dtype=val.dtype, device=val.device).index_add_(0, row[row == col], val[row == col])
        else:
            diag = A.diagonal() if not A.is_sparse else A.to_dense().diagonal()
        Minv = torch.where(diag != 0, 1.0 / diag, torch.ones_like(diag)) if M is None else None
        prec = (lambda r: Minv * r) if M is None else (lambda r: (M @ r.unsqueeze(-1)).squeeze(-1))
        mv = lambda v: (A @ v.unsqueeze(-1)).squeeze(-1)
        x = torch.zeros_like(b) if x is None else x.reshape(-1).clone()
        r = b - mv(x)
        z = prec(r)
        p = z.clone()
        rz = torch.dot(r, z)
        bnorm = b.norm()
        self.iterations = 0
        if float(bnorm) == 0.0:
            return x.unsqueeze(-1)
        for it in range(maxiter):
            Ap = mv(p)
            alpha = rz / torch.dot(p, Ap)
            x = x + alpha * p
            r = r - alpha * Ap
            self.iterations = it + 1
            if float(r.norm()) <= self.tol * float(bnorm):
                break
            z = prec(r)
            rz_new = torch.dot(r, z)
            p = z + (rz_new / rz) * p
            rz = rz_new
        assert not torch.any(torch.isnan(x)), 'Conjugate gradient produced NaN. Check your matrix (may not be positive-definite)'
        return x.unsqueeze(-1)
