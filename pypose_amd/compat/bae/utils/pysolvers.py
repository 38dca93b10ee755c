"""``PCG``: Jacobi-preconditioned conjugate gradient on a sparse (CSR) system -- what ``pypose.optim.solver.PCG`` resolves
to (solver.py:358-364); call shape of the reference's solvers, ``solver(A=, b=) -> x`` (solver.py:204)."""
import torch
from torch import nn


class PCG(nn.Module):
    def __init__(self, maxiter=None, tol=1e-5, check_every=8):
        super().__init__()
        self.maxiter, self.tol, self.check_every = maxiter, tol, max(1, int(check_every))
        self.iterations = 0

    def forward(self, A, b, x=None, M=None):
        b = b.reshape(-1)
        n = b.numel()
        maxiter = self.maxiter if self.maxiter is not None else 10 * n
        if A.layout == torch.sparse_csr:
            crow, col, val = A.crow_indices(), A.col_indices(), A.values()
            row = torch.repeat_interleave(torch.arange(n, device=col.device), crow[1:] - crow[:-1])
            on_diag = row == col
            diag = torch.zeros(n, dtype=val.dtype, device=val.device).index_add_(0, row[on_diag], val[on_diag])
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
        # The stop test `|r| <= tol |b|` runs on the DEVICE every iteration and the host looks at it every `check` iterations: the
        # iteration that meets it is the last one that moves x and r (`live` multiplies the step from then on), so the result and
        # the iteration count are those of a loop that tests on the host every time -- without a synchronisation per iteration.
        check = self.check_every
        thresh = self.tol * bnorm
        live = torch.ones((), dtype=b.dtype, device=b.device)
        zero = torch.zeros((), dtype=b.dtype, device=b.device)
        count = torch.zeros((), dtype=torch.int64, device=b.device)
        for it in range(maxiter):
            Ap = mv(p)
            alpha = torch.where(live > 0, rz / torch.dot(p, Ap), zero)      # (a select: whatever the idle iterations divide by)
            x = x + alpha * p
            r = r - alpha * Ap
            count = count + live.to(torch.int64)
            live = live * (r.norm() > thresh).to(b.dtype)
            if (it + 1) % check == 0 and float(live) == 0.0:
                break
            z = prec(r)
            rz_new = torch.dot(r, z)
            on = live > 0                                            # (idle iterations leave p and rz where they were: all finite)
            p = torch.where(on, z + (rz_new / rz) * p, p)
            rz = torch.where(on, rz_new, rz)
        self.iterations = int(count)
        assert not torch.any(torch.isnan(x)), 'Conjugate gradient produced NaN. Check your matrix (may not be positive-definite)'
        return x.unsqueeze(-1)
