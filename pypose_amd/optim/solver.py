"""Linear solvers for the (damped) normal equations (reference pypose/optim/solver.py).

``PINV``, ``LSTSQ``, ``Cholesky`` and ``CG`` keep the reference's call shape ``solver(A=, b=) -> x``
on dense tensors (torch.linalg runs them on the GPU through rocSOLVER/hipBLAS).  The structured
LM paths of this package do not build a global dense ``A``: block-diagonal problems are solved
per block by ``pplie_block_chol_solve`` and pose graphs by the matrix-free block-Jacobi ``PCG`` in
:mod:`pypose_amd.optim.posegraph`.
"""
from typing import Optional

import torch
from torch import Tensor, nn
from torch.linalg import cholesky_ex, lstsq, pinv


class PINV(nn.Module):
    """x = pinv(A) b (reference solver.py:10-67)."""

    def __init__(self, atol=None, rtol=None, hermitian=False):
        super().__init__()
        self.atol, self.rtol, self.hermitian = atol, rtol, hermitian

    def forward(self, A: Tensor, b: Tensor) -> Tensor:
        return pinv(A, atol=self.atol, rtol=self.rtol, hermitian=self.hermitian) @ b


class LSTSQ(nn.Module):
    """Least-squares solve (reference solver.py:71-152)."""

    def __init__(self, rcond=None, driver=None):
        super().__init__()
        self.rcond, self.driver = rcond, driver

    def forward(self, A: Tensor, b: Tensor) -> Tensor:
        self.out = lstsq(A, b, rcond=self.rcond, driver=self.driver)
        assert not torch.any(torch.isnan(self.out.solution)), 'Linear Solver Failed Using LSTSQ. Using PINV() instead'
        return self.out.solution


class Cholesky(nn.Module):
    """cholesky_ex + cholesky_solve; asserts a NaN-free factor (reference solver.py:155-216)."""

    def __init__(self, upper=False):
        super().__init__()
        self.upper = upper

    def forward(self, A: Tensor, b: Tensor) -> Tensor:
        L, info = cholesky_ex(A, upper=self.upper)
        assert not torch.any(torch.isnan(L)), \
            'Cholesky decomposition failed. Check your matrix (may not be positive-definite)'
        return b.cholesky_solve(L, upper=self.upper)


class CG(nn.Module):
    """Conjugate gradient with scipy.sparse.linalg.cg's stopping rule: atol = tol * ||b||,
    maxiter = 10 n, optional preconditioner *matrix* M (reference solver.py:219-340).
    Accepts dense, batched dense, CSR and BSR ``A``."""

    def __init__(self, maxiter=None, tol=1e-5):
        super().__init__()
        self.maxiter, self.tol = maxiter, tol

    def forward(self, A: Tensor, b: Tensor, x: Optional[Tensor] = None, M: Optional[Tensor] = None) -> Tensor:
        if A.ndim == b.ndim + 1:
            b = b.unsqueeze(-1)
        else:
            assert A.ndim == b.ndim, 'The number of dimensions of A and b must be the same or one more than b'
        if x is None:
            x = torch.zeros_like(b)
        bnrm2 = torch.linalg.norm(b, dim=0)
        if (bnrm2 == 0).all():
            return b
        atol = self.tol * bnrm2
        maxiter = b.shape[-2] * 10 if self.maxiter is None else self.maxiter
        r = b - A @ x if x.any() else b.clone()
        rho_prev, p = None, None
        for it in range(maxiter):
            if (torch.linalg.norm(r, dim=0) < atol).all():
                return x
            z = M @ r if M is not None else r
            rho = r.mT @ z
            p = z.clone() if it == 0 else p.mul_(rho / rho_prev).add_(z)
            q = A @ p
            alpha = rho / (p.mT @ q)
            x = x + alpha * p
            r = r - alpha * q
            rho_prev = rho
        return x


__all__ = ["PINV", "LSTSQ", "Cholesky", "CG"]
