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
    """Conjugate gradient with scipy.sparse.linalg.cg's stopping rule -- atol = tol * ||b||, tested BEFORE every update,
    maxiter = 10 n, optional preconditioner *matrix* M -- on dense, batched dense, CSR and BSR ``A`` (reference
    solver.py:219-340: same iterates, same iteration in which it stops).

    The reference asks the host ``(norm(r) < atol).all()`` in every iteration: on a GPU that is a stream synchronisation per
    iteration, more than the iteration itself for the systems this solver sees.  Here the test runs on the device and steers a
    ``live`` flag that gates the updates (``torch.where``: the iteration that meets the test is the last one that moves x and r,
    whatever the idle ones compute); the host looks at the flag every ``check_every`` iterations -- 1 on host tensors, where a
    look costs nothing."""

    def __init__(self, maxiter=None, tol=1e-5, check_every=None):
        super().__init__()
        self.maxiter, self.tol, self.check_every = maxiter, tol, check_every

    def forward(self, A: Tensor, b: Tensor, x: Optional[Tensor] = None, M: Optional[Tensor] = None) -> Tensor:
        if A.ndim == b.ndim + 1:
            b = b.unsqueeze(-1)
        else:
            assert A.ndim == b.ndim, 'The number of dimensions of A and b must be the same or one more than b'
        x = torch.zeros_like(b) if x is None else x
        bnrm2 = torch.linalg.norm(b, dim=0)
        if (bnrm2 == 0).all():
            return b
        atol = self.tol * bnrm2
        maxiter = b.shape[-2] * 10 if self.maxiter is None else self.maxiter
        every = self.check_every if self.check_every is not None else (8 if b.is_cuda else 1)
        r = b - A @ x if x.any() else b.clone()
        live = torch.ones((), dtype=torch.bool, device=b.device)
        p = rho_prev = None
        for it in range(maxiter):
            live = live & ~(torch.linalg.norm(r, dim=0) < atol).all()
            if it % every == 0 and not bool(live):
                break
            z = M @ r if M is not None else r
            rho = r.mT @ z
            p_new = z if p is None else z + (rho / rho_prev) * p
            q = A @ p_new
            alpha = rho / (p_new.mT @ q)
            x = torch.where(live, x + alpha * p_new, x)
            r = torch.where(live, r - alpha * q, r)
            p = p_new if p is None else torch.where(live, p_new, p)
            rho_prev = rho if rho_prev is None else torch.where(live, rho, rho_prev)
        return x


__all__ = ["PINV", "LSTSQ", "Cholesky", "CG"]
