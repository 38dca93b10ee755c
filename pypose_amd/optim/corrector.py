"""Residual / Jacobian re-weighting for robust kernels (reference pypose/optim/corrector.py).

Both correctors are row-local (one scale -- and for Triggs one rank-1 correction -- per residual
row), so they apply unchanged to the per-row Jacobian blocks of the structured LM paths: ``J``
may be the dense ``[N_res, N_par]`` matrix or blocks ``[rows, d_res, d_par]``.
"""
import torch
from torch import Tensor, nn
from torch.autograd import grad


def _rho_derivatives(kernel, x, second=False):
    """rho'(x) (and rho''(x)) elementwise, by autograd on sum(rho(x)) (rho acts elementwise)."""
    with torch.enable_grad():
        x = x.detach().requires_grad_(True)
        y = kernel(x).sum()
        g1 = grad(y, x, create_graph=second)[0]
        if not second:
            return x.detach(), g1.detach(), None
        g2 = grad(g1.sum(), x)[0]
    return x.detach(), g1.detach(), g2.detach()


class FastTriggs(nn.Module):
    """Scale R and J by sqrt(rho'(||r||^2)) (reference corrector.py:7-96)."""

    def __init__(self, kernel):
        super().__init__()
        self.kernel = kernel

    def forward(self, R: Tensor, J: Tensor):
        assert not torch.is_inference_mode_enabled(), "FastTriggs modifier does not work in torch.inference_mode."
        _, g1, _ = _rho_derivatives(self.kernel, R.square().sum(-1, keepdim=True))
        s = g1.sqrt()
        if J.dim() == 2:                       # dense: one scale per scalar residual row
            return s * R, s.expand_as(R).reshape(-1, 1) * J
        return s * R, s.reshape(-1, 1, 1) * J  # blocks [rows, d_res, d_par]


class Triggs(nn.Module):
    """Second-order (Triggs) correction (reference corrector.py:98-167)."""

    def __init__(self, kernel):
        super().__init__()
        self.kernel = kernel

    def forward(self, R: Tensor, J: Tensor):
        x, g1, g2 = _rho_derivatives(self.kernel, R.square().sum(-1, keepdim=True), second=True)
        se = g1.sqrt()
        Jb = J.reshape(R.shape + (J.shape[-1],))
        sR, sJ = se * R, se.unsqueeze(-1) * Jb
        M = ~((x == 0) | (g2 <= 0)).squeeze(-1)
        alpha = 1 - (1 + 2 * x[M] * g2[M] / g1[M]).clamp(min=0).sqrt()
        sR[M] = se[M] / (1 - alpha)            # sic: exactly the reference's expression (corrector.py:164)
        Q = torch.einsum('...d,...k,...kl->...dl', R[M], R[M], sJ[M])
        sJ[M] = sJ[M] - (alpha / x[M]).unsqueeze(-1) * Q
        return sR, sJ.reshape(J.shape)
