"""Residual / Jacobian re-weighting for robust kernels (reference pypose/optim/corrector.py).

Both correctors are row-local (one scale -- and for Triggs one rank-1 correction -- per residual
row), so they apply unchanged to the per-row Jacobian blocks of the structured LM paths: ``J``
may be the dense ``[N_res, N_par]`` matrix or blocks ``[rows, d_res, d_par]``.

With a built-in kernel (optim/kernel.py) and GPU tensors both correctors are ONE HIP launch, ``pplie_robust_scale_rows``
(csrc/robust.hip): rho' in closed form, R and J scaled in a single pass over J -- the reference builds and differentiates an
autograd graph over ``kernel(x).sum()`` every step and scales with element-wise launches (corrector.py:91-96).  Every built-in
kernel is concave (rho'' <= 0), so Triggs' second-order row mask (reference :160) is empty for them and Triggs is the same
scaling; user-defined kernels keep the autograd formulation below.
"""
import ctypes

import torch
from torch import Tensor, nn
from torch.autograd import grad

from .. import _C
from .kernel import robust_code

_SCALE_SIG = [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_void_p]


def fused_code(corrector):
    """(kind, p0, p1) if ``corrector`` is one the HIP kernels reproduce -- exactly FastTriggs or Triggs (no subclass) over a
    built-in kernel -- else None.  The ONE eligibility rule of every fused route (the correctors' own forward, the per-edge
    blocks of optim/posegraph.py and multigraph.py, the robust linearisation kernel of optim/fused.py).  Triggs over Scale is
    excluded everywhere: rho' is a constant there and the reference's second autograd.grad raises (corrector.py:155-157), so
    the autograd formulation below must be the one that runs (and raises)."""
    from .kernel import Scale
    if type(corrector) not in (FastTriggs, Triggs):
        return None
    if type(corrector) is Triggs and type(corrector.kernel) is Scale:
        return None
    return robust_code(corrector.kernel)


def fused_scale_rows(kernel, R, J, inplace=False):
    """(s R, s J) with s_i = sqrt(rho'(|R_i|^2)) by one kernel, or None if this call cannot take the route.
    R [n, dr]; J any tensor whose leading dimension splits into n row blocks.  ``inplace``: J is overwritten (the caller owns it)."""
    code = robust_code(kernel)
    if code is None or _C._test_backend is not None or not (R.is_cuda and J.is_cuda) or R.dtype != J.dtype \
            or R.dtype not in (torch.float32, torch.float64) or R.dim() != 2 or R.shape[1] > 64 or R.shape[0] == 0 \
            or J.numel() == 0 or J.numel() % R.shape[0] or (torch.is_grad_enabled() and (R.requires_grad or J.requires_grad)) \
            or torch._C._are_functorch_transforms_active():
        return None
    n, dr = R.shape
    Rc, Jc = R.contiguous(), J.contiguous()
    Rout = torch.empty_like(Rc)
    Jout = Jc if (inplace and Jc.data_ptr() == J.data_ptr()) or Jc.data_ptr() != J.data_ptr() else torch.empty_like(Jc)
    fn = _C.library().symbol("pplie_robust_scale_rows" + ("_f32" if R.dtype == torch.float32 else "_f64"), _SCALE_SIG)
    with _C._on_device(R.device):
        rc = fn(Rc.data_ptr(), Rout.data_ptr(), Jc.data_ptr(), Jout.data_ptr(), n, dr, Jc.numel() // n, code[0], code[1], code[2],
                _C.stream_ptr(R.device))
    _C.check(rc, "pplie_robust_scale_rows")
    if Jout.data_ptr() == J.data_ptr():
        _C.mark_written(J)
    return Rout, Jout.view(J.shape)


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

    def forward(self, R: Tensor, J: Tensor, inplace=False):
        done = fused_scale_rows(self.kernel, R, J, inplace)
        if done is not None:
            return done
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

    def forward(self, R: Tensor, J: Tensor, inplace=False):
        # built-in kernels are concave (rho'' <= 0): the second-order mask M below is empty and Triggs is FastTriggs' scaling --
        # except Scale, whose rho' is a constant the reference cannot differentiate again (it raises; so does the route below)
        from .kernel import Scale
        done = None if type(self.kernel) is Scale else fused_scale_rows(self.kernel, R, J, inplace)      # (cf. fused_code)
        if done is not None:
            return done
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
