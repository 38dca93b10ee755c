"""Robust kernels rho(x) on the squared residual norm x = ||r||^2 (API of pypose/optim/kernel.py).

Every kernel is a ``RobustKernel``: the scale parameter is validated once in the base class and ``forward`` checks the
argument's sign before handing it to the subclass's ``rho``.  The correctors differentiate ``rho`` by autograd
(optim/corrector.py), so each ``rho`` must have a finite derivative wherever it can be evaluated.

On the GPU with no gradient being recorded, ``forward`` is ONE element-wise HIP kernel (``pplie_robust_rho``, csrc/robust.hip)
-- the torch formulas below are 3-6 launches each, Huber's two boolean-mask writes a host synchronisation -- and
``robust_code`` hands the linearisation kernels the (kind, p0, p1) triple of csrc/robust.h (include/pplie.h PPLIE_ROBUST_*).
"""
import ctypes
import math

import torch
from torch import nn

from .. import _C

_RHO_SIG = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]


def robust_code(kernel):
    """(kind, p0, p1) of a built-in kernel for the HIP kernels, None for anything else (subclasses and user kernels keep the
    torch / autograd route: their ``rho`` may differ)"""
    return getattr(kernel, "_code", lambda: None)() if type(kernel) in _BUILTIN else None


def _fused_rho(kernel, x):
    """rho(x) by pplie_robust_rho, or None if this call is not a plain GPU evaluation"""
    code = robust_code(kernel)
    if code is None or not isinstance(x, torch.Tensor) or not x.is_cuda or _C._test_backend is not None \
            or x.dtype not in (torch.float32, torch.float64) or (torch.is_grad_enabled() and x.requires_grad) \
            or torch._C._are_functorch_transforms_active():
        return None
    xc = torch.Tensor.as_subclass(x, torch.Tensor).contiguous()
    out = torch.empty_like(xc)
    if xc.numel():
        fn = _C.library().symbol("pplie_robust_rho" + ("_f32" if xc.dtype == torch.float32 else "_f64"), _RHO_SIG)
        with _C._on_device(xc.device):
            rc = fn(xc.data_ptr(), out.data_ptr(), xc.numel(), code[0], code[1], code[2], _C.stream_ptr(xc.device))
        _C.check(rc, "pplie_robust_rho")
    return out


class RobustKernel(nn.Module):
    scale_message = "delta has to be positive: {}"

    def __init__(self, delta=1.0):
        super().__init__()
        assert self.valid_scale(delta), ValueError(self.scale_message.format(delta))
        self.delta, self.delta2 = delta, delta * delta

    @staticmethod
    def valid_scale(delta):
        return delta > 0

    def forward(self, input):
        # the public call keeps the reference's contract (kernel.py:43-53: AssertionError on a negative argument) on every
        # device; the optimizer's loss, whose argument is |r|^2 by construction, enters through `of_squared_norm` instead
        assert torch.all(input >= 0), 'input has to be non-negative.'
        return self.of_squared_norm(input)

    def of_squared_norm(self, x):
        """rho(x) for a caller that GUARANTEES x >= 0 (x = |r|^2): no sign check, i.e. no device round trip per evaluation"""
        out = _fused_rho(self, x)
        return self.rho(x) if out is None else out

    def rho(self, x):
        raise NotImplementedError


class Huber(RobustKernel):
    """x inside sqrt(x) < delta, the tangent line in sqrt(x) outside: 2 delta sqrt(x) - delta^2 (kernel.py:37-55)."""

    def _code(self):
        return (1, float(self.delta), 0.0)

    def rho(self, x):
        inside = x.detach().sqrt() < self.delta
        # masked writes, not torch.where, and the root taken of the OUTSIDE elements only: the outer branch has an infinite slope
        # at x = 0, and a root over all of x would feed 0 * inf = NaN into rho'(0) when the correctors differentiate through it
        out = torch.zeros_like(x)
        out[inside] = x[inside]
        out[~inside] = 2 * self.delta * x[~inside].sqrt() - self.delta2
        return out


class PseudoHuber(RobustKernel):
    """2 delta^2 (sqrt(1 + x / delta^2) - 1) (kernel.py:83-95)."""

    def _code(self):
        return (2, float(self.delta), 0.0)

    def rho(self, x):
        return 2 * self.delta2 * ((x / self.delta2 + 1).sqrt() - 1)


class Cauchy(RobustKernel):
    """delta^2 log(1 + x / delta^2) (kernel.py:123-135)."""

    def _code(self):
        return (3, float(self.delta), 0.0)

    def rho(self, x):
        return self.delta2 * (x / self.delta2 + 1).log()


class SoftLOne(RobustKernel):
    """2 (delta sqrt(x + 1 / delta^2) - 1) (kernel.py:163-176)."""

    def _code(self):
        return (4, float(self.delta), 0.0)

    def rho(self, x):
        return 2 * (self.delta * (1 / self.delta2 + x).sqrt() - 1)


class Arctan(RobustKernel):
    """delta^2 arctan(x / delta^2); the reference accepts any delta here (kernel.py:204-215)."""

    @staticmethod
    def valid_scale(delta):
        return True

    def _code(self):
        return (5, float(self.delta), 0.0)

    def rho(self, x):
        return self.delta2 * (x / self.delta2).arctan()


class Scale(RobustKernel):
    """delta x with 0 < delta <= 1, and no sign check on x (kernel.py:287-297)."""
    scale_message = "delta has to be between 0 and 1: {}"

    @staticmethod
    def valid_scale(delta):
        return 0 < delta <= 1

    def _code(self):
        return (6, float(self.delta), 0.0)

    def forward(self, input):
        return self.delta * input

    def rho(self, x):
        return self.delta * x


class Tolerant(RobustKernel):
    """b log(1 + e^((x - a) / b)) - b log(1 + e^(-a / b)) with a > 0 > b (kernel.py:244-259)."""

    def __init__(self, a=1.0, b=-1.0):
        nn.Module.__init__(self)
        assert a > 0, ValueError("a has to be positive: {}".format(a))
        assert b < 0, ValueError("b has to be negative: {}".format(b))
        self.a, self.b = a, b
        self.at_zero = b * math.log(1 + math.exp(-a / b))

    def _code(self):
        return (7, float(self.a), float(self.b))

    def rho(self, x):
        return self.b * (1 + ((x - self.a) / self.b).exp()).log() - self.at_zero


_BUILTIN = (Huber, PseudoHuber, Cauchy, SoftLOne, Arctan, Scale, Tolerant)
