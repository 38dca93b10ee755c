"""Robust kernels rho(x) on the squared residual norm x = ||r||^2 (API of pypose/optim/kernel.py).

Every kernel is a ``RobustKernel``: the scale parameter is validated once in the base class and ``forward`` checks the
argument's sign before handing it to the subclass's ``rho``.  The correctors differentiate ``rho`` by autograd
(optim/corrector.py), so each ``rho`` must have a finite derivative wherever it can be evaluated.
"""
import math

import torch
from torch import nn


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
        assert torch.all(input >= 0), 'input has to be non-negative.'
        return self.rho(input)

    def rho(self, x):
        raise NotImplementedError


class Huber(RobustKernel):
    """x inside sqrt(x) < delta, the tangent line in sqrt(x) outside: 2 delta sqrt(x) - delta^2 (kernel.py:37-55)."""

    def rho(self, x):
        root = x.sqrt()
        inside = root < self.delta
        # masked writes, not torch.where: the outer branch has an infinite slope at x = 0, and where() would feed
        # 0 * inf = NaN into rho'(0) when the correctors differentiate through it
        out = torch.zeros_like(x)
        out[inside] = x[inside]
        out[~inside] = 2 * self.delta * root[~inside] - self.delta2
        return out


class PseudoHuber(RobustKernel):
    """2 delta^2 (sqrt(1 + x / delta^2) - 1) (kernel.py:83-95)."""

    def rho(self, x):
        return 2 * self.delta2 * ((x / self.delta2 + 1).sqrt() - 1)


class Cauchy(RobustKernel):
    """delta^2 log(1 + x / delta^2) (kernel.py:123-135)."""

    def rho(self, x):
        return self.delta2 * (x / self.delta2 + 1).log()


class SoftLOne(RobustKernel):
    """2 (delta sqrt(x + 1 / delta^2) - 1) (kernel.py:163-176)."""

    def rho(self, x):
        return 2 * (self.delta * (1 / self.delta2 + x).sqrt() - 1)


class Arctan(RobustKernel):
    """delta^2 arctan(x / delta^2); the reference accepts any delta here (kernel.py:204-215)."""

    @staticmethod
    def valid_scale(delta):
        return True

    def rho(self, x):
        return self.delta2 * (x / self.delta2).arctan()


class Scale(RobustKernel):
    """delta x with 0 < delta <= 1, and no sign check on x (kernel.py:287-297)."""
    scale_message = "delta has to be between 0 and 1: {}"

    @staticmethod
    def valid_scale(delta):
        return 0 < delta <= 1

    def forward(self, input):
        return self.delta * input


class Tolerant(RobustKernel):
    """b log(1 + e^((x - a) / b)) - b log(1 + e^(-a / b)) with a > 0 > b (kernel.py:244-259)."""

    def __init__(self, a=1.0, b=-1.0):
        nn.Module.__init__(self)
        assert a > 0, ValueError("a has to be positive: {}".format(a))
        assert b < 0, ValueError("b has to be negative: {}".format(b))
        self.a, self.b = a, b
        self.at_zero = b * math.log(1 + math.exp(-a / b))

    def rho(self, x):
        return self.b * (1 + ((x - self.a) / self.b).exp()).log() - self.at_zero
