"""Robust kernel functions rho(x) applied to x = ||r||^2 (reference pypose/optim/kernel.py)."""
import math

import torch
from torch import Tensor, nn


def _nonneg(x):
    assert torch.all(x >= 0), 'input has to be non-negative'


class Huber(nn.Module):
    """x if sqrt(x) < delta else 2 delta sqrt(x) - delta^2 (reference kernel.py:5-53)."""

    def __init__(self, delta: float = 1.0) -> None:
        super().__init__()
        assert delta > 0, ValueError("delta has to be positive: {}".format(delta))
        self.delta, self.delta2 = delta, delta ** 2

    def forward(self, input: Tensor) -> Tensor:
        assert torch.all(input >= 0), 'input has to be non-negative.'
        inlier = input.sqrt() < self.delta
        # masked writes (not torch.where): d/dx of the outlier branch is infinite at x = 0 and must
        # not leak a 0 * inf = NaN into rho'(0) when the correctors differentiate through here
        output = torch.zeros_like(input)
        output[inlier] = input[inlier]
        output[~inlier] = 2 * self.delta * input[~inlier].sqrt() - self.delta2
        return output


class PseudoHuber(nn.Module):
    """2 delta^2 (sqrt(x/delta^2 + 1) - 1) (reference kernel.py:56-94)."""

    def __init__(self, delta: float = 1.0) -> None:
        super().__init__()
        assert delta > 0, ValueError("delta has to be positive: {}".format(delta))
        self.delta2 = delta ** 2

    def forward(self, input: Tensor) -> Tensor:
        _nonneg(input)
        return 2 * self.delta2 * ((input / self.delta2 + 1).sqrt() - 1)


class Cauchy(nn.Module):
    """delta^2 log(x/delta^2 + 1) (reference kernel.py:97-134)."""

    def __init__(self, delta: float = 1.0) -> None:
        super().__init__()
        assert delta > 0, ValueError("delta has to be positive: {}".format(delta))
        self.delta2 = delta ** 2

    def forward(self, input: Tensor) -> Tensor:
        _nonneg(input)
        return self.delta2 * (input / self.delta2 + 1).log()


class SoftLOne(nn.Module):
    """2 (delta sqrt(1/delta^2 + x) - 1) (reference kernel.py:137-175)."""

    def __init__(self, delta: float = 1.0) -> None:
        super().__init__()
        assert delta > 0, ValueError("delta has to be positive: {}".format(delta))
        self.delta1, self.delta2 = delta, delta ** 2

    def forward(self, input: Tensor) -> Tensor:
        _nonneg(input)
        return 2 * (self.delta1 * (1 / self.delta2 + input).sqrt() - 1)


class Arctan(nn.Module):
    """delta^2 arctan(x / delta^2) (reference kernel.py:178-214)."""

    def __init__(self, delta: float = 1.0) -> None:
        super().__init__()
        self.delta2 = delta ** 2

    def forward(self, input: Tensor) -> Tensor:
        _nonneg(input)
        return self.delta2 * (input / self.delta2).arctan()


class Tolerant(nn.Module):
    """b log(1 + exp((x - a)/b)) - b log(1 + exp(-a/b)) (reference kernel.py:217-258)."""

    def __init__(self, a: float = 1.0, b: float = -1.0) -> None:
        super().__init__()
        assert a > 0, ValueError("a has to be positive: {}".format(a))
        assert b < 0, ValueError("b has to be negative: {}".format(b))
        self.a, self.b = a, b

    def forward(self, input: Tensor) -> Tensor:
        _nonneg(input)
        offset = self.b * math.log(1 + math.exp(-self.a / self.b))
        return self.b * (1 + ((input - self.a) / self.b).exp()).log() - offset


class Scale(nn.Module):
    """delta * x (reference kernel.py:261-297)."""

    def __init__(self, delta: float = 1.0) -> None:
        super().__init__()
        assert 0 < delta <= 1, ValueError("delta has to be between 0 and 1: {}".format(delta))
        self.delta = delta

    def forward(self, input):
        return self.delta * input
