"""Small free functions of the LieTensor API (reference pypose/lietensor/basics.py)."""
import torch


def vec2skew(input: torch.Tensor) -> torch.Tensor:
    """[..., 3] -> [..., 3, 3] skew-symmetric matrices (reference lietensor/basics.py:7-41)."""
    v = input.tensor() if hasattr(input, 'ltype') else input
    assert v.shape[-1] == 3, "Last dim should be 3"
    K = torch.zeros(v.shape[:-1] + (3, 3), dtype=v.dtype, device=v.device)
    x, y, z = v.unbind(-1)
    K[..., 0, 1], K[..., 0, 2] = -z, y
    K[..., 1, 0], K[..., 1, 2] = z, -x
    K[..., 2, 0], K[..., 2, 1] = -y, x
    return K


def add(input, other, alpha=1):
    """``input + alpha * other`` with LieTensor semantics (group: Exp(other) * input)."""
    return input.add(other, alpha)


def add_(input, other, alpha=1):
    return input.add_(other, alpha)


def mul(input, other):
    return input * other
