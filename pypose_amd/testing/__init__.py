"""``pp.testing`` (reference pypose/testing/comparison.py): closeness assertion that also takes LieTensors."""
import torch

from ..function.checking import is_lietensor

__all__ = ["assert_close"]


def assert_close(actual, expected, *args, **kwargs):
    """``torch.testing.assert_close`` for tensors; two LieTensors are close when the tangent of their
    relative transform, ``Log(actual^-1 * expected)``, is close to zero (comparison.py:38-42)."""
    if is_lietensor(actual) and is_lietensor(expected):
        delta = (actual.Inv() @ expected).Log().tensor()
        return torch.testing.assert_close(delta, torch.zeros_like(delta), *args, **kwargs)
    return torch.testing.assert_close(actual, expected, *args, **kwargs)
