"""Type / value predicates of pypose/function/checking.py (is_lietensor :6, is_SE3 :19, hasnan :32)."""
import math

import torch

from ..lietensor import LieTensor, SE3_type


def is_lietensor(obj):
    """True iff ``obj`` is a :class:`LieTensor`."""
    return isinstance(obj, LieTensor)


def is_SE3(obj):
    """True iff ``obj`` carries the SE3 group type."""
    return getattr(obj, "ltype", None) is SE3_type


def hasnan(obj):
    """True iff a tensor / number -- or any element of a (nested) list or tuple of them -- is NaN."""
    if isinstance(obj, (list, tuple)):
        return any(hasnan(o) for o in obj)
    return torch.isnan(obj).any() if torch.is_tensor(obj) else math.isnan(obj)
