"""``pp.func`` (reference pypose/func/jac.py): functional transforms that accept LieTensor arguments."""
import functools

import torch

from ..lietensor.lietensor import retain_ltype

__all__ = ["jacrev"]


def jacrev(func, argnums=0, *, has_aux=False, chunk_size=None, _preallocate_and_copy=False):
    """``torch.func.jacrev`` whose transformed function may take LieTensors: the call runs under
    :func:`pypose_amd.retain_ltype`, so the group type survives functorch's tensor wrapping (jac.py:53-58).
    The derivative is with respect to the stored coordinates (last dimension = storage width)."""
    transformed = torch.func.jacrev(func, argnums, has_aux=has_aux, chunk_size=chunk_size,
                                    _preallocate_and_copy=_preallocate_and_copy)

    @functools.wraps(transformed)
    def with_ltype(*args, **kwargs):
        with retain_ltype():
            return transformed(*args, **kwargs)

    return with_ltype
