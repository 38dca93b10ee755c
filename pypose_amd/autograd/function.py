"""``parallel_for_sparse_jacobian`` / ``psjac`` (reference pypose/autograd/function.py:13-83).

In the reference this decorator hands a row-independent batched function to the optional ``bae``
plugin so that it can trace sparse Jacobians.  ``pypose_amd.optim`` discovers row independence and
gather structure by itself (probe-verified, see optim/blocks.py and optim/posegraph.py), so the
decorator only has to keep the function unchanged -- which is also all the reference's does to the
function's behaviour ("This decorator doesn't change the function behavior").
"""
from functools import wraps


def parallel_for_sparse_jacobian(function):
    @wraps(function)
    def wrapped(*args, **kwargs):
        return function(*args, **kwargs)
    return wrapped


psjac = parallel_for_sparse_jacobian
__all__ = ["parallel_for_sparse_jacobian", "psjac"]
