"""Model Jacobians (reference pypose/optim/functional.py).

``modjac`` differentiates ``model(input)`` w.r.t. all parameters with
``torch.autograd.functional.jacobian`` over ``torch.func.functional_call`` -- the same contract as
the reference (:8-153).  With ``vectorize=True`` the backward runs under vmap; the HIP-backed Lie
ops fold the vmapped dimension into their row dimension (lietensor/operation.py).
"""
from functools import partial

import torch
from torch.autograd.functional import jacobian
from torch.func import functional_call, jacfwd, jacrev


def _hasnan(obj):
    if isinstance(obj, (tuple, list)):
        return any(_hasnan(o) for o in obj)
    return bool(torch.isnan(obj).any())


@torch.enable_grad()
def modjac(model, input=None, create_graph=False, strict=False, vectorize=False,
           strategy='reverse-mode', flatten=False):
    params, buffers = dict(model.named_parameters()), dict(model.named_buffers())
    names, values = list(params.keys()), tuple(params.values())
    input = tuple() if input is None else input

    def as_function_of_parameters(*new_values):
        return functional_call(model, (dict(zip(names, new_values)), buffers), input)

    J = jacobian(as_function_of_parameters, values, create_graph=create_graph, strict=strict,
                 vectorize=vectorize, strategy=strategy)
    assert not _hasnan(J), 'Jacobian contains Nan! Check your model and input!'
    if flatten and isinstance(J, tuple):
        def row(blocks):
            return torch.cat([j.view(-1, p.numel()) for j, p in zip(blocks, values)], dim=1)
        J = torch.cat([row(Jr) for Jr in J]) if any(isinstance(j, tuple) for j in J) else row(J)
    return J


@torch.enable_grad()
def modjacrev(model, input, argnums=0, *, has_aux=False):
    return jacrev(partial(functional_call, model), argnums=argnums, has_aux=has_aux)(dict(model.named_parameters()), input)


@torch.enable_grad()
def modjacfwd(model, input, argnums=0, *, has_aux=False):
    return jacfwd(partial(functional_call, model), argnums=argnums, has_aux=has_aux)(dict(model.named_parameters()), input)
