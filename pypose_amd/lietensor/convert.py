"""Conversions between matrices / Euler angles and LieTensors (host-side mirror of
pypose/lietensor/convert.py).

The per-row arithmetic runs in HIP row kernels (csrc/convert.hip: ``pplie_mat2so3_*``,
``pplie_euler2so3_*``, ``pplie_so3_euler_*``), wrapped in autograd Functions whose backward kernels
differentiate the branch the value takes (the reference: autograd through masked sums / ``torch.where``).
Argument checks, warnings and error messages follow the reference function by function.
"""
from __future__ import annotations

import warnings

import torch
from torch.nn.functional import normalize

from .. import _C
from .lietensor import LieTensor, SO3_type, SE3_type, Sim3_type, RxSO3_type, liegroup
from .operation import _fold_vmap, _launch
from .utils import SO3, SE3, Sim3, RxSO3


def _make_fn(name, kernel, win, wout, has_prm):
    """Function for a conversion row op  out = f(x[; prm]),  gradient by the ``_bwd`` kernel (Dual sweeps).
    Both directions launch through ``operation._launch`` (leading-dim flattening, legacy-vmap peeling)."""

    class _Bwd(torch.autograd.Function):
        @staticmethod
        def forward(x, g, prm):
            return _launch(kernel + "_bwd", (x, g), (win, wout), (win,), prm)[0]

        @staticmethod
        def setup_context(ctx, inputs, output):
            return

        @staticmethod
        def backward(ctx, *grads):
            raise NotImplementedError(f"{name}: double backward is not supported")

        @staticmethod
        def vmap(info, in_dims, x, g, prm):
            x, g = _fold_vmap(in_dims[:2], (x, g))
            return _Bwd.apply(x, g, prm), 0

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(x, prm):
            return _launch(kernel + "_fwd", (x,), (win,), (wout,), prm)[0]

        @staticmethod
        def setup_context(ctx, inputs, output):
            ctx.save_for_backward(inputs[0])
            ctx.prm = inputs[1]

        @staticmethod
        def backward(ctx, g):
            return _Bwd.apply(ctx.saved_tensors[0], g, ctx.prm), None

        @staticmethod
        def vmap(info, in_dims, x, prm):
            return _Fn.apply(x.movedim(in_dims[0], 0), prm), 0

    _Fn.__name__ = _Fn.__qualname__ = name
    _Bwd.__name__ = _Bwd.__qualname__ = name + "_Bwd"
    return _Fn


_Mat2SO3 = _make_fn("Mat2SO3", "mat2so3", 9, 4, True)
_SO3Euler = _make_fn("SO3_Euler", "so3_euler", 4, 3, True)
_Euler2SO3 = _make_fn("Euler2SO3", "euler2so3", 3, 4, False)


def _check_matrix_shape(mat):
    if not torch.is_tensor(mat):
        mat = torch.tensor(mat)
    if len(mat.shape) < 2:
        raise ValueError("Input size must be at least 2 dimensions. Got {}".format(mat.shape))
    if not (mat.shape[-2:] == (3, 3) or mat.shape[-2:] == (3, 4) or mat.shape[-2:] == (4, 4)):
        raise ValueError("Input size must be a * x 3 x 3 or * x 3 x 4 or * x 4 x 4  tensor. \
                Got {}".format(mat.shape))
    return mat


def _check_last_row(mat, check, rtol, atol, what):
    if mat.shape[-2:] == (4, 4) and check is True:
        zerosone = torch.tensor([0, 0, 0, 1], dtype=mat.dtype, device=mat.device)
        if not torch.allclose(mat[..., 3, :], zerosone.expand_as(mat[..., 3, :]), rtol=rtol, atol=atol):
            warnings.warn(what + " of shape 4x4 last rows are not all equal [0, 0, 0, 1]")


def mat2SO3(mat, check=True, rtol=1e-5, atol=1e-5):
    """Rotation / transformation matrices ``(*, 3, 3 | 3, 4 | 4, 4)`` -> SO3 ``(*, 4)`` (reference convert.py:8-146)."""
    mat = _check_matrix_shape(mat)
    mat = mat[..., :3, :3]
    shape = mat.shape
    with torch.no_grad():
        if check:
            e0 = mat @ mat.mT
            e1 = torch.eye(3, dtype=mat.dtype, device=mat.device)
            if not torch.allclose(e0, e1.expand_as(e0), rtol=rtol, atol=atol):
                raise ValueError("Input rotation matrices are not all orthogonal matrix")
            ones = torch.ones(shape[:-2], dtype=mat.dtype, device=mat.device)
            if not torch.allclose(torch.det(mat), ones, rtol=rtol, atol=atol):
                raise ValueError("Input rotation matrices' determinant are not all equal to 1")
    q = _Mat2SO3.apply(mat.reshape(shape[:-2] + (9,)), float(atol))
    return SO3(q)


def _translation_of(mat):
    if mat.shape[-1] == 3:
        return torch.zeros(mat.shape[:-2] + (3,), dtype=mat.dtype, device=mat.device, requires_grad=mat.requires_grad)
    return mat[..., :3, 3]


def mat2SE3(mat, check=True, rtol=1e-5, atol=1e-5):
    """``(*, 3, 3 | 3, 4 | 4, 4)`` -> SE3 ``(*, 7)`` (reference convert.py:148-258)."""
    mat = _check_matrix_shape(mat)
    _check_last_row(mat, check, rtol, atol, "input")
    q = mat2SO3(mat[..., :3, :3], check=check, rtol=rtol, atol=atol).tensor()
    return SE3(torch.cat([_translation_of(mat), q], dim=-1))


def _scale_of(mat, rtol, atol):
    rot = mat[..., :3, :3]
    s = torch.pow(torch.det(rot), 1 / 3).unsqueeze(-1)
    zeros = torch.zeros(mat.shape[:-2], dtype=mat.dtype, device=mat.device)
    if torch.allclose(s, zeros, rtol=rtol, atol=atol):
        raise ValueError("Rotation matrix not full rank.")
    return rot, s


def mat2Sim3(mat, check=True, rtol=1e-5, atol=1e-5):
    """``(*, 3, 3 | 3, 4 | 4, 4)`` with uniformly scaled rotation -> Sim3 ``(*, 8)`` (reference convert.py:261-412)."""
    mat = _check_matrix_shape(mat)
    _check_last_row(mat, check, rtol, atol, "Input")
    rot, s = _scale_of(mat, rtol, atol)
    q = mat2SO3(rot / s.unsqueeze(-1), check=check, rtol=rtol, atol=atol).tensor()
    return Sim3(torch.cat([_translation_of(mat), q, s], dim=-1))


def mat2RxSO3(mat, check=True, rtol=1e-5, atol=1e-5):
    """``(*, 3, 3 | 3, 4 | 4, 4)`` with uniformly scaled rotation -> RxSO3 ``(*, 5)`` (reference convert.py:415-513)."""
    mat = _check_matrix_shape(mat)
    rot, s = _scale_of(mat, rtol, atol)
    q = mat2SO3(rot / s.unsqueeze(-1), check=check, rtol=rtol, atol=atol).tensor()
    return RxSO3(torch.cat([q, s], dim=-1))


def from_matrix(mat, ltype, check=True, rtol=1e-5, atol=1e-5):
    """Dispatch on the target group type (reference convert.py:516-599)."""
    mat = _check_matrix_shape(mat)
    if ltype == SO3_type:
        return mat2SO3(mat, check=check, rtol=rtol, atol=atol)
    elif ltype == SE3_type:
        return mat2SE3(mat, check=check, rtol=rtol, atol=atol)
    elif ltype == Sim3_type:
        return mat2Sim3(mat, check=check, rtol=rtol, atol=atol)
    elif ltype == RxSO3_type:
        return mat2RxSO3(mat, check=check, rtol=rtol, atol=atol)
    raise ValueError("Input ltype must be one of SO3_type, SE3_type, Sim3_type or RxSO3_type.\
                Got {}".format(ltype))


def euler2SO3(euler):
    """``[roll, pitch, yaw]`` ``(*, 3)`` -> SO3 ``(*, 4)`` (reference convert.py:607-663)."""
    if not torch.is_tensor(euler):
        euler = torch.tensor(euler)
    assert euler.shape[-1] == 3
    return SO3(_Euler2SO3.apply(euler, None))


def so3_euler(q, eps=2e-4):
    """Quaternion rows ``(*, 4)`` -> ``[roll, pitch, yaw]`` (the kernel behind ``LieTensor.euler``)."""
    return _SO3Euler.apply(q, float(eps))


def quat2unit(input, eps=1e-12):
    """Normalise the quaternion part of a Lie group LieTensor (reference convert.py:830-862)."""
    if isinstance(input, LieTensor) and (input.ltype in liegroup):
        data = input.tensor()
        if input.ltype in [SO3_type, RxSO3_type]:
            data[..., :4] = normalize(data[..., :4], p=2, dim=-1, eps=eps)
        elif input.ltype in [SE3_type, Sim3_type]:
            data[..., 3:7] = normalize(data[..., 3:7], p=2, dim=-1, eps=eps)
        output = LieTensor(data, ltype=input.ltype)
        if (output.rotation().tensor().norm(p=2, dim=-1) < eps).any():
            raise ValueError("Detected zero quaternions, which cannot be normalized.")
        return output
    warnings.warn("Input is not Lie group, doing thing and returning input..")
    return input
