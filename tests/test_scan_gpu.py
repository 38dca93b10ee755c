"""pplie_scan_mat: pp.cumprod on plain stacks of small square matrices (SURVEY 8b `scan_mat9`; the reference's use is
module/imu_preintegrator.py:462) through the C ABI, against the reference's Hillis-Steele formulation (pypose/basics/ops.py:27-56)."""
import pytest
import torch

import pypose_amd as pp

pytestmark = pytest.mark.gpu


# (fp32: a sequential product of 257 factors against a tree of the same factors -- the two associate differently)
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 2e-4)])
@pytest.mark.parametrize("left", [True, False])
@pytest.mark.parametrize("shape", [(3, 1, 9, 9), (5, 7, 9, 9), (2, 3, 130, 4, 4), (1, 257, 3, 3), (4, 64, 6, 6), (2, 33, 2, 2), (0, 5, 9, 9)])
def test_matrix_cumprod_is_one_launch_and_equals_the_reference_formulation(shape, left, dtype, tol):
    """pp.cumprod on a plain stack of square matrices along the axis in front of them (the reference's use: [B, F + 1, 9, 9],
    module/imu_preintegrator.py:462; SURVEY 8b scan_mat9): pplie_scan_mat against the reference's own Hillis-Steele rounds
    (basics/ops.py:27-36) built from torch ops."""
    from pypose_amd.basics import ops as O
    torch.manual_seed(0)
    d = shape[-1]
    dim = len(shape) - 3
    # (factors I + noise with the noise scaled so that the product of all L of them stays well-conditioned: the two formulations
    #  associate the same factors differently, and their fp32 difference is the product's condition number times L eps)
    noise = 0.5 / (max(shape[dim], 1) * d) ** 0.5
    x = (torch.eye(d, dtype=dtype, device="cuda") + noise * torch.randn(*shape, dtype=dtype, device="cuda")).contiguous()
    want = O.cumops_(x.clone(), dim, (lambda a, b: b @ a) if left else (lambda a, b: a @ b))
    calls = []
    from pypose_amd import _C
    real = _C.check
    _C.check = lambda code, what: (calls.append(what), real(code, what))[1]
    try:
        got = pp.cumprod(x, dim, left=left)
    finally:
        _C.check = real
    assert calls == (["pplie_scan_mat"] if x.numel() else ["pplie_scan_mat"]) or x.numel() == 0
    assert got.shape == want.shape and got is not x
    if x.numel():
        err = (got - want).abs().amax((-1, -2)) / want.abs().amax((-1, -2)).clamp_min(1e-30)
        assert float(err.max()) <= tol, float(err.max())
    # elementwise cummul of the same tensor is NOT the matrix route, and a gradient keeps the differentiable formulation
    torch.testing.assert_close(pp.cummul(x, dim, left=left), torch.cumprod(x, dim), rtol=1e-4 if dtype == torch.float32 else 1e-10, atol=1e-30)     # (off-diagonal entries underflow)
    if x.numel():
        xg = x.clone().requires_grad_(True)
        y = pp.cumprod(xg, dim, left=left)
        assert y.requires_grad
        y.sum().backward()
        assert xg.grad is not None and torch.isfinite(xg.grad).all()
