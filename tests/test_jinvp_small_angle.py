"""Jinvp backward where nothing in fp64 is an anchor (VERDICT r04 "what's weak" 2).

Below theta ~ 1e-3 the reference's own fp64 autograd through ``so3_Jl_inv`` / ``calcQ`` (lietensor.py:422-429,
operation.py:23-58) loses digits (4e-4 at theta = 1e-5) and the finite-difference oracle loses more; the kernels' reverse
sweep (csrc/lie_math.h, series below the switch, cancellation-free forms above) does not.  The comparator here is
oracle/jinvp_mp.py: the reference's closed forms in 40-digit arithmetic.  The host build of lie_math.h runs in the CPU
suite, the HIP kernels (through the C ABI) in the GPU suite, on the same 240 rows with theta from 1e-6 to 1e-1.
"""
import numpy as np
import pytest

from oracle import jinvp_mp, lie_np
from tests.golden_util import row_rel_err

N = 240


def _case(group):
    rng = np.random.default_rng(7)
    d = rng.standard_normal((N, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    phi = d * (10 ** np.linspace(-6, -1, N))[:, None]
    x = phi if group == "so3" else np.concatenate([rng.standard_normal((N, 3)), phi], -1)
    w = x.shape[1]
    X = lie_np.OPS[f"{group}_exp_fwd"](x)[0].astype(np.float32)
    p, g = rng.standard_normal((N, w)).astype(np.float32), rng.standard_normal((N, w)).astype(np.float32)
    # the truth is a function of the fp32-rounded X the kernel sees: x = Log(X) in fp64, gradient there, then Log's backward
    xl = lie_np.OPS[f"{group}_log_fwd"](X.astype(np.float64))[0]
    gx, gp = jinvp_mp.jinvp_algebra_grad(group, xl, p.astype(np.float64), g.astype(np.float64))
    gX = lie_np.OPS[f"{group}_log_bwd"](xl, gx)[0]
    return (X, p, g), (gX, gp)


_CASES = {}


def case(group):
    if group not in _CASES:
        _CASES[group] = _case(group)
    return _CASES[group]


def check(run, group):
    (X, p, g), (gX, gp) = case(group)
    name = f"{group}_jinvp_bwd"
    for dtype, tol in ((np.float32, 2e-6), (np.float64, 1e-7)):
        oX, op = run(name, [X.astype(dtype), p.astype(dtype), g.astype(dtype)])
        eX, _ = row_rel_err(oX[:, :gX.shape[1]], gX)
        ep, _ = row_rel_err(op, gp)
        assert eX.max() < tol, (name, dtype, eX.max(), int(np.argmax(eX)))
        assert ep.max() < tol, (name, dtype, ep.max())


@pytest.mark.parametrize("group", ["se3", "so3"])
def test_host_build_of_the_kernel_arithmetic_against_40_digits(group):
    from tests.hostmath_util import hostmath_op
    check(hostmath_op, group)


def test_the_fp64_comparators_are_not_anchors_below_1e_3():
    """what the docstring claims, measured: the finite-difference oracle is off by more than 1e-4 somewhere below theta = 1e-3
    (so it cannot gate a kernel at 1e-5 there) while it is good to 2e-5 from 1e-2 up (where test_lie_parity_gpu.py uses it)"""
    (X, p, g), (gX, _) = case("se3")
    fd = lie_np.se3_jinvp_bwd(X.astype(np.float64), p.astype(np.float64), g.astype(np.float64))[0]
    e, _ = row_rel_err(fd[:, :6], gX[:, :6])
    theta = np.linalg.norm(lie_np.se3_log_fwd(X.astype(np.float64))[0][:, 3:], axis=-1)
    assert e[theta < 1e-3].max() > 1e-4
    assert e[theta >= 1e-2].max() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("group", ["se3", "so3"])
def test_hip_kernels_against_40_digits(group):
    from tests.test_lie_parity_gpu import run_hip
    check(run_hip, group)
