"""The kernels' per-row arithmetic (pypose_amd/csrc/lie_math.h, compiled here for the host)
against the reference's golden vectors -- the fp64-anchored protocol of SURVEY.md section 7:

  (1) new-fp32 vs reference-fp64 (inputs = the fp32 golden inputs, up-cast for the reference)
      <= 1e-5 row-relative on all rows,
  (2) new-fp64 vs reference-fp64 tight, except where the reference's own closed forms lose
      digits (tiny theta), where the bound is the reference's cancellation envelope.

This is a check of the arithmetic header on the CPU; the GPU build of the same header is
checked through the C ABI in tests/test_lie_parity_gpu.py.
"""
import numpy as np
import pytest

from oracle import lie_np
from tests.golden_util import AUTOGRAD_OPS, golden_case, row_rel_err, well_conditioned_rows
from tests.hostmath_util import hostmath_op

ALL_OPS = sorted(lie_np.OPS)

# ops whose reference formulation itself is ill-conditioned on part of the adversarial set
# sim3_Exp: the reference's C = (exp(s)-1)/s (operation.py:112) loses ~eps/|s| digits at |s| -> 0 (golden row
# 97 has s = 1e-9 -> 1e-7 relative noise in the reference's own fp64 result); the kernels use expm1.
LOOSE64 = {"sim3_exp_fwd": 1e-6}


@pytest.mark.parametrize("name", ALL_OPS)
def test_fp64_vs_reference(golden, name):
    ins, refs = golden_case(golden, "f64", name)
    outs = hostmath_op(name, ins)
    m = well_conditioned_rows(name, ins)
    for o, r in zip(outs, refs):
        assert np.isfinite(o).all()              # also where the reference's autograd returns NaN (theta = 0)
        e, ok = row_rel_err(o[m], r[m])
        assert e.max() < LOOSE64.get(name, 2e-9), (name, e.max(), int(np.argmax(e)))
        assert np.median(e) < 1e-14, (name, np.median(e))


@pytest.mark.parametrize("name", ALL_OPS)
def test_fp32_vs_reference_fp64(golden, name):
    ins32, _ = golden_case(golden, "f32", name)
    # reference-quality answer for exactly these fp32 inputs: the oracle in fp64 (pinned to 1e-11)
    refs = lie_np.OPS[name](*[a.astype(np.float64) for a in ins32])
    outs = hostmath_op(name, ins32)
    m = well_conditioned_rows(name, ins32, theta_min=1e-4)     # the oracle's central differences need theta >> h
    for o, r in zip(outs, refs):
        assert o.dtype == np.float32 and np.isfinite(o).all()
        e, ok = row_rel_err(o[m], r[m])
        assert e.max() < (2e-5 if name in AUTOGRAD_OPS else 1e-5), (name, e.max(), int(np.argmax(e)))


@pytest.mark.parametrize("name", ["sim3_exp_fwd", "sim3_log_fwd"])
def test_sim3_small_sigma_and_theta_fp32(name):
    """The reference's closed forms for rxso3_Ws's A, B (operation.py:117-122) cancel in fp32 when sigma AND theta are small
    (a sigma against (1 - b) theta: A wrong by ~2e-7 / (theta^2 + sigma^2), i.e. 1e-5 of the translation at sigma = theta = 5e-3;
    found by the max-over-rows gate of tests/test_lie_parity_gpu.py, 3 rows in 300 k).  lie_math.h's ws_coef uses the
    coefficients' series there in fp32: every row within 1e-5 of the reference's formulas evaluated in fp64."""
    rng = np.random.default_rng(11)
    n = 20_000
    d = rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    theta = 10.0 ** rng.uniform(-5, -0.8, (n, 1))
    sigma = 10.0 ** rng.uniform(-5, -0.8, (n, 1)) * rng.choice([-1.0, 1.0], (n, 1))
    x = np.concatenate([rng.standard_normal((n, 3)), d * theta, sigma], -1).astype(np.float32)
    if name == "sim3_exp_fwd":
        ins = [x]
    else:
        ins = [lie_np.sim3_exp_fwd(x.astype(np.float64))[0].astype(np.float32)]
    ref = lie_np.OPS[name](*[a.astype(np.float64) for a in ins])[0]
    out = hostmath_op(name, ins)[0]
    e, _ = row_rel_err(out, ref)
    assert e.max() < 1e-5, (name, e.max(), ins[0][int(np.argmax(e))])
