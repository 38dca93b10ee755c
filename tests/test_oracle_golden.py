"""Pin the numpy oracle (oracle/lie_np.py) against outputs of the real reference.

fp64: the oracle follows the reference's formulation, so it must agree to ~1e-12 row-relative.
fp32: both evaluate cancellation-prone closed forms in fp32 with different libm's, so agreement
is only expected inside the reference's own fp32-vs-fp64 noise envelope.
"""
import numpy as np
import pytest

from oracle import lie_np
from tests.golden_util import AUTOGRAD_OPS, golden_case, row_rel_err, well_conditioned_rows

ALL_OPS = sorted(lie_np.OPS)


@pytest.mark.parametrize("name", ALL_OPS)
def test_oracle_matches_reference_fp64(golden, name):
    ins, refs = golden_case(golden, "f64", name)
    outs = lie_np.OPS[name](*ins)
    assert len(outs) == len(refs)
    m = well_conditioned_rows(name, ins)
    tol = 1e-6 if name in AUTOGRAD_OPS else 1e-11        # those oracle entries use central differences
    for o, r in zip(outs, refs):
        assert o.dtype == np.float64 and o.shape == r.shape
        e, ok = row_rel_err(o[m], r[m])
        assert ok.sum() > 80
        assert e.max() < tol, (name, e.max(), np.argmax(e))


@pytest.mark.parametrize("name", ALL_OPS)
def test_oracle_matches_reference_fp32(golden, name):
    ins, refs = golden_case(golden, "f32", name)
    ins64, refs64 = golden_case(golden, "f64", name)
    outs = lie_np.OPS[name](*ins)
    m = well_conditioned_rows(name, ins)
    for o, r in zip(outs, refs):
        assert o.dtype == np.float32 and o.shape == r.shape
        e, ok = row_rel_err(o[m], r[m])
        # median must be at fp32 rounding level; the tail is bounded by the closed forms' noise
        assert np.median(e) < 5e-7, (name, np.median(e))
        assert np.quantile(e, 0.9) < 1e-4, (name, np.quantile(e, 0.9))


def test_signature_table_complete():
    for name in ALL_OPS:
        iw, ow = lie_np.op_signature(name)
        assert 1 <= len(iw) <= 3 and 1 <= len(ow) <= 2
