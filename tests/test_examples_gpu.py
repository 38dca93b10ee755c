"""The example scripts (examples/) run end to end on the GPU at small sizes."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))


def test_pgo_example_synthetic_and_g2o(tmp_path):
    import pgo
    assert pgo.main(["--synthetic", "300", "900", "--steps", "6"]) < 5.0
    out = tmp_path / "opt.g2o"
    loss = pgo.main(["--g2o", os.path.join(ROOT, "tests", "golden", "sample.g2o"), "--steps", "3", "--save", str(out)])
    assert loss == loss and out.exists()


def test_imu_example():
    import imu
    out = imu.main(["--batch", "64", "--steps", "100"])
    assert out["cov"].shape == (64, 1, 9, 9) or out["cov"].shape[-2:] == (9, 9)


def test_ba_example():
    import ba
    l0, l1 = ba.main(["--cameras", "12", "--points", "400", "--per-point", "4", "--steps", "3"])
    assert l1 < 0.5 * l0
