"""Drop-in check: the reference's OWN test files, run unmodified from /root/reference against this package aliased
as ``pypose`` (tests/run_reference_tests.py).  Only possible where the reference is mounted (this container); on the
GPU box the test skips -- nothing under -m gpu reads /root/reference.

Known, intended difference (everything else must pass):
  * test_lietensor.py::test_parameter_dispatch monkeypatches ``pypose._require_backend_attr``, the loader of the
    external ``bae`` plugin (pypose/__init__.py) -- out of scope, DESIGN.md section 7.
"""
import os
import re
import subprocess
import sys
from pathlib import Path

import pytest

REF = Path("/root/reference/tests")
ROOT = Path(__file__).resolve().parents[1]
FILES = ["lietensor/test_lietensor.py", "optim/test_optimizer.py", "optim/test_jacobian.py", "optim/test_solver.py",
         "optim/test_scheduler.py", "optim/test_sparse_lm.py", "basics/test_ops.py", "basics/test_func.py",
         "function/test_checking.py", "function/test_spline.py", "module/test_loss.py"]
KNOWN = re.compile(r"test_parameter_dispatch")


@pytest.mark.skipif(not REF.exists(), reason="reference checkout not mounted")
def test_reference_tests_pass_against_this_package():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, str(ROOT / "tests" / "run_reference_tests.py"), "--cpu-oracle",
                          *[str(REF / f) for f in FILES]], capture_output=True, text=True, env=env, cwd="/tmp", timeout=1500)
    text = out.stdout + out.stderr
    failed = [l for l in text.splitlines() if l.startswith(("FAILED", "ERROR"))]
    unexpected = [l for l in failed if not KNOWN.search(l)]
    assert not unexpected, "\n".join(unexpected) + "\n" + text[-3000:]
    m = re.search(r"(\d+) passed", text)
    assert m and int(m.group(1)) >= 60, text[-3000:]
