"""Run the REFERENCE's own pytest files against this package, imported under the name ``pypose``.

    python tests/run_reference_tests.py [--cpu-oracle] /root/reference/tests/lietensor/test_lietensor.py ...

This is the drop-in check of SURVEY.md section 8(b): the files are executed where they lie (nothing is copied) with
``sys.modules["pypose"]`` pointing at ``pypose_amd``.  On a GPU box the kernels run; ``--cpu-oracle`` installs the
test-only oracle backend so that the host-side API surface can be checked in the CPU container.  ``torchvision``
(imported by one reference test for ``Compose`` only) is absent from this image and is stubbed.
"""
import importlib
import sys
import types
from pathlib import Path

sys.dont_write_bytecode = True
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import torch  # noqa: E402,F401
import pypose_amd  # noqa: E402

SUBMODULES = ("optim", "optim.solver", "optim.strategy", "optim.kernel", "optim.corrector", "optim.scheduler",
              "optim.functional", "optim.optimizer", "lietensor", "lietensor.lietensor", "lietensor.operation",
              "lietensor.utils", "basics", "module", "autograd", "autograd.function", "function", "testing", "func", "metric")


def alias_as_pypose():
    sys.modules["pypose"] = pypose_amd
    for sub in SUBMODULES:
        sys.modules["pypose." + sub] = importlib.import_module("pypose_amd." + sub)


def stub_torchvision():
    if "torchvision" in sys.modules:
        return
    tv, tr = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")

    class Compose:
        def __init__(self, fns):
            self.fns = fns

        def __call__(self, x):
            for f in self.fns:
                x = f(x)
            return x

    tr.Compose, tv.transforms = Compose, tr
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tr


def main(argv):
    import contextlib
    import pytest
    use_oracle = "--cpu-oracle" in argv
    argv = [a for a in argv if a != "--cpu-oracle"]
    alias_as_pypose()
    stub_torchvision()
    ctx = contextlib.nullcontext()
    if use_oracle:
        from tests.oracle_backend import oracle_backend
        ctx = oracle_backend()
    with ctx:
        return int(pytest.main(["-q", "-p", "no:cacheprovider", "--rootdir=/tmp", "-rf", *argv]))


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
