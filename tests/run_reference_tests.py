"""Run the REFERENCE's own pytest files against this package, imported under the name ``pypose``.

    python tests/run_reference_tests.py [--cpu-oracle] /root/reference/tests/lietensor/test_lietensor.py ...

This is the drop-in check of SURVEY.md section 8(b): the files are executed where they lie (nothing is copied) with
``sys.modules["pypose"]`` pointing at ``pypose_amd``.  On a GPU box the kernels run (the files pick ``cuda`` when it is
available; host tensors they still create are staged through the kernels); ``--default-cuda`` additionally makes ``cuda`` torch's
default device.  ``--cpu-oracle`` installs the test-only oracle backend so that the host-side API surface can be checked in
the CPU container.  On the GPU box the files are the copies ``oracle/Makefile`` placed under ``oracle/_ref/tests``.  ``torchvision``
(imported by one reference test for ``Compose`` only) is absent from this image and is stubbed.
"""
import importlib
import os
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
    default_cuda = "--default-cuda" in argv
    argv = [a for a in argv if a not in ("--cpu-oracle", "--default-cuda")]
    alias_as_pypose()
    stub_torchvision()
    ctx = contextlib.nullcontext()
    if use_oracle:
        from tests.oracle_backend import oracle_backend
        ctx = oracle_backend()
    if default_cuda:
        # every factory call of the test files (torch.randn, pp.randn_SE3, nn.Module parameters ...) lands on the MI355X
        torch.set_default_device("cuda")
    with ctx:
        rc = int(pytest.main(["-q", "-p", "no:cacheprovider", "--rootdir=/tmp", "-rf", *argv]))
    if os.environ.get("PPLIE_REPORT_LIBS"):
        # evidence for the caller that the kernels -- not a stand-in -- did the arithmetic
        with open("/proc/self/maps") as f:
            print("libpplie.so mapped:", "libpplie.so" in f.read(), "| stand-in backend:", pypose_amd._C._test_backend is not None)
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
