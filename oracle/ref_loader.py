"""ORACLE (test infrastructure only): import the reference package shipped as oracle/_ref/pypose (see oracle/Makefile).

Only tests/, bench.py's cpu_baseline leg, __graft_entry__.smoke() and tools/ use this; pypose_amd never does."""
import importlib
import os
import sys

_REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def available() -> bool:
    return os.path.isdir(os.path.join(_REF, "pypose"))


def load():
    """The reference's `pypose` module (imported from oracle/_ref, without writing bytecode there)."""
    if not available():
        raise ImportError("oracle/_ref/pypose is missing: run `make -C oracle` where /root/reference exists")
    mod = sys.modules.get("pypose")
    if mod is not None and os.path.dirname(os.path.dirname(os.path.abspath(mod.__file__))) == _REF:
        return mod
    old = sys.dont_write_bytecode
    sys.dont_write_bytecode = True
    sys.path.insert(0, _REF)
    try:
        return importlib.import_module("pypose")
    finally:
        sys.path.remove(_REF)
        sys.dont_write_bytecode = old
