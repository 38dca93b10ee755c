"""The C-ABI shared library loads and exports every symbol include/pplie.h declares (no compute)."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "pplie.h")
LIB = os.path.join(ROOT, "pypose_amd", "lib", "libpplie.so")


def declared_symbols():
    pre = subprocess.run(["gcc", "-E", "-P", HDR], capture_output=True, text=True, check=True).stdout
    return sorted(set(re.findall(r"\bint\s+(pplie_[a-z0-9_]+)\s*\(", pre)))


def test_header_is_valid_c():
    subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Werror", "-x", "c", HDR], check=True)
    subprocess.run(["g++", "-fsyntax-only", "-Wall", "-Werror", "-x", "c++", HDR], check=True)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        import pypose_amd.build as b
        b.build(verbose=False)
    syms = declared_symbols()
    assert len(syms) >= 138
    nm = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (pplie_[a-z0-9_]+)", nm))
    missing = [s for s in syms if s not in exported]
    assert not missing, missing


def test_library_loads():
    import torch  # noqa: F401  (brings PyTorch's libamdhip64 in first, as the product does)
    lib = ctypes.CDLL(LIB)
    for s in declared_symbols():
        assert getattr(lib, s) is not None


def test_product_has_no_cpu_path():
    import pytest
    import torch
    import pypose_amd as pp
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU compute path"):
        pp.randn_so3(2).Exp()


def test_torch_extension_builds_and_imports():
    """pypose_amd/csrc_torch/pplie_autograd.cpp (native autograd nodes around the row kernels) compiles against the installed torch
    without a GPU and exposes its two entry points; the product uses it only on HIP tensors (tests/test_native_autograd_gpu.py)."""
    import importlib.util
    from pypose_amd.build import build_torch_ext
    path = build_torch_ext(verbose=False)
    assert path is not None and path.exists()
    import torch  # noqa: F401  (the extension links against libtorch)
    spec = importlib.util.spec_from_file_location("pplie_torch_ext", str(path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert callable(mod.row_op) and callable(mod.set_rule)


def test_prototype_cache_is_not_poisoned_by_address_lookups():
    """The native autograd nodes resolve entry ADDRESSES (HipLibrary.address); a later ctypes caller of the same entry must still get
    its own prototype, also when someone resolved it without one first (18-argument entries called through the 7-argument row
    prototype pass truncated pointers)."""
    import ctypes
    from pypose_amd import _C
    from pypose_amd.module import imu_preintegrator as im
    lib = _C.HipLibrary()
    addr = lib.address("pplie_imu_integrate_f32")
    assert isinstance(addr, int) and addr != 0 and "pplie_imu_integrate_f32" not in lib._fns
    fn = lib.symbol("pplie_imu_integrate_f32", im._INT_SIG)
    assert len(fn.argtypes) == len(im._INT_SIG) == 18
    lib2 = _C.HipLibrary()
    assert len(lib2.symbol("pplie_imu_integrate_bwd_f64").argtypes) == 7                  # (resolved blind: the row prototype)
    assert len(lib2.symbol("pplie_imu_integrate_bwd_f64", im._INTB_SIG).argtypes) == len(im._INTB_SIG) == 17
    assert ctypes.cast(fn, ctypes.c_void_p).value == addr
