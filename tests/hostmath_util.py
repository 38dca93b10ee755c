"""Build + call tests/hostmath (lie_math.h compiled for the host). Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

from oracle import lie_np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "hostmath", "hostmath.cpp")
_OUT = os.path.join(_HERE, "hostmath", "_build", "libhostmath.so")
_HDR = os.path.join(os.path.dirname(_HERE), "pypose_amd", "csrc", "lie_math.h")
_lib = None


def hostmath_lib():
    global _lib
    if _lib is None:
        os.makedirs(os.path.dirname(_OUT), exist_ok=True)
        stale = (not os.path.exists(_OUT)) or os.path.getmtime(_OUT) < max(os.path.getmtime(_SRC), os.path.getmtime(_HDR))
        if stale:
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                                   "-I" + os.path.dirname(_HDR), _SRC, "-o", _OUT])
        _lib = ctypes.CDLL(_OUT)
    return _lib


def hostmath_op(name, ins):
    """Run op ``name`` of lie_math.h on the host for numpy inputs [N,W]; returns tuple of outputs."""
    lib = hostmath_lib()
    dt = ins[0].dtype
    suffix = {np.dtype(np.float32): "_f32", np.dtype(np.float64): "_f64"}[np.dtype(dt)]
    fn = getattr(lib, "hm_" + name + suffix)
    fn.restype = None
    iw, ow = lie_np.op_signature(name)
    n = ins[0].shape[0]
    ins = [np.ascontiguousarray(a, dtype=dt) for a in ins]
    for a, w in zip(ins, iw):
        assert a.shape == (n, w), (name, a.shape, w)
    outs = [np.empty((n, w), dtype=dt) for w in ow]
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    args = [P(a) for a in ins] + [None] * (3 - len(ins)) + [P(o) for o in outs] + [None] * (2 - len(outs))
    fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64]
    fn(*args, ctypes.c_int64(n))
    return tuple(outs)
