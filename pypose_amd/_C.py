"""ctypes binding of the C ABI declared in ``include/pplie.h`` (lib/libpplie.so).

The library is pure HIP behind ``extern "C"`` entry points: plain pointers, row counts and a
``hipStream_t``; it never allocates, frees or synchronises.  PyTorch is used here only as the
owner of device memory and streams (``tensor.data_ptr()``, ``current_stream().cuda_stream``).

There is **no CPU fallback**: if the shared library is missing, or a tensor is not on a HIP
device, the call raises.  (Tests that exercise host-side logic without a GPU install their own
backend through :func:`set_backend_for_testing`; the product never does.)
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import torch

_LIB_PATH = Path(__file__).resolve().parent / "lib" / "libpplie.so"
if os.environ.get("PPLIE_LIBRARY_FILE"):       # measurement tools only: an alternative build of the same ABI, next to libpplie.so
    _LIB_PATH = _LIB_PATH.with_name(os.environ["PPLIE_LIBRARY_FILE"])

_ERRORS = {-1: "bad argument (negative row count or null pointer)", -2: "kernel launch failed (hipGetLastError)",
           -3: "the problem does not fit the device-resident variant (nothing launched)"}
ECAPACITY = -3

_ROW_SIG = [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_void_p]


class HipLibrary:
    """Lazy handle on libpplie.so."""

    def __init__(self, path: Path = _LIB_PATH):
        self.path = Path(path)
        self._cdll = None
        self._fns = {}

    @property
    def cdll(self):
        if self._cdll is None:
            if not self.path.exists():
                raise ImportError(
                    f"pypose_amd: HIP library {self.path} not found. Build it with "
                    f"`python -m pypose_amd.build` (needs hipcc, gfx950 cross-compiles without a GPU). "
                    f"There is no CPU fallback.")
            # torch must already be imported so that libamdhip64.so.7 resolves to the runtime
            # PyTorch itself uses (one HIP runtime per process: streams/pointers are shared).
            self._cdll = ctypes.CDLL(str(self.path), mode=os.RTLD_NOW | os.RTLD_LOCAL)
        return self._cdll

    def symbol(self, name: str, argtypes=None):
        fn = self._fns.get(name)
        if fn is None:
            try:
                fn = getattr(self.cdll, name)
            except AttributeError as e:
                raise AttributeError(f"pypose_amd: symbol {name} missing from {self.path}") from e
            fn.argtypes = _ROW_SIG if argtypes is None else argtypes
            fn.restype = ctypes.c_int
            self._fns[name] = fn
        elif argtypes is not None and len(fn.argtypes) != len(argtypes):
            fn.argtypes = argtypes          # (first resolved without a prototype: the caller that knows the signature wins)
        return fn

    def address(self, name: str) -> int:
        """entry address of an export (for the native autograd nodes, which call through function pointers); leaves the
        ctypes prototype cache of ``symbol`` alone -- an entry first seen here must not get the row-operator signature"""
        try:
            return ctypes.cast(getattr(self.cdll, name), ctypes.c_void_p).value
        except AttributeError as e:
            raise AttributeError(f"pypose_amd: symbol {name} missing from {self.path}") from e

    def has(self, name: str) -> bool:
        try:
            getattr(self.cdll, name)
            return True
        except AttributeError:
            return False


_lib = HipLibrary()
_SUFFIX = {torch.float32: "_f32", torch.float64: "_f64"}
_test_backend = None


def tune_library() -> HipLibrary:
    """lib/libpplie_tune.so (build.build_tune): launch-shape variants for the measurement tools; never loaded by the product"""
    return HipLibrary(_LIB_PATH.with_name("libpplie_tune.so"))


def library() -> HipLibrary:
    return _lib


def set_backend_for_testing(backend):
    """Install a stand-in for the HIP row-op launcher (tests only; ``None`` restores HIP).

    ``backend(name, ins, out_widths) -> tuple[Tensor, ...]`` with ``name`` like ``"se3_exp_fwd"``.
    """
    global _test_backend
    _test_backend = backend


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def check(code: int, what: str):
    if code != 0:
        raise RuntimeError(f"pypose_amd: {what} failed: {_ERRORS.get(code, code)}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


class DryTraceEscape(RuntimeError):
    """a kernel launch was attempted during a dry trace (optim/fused.py DryTracer): the model does something a dry run
    cannot follow; the caller falls back to a real forward"""


import threading as _threading
_tls = _threading.local()          # .dry > 0: this thread is running a model's Python WITHOUT launching kernels (optim/fused.py)


def dry_tracing() -> bool:
    return getattr(_tls, "dry", 0) > 0


def stream_ptr(device) -> ctypes.c_void_p:
    """hipStream_t of torch's current stream on ``device`` (the raw-stream query is ~10x cheaper than building a
    torch.cuda.Stream object; this sits on the launch path of every op).  Every launch of the library asks for its stream
    here, which makes this the one place where a dry trace stops anything that is not a traced row op."""
    if getattr(_tls, "dry", 0):
        raise DryTraceEscape("pypose_amd: kernel launch during a dry trace")
    if _raw_stream is not None:
        idx = device.index if isinstance(device, torch.device) else device
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device() if idx is None else idx))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _on_device:
    """``with torch.cuda.device(d)`` only when ``d`` is not already current (the context manager costs ~5 us)."""
    __slots__ = ("ctx",)

    def __init__(self, device):
        self.ctx = None if device.index is None or device.index == torch.cuda.current_device() else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


class graph_capture:
    """``with torch.cuda.graph(g)`` without a cyclic-GC pass inside the capture: a collection may finalise unrelated device
    objects (another captured graph, events, streams) whose destructors call HIP APIs that are illegal while a stream is
    capturing -- the process aborts.  (torch.cuda.graph collects once on entry; a Python-heavy capture body can trigger
    another.)"""

    def __init__(self, g):
        # thread_local: only THIS thread's calls are policed while the stream captures -- another thread's event query or
        # allocation (the RCCL watchdog of a process that also holds a process group, a pinned-memory loader) must not
        # invalidate a capture it has nothing to do with
        self.ctx = torch.cuda.graph(g, capture_error_mode="thread_local")

    def __enter__(self):
        import gc
        self.was_enabled = gc.isenabled()
        self.ctx.__enter__()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        try:
            return self.ctx.__exit__(*exc)
        finally:
            if self.was_enabled:
                gc.enable()


def mark_written(t):
    """Bump the version counter of a tensor a kernel has just written through its raw pointer.  Autograd detects in-place
    modification of tensors saved for backward by comparing version counters; a raw-pointer write that skipped this
    would turn "one of the variables needed for gradient computation has been modified" into silently wrong gradients."""
    if _test_backend is None:          # (the stand-in writes with copy_, which already bumps it)
        torch.autograd.graph.increment_version(t)


_staging_warned = False


def _warn_staging(name):
    """once per process: host tensors are computed on the GPU and copied back (the reference computes them on the CPU)"""
    global _staging_warned
    if not _staging_warned and not os.environ.get("PPLIE_QUIET_STAGING"):
        _staging_warned = True
        import warnings
        warnings.warn(f"pypose_amd: op {name} received host tensors; they are staged through the current HIP device (there is no "
                      f"CPU compute path) -- move the data to the device once instead (set PPLIE_QUIET_STAGING=1 to silence)",
                      stacklevel=4)


def row_op(name: str, ins, out_widths, out=None):
    """Launch ``pplie_<name>_{f32,f64}`` on contiguous ``[N, W]`` inputs.

    ins: 1-3 tensors, same dtype/device, same N, contiguous. out_widths: 1-2 ints.
    Returns a tuple of freshly allocated ``[N, W_out]`` tensors, or of the caller's ``out`` tensors (contiguous
    ``[N, W_out]``; an output may alias an input of the same width: every row kernel reads a tile's rows into LDS /
    registers before it writes that tile).
    """
    if _test_backend is not None:
        res = _test_backend(name, ins, out_widths)
        if out is None:
            return res
        for o, r in zip(out, res):
            o.copy_(r)
        return tuple(out)
    x0 = ins[0]
    if not x0.is_cuda:
        # Host tensors are staged through the GPU (the arithmetic still runs in the HIP kernel);
        # with no GPU present this raises -- there is no CPU implementation to fall back to.
        if not torch.cuda.is_available():
            raise RuntimeError(
                f"pypose_amd: op {name} needs a HIP device (got {x0.device} tensors and no GPU is visible); "
                f"there is no CPU compute path.")
        _warn_staging(name)
        dev = torch.device("cuda", torch.cuda.current_device())
        outs = row_op(name, [t.to(dev) for t in ins], out_widths)
        if out is None:
            return tuple(o.to(x0.device) for o in outs)
        for o, r in zip(out, outs):
            o.copy_(r)
        return tuple(out)
    suffix = _SUFFIX.get(x0.dtype)
    if suffix is None:
        raise TypeError(f"pypose_amd: op {name} supports float32/float64, got {x0.dtype}")
    n = x0.shape[0]
    for t in ins:
        if t.dtype != x0.dtype or t.device != x0.device or t.shape[0] != n or t.dim() != 2 or not t.is_contiguous():
            raise ValueError(f"pypose_amd: op {name}: inputs must be contiguous [N,W], same N/dtype/device")
    if out is None:
        outs = tuple(torch.empty((n, w), dtype=x0.dtype, device=x0.device) for w in out_widths)
    else:
        outs = tuple(out)
        for o, w in zip(outs, out_widths):
            if o.shape != (n, w) or o.dtype != x0.dtype or o.device != x0.device or not o.is_contiguous():
                raise ValueError(f"pypose_amd: op {name}: `out` must be contiguous [N,{w}] of the inputs' dtype / device")
    if n == 0:
        return outs
    fn = _lib.symbol("pplie_" + name + suffix)
    pi = [t.data_ptr() for t in ins] + [None] * (3 - len(ins))
    po = [t.data_ptr() for t in outs] + [None] * (2 - len(outs))
    with _on_device(x0.device):
        code = fn(*pi, *po, n, stream_ptr(x0.device))
    if code:
        check(code, "pplie_" + name + suffix)
    return outs


def param_op(name: str, ins, out_width: int, prm: float):
    """Row ops with one launch-wide scalar parameter (``pplie_mat2so3_*``, ``pplie_so3_euler_*``):
    ``fn(in0[, in1], out, prm, n, stream)``.  Same staging / error rules as :func:`row_op`."""
    if _test_backend is not None:
        return _test_backend(name, ins, (out_width,), prm)[0]
    x0 = ins[0]
    if not x0.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError(
                f"pypose_amd: op {name} needs a HIP device (got {x0.device} tensors and no GPU is visible); "
                f"there is no CPU compute path.")
        _warn_staging(name)
        dev = torch.device("cuda", torch.cuda.current_device())
        return param_op(name, [t.to(dev) for t in ins], out_width, prm).to(x0.device)
    suffix = _SUFFIX.get(x0.dtype)
    if suffix is None:
        raise TypeError(f"pypose_amd: op {name} supports float32/float64, got {x0.dtype}")
    n = x0.shape[0]
    for t in ins:
        if t.dtype != x0.dtype or t.device != x0.device or t.shape[0] != n or t.dim() != 2 or not t.is_contiguous():
            raise ValueError(f"pypose_amd: op {name}: inputs must be contiguous [N,W], same N/dtype/device")
    out = torch.empty((n, out_width), dtype=x0.dtype, device=x0.device)
    if n == 0:
        return out
    sig = [ctypes.c_void_p] * (len(ins) + 1) + [ctypes.c_double, ctypes.c_int64, ctypes.c_void_p]
    fn = _lib.symbol("pplie_" + name + suffix, sig)
    with _on_device(x0.device):
        code = fn(*[t.data_ptr() for t in ins], out.data_ptr(), float(prm), n, stream_ptr(x0.device))
    if code:
        check(code, "pplie_" + name + suffix)
    return out
