"""Drop the HIP kernels into an installed ``pypose`` (the reference) without touching its sources.

PyPose has no operator registry: the seam is Python name resolution.  ``pypose/lietensor/lietensor.py``
imports the 32 ``torch.autograd.Function`` classes of ``pypose/lietensor/operation.py`` by name
(:10-22) and resolves them at call time (``SE3_Log.apply(X)`` at :360 etc.); Functions also call each
other through the same module globals (operation.py:378, 404, 550, 860-861, 957-958, 1029).
:func:`activate` rebinds those names, in both modules, to small dispatchers:

    tensors on a HIP device, float32/float64  ->  pypose_amd's Function (one HIP kernel)
    anything else                             ->  the reference's original Function

so ``pypose.LieTensor``, ``pypose.Parameter``, ``pp.optim.*`` and user code keep working unchanged,
on the same objects, and pick up the kernels whenever their data lives on the GPU.
``deactivate()`` restores the originals.  See INTEGRATION.md.
"""
from __future__ import annotations

import importlib

import torch

from .lietensor import operation as _hip_ops

_FUNCTION_NAMES = list(_hip_ops.__all__)          # SO3_Log, so3_Exp, ..., Sim3_AdjTXa (32 names)
_saved = {}


class _Dispatch:
    """Stands in for one reference Function class: only ``.apply`` is ever used by pypose."""

    def __init__(self, name, ours, theirs, force):
        self.__name__ = name
        self._ours, self._theirs, self._force = ours, theirs, force

    def apply(self, *args):
        t = args[0]
        if self._force or (t.is_cuda and t.dtype in (torch.float32, torch.float64)):
            return self._ours.apply(*args)
        return self._theirs.apply(*args)

    def __getattr__(self, item):                  # forward / backward / ... of the original class
        return getattr(self._theirs, item)


def activate(pypose=None, force: bool = False, optim: bool = False):
    """Rebind the Lie-op Functions inside ``pypose`` to the HIP-backed ones.

    pypose: the imported reference package (default: ``import pypose``).
    force:  route every call to pypose_amd regardless of device (used by the test-suite, where
            the HIP launcher is replaced by a stand-in).
    optim:  also rebind ``pypose.optim.LM / LevenbergMarquardt / GN / GaussNewton`` to pypose_amd's optimizers, which
            keep the reference's constructor and ``step`` contract and add the structured (block / pose-graph /
            multi-parameter / fused) linearisations; the reference's own solver, strategy, kernel and corrector objects
            are accepted as they are (same call shapes), its LieTensor parameters are recognised by their ``ltype``.
    """
    pypose = importlib.import_module("pypose") if pypose is None else pypose
    mods = [importlib.import_module(pypose.__name__ + ".lietensor.operation"),
            importlib.import_module(pypose.__name__ + ".lietensor.lietensor")]
    if _saved:
        deactivate()
    for name in _FUNCTION_NAMES:
        theirs = getattr(mods[0], name)
        shim = _Dispatch(name, getattr(_hip_ops, name), theirs, force)
        for m in mods:
            if hasattr(m, name):
                _saved[(m, name)] = getattr(m, name)
                setattr(m, name, shim)
    if optim:
        from .optim import optimizer as _ours
        omods = [importlib.import_module(pypose.__name__ + ".optim"), importlib.import_module(pypose.__name__ + ".optim.optimizer")]
        for name, cls in (("LM", _ours.LevenbergMarquardt), ("LevenbergMarquardt", _ours.LevenbergMarquardt),
                          ("GN", _ours.GaussNewton), ("GaussNewton", _ours.GaussNewton)):
            for m in omods:
                if hasattr(m, name):
                    _saved[(m, name)] = getattr(m, name)
                    setattr(m, name, cls)
    return pypose


def deactivate():
    for (m, name), orig in _saved.items():
        setattr(m, name, orig)
    _saved.clear()
