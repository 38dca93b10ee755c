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

``module=True`` extends the same device dispatch to the parts of the path that are NOT Functions in the reference:

* ``pypose.basics.ops.cumprod / cumprod_ / cummul / cummul_`` (basics/ops.py:59-204: Hillis-Steele, log2(L) rounds of
  gather + product + scatter) -> the single-pass wave scan ``pplie_scan_*`` on group LieTensors;
* ``<Group>Type.Jinvp`` (lietensor.py:257-264, 422-429, 556-563, 700-707: eager ``so3_Jl_inv(Log X) @ p``) -> the fused
  ``pplie_<g>_jinvp_{fwd,bwd}`` Functions; ``so3Type.Jr`` (:343-351) -> ``pplie_so3_jr_{fwd,bwd}``;
* ``pypose.module.IMUPreintegrator.forward`` (module/imu_preintegrator.py:128-312) -> the two fused kernels
  ``pplie_imu_integrate`` + ``pplie_imu_cov2`` on the reference module's own buffers (BASELINE configs[4]).
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
        if self._force or ((t.is_cuda or t.device.type == "meta") and t.dtype in (torch.float32, torch.float64)):
            # (`meta`: results of a dry trace of the model, optim/fused.py DryTracer -- the op is recorded, nothing runs)
            return self._ours.apply(*args)
        return self._theirs.apply(*args)

    def __getattr__(self, item):                  # forward / backward / ... of the original class
        return getattr(self._theirs, item)


def _to_ours(t):
    """a reference LieTensor as ours (same storage), anything else unchanged"""
    from .lietensor import lietensor as L
    lt = getattr(t, "ltype", None)
    if t is None or lt is None or isinstance(t, L.LieTensor):
        return t
    return L._wrap(torch.Tensor.as_subclass(t, torch.Tensor), getattr(L, type(lt).__name__.replace("Type", "_type")))


def _ours_cached(module, slot, t):
    """``_to_ours(t)``, the same object for the same source object (the fused route caches what it derives from the initial
    state -- r0^-1, broadcast copies -- by object identity + version)"""
    cache = module.__dict__.setdefault('_pplie_ours', {})
    hit = cache.get(slot)
    if hit is not None and hit[0] is t:
        return hit[1]
    out = _to_ours(t)
    cache[slot] = (t, out)
    return out


def _to_theirs(pypose, t):
    from .lietensor import lietensor as L
    if isinstance(t, L.LieTensor):
        return pypose.LieTensor(t.tensor(), ltype=getattr(pypose, type(t.ltype).__name__.replace("Type", "_type")))
    return t


def _on_device(t, force):
    t = torch.Tensor.as_subclass(t, torch.Tensor)
    return force or (t.is_cuda and t.dtype in (torch.float32, torch.float64))


def _rebind_everywhere(pypose, name, orig, new):
    """every loaded module of the package that imported ``orig`` by name gets ``new``"""
    import sys
    prefix = pypose.__name__
    for modname, mod in list(sys.modules.items()):
        if mod is not None and (modname == prefix or modname.startswith(prefix + ".")) and mod.__dict__.get(name) is orig:
            _saved[(mod, name)] = orig
            setattr(mod, name, new)


def _activate_scans(pypose, force):
    from .basics.scan import try_scan_
    ops = importlib.import_module(pypose.__name__ + ".basics.ops")
    for base in ("cumprod", "cummul"):
        orig_ip, orig = getattr(ops, base + "_"), getattr(ops, base)

        def inplace(input, dim, left=True, _orig=orig_ip):
            done = try_scan_(input, dim, left) if _on_device(input, False) else None     # (one launch, O(L) work)
            return done if done is not None else _orig(input, dim, left)

        def outofplace(input, dim, left=True, _ip=inplace, _orig=orig):
            if _on_device(input, False):          # (with a gradient: one autograd node, backward = pplie_scan_<g>_bwd)
                return _ip(input.clone(), dim, left)
            return _orig(input, dim, left)
        inplace.__name__, outofplace.__name__ = base + "_", base
        _rebind_everywhere(pypose, base + "_", orig_ip, inplace)
        _rebind_everywhere(pypose, base, orig, outofplace)


def _activate_jinvp_jr(pypose, force):
    from .lietensor.operation import broadcast_inputs
    lt = importlib.import_module(pypose.__name__ + ".lietensor.lietensor")
    LieTensor = lt.LieTensor
    for cap, alg in (("SO3", "so3"), ("SE3", "se3"), ("Sim3", "sim3"), ("RxSO3", "rxso3")):
        cls, fn, alg_type = getattr(lt, cap + "Type"), getattr(_hip_ops, cap + "_Jinvp"), getattr(lt, alg + "_type")
        orig = cls.__dict__["Jinvp"]

        def Jinvp(self, X, p, _orig=orig, _fn=fn, _alg=alg_type):
            Xt = X.tensor() if isinstance(X, LieTensor) else X
            if not _on_device(Xt, force):
                return _orig(self, X, p)
            pt = p.tensor() if isinstance(p, LieTensor) else p
            (x, y), out_shape = broadcast_inputs(Xt, pt)
            out = _fn.apply(x, y)
            return LieTensor(out.view(tuple(out_shape) + (-1 if out.nelement() != 0 else y.shape[-1],)), ltype=_alg)
        _saved[(cls, "Jinvp")] = orig
        cls.Jinvp = Jinvp
    cls = lt.so3Type
    orig = cls.__dict__["Jr"]

    def Jr(self, X, _orig=orig):
        Xt = X.tensor() if isinstance(X, LieTensor) else X
        return _hip_ops.so3_Jr.apply(Xt) if _on_device(Xt, force) else _orig(self, X)
    _saved[(cls, "Jr")] = orig
    cls.Jr = Jr


_IMU_HELPERS = ("_fused_ok", "_rij_offset", "_bcast", "_gravity_host", "_fused_cov2", "_fused_integrate", "_launch_integrate", "_isotropic")


def _activate_imu(pypose):
    from .module.imu_preintegrator import IMUPreintegrator as Ours
    mod = importlib.import_module(pypose.__name__ + ".module.imu_preintegrator")
    cls = mod.IMUPreintegrator
    orig = cls.__dict__["forward"]

    def forward(self, dt, gyro, acc, rot=None, gyro_cov=None, acc_cov=None, init_state=None):
        """the fused route of pypose_amd's module on THIS module's buffers when everything lives on the GPU (with gradients
        w.r.t. dt / gyro / acc / the initial state: one autograd node, backward = pplie_imu_integrate_bwd); the reference's
        own forward otherwise"""
        st = init_state if init_state is not None else {'pos': self.pos, 'rot': self.rot, 'vel': self.vel}
        c = self._check
        if not Ours._fused_ok(self, c(dt), c(gyro), c(acc), c(rot) if rot is not None else None, st):
            return orig(self, dt, gyro, acc, rot=rot, gyro_cov=gyro_cov, acc_cov=acc_cov, init_state=init_state)
        mine = dict(st)
        mine['rot'] = _ours_cached(self, 'rot', st['rot'])
        if self.prop_cov:
            mine['Rij'] = _ours_cached(self, 'Rij', st['Rij'] if 'Rij' in st else self.Rij)
        out = Ours.forward(self, dt, gyro, acc, rot=_to_ours(rot), gyro_cov=gyro_cov, acc_cov=acc_cov, init_state=mine)
        if not self.reset:                        # the state the module carries to the next call, in the reference's own types
            self.rot = _to_theirs(pypose, self.rot)
            self.Rij = _to_theirs(pypose, self.Rij)
        return {k: _to_theirs(pypose, v) for k, v in out.items()}
    for name in _IMU_HELPERS:
        _saved[(cls, name)] = _ABSENT
        setattr(cls, name, Ours.__dict__[name])
    _saved[(cls, "forward")] = orig
    cls.forward = forward


_ABSENT = object()


def activate(pypose=None, force: bool = False, optim: bool = False, module: bool = False):
    """Rebind the Lie-op Functions inside ``pypose`` to the HIP-backed ones.

    pypose: the imported reference package (default: ``import pypose``).
    force:  route every call to pypose_amd regardless of device (used by the test-suite, where
            the HIP launcher is replaced by a stand-in).
    optim:  also rebind ``pypose.optim.LM / LevenbergMarquardt / GN / GaussNewton`` to pypose_amd's optimizers, which
            keep the reference's constructor and ``step`` contract and add the structured (block / pose-graph /
            multi-parameter / fused) linearisations; the reference's own solver, strategy, kernel and corrector objects
            are accepted as they are (same call shapes), its LieTensor parameters are recognised by their ``ltype``.
    module: also rebind the non-Function parts of the path: the scans of ``pypose.basics.ops``, ``<Group>Type.Jinvp``,
            ``so3Type.Jr`` and ``pypose.module.IMUPreintegrator.forward`` (see the module docstring).
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
    if module:
        _activate_scans(pypose, force)
        _activate_jinvp_jr(pypose, force)
        _activate_imu(pypose)
    return pypose


def deactivate():
    for (m, name), orig in _saved.items():
        if orig is _ABSENT:
            if name in m.__dict__:
                delattr(m, name)
        else:
            setattr(m, name, orig)
    _saved.clear()
