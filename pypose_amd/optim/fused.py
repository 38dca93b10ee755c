"""Whole-step fused Levenberg-Marquardt for recognised residual programs.

The block path (optim/blocks.py) differentiates ANY row-independent model with ``d_res`` batched
backward sweeps and then runs per-problem kernels; every intermediate (J blocks, A, g, D) makes a
round trip through HBM.  For a model whose forward pass is exactly a known chain of Lie kernels the
whole trial step -- Jacobian, normal equations, damped Cholesky, retraction and the new loss -- fits in
registers, one problem per lane (csrc/lm_fused.hip).

Recognition is by *execution trace*, not by model class: :class:`OpTracer` records the HIP kernels the
model's forward launches (name, input tensors, output tensors); a program matches only if the trace is
exactly the expected chain and its last output IS the residual the optimizer sees.  The first fused
step of a given signature is cross-checked against the generic block linearisation.

Programs:

``se3inv``   r = Log(P * X), P an SE3 ``pp.Parameter`` [n,7], X a constant SE3 batch -- the reference's
             README InvNet (README.md:120-129; BASELINE configs[2]), J = [Jl_inv(r) | 0].
``pgo``      r = Log(Z^-1 * nodes[i]^-1 * nodes[j]) -- the reference's PoseGraph (examples/module/pgo/pgo.py:15-25;
             BASELINE metric / configs[3]): residuals and both Jacobian blocks per edge from ONE kernel
             (csrc/pgo_fused.hip) instead of six batched backward sweeps; everything downstream (correctors,
             weights, assembly, PCG, sharding) is the ordinary pose-graph linearisation (optim/posegraph.py).
"""
from __future__ import annotations

import ctypes
import os

import torch

from .. import _C
from ..lietensor import lietensor as _lt
from ..lietensor import operation as _op
from . import blocks as _blocks

_no_tf = torch._C.DisableTorchFunctionSubclass
_PARTIALS = 4096          # PPLIE_LM_TRIAL_PARTIALS: rows of per-workgroup partial sums the kernel may write
_TRIAL_SIG = [ctypes.c_void_p] * 7 + [ctypes.c_double] * 3 + [ctypes.c_int64, ctypes.c_void_p]


class OpTracer:
    """Context manager recording every row kernel launched by the Lie Functions."""

    def __init__(self):
        self.events = []        # (kernel name, inputs, outputs); holds the tensors alive -> stable data_ptrs

    def __enter__(self):
        _op._op_tracers.append(self)
        return self

    def __exit__(self, *exc):
        _op._op_tracers.remove(self)

    def note(self, name, ins, outs):
        self.events.append((name, tuple(ins), tuple(outs)))


class _DryState:
    """What a dry trace keeps from one step to the next (per optimizer): the `meta` outputs by trace position (the same
    program produces the same shapes every step: no allocation), and the signature + match of the latest trace."""
    __slots__ = ("pool", "wrapped", "sig", "match", "base", "touched", "toks")

    def __init__(self):
        self.pool, self.wrapped, self.sig, self.match, self.touched = {}, {}, None, None, False
        # signature tokens of the pooled outputs by id() (the pool keeps them alive, so an id cannot be re-used while it is here):
        # a token is four attribute reads and two tuples, ~1.3 us, and a step asks for the same ones every time
        self.toks = {}
        # every dry output is a view into ONE storage-less `meta` buffer at its own offset: the offset identifies the
        # value across as_subclass / view re-wrappings (a storage_offset() read is ~0.1 us; a storage handle ~1 us)
        self.base = torch.empty((1 << 60,), dtype=torch.uint8, device="meta")


def _meta_id(t):
    """identity of a dry-trace value: its BYTE offset in the tracer's buffer (one window per trace position; element offsets of
    different dtypes can coincide: an fp32 value at position 1 and an fp64 one at position 3)"""
    return -1 - t.storage_offset() * t.element_size()


def _tok(t, toks=None):
    """what a trace signature records of an operand: a dry value's identity AND view (shape / stride / dtype: a model that
    returns r[..., :3] of the same value one step and r the next must not look unchanged), or a real tensor's memory + layout.
    ``toks``: the tracer state's cache of its pooled outputs' tokens (by object identity)"""
    if toks is not None:
        hit = toks.get(id(t))
        if hit is not None:
            return hit
    if t.device.type == "meta":
        return (_meta_id(t), t.shape, t.stride(), t.dtype)
    return (t.data_ptr(), t.shape, t.dtype, t.requires_grad, t.stride(), t.device.index)


class DryTracer(OpTracer):
    """The same recording with NOTHING launched: while it is active on this thread the Lie Functions return `meta` tensors
    (lietensor/operation.py:_launch), row gathers on tracked parameters are noted but not executed (LieTensor.__torch_function__)
    and any other launch of the library raises (_C.stream_ptr).  The model's own Python runs for real -- every step, as in
    the reference (optimizer.py:631, 646) -- so a change that touches no tensor (an attribute flipped, a buffer rebound, a
    different branch taken) shows up in the trace of the very next step; what a dry run cannot follow (a model that looks at
    VALUES of intermediate results) raises out of it and the caller runs a real forward instead."""

    def __init__(self, state=None):
        super().__init__()
        self.state = _DryState() if state is None else state
        self.pos, self.next_offset, self.sig = 0, 16, []
        self.touched = False           # the model applied a torch function to a REAL LieTensor (it may have read parameter values)

    def __enter__(self):
        _C._tls.dry = getattr(_C._tls, "dry", 0) + 1
        return super().__enter__()

    def __exit__(self, *exc):
        super().__exit__(*exc)
        _C._tls.dry -= 1

    def _out(self, shape, dtype):
        """the `meta` output of this trace position (kept across steps while shape and dtype stay the same)"""
        pos = self.pos
        self.pos = pos + 1
        hit = self.state.pool.get(pos)
        if hit is not None and hit[0] == shape and hit[1] is dtype:
            return hit[2]
        if hit is not None:
            self.state.toks.pop(id(hit[2]), None)
        n = 1
        for v in shape:
            n *= v
        esz = torch.empty((), dtype=dtype, device="meta").element_size()
        off = (pos + 1) << 44                                        # byte offset: one 16 TB window per trace position
        t = self.state.base[off:off + max(n, 1) * esz].view(dtype)[:n].view(shape)
        self.state.pool[pos] = (shape, dtype, t)
        self.state.toks[id(t)] = _tok(t)
        return t

    def dry_launch(self, name, ins, in_widths, out_widths, lead):
        x0 = ins[0]
        for t, w in zip(ins, in_widths):
            if t.shape[-1] != w:
                raise ValueError(f"expected last dimension {w}, got shape {tuple(t.shape)}")
            if t.dtype != x0.dtype or (t.device != x0.device and "meta" not in (t.device.type, x0.device.type)):
                raise ValueError(f"pypose_amd: op {name}: inputs must share dtype / device")
        lead = tuple(lead)
        outs = tuple(self._out(lead + (w,), x0.dtype) for w in out_widths)
        self.events.append((name, tuple(ins), outs))
        toks = self.state.toks
        self.sig.append((name,) + tuple(_tok(t, toks) for t in ins) + tuple(_tok(o, toks) for o in outs))
        return outs

    def dry_lie(self, fn, xs, out_ltype):
        """the LieTensor-level entry (lietensor.LieType._dry): operands may be LieTensors; the wrapped result of this trace
        position is kept across steps together with its plain alias"""
        name, in_widths, out_width = fn._dry_kernel
        ins = []
        for x in xs:
            pl = x.__dict__.get("_pl") if type(x) is not torch.Tensor else x
            ins.append(pl if pl is not None else torch.Tensor.as_subclass(x, torch.Tensor))
        x0 = ins[0]
        lead = x0.shape[:-1]
        for t, w in zip(ins, in_widths):
            if t.shape[-1] != w:
                raise ValueError(f"expected last dimension {w}, got shape {tuple(t.shape)}")
            if t.dtype != x0.dtype or (t.device != x0.device and "meta" not in (t.device.type, x0.device.type)):
                raise ValueError(f"pypose_amd: op {name}: inputs must share dtype / device")
            if t.shape[:-1] != lead:
                lead = torch.broadcast_shapes(lead, t.shape[:-1])
        pos = self.pos
        out = self._out(tuple(lead) + (out_width,), x0.dtype)
        ins = tuple(ins)
        self.events.append((name, ins, (out,)))
        toks = self.state.toks
        self.sig.append((name,) + tuple(_tok(t, toks) for t in ins) + (_tok(out, toks),))
        wrapped = self.state.wrapped.get(pos)
        if wrapped is None or wrapped._pl is not out or wrapped.ltype is not out_ltype:
            wrapped = _lt._wrap(out, out_ltype)
            wrapped._pl = out
            self.state.wrapped[pos] = wrapped
        return wrapped

    def dry_gather(self, source, index, recorders):
        """``source[index]`` of a tracked 2-D parameter with an int64 index tensor: a `meta` result, or None (not ours)"""
        if not isinstance(index, torch.Tensor) or index.dtype != torch.int64:
            return None
        src = None
        for rec in recorders:
            src = rec.plain.get(id(source))
            if src is not None:
                break
        if src is None or src.dim() != 2:
            return None
        out = self._out(tuple(index.shape) + (src.shape[-1],), src.dtype)
        self.sig.append(("gather", id(source), _tok(index), _tok(out, self.state.toks)))
        return out


def _key(t):
    """identity of a tensor's memory: the address, or (dry-trace outputs have none) the offset in the dry trace's buffer"""
    if type(t) is not torch.Tensor:
        t = torch.Tensor.as_subclass(t, torch.Tensor)          # (attribute reads on a LieTensor are __torch_function__ round trips)
    if t.device.type == "meta":
        return _meta_id(t)
    return t.data_ptr()


def _is_se3_group(P):
    """an SE3 group LieTensor of this package or of an activated reference pypose (same type name, 7-wide, 6 dof)"""
    lt = getattr(P, "ltype", None)
    return lt is _lt.SE3_type or (lt is not None and type(lt).__name__ == "SE3Type" and tuple(getattr(lt, "dimension", ())) == (7,)
                                  and tuple(getattr(lt, "manifold", ())) == (6,))


def _same(a, b):
    # (the Lie methods flatten leading dims to rows before launching: same storage, same element count)
    a = a if type(a) is torch.Tensor else torch.Tensor.as_subclass(a, torch.Tensor)
    b = b if type(b) is torch.Tensor else torch.Tensor.as_subclass(b, torch.Tensor)
    return a.device == b.device and _key(a) == _key(b) and a.numel() == b.numel() and a.dtype == b.dtype \
        and a.is_contiguous() and b.is_contiguous()


def match_se3inv(trace, R, params):
    """(P, X, r) if the traced forward is exactly  r = se3_log(se3_mul(P, X))  with P the only parameter."""
    if len(trace.events) != 2 or len(R) != 1 or len(params) != 1:
        return None
    (n0, i0, o0), (n1, i1, o1) = trace.events
    P = params[0]
    if n0 != "se3_mul_fwd" or n1 != "se3_log_fwd" or not _is_se3_group(P):
        return None
    A, X = i0
    if not _same(A, P) or X.requires_grad or X.numel() != P.numel() or P.dim() < 2 or not _same(i1[0], o0[0]) \
            or X.dtype != P.dtype or X.device != P.device or not X.is_contiguous():
        return None
    if not _same(o1[0], R[0]) or R[0].shape[-1] != 6:
        return None
    return P, X, o1[0]


class Se3InvLinearization:
    """LM trial steps of ``r = Log(P X)`` in one kernel each (pplie_lm_se3inv_trial)."""

    kind = "fused:se3inv"
    reference_kind = "block"

    def __init__(self, opt, P, X, r, input=None):
        self.opt, self.P = opt, P
        self.n = P.numel() // 7
        self.X_src, self.input = X, input
        self.X = X.detach().reshape(self.n, 7).contiguous()
        # r = None (LM(static=True)): the first trial kernel of the step computes Log(P X) itself and leaves it in a
        # buffer for the retries, so the step needs no separate residual evaluation at all
        self.R = r.detach().reshape(self.n, 6).contiguous() if r is not None else None
        self.scale = 1.0
        self.sums = None

    def build_normal_equations(self, dmin, dmax):
        self.dmin, self.dmax = float(dmin), float(dmax)

    def damp(self, damping):
        self.scale *= 1.0 + float(damping)          # A.diag += A.diag * damping, compounding (optimizer.py:666)

    def _trial(self, out, scale):
        pt = self.P.detach().reshape(self.n, 7)
        assert pt.is_contiguous()
        out = pt if out is None else out
        D = torch.empty((self.n, 7), dtype=pt.dtype, device=pt.device)
        part = torch.zeros((_PARTIALS, 4), dtype=pt.dtype, device=pt.device)
        fn = _C.library().symbol("pplie_lm_se3inv_trial" + _blocks._suffix(pt), _TRIAL_SIG)
        rbuf = torch.empty((self.n, 6), dtype=pt.dtype, device=pt.device) if self.R is None else None
        with _C._on_device(pt.device):
            code = fn(self.R.data_ptr() if self.R is not None else None, rbuf.data_ptr() if rbuf is not None else None,
                      pt.data_ptr(), self.X.data_ptr(), out.data_ptr(), D.data_ptr(), part.data_ptr(),
                      scale, self.dmin, self.dmax, self.n, _C.stream_ptr(pt.device))
        _C.check(code, "pplie_lm_se3inv_trial")
        if out is pt:
            _C.mark_written(self.P)
        if rbuf is not None:
            self.R = rbuf                            # retries of this step linearise at the same point
        return D, part.sum(0)

    def verify(self, ref, dmin, dmax, rtol=1e-3):
        """Undamped trial into a scratch buffer vs the generic block linearisation ``ref`` of the same model."""
        from .solver import Cholesky
        self.build_normal_equations(dmin, dmax)
        D, sums = self._trial(torch.empty_like(self.P.detach().reshape(self.n, 7)), 1.0)
        ref.build_normal_equations(dmin, dmax)
        Dr = ref.solve(Cholesky()).view(self.n, 7)
        Rr = ref.Rb
        ok = bool((D - Dr).abs().max() <= rtol * Dr.abs().max().clamp_min(1e-30))
        return ok and bool((sums[1] - Rr.square().sum()).abs() <= rtol * Rr.square().sum().clamp_min(1e-30))

    def run_trials(self, opt, pg):
        """The trial loop of LM.step (reference optimizer.py:662-678) on the device: see :class:`DeviceLM`."""
        dev = opt.__dict__.get('_device_lm')
        if dev is None or dev.P is not self.P or not dev.same_operand(self.X_src):
            dev = opt._device_lm = DeviceLM(opt, self.P, self.X_src, self.input)
        else:
            dev.rearm(self.input)
        return dev.step()


# ---------------------------------------------------------------------------------------------
# the LM step with its loop state in device memory (csrc/lm_step.hip)
# ---------------------------------------------------------------------------------------------
def _strategy_kind(strategy):
    """0 / 1 / 2 for exactly the built-in Constant / Adaptive / TrustRegion policies -- of this package or of an
    activated reference pypose (same class, same module name, same attributes: strategy.py:41, :134, :248) -- else None:
    the device-side decision implements these three and nothing else (a subclass may override update())."""
    t = type(strategy)
    if t.__module__.split(".")[-2:] != ["optim", "strategy"] or t.__module__.split(".")[0] not in ("pypose", "pypose_amd"):
        return None
    return {"Constant": 0, "Adaptive": 1, "TrustRegion": 2}.get(t.__name__)


class _LmCfg(ctypes.Structure):       # = pplie_lm_cfg (include/pplie.h)
    _fields_ = [(k, ctypes.c_double) for k in ("high", "low", "up", "factor", "smin", "smax", "sdown", "dmin", "dmax",
                                               "host_damping", "host_down")] + \
               [(k, ctypes.c_int) for k in ("strategy", "reject", "flags", "grid_cap", "plateau_patience", "plateau_max_steps")] + \
               [("plateau_decreasing", ctypes.c_double), ("plateau_flag", ctypes.c_uint64)]


_STEP_SIG = [ctypes.c_void_p] * 7 + [ctypes.POINTER(_LmCfg), ctypes.c_int64] + [ctypes.c_void_p] * 3
_SUMS_SIG = [ctypes.c_void_p] * 5 + [ctypes.POINTER(_LmCfg), ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 2
_DECIDE_SIG = [ctypes.c_void_p] * 2 + [ctypes.POINTER(_LmCfg), ctypes.c_int] + [ctypes.c_void_p] * 4
_ST_DAMPING, _ST_RADIUS, _ST_DOWN, _ST_SCALE, _ST_LAST, _ST_LOSS, _ST_REJECTS, _ST_DONE, _ST_FAILED, _ST_TRIALS = range(10)
_ST_PL_STEPS, _ST_PL_COUNT, _ST_PL_STOP = 11, 12, 13        # device-side StopOnPlateau (csrc/lm_common.h)
_LM_STATE = 16            # PPLIE_LM_STATE
_LOSS_BLOCK = 1024        # loss / last scalars handed out as views of one allocation per 1024 steps
_LAZY_KEYS = frozenset(("damping", "radius", "down"))
import os as _os
_TUNE_FLAGS = int(_os.environ.get("PPLIE_LM_FLAGS", "0"))       # tuning knobs of the trial kernel (tools/time_c3_device.py)
_TUNE_GRID = int(_os.environ.get("PPLIE_LM_GRID", "0"))


class LazyGroup(dict):
    """The optimizer's param group while its ``damping`` / ``radius`` / ``down`` entries live in device memory: reading
    one of them first brings the host copy up to date (one small read-back), writing one makes the host copy the
    authority again.  Everything else is a plain dict, so ``state_dict()``, ``repr`` and user code keep working."""

    __slots__ = ("sync",)

    def _pull(self):
        s = self.sync
        if s is not None and s.pending:
            s.flush()

    def _push(self):
        s = self.sync
        if s is not None:
            if s.pending:
                s.flush()
            s.host_dirty = True

    def __getitem__(self, k):
        if k in _LAZY_KEYS:
            self._pull()
        return dict.__getitem__(self, k)

    def get(self, k, default=None):
        if k in _LAZY_KEYS:
            self._pull()
        return dict.get(self, k, default)

    def __setitem__(self, k, v):
        if k in _LAZY_KEYS:
            self._push()
        dict.__setitem__(self, k, v)

    def update(self, *a, **kw):
        self._push()
        dict.update(self, *a, **kw)

    def setdefault(self, k, default=None):
        self._push()
        return dict.setdefault(self, k, default)

    def pop(self, *a):
        self._push()
        return dict.pop(self, *a)

    def items(self):
        self._pull()
        return dict.items(self)

    def values(self):
        self._pull()
        return dict.values(self)

    def copy(self):
        self._pull()
        return dict(self)

    def __iter__(self):
        # (also takes CPython's dict(pg) / {**pg} off their raw-storage fast path: they iterate, and every read pulls)
        self._pull()
        return dict.__iter__(self)

    def keys(self):
        self._pull()
        return dict.keys(self)

    def __repr__(self):
        self._pull()
        return dict.__repr__(self)

    def __reduce__(self):
        self._pull()
        return (dict, (dict(self),))


class DeviceLM:
    """Levenberg-Marquardt steps of the ``se3inv`` program whose loop state -- damping, trust-region radius, last /
    current loss, reject count -- stays in device memory (``pplie_lm_se3inv_step``): a step is two kernel launches
    and NO host synchronisation; ``opt.loss`` is a device scalar as in the reference, ``pg['damping']`` /
    ``pg['radius']`` / ``opt.reject_count`` are read back on first use (:class:`LazyGroup`).

    The reference's semantics are kept (optimizer.py:659-678): ONE loss, ONE damping, ONE accept / reject decision
    per trial over all problems; the damping compounds over rejected retries; a rejected trial restarts from the
    linearisation point (the reference returns there by ``Exp(-d)``, i.e. up to rounding)."""

    native = None        # (the prepared native launch of the se3inv program; subclasses with their own launch keep the ctypes route)

    def __init__(self, opt, P, X_src, input):
        self.opt, self.P, self.X_src = opt, P, X_src
        self.kind = _strategy_kind(opt.strategy)
        self.strategy = opt.strategy
        pt = P.detach()
        self.n = pt.numel() // 7
        self.dtype, self.device = pt.dtype, pt.device
        self.X = X_src.detach().reshape(self.n, 7)
        assert self.X.is_contiguous()
        self.x_ptr, self.x_version = self.X.data_ptr(), X_src._version
        self.p_ptr = pt.data_ptr()
        z = dict(dtype=self.dtype, device=self.device)
        self.save = torch.empty((self.n, 7), **z)
        self.partials = torch.empty((_PARTIALS, 4), **z)
        self.sums = torch.zeros(4, **z)
        # double-buffered loop state: a step reads the buffer the previous step wrote and writes the other one
        self.state = torch.zeros((2, _LM_STATE), dtype=torch.float64, device=self.device)
        self.cur = 0                  # buffer holding the latest state
        self.sync = torch.zeros(8, dtype=torch.int32, device=self.device)
        self.cfg = _LmCfg()
        sfx = _blocks._suffix(pt)
        self.fn = _C.library().symbol("pplie_lm_se3inv_step" + sfx, _STEP_SIG)
        self.kind_name = Se3InvLinearization.kind
        self.fn_sums = _C.library().symbol("pplie_lm_se3inv_trial_sums" + sfx, _SUMS_SIG)
        self.fn_decide = _C.library().symbol("pplie_lm_decide" + sfx, _DECIDE_SIG)
        sp = self.state.data_ptr()
        self.ptrs = (self.save.data_ptr(), self.partials.data_ptr(), (sp, sp + 8 * _LM_STATE), self.sync.data_ptr())
        self.item = pt.element_size()
        self.k = _LOSS_BLOCK
        # the launch behind ONE native call (csrc_torch/pplie_autograd.cpp LmStepHandle: stream lookup, device guard and the C entry
        # with every fixed pointer bound); None: the ctypes route of _launch
        self.native = None
        nat = _op._native()
        if nat is not None and hasattr(nat, "LmStepHandle") and type(self) is DeviceLM and _C._test_backend is None:
            try:
                self.native = nat.LmStepHandle(_C.library().address("pplie_lm_se3inv_step" + sfx), self.p_ptr, self.x_ptr,
                                               self.save.data_ptr(), self.partials.data_ptr(), sp, sp + 8 * _LM_STATE,
                                               self.sync.data_ptr(), ctypes.addressof(self.cfg), self.n,
                                               self.device.index if self.device.index is not None else torch.cuda.current_device())
            except Exception:
                self.native = None
        self.pending = False          # device state newer than the host mirrors
        self.host_dirty = True        # host param group newer than the device state
        self.last_loss = None         # the loss tensor handed out by the previous step
        self.rearm(input)
        self._wrap_group()

    # ---- validity of the shortcut that skips the traced forward ------------------------------------------------
    def same_operand(self, X_src):
        return X_src.data_ptr() == self.x_ptr and X_src.numel() == self.n * 7 and X_src.dtype == self.dtype \
            and X_src.device == self.device and X_src.is_contiguous()

    def rearm(self, input):
        self.input = input

    def _wrap_group(self):
        opt = self.opt
        pg = opt.param_groups[0]
        if type(pg) is not LazyGroup:
            new = LazyGroup(pg)
            new.sync = self
            opt.param_groups[0] = new
            self.host_dirty = True
        elif pg.sync is not self:
            if pg.sync is not None and pg.sync.pending:
                pg.sync.flush()
            pg.sync = self
            self.host_dirty = True
        return opt.param_groups[0]

    def _step_conditions(self, target, weight, m=None):
        opt = self.opt
        if target is not None and (m is None or m[0] != "lpr" or m[1].b is None or not _same(m[1].b, target)):
            return False
        with _no_tf():
            p_ptr = self.P.data_ptr()
        return not (weight is not None or opt.weight is not None or p_ptr != self.p_ptr
                    or opt.strategy is not self.strategy or not opt.fused or not getattr(opt, 'structured', True)
                    or len(opt.param_groups) != 1 or torch.is_inference_mode_enabled())

    def try_step(self, input, target, weight):
        """LM(static=True) only -- the caller's promise that the residual program does not change: the whole of ``LM.step``
        without running the model, while the same ``input`` object(s) and operand storage are passed; else None."""
        if not self._step_conditions(target, weight) or not _same_input(self.input, input) \
                or self.X_src._version != self.x_version or self.X_src.data_ptr() != self.x_ptr:
            return None
        self.opt.linearization = Se3InvLinearization.kind
        return self.step()

    def checked_step(self, m, target, weight):
        """The default: ``m`` is the program this step's (dry) run of the model was matched to.  If it is the program this
        state was built on -- same parameter, same constant operand storage -- the step is two launches; else None."""
        if not self.matches(m) or not self._step_conditions(target, weight, m):
            return None
        self.opt.linearization = self.kind_name
        return self.step()

    # ---- one step ----------------------------------------------------------------------------------------------
    def _fill_cfg(self, pg):
        cfg, opt, st = self.cfg, self.opt, self.strategy
        g = dict.get
        cfg.high, cfg.low, cfg.up, cfg.factor = g(pg, 'high', 0.5), g(pg, 'low', 1e-3), g(pg, 'up', 2.0), g(pg, 'factor', 0.5)
        cfg.smin, cfg.smax, cfg.sdown = getattr(st, 'min', 0.0), getattr(st, 'max', 0.0), getattr(st, 'down', 0.5)
        cfg.dmin, cfg.dmax = g(pg, 'min'), g(pg, 'max')
        cfg.strategy, cfg.reject = self.kind, int(opt.reject)
        flags = 0
        if self.host_dirty:
            flags |= 1
            cfg.host_damping, cfg.host_down = g(pg, 'damping'), g(pg, 'down', 0.5)
        loss = opt.__dict__.get('loss')
        if loss is None:
            flags |= 2
        elif loss is not self.last_loss:            # set by the user or by another linearisation's step
            self.state[self.cur, _ST_LOSS:_ST_LOSS + 1].copy_(torch.as_tensor(loss).detach().reshape(1))
        cfg.flags = flags | _TUNE_FLAGS
        cfg.grid_cap = _TUNE_GRID

    def _slot(self):
        if self.k == _LOSS_BLOCK:
            self.block = torch.empty((2, _LOSS_BLOCK), dtype=self.dtype, device=self.device)
            self.loss_views, self.last_views = self.block[0].unbind(0), self.block[1].unbind(0)
            self.block_ptr = self.block.data_ptr()
            self.k = 0
        k = self.k
        self.k = k + 1
        return k, self.block_ptr + k * self.item, self.block_ptr + (_LOSS_BLOCK + k) * self.item

    def step(self):
        opt = self.opt
        pg = opt.param_groups[0]
        if type(pg) is not LazyGroup or pg.sync is not self:
            pg = self._wrap_group()
        self._fill_cfg(pg)
        k, loss_ptr, last_ptr = self._slot()
        if self.native is not None and opt.group is None:
            code = self.native.launch(self.cur, loss_ptr, last_ptr)
            if code:
                _C.check(code, "pplie_lm_se3inv_step")
            self.cur = 1 - self.cur
        else:
            save, partials, state, sync = self.ptrs
            st_in, st_out = state[self.cur], state[1 - self.cur]
            stream = _C.stream_ptr(self.device)
            if opt.group is None:
                with _C._on_device(self.device):
                    code = self._launch(save, partials, st_in, st_out, sync, loss_ptr, last_ptr, stream)
                if code:
                    _C.check(code, "pplie_lm_*_step")
                self.cur = 1 - self.cur
            else:
                self._sharded(st_in, st_out, loss_ptr, last_ptr, stream)
        _C.mark_written(self.P)                              # P was rewritten through its raw pointer
        self.host_dirty, self.pending = False, True
        opt.loss = self.last_loss = self.loss_views[k]
        opt._last_view = self.last_views[k]
        opt.__dict__.pop('_host_loss', None)
        return opt.loss

    def _launch(self, save, partials, st_in, st_out, sync, loss_ptr, last_ptr, stream):
        return self.fn(self.p_ptr, self.x_ptr, save, partials, st_in, st_out, sync, self.cfg, self.n, loss_ptr, last_ptr, stream)

    def matches(self, m):
        """``m`` (a freshly matched program) is the program this state was built on"""
        return m[0] == "se3inv" and m[1] is self.P and self.same_operand(m[2])

    def _sharded(self, st_in, st_out, loss_ptr, last_ptr, stream):
        """LM(group=...): every rank runs the trial on its problems; the four sums are all-reduced so that all ranks
        take the same decision (3 scalars per trial); retries are driven from the host (one flag read per
        trial) because the all-reduce cannot run inside the finish kernel."""
        import torch.distributed as dist
        save, partials, _, _ = self.ptrs
        first, sums = 1, self.sums.data_ptr()
        self.cur = 1 - self.cur
        while True:
            with _C._on_device(self.device):
                _C.check(self.fn_sums(self.p_ptr, self.x_ptr, save, partials, st_in if first else st_out, self.cfg, first, self.n, sums,
                                      stream), "pplie_lm_se3inv_trial_sums")
                dist.all_reduce(self.sums, group=self.opt.group)
                _C.check(self.fn_decide(st_in if first else st_out, st_out, self.cfg, first, sums, loss_ptr, last_ptr, stream),
                         "pplie_lm_decide")
            st = self.state[self.cur].tolist()
            if st[_ST_FAILED]:
                self.P.detach().reshape(self.n, 7).copy_(self.save)
                break
            if st[_ST_DONE]:
                break
            first = 0

    # ---- StopOnPlateau on the device (optim/scheduler.py optimize()) -----------------------------------------------
    def set_plateau(self, decreasing, patience, max_steps, steps, count):
        """arm the device-side stop rules: from the next step on the finish kernel counts steps / plateau steps and raises the stop flag
        (lm_common.h lm_decide); steps enqueued behind the stop return at once.  ``steps`` / ``count``: the scheduler's counters so far."""
        if type(self) is not DeviceLM or self.opt.group is not None:
            return False
        self.cfg.plateau_decreasing, self.cfg.plateau_patience, self.cfg.plateau_max_steps = float(decreasing), int(patience), int(max_steps)
        self.state[self.cur, _ST_PL_STEPS:_ST_PL_STOP + 1].copy_(torch.tensor([float(steps), float(count), 0.0], dtype=torch.float64))
        # the stopping step announces itself in host-pinned memory (a system-scope store by the finish kernel): the host looks at it
        # before enqueuing each further step -- no synchronisation, and at most the steps already in flight run as no-ops
        flag = self.__dict__.get('_plateau_flag')
        if flag is None:
            flag = self._plateau_flag = torch.zeros(1, dtype=torch.float64).pin_memory()
            self._plateau_flag_np = flag.numpy()
        self._plateau_flag_np[0] = 0.0
        self.cfg.plateau_flag = flag.data_ptr()
        self.plateau = None
        return True

    def plateau_stopped(self):
        return self._plateau_flag_np[0] != 0.0

    def clear_plateau(self):
        self.cfg.plateau_max_steps = 0
        self.cfg.plateau_flag = 0

    def flush(self):
        if not self.pending:
            return
        self.pending = False
        st = self.state[self.cur].tolist()                    # the one read-back
        self.plateau = (int(st[_ST_PL_STEPS]), int(st[_ST_PL_COUNT]), bool(st[_ST_PL_STOP]))
        opt = self.opt
        pg = opt.param_groups[0]
        dict.__setitem__(pg, 'damping', st[_ST_DAMPING])
        if self.kind == 2:
            dict.__setitem__(pg, 'radius', st[_ST_RADIUS])
            dict.__setitem__(pg, 'down', st[_ST_DOWN])
        opt.__dict__['_reject_count'] = int(st[_ST_REJECTS])
        opt.__dict__['_trials'] = int(st[_ST_TRIALS])
        if opt.__dict__.get('loss') is self.last_loss and self.last_loss is not None:
            opt._host_loss = (opt.loss, st[_ST_LOSS])
        if st[_ST_FAILED]:
            print('Cholesky decomposition failed. Check your matrix (may not be positive-definite)',
                  "\nLinear solver failed. Breaking optimization step...")


# ---------------------------------------------------------------------------------------------
# the normal form  r = Log(L * P^s * R) [- b]  /  r = (L * P^s * R) . a [- b]   (csrc/lm_generic.hip)
# ---------------------------------------------------------------------------------------------
_GROUP_OF = {"SO3Type": ("so3", 0, 3, 4), "SE3Type": ("se3", 1, 6, 7), "Sim3Type": ("sim3", 2, 7, 8), "RxSO3Type": ("rxso3", 3, 4, 5)}
_LPR_SIG = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 10 + [ctypes.POINTER(_LmCfg), ctypes.c_int64] + [ctypes.c_void_p] * 3


class LprProgram:
    """One residual program in normal form: the parameter ``P`` (a group LieTensor, one element per problem), ``sign`` (+1: P,
    -1: P^-1), the constant factors to its left and right in the product (``(tensor, inverted)`` pairs, folded into one
    L / R tensor each by real kernel launches -- once, and again only when one of them is written to), ``kind`` 0 (Log)
    or 1 (Act on points ``a``) and the target ``b`` (or None)."""

    def __init__(self, P, gkey, kind, sign, left, right, a, b):
        self.P, self.kind, self.sign, self.left, self.right, self.a, self.b = P, kind, sign, left, right, a, b
        self.g, self.gid, self.da, self.dg = gkey
        self.n = P.numel() // self.dg
        self._folded = None

    def key(self):
        t = lambda x: None if x is None else (x.data_ptr(), tuple(x.shape), x.dtype)
        return (id(self.P), self.kind, self.sign, tuple((t(c), inv) for c, inv in self.left), tuple((t(c), inv) for c, inv in self.right),
                t(self.a), t(self.b))

    def _versions(self):
        return tuple(c._version for c, _ in self.left + self.right)

    def _fold(self, factors):
        if not factors:
            return None
        g, dg, n = self.g, self.dg, self.n
        out = None
        for c, inv in factors:
            c = c.detach().reshape(-1, dg)
            if c.shape[0] != n:
                c = c.expand(n, dg)
            c = c.contiguous()
            if inv:
                c = _C.row_op(g + "_inv_fwd", [c], (dg,))[0]
            out = c if out is None else _C.row_op(g + "_mul_fwd", [out, c], (dg,))[0]
        return out

    def operands(self):
        """(L, R, a, b) as contiguous [n, w] tensors (None where absent); the folds are redone only when a factor was edited"""
        ver = self._versions()
        if self._folded is None or self._folded[0] != ver:
            row = lambda t, w: None if t is None else (t.detach().reshape(-1, w).expand(self.n, w).contiguous())
            dr = self.da if self.kind == 0 else 3
            self._folded = (ver, self._fold(self.left), self._fold(self.right), row(self.a, 3), row(self.b, dr),
                            (None if self.a is None else self.a._version, None if self.b is None else self.b._version))
        elif self._folded[5] != (None if self.a is None else self.a._version, None if self.b is None else self.b._version):
            self._folded = None
            return self.operands()
        return self._folded[1:5]


def match_lpr(trace, outputs, params, target, gathers):
    """An :class:`LprProgram` if the traced forward is  Log / Act  of a product chain (Mul / Inv) over exactly one occurrence
    of the parameter and any number of constant group elements, its output the model's only output; else None."""
    if gathers or len(outputs) != 1 or len(params) != 1 or not trace.events:
        return None
    P = params[0]
    gkey = _GROUP_OF.get(type(getattr(P, "ltype", None)).__name__)
    if gkey is None or P.dim() < 2 or tuple(getattr(P.ltype, "dimension", ())) != (gkey[3],):
        return None
    g, dg = gkey[0], gkey[3]
    Pp = torch.Tensor.as_subclass(P, torch.Tensor)
    if not Pp.is_contiguous():
        return None
    n = Pp.numel() // dg
    by_out = {_key(o[0]): (name, ins) for name, ins, o in trace.events}
    top_out = {_key(o[0]): o[0] for name, ins, o in trace.events}.get(_key(outputs[0]))
    top = by_out.get(_key(outputs[0]))
    # the model's output has to BE the top event's result -- same element count, contiguous: not a slice or a strided view of it
    # (r[..., :3] starts at the same offset), as match_se3inv / match_pgo require of theirs
    if top is None or len(by_out) != len(trace.events) or not _same(top_out, outputs[0]):
        return None
    used = [0]

    def const_ok(t, w):
        return (not t.requires_grad and t.device == Pp.device and t.dtype == Pp.dtype and t.shape[-1] == w and t.numel() in (w, n * w)
                and t.device.type != "meta")

    def factors(t, inverted):
        ev = by_out.get(_key(t)) if t.device.type == "meta" else None
        if ev is not None:
            used[0] += 1
            name, ins = ev
            if name == g + "_mul_fwd":
                a, b = factors(ins[0], inverted), factors(ins[1], inverted)
                if a is None or b is None:
                    return None
                return b + a if inverted else a + b
            if name == g + "_inv_fwd":
                return factors(ins[0], not inverted)
            return None
        if t.device.type == "meta":
            return None
        if _same(t, Pp):
            return [("P", inverted)]
        return [(t, inverted)] if const_ok(t, dg) else None
    name, ins = top
    a = None
    if name == g + "_log_fwd":
        kind = 0
    elif name == g + "_act_fwd":
        kind, a = 1, ins[1]
        if not const_ok(a, 3):
            return None
    else:
        return None
    chain = factors(ins[0], False)
    if chain is None or used[0] + 1 != len(trace.events):
        return None
    where = [k for k, f in enumerate(chain) if f[0] == "P" if isinstance(f[0], str)]
    if len(where) != 1:
        return None
    k = where[0]
    b = None
    if target is not None:
        b = target
        dr = gkey[2] if kind == 0 else 3
        if not isinstance(b, torch.Tensor) or not const_ok(torch.Tensor.as_subclass(b, torch.Tensor), dr):
            return None
        b = torch.Tensor.as_subclass(b, torch.Tensor)
    return LprProgram(P, gkey, kind, -1 if chain[k][1] else 1, chain[:k], chain[k + 1:], a, b)


class DeviceLMLpr(DeviceLM):
    """:class:`DeviceLM` for an :class:`LprProgram` (``pplie_lm_lpr_step``): same loop state, decisions and lazy mirrors."""

    def __init__(self, opt, prog, input):
        self.prog = prog
        self.opt, self.P, self.X_src = opt, prog.P, prog.P
        self.kind = _strategy_kind(opt.strategy)
        self.strategy = opt.strategy
        pt = prog.P.detach()
        self.n, dg = prog.n, prog.dg
        self.dtype, self.device = pt.dtype, pt.device
        self.p_ptr = pt.data_ptr()
        self.x_ptr, self.x_version = 0, 0
        z = dict(dtype=self.dtype, device=self.device)
        self.save = torch.empty((self.n, dg), **z)
        self.partials = torch.empty((_PARTIALS, 4), **z)
        self.sums = torch.zeros(4, **z)
        self.state = torch.zeros((2, _LM_STATE), dtype=torch.float64, device=self.device)
        self.cur = 0
        self.sync = torch.zeros(8, dtype=torch.int32, device=self.device)
        self.cfg = _LmCfg()
        self.fn = _C.library().symbol("pplie_lm_lpr_step" + _blocks._suffix(pt), _LPR_SIG)
        self.kind_name = LprLinearization.kind
        sp = self.state.data_ptr()
        self.ptrs = (self.save.data_ptr(), self.partials.data_ptr(), (sp, sp + 8 * _LM_STATE), self.sync.data_ptr())
        self.item = pt.element_size()
        self.k = _LOSS_BLOCK
        self.pending = False
        self.host_dirty = True
        self.last_loss = None
        self.rearm(input)
        self._wrap_group()

    def _launch(self, save, partials, st_in, st_out, sync, loss_ptr, last_ptr, stream):
        pr = self.prog
        L, R, a, b = self._ops = pr.operands()                       # (held until the next step: the launch reads them)
        p = lambda t: None if t is None else t.data_ptr()
        return self.fn(pr.gid, pr.kind, pr.sign, self.p_ptr, p(L), p(R), p(a), p(b), save, partials, st_in, st_out, sync, self.cfg,
                       self.n, loss_ptr, last_ptr, stream)

    def matches(self, m):
        return m[0] == "lpr" and m[1].key() == self.prog.key()

    def try_step(self, input, target, weight):
        """LM(static=True): the program is taken on trust while the same input object(s) are passed"""
        if not self._step_conditions(None if (self.prog.b is not None and target is not None and _same(self.prog.b, target)) else target,
                                     weight) or not _same_input(self.input, input):
            return None
        self.opt.linearization = self.kind_name
        return self.step()

    def _sharded(self, *a):
        raise RuntimeError("LM(group=...) is not available for this residual program; use the block path (opt.fused = False)")


class LprLinearization:
    """LM trial steps of an :class:`LprProgram` on the device (csrc/lm_generic.hip)."""

    kind = "fused:lpr"
    reference_kind = "block"

    def __init__(self, opt, prog, input):
        self.opt, self.prog, self.input, self.P = opt, prog, input, prog.P

    def build_normal_equations(self, dmin, dmax):
        self.dmin, self.dmax = float(dmin), float(dmax)

    def verify(self, ref, dmin, dmax, rtol=1e-3):
        """One device step (damping 1e-2: fp32 normal equations of far-from-converged problems can be singular to rounding
        without it) on a COPY of the parameter against the generic block linearisation ``ref`` of the same model: the
        candidate poses Exp(d) P and the residual norm."""
        from .solver import Cholesky
        pr = self.prog
        pt = pr.P.detach().reshape(pr.n, pr.dg)
        scratch = pt.clone()
        z = dict(dtype=pt.dtype, device=pt.device)
        save, partials = torch.empty_like(scratch), torch.empty((_PARTIALS, 4), **z)
        state = torch.zeros((2, _LM_STATE), dtype=torch.float64, device=pt.device)
        sync = torch.zeros(8, dtype=torch.int32, device=pt.device)
        out = torch.zeros(2, **z)
        cfg = _LmCfg()
        cfg.high, cfg.low, cfg.up, cfg.factor, cfg.smin, cfg.smax, cfg.sdown = 0.5, 1e-3, 2.0, 0.5, 1e-6, 1e16, 0.5
        cfg.dmin, cfg.dmax, cfg.host_damping, cfg.host_down = float(dmin), float(dmax), 1e-2, 0.5
        cfg.strategy, cfg.reject, cfg.flags, cfg.grid_cap = 0, 0, 3, 0
        L, R, a, b = pr.operands()
        p = lambda t: None if t is None else t.data_ptr()
        fn = _C.library().symbol("pplie_lm_lpr_step" + _blocks._suffix(pt), _LPR_SIG)
        with _C._on_device(pt.device):
            code = fn(pr.gid, pr.kind, pr.sign, scratch.data_ptr(), p(L), p(R), p(a), p(b), save.data_ptr(), partials.data_ptr(),
                      state[0].data_ptr(), state[1].data_ptr(), sync.data_ptr(), cfg, pr.n, out.data_ptr(),
                      out.data_ptr() + pt.element_size(), _C.stream_ptr(pt.device))
        _C.check(code, "pplie_lm_lpr_step")
        ref.build_normal_equations(dmin, dmax)
        ref.damp(1e-2)
        Dr = ref.solve(Cholesky()).view(pr.n, pr.dg).contiguous()
        want = _C.row_op(pr.g + "_retract", [Dr, pt.contiguous()], (pr.dg,))[0]
        moved = (want - pt).abs().max().clamp_min(1e-30)
        ok = bool((scratch - want).abs().max() <= rtol * moved + 1e-6 * want.abs().max())
        old = ref.Rb.square().sum()
        return ok and bool((state[1, _ST_LAST] - old.double()).abs() <= rtol * old.double().clamp_min(1e-30))

    def run_trials(self, opt, pg):
        dev = opt.__dict__.get('_device_lm')
        if not isinstance(dev, DeviceLMLpr) or dev.prog.key() != self.prog.key():
            dev = opt._device_lm = DeviceLMLpr(opt, self.prog, self.input)
        else:
            dev.prog = self.prog if dev.prog._folded is None else dev.prog
            dev.rearm(self.input)
        return dev.step()


# the recognised pose-graph program's linearisation also assembles the edges' shares of the normal equations (pplie_pgo_linearize_lap);
# PPLIE_FUSE_PGO_ASSEMBLY=0: the linearisation and pplie_graph_assemble_lap's two launches, as before round 6
FUSE_PGO_ASSEMBLY = os.environ.get("PPLIE_FUSE_PGO_ASSEMBLY", "1") != "0"
_PGO_SIG = [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_void_p]
_PGO_LAP_SIG = [ctypes.c_void_p] * 8 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_int64,
                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
_PGO_LOSS_SIG = [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
_PGO_ROBUST_SIG = [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
_PGO_PARTIALS = 1024      # PPLIE_PGO_PARTIALS


def _unchanged(old, new, old_version):
    return old.data_ptr() == new.data_ptr() and old.shape == new.shape and old.stride() == new.stride() \
        and old.dtype == new.dtype and new._version == old_version


def match_pgo(trace, gathers, R, params):
    """(param, idx0, idx1, Z) if the traced forward is exactly  Log(Inv(Z) * Inv(param[idx0]) * param[idx1])."""
    if len(trace.events) != 5 or len(gathers) != 2 or len(R) != 1 or len(params) != 1:
        return None
    P = params[0]
    if not _is_se3_group(P) or P.dim() != 2 or not P.is_contiguous() \
            or any(src is not P for src, _, _ in gathers):
        return None
    by_out = {_key(o[0]): (n, i) for n, i, o in trace.events}
    gat = {_key(out): ix for _, ix, out in gathers}
    if len(by_out) != 5 or len(gat) != 2:
        return None
    log = [(n, i, o) for n, i, o in trace.events if n == "se3_log_fwd"]
    if len(log) != 1 or not _same(log[0][2][0], R[0]):
        return None
    top = by_out.get(_key(log[0][1][0]))
    if top is None or top[0] != "se3_mul_fwd":
        return None
    ev = lambda t: by_out.get(_key(t))
    is_inv_of = lambda e, pred: e is not None and e[0] == "se3_inv_fwd" and pred(e[1][0])
    gathered = lambda t: _key(t) in gat
    const = lambda t: not t.requires_grad and not gathered(t) and _key(t) not in by_out and t.device == P.device
    a, b = top[1]
    if gathered(b) and ev(a) is not None and ev(a)[0] == "se3_mul_fwd":
        # (Inv(Z) * Inv(n_i)) * n_j   -- the association of the reference example, pgo.py:24
        n_j, (l, r) = b, ev(a)[1]
        zi, ni = ev(l), ev(r)
    elif is_inv_of(ev(a), const) and ev(b) is not None and ev(b)[0] == "se3_mul_fwd":
        # Inv(Z) * (Inv(n_i) * n_j)
        zi, (l, n_j) = ev(a), ev(b)[1]
        ni = ev(l)
        if not gathered(n_j):
            return None
    else:
        return None
    if not is_inv_of(zi, const) or not is_inv_of(ni, gathered) or _key(ni[1][0]) == _key(n_j):
        return None
    idx0, idx1, Z = gat[_key(ni[1][0])], gat[_key(n_j)], zi[1][0]
    E = R[0].numel() // 6
    if idx0.numel() != E or idx1.numel() != E or Z.numel() != E * 7 or Z.dtype != P.dtype or idx0.device != P.device \
            or idx1.device != P.device:
        return None
    return P, idx0.reshape(-1), idx1.reshape(-1), Z


class PgoProgram:
    """Fused evaluation of the recognised pose-graph residual program (csrc/pgo_fused.hip)."""

    def __init__(self, P, idx0, idx1, Z, cache=None):
        self.P = P
        # the stacked edge list is rebuilt only when the user's index tensors change (the gathered views are held, so
        # an equal data_ptr is the same storage, and an equal version counter means it was not written to): the same
        # tensor object then also hits the incidence-list cache without a compare kernel
        hit = cache.get("pgo_idx") if cache is not None else None
        if hit is not None and all(_unchanged(a, b, v) for a, b, v in zip(hit[0], (idx0, idx1), hit[1])):
            self.idx = hit[2]
        else:
            self.idx = torch.stack([idx0, idx1], dim=-1).contiguous()
            # gather indices may be negative (nodes[-1] is the last node, as in torch indexing and the reference's dense
            # path): the kernels address rows directly, so they are normalised once per edge list
            self.idx = torch.where(self.idx < 0, self.idx + P.shape[0], self.idx)
            if cache is not None:
                cache["pgo_idx"] = ((idx0, idx1), (idx0._version, idx1._version), self.idx)
        self.Z = Z.detach().reshape(-1, 7).contiguous()
        self.E = self.idx.shape[0]
        self.sources = _sources(idx0, idx1, Z)

    def matches(self, P, idx0, idx1, Z):
        """the operands of a freshly matched program are the tensors this one was built from, unmodified"""
        return P is self.P and all(_unchanged(old, new, ver) and old.data_ptr() == ptr
                                   for (old, ptr, ver), new in zip(self.sources, (idx0, idx1, Z)))

    def outputs(self):
        """uninitialised residuals [E, 6] and blocks [E, 2, 6, 6] for :meth:`linearize` / :meth:`linearize_lap` to fill"""
        nodes = torch.Tensor.as_subclass(self.P, torch.Tensor).detach()
        return (torch.empty((self.E, 6), dtype=nodes.dtype, device=nodes.device),
                torch.empty((self.E, 2, 6, 6), dtype=nodes.dtype, device=nodes.device))

    def linearize_lap(self, R, J, inc, HB, gg, pack, robust=None, begin=None):
        """:meth:`linearize` into ``R`` / ``J`` that ALSO leaves the edges' shares of the normal equations (``pplie_pgo_linearize_lap``):
        -J_1^T J_1 at both incidence slots ``inc[e]`` of ``HB`` (full or packed blocks) and -+J_1^T r in ``gg`` -- what the first
        launch of ``pplie_graph_assemble_lap`` computed from the J blocks it read back"""
        nodes = torch.Tensor.as_subclass(self.P, torch.Tensor).detach()
        assert nodes.is_contiguous() and inc.shape == (self.E, 2) and inc.dtype == torch.int32 and inc.is_contiguous()
        kind, p0, p1 = (0, 0.0, 0.0) if robust is None else robust
        # begin = (control block, host-pinned damping factor or None, device scalar): pplie_pcg_begin's work for the solve that follows
        ctl, s_src, s_dst = (None, None, None) if begin is None else begin
        with _C._on_device(nodes.device):
            fn = _C.library().symbol("pplie_pgo_linearize_lap" + _blocks._suffix(nodes), _PGO_LAP_SIG)
            code = fn(nodes.data_ptr(), self.idx.data_ptr(), self.Z.data_ptr(), R.data_ptr(), J.data_ptr(), inc.data_ptr(), HB.data_ptr(),
                      gg.data_ptr(), self.E, 1 if pack else 0, kind, p0, p1, None if ctl is None else ctl.data_ptr(),
                      0 if ctl is None else ctl.numel() * ctl.element_size(), None if s_src is None else s_src.data_ptr(),
                      None if s_dst is None else s_dst.data_ptr(), _C.stream_ptr(nodes.device))
        _C.check(code, "pplie_pgo_linearize_lap")

    def linearize(self, robust=None, out=None):
        """residuals [E, 6] and blocks [E, 2, 6, 6]; with ``robust`` = (kind, p0, p1) of a built-in kernel (optim/kernel.py
        robust_code) they come out already corrected -- sqrt(rho'(|r_e|^2)) applied to r_e and J_e in the kernel's registers"""
        nodes = torch.Tensor.as_subclass(self.P, torch.Tensor).detach()     # (plain: no __torch_function__ round trips below)
        assert nodes.is_contiguous()
        R, J = self.outputs() if out is None else out
        with _C._on_device(nodes.device):
            if robust is None:
                fn = _C.library().symbol("pplie_pgo_linearize" + _blocks._suffix(nodes), _PGO_SIG)
                code = fn(nodes.data_ptr(), self.idx.data_ptr(), self.Z.data_ptr(), R.data_ptr(), J.data_ptr(), self.E,
                          _C.stream_ptr(nodes.device))
            else:
                fn = _C.library().symbol("pplie_pgo_linearize_robust" + _blocks._suffix(nodes), _PGO_ROBUST_SIG)
                code = fn(nodes.data_ptr(), self.idx.data_ptr(), self.Z.data_ptr(), R.data_ptr(), J.data_ptr(), self.E,
                          robust[0], robust[1], robust[2], _C.stream_ptr(nodes.device))
        _C.check(code, "pplie_pgo_linearize")
        return R, J

    def loss(self, group=None, robust=None):
        """sum_e rho(|r_e|^2) at the current parameter values (optimizer.py:118-125; rho = identity without ``robust``)."""
        nodes = torch.Tensor.as_subclass(self.P, torch.Tensor).detach()     # (plain: no __torch_function__ round trips below)
        # one launch (round 6; was a zero fill, the residual kernel and a torch reduction): the last workgroup adds the partial sums.
        # The workspace (partials + arrival ticket) belongs to the stream the calls are ordered on
        stream = _C.stream_ptr(nodes.device)
        key = (nodes.dtype, nodes.device, stream.value)
        ws = self.__dict__.setdefault('_loss_ws', {}).get(key)
        if ws is None:
            nbytes = _C.library().symbol("pplie_pgo_loss_ws_bytes", [])()
            ws = self._loss_ws[key] = torch.zeros(nbytes // 8, dtype=torch.int64, device=nodes.device)
        loss = torch.empty((), dtype=nodes.dtype, device=nodes.device)
        kind, p0, p1 = (0, 0.0, 0.0) if robust is None else robust
        with _C._on_device(nodes.device):
            code = _C.library().symbol("pplie_pgo_loss" + _blocks._suffix(nodes), _PGO_LOSS_SIG)(
                nodes.data_ptr(), self.idx.data_ptr(), self.Z.data_ptr(), ws.data_ptr(), loss.data_ptr(), self.E, kind, p0, p1, stream)
        _C.check(code, "pplie_pgo_loss")
        if group is not None:
            import torch.distributed as dist
            dist.all_reduce(loss, group=group)
        return loss


def try_fused(opt, pg, input, target, weight, cache):
    """A fused linearisation if the model's forward is a recognised program, else None."""
    from .optimizer import Trivial
    from .solver import Cholesky
    from . import posegraph as _pg
    params = [p for p in pg['params'] if p.requires_grad]
    if cache.get("fused") is False or len(params) != 1 or _C._test_backend is not None:
        return None
    P = params[0]
    if not (P.is_cuda and _blocks._suffix(P) and len(opt.param_groups) == 1):
        return None
    trivial = all(isinstance(c, Trivial) for c in opt.corrector) and all(isinstance(k, Trivial) for k in opt.model.kernel)
    solver_ok = (isinstance(opt.solver, Cholesky) and not opt.solver.upper) or isinstance(opt.solver, _pg.PCG)
    # the device-side decision implements exactly the three built-in damping policies; a user-defined or subclassed
    # strategy sees the real J, D, R of the block linearisation instead
    builtin = _strategy_kind(opt.strategy) is not None
    static = getattr(opt, 'static', False)
    hit = cache.get("program")
    if static and cache.get("fused") is True and hit is not None and _same_input(hit[0], input) and hit[1] is P \
            and all(t.data_ptr() == ptr and t._version == ver for t, ptr, ver in hit[4]):
        # LM(static=True): the caller's promise that the residual program does not change between steps.  The verified
        # program of the previous step is evaluated directly, without running the model, while the same `input` object(s)
        # are passed and every operand tensor sits at the same address with the same version counter.
        kind, operands = hit[2], hit[3]
        if kind == "se3inv" and weight is None and trivial and solver_ok and builtin and target is None:
            return Se3InvLinearization(opt, P, operands, None, input)
        if kind == "pgo" and len(opt.corrector) == 1 and target is None:
            return _pgo_linearization(opt, operands, weight, P, trivial)
        if kind == "lpr" and weight is None and trivial and solver_ok and builtin \
                and ((operands.b is None) if target is None else (operands.b is not None and _same(operands.b, target))):
            return LprLinearization(opt, operands, input)
    # Default: the model runs EVERY step, as in the reference (optimizer.py:631, 646) -- as a dry trace (DryTracer): its
    # Python executes, the Lie kernels it would launch are recorded instead of launched, and the program is re-matched
    # from that trace; the fused kernels below compute the residual themselves.  Only a model a dry run cannot follow
    # (it looks at values of intermediates) gets a real traced forward.
    m = opt.__dict__.pop('_dry_hint', None)
    m = m[1] if m is not None and m[0] is input else None
    if m is None:
        m = dry_program(opt, params, input, target) if cache.get("dry") is not False else False
    if m is False:
        cache["dry"] = False
        with torch.no_grad(), OpTracer() as tr, _pg.GatherRecorder(params) as rec:
            R = _outputs(opt, input)
        m = _match(tr, rec, R, params, target)
    if m is not None and m[0] == "lpr" and weight is None and trivial and solver_ok and builtin:
        cache["program"] = (input, P, "lpr", m[1], [], None)
        return LprLinearization(opt, m[1], input)
    if m is not None and m[0] == "se3inv" and weight is None and trivial and solver_ok and builtin:
        cache["program"] = (input, P, "se3inv", m[2], _sources(m[2]), None)     # (DeviceLM is this program's shortcut)
        return Se3InvLinearization(opt, P, m[2], None, input)
    if m is not None and m[0] == "pgo" and len(opt.corrector) == 1:
        prog = hit[3] if hit is not None and hit[2] == "pgo" and hit[3].matches(*m[1:]) else PgoProgram(*m[1:], cache=cache)
        cache["program"] = (input, P, "pgo", prog, _sources(*m[2:]), None)
        return _pgo_linearization(opt, prog, weight, P, trivial)
    cache["fused"] = False
    return None


def _outputs(opt, input):
    """the model's raw outputs (RobustModel subtracts the target afterwards, optimizer.py:94-101), as a list"""
    out = opt.model.model_forward(input)
    return list(out) if isinstance(out, (tuple, list)) else [out]


def _match(tr, rec, R, params, target=None):
    """``R``: the model's raw outputs (before ``- target``)"""
    if target is None:
        m = match_se3inv(tr, R, params) if not rec.events else None
        if m is not None:
            return ("se3inv", m[0], m[1])
        m = match_pgo(tr, rec.events, R, params)
        if m is not None:
            return ("pgo",) + m
    m = match_lpr(tr, R, params, target, rec.events)
    return ("lpr", m) if m is not None else None


def dry_program(opt, params, input, target):
    """Run the model's Python for this step without launching its kernels and match the recorded chain:
    ("se3inv", P, X) / ("pgo", P, idx0, idx1, Z), None if it is no recognised program, False if the model cannot be
    followed dry (it raised: a real forward will say whether that was the dry run's fault).  A trace whose signature
    (kernel names, operand memory / layout, data flow) equals the previous step's is the previous step's program."""
    from . import posegraph as _pg
    if len(params) != 1 or type(getattr(params[0], "ltype", None)).__name__ not in _GROUP_OF:
        return None
    st = opt.__dict__.get('_dry_state')
    if st is None:
        st = opt.__dict__['_dry_state'] = _DryState()
    try:
        st.touched = True              # (until this run of the model completes without touching a value)
        with torch.no_grad(), DryTracer(st) as tr, _pg.GatherRecorder(params) as rec:
            R = _outputs(opt, input)
        st.touched = tr.touched
        # (a wrapped dry result carries its pooled plain alias: no re-wrapping, and its signature token is cached)
        plain = [r if type(r) is torch.Tensor else (r.__dict__.get("_pl") if r.__dict__.get("_pl") is not None
                                                    else torch.Tensor.as_subclass(r, torch.Tensor)) for r in R]
        if any(r.device.type != "meta" for r in plain):
            return None
        tt = None if target is None else (_tok(torch.Tensor.as_subclass(target, torch.Tensor)) if isinstance(target, torch.Tensor) else id(target))
        sig = (tuple(tr.sig), tuple(_tok(r, st.toks) for r in plain), id(params[0]), tt)
        if st.sig == sig:
            return st.match
        m = _match(tr, rec, plain, params, target)
        st.sig, st.match = sig, m
        return m
    except Exception:
        return False


def checked_shortcut(opt, dev, gs, input, target, weight):
    """``LM.step`` of the default (non-static) optimizer when a device-resident step exists: the model runs dry, and the
    shortcut is taken only if THIS step's program is the one it was built on."""
    pg = opt.param_groups[0]
    cache = opt.__dict__.get('_structure_cache') or {}
    if cache.get("fused") is not True or cache.get("dry") is False or torch.is_inference_mode_enabled():
        return None
    if dev is None and gs is not None and getattr(opt, 'speculate', True):
        # (round 6) the replay goes out before ANY of the step's checks but the two `quick` makes: what stands between the previous
        # trial's verdict and this trial's first kernel is host time the GPU idles through (profiles/r06/EXPERIMENTS.md "host segments")
        with _no_tf():
            early = gs.quick(target)
        if early:
            gs.launch(pg)
            w = opt.weight if weight is None else weight
            m = None
            if gs.usable(pg, input, target, w, checked=True):
                with _no_tf():
                    params = [p for p in dict.__getitem__(pg, 'params') if p.requires_grad]
                m = dry_program(opt, params, input, target)
                st = opt.__dict__.get('_dry_state')
                if m and m[0] == "pgo" and gs.prog.matches(*m[1:]) and not (st is not None and st.touched):
                    return gs.finish(pg)
                if st is not None and st.touched:
                    opt.speculate = False
            gs.cancel()
            if m:
                opt._dry_hint = (input, m)
            return None
    with _no_tf():              # (attribute reads on a LieTensor parameter are __torch_function__ round trips: ~3 us each)
        params = [p for p in dict.__getitem__(pg, 'params') if p.requires_grad]
    if dev is None and gs is not None and getattr(opt, 'speculate', True):
        # Pose graphs: a step is one hipGraph replay followed by a read-back the host has to wait for anyway (~0.5 ms).  The
        # replay is enqueued FIRST and the model's (dry) run -- this step's check that the program is still the one captured --
        # happens while the GPU works; the captured trial starts by saving the parameters, so a mismatch (rare: the model
        # changed) is undone exactly: wait, copy back, take the ordinary path.  A model that reads parameter VALUES during
        # its forward would see them mid-update: the dry tracer notices any such access (DryTracer.touched) and the
        # speculation is then cancelled and never tried again for this optimizer.
        w = opt.weight if weight is None else weight
        if gs.usable(pg, input, target, w, checked=True):
            with torch.no_grad():
                gs.launch(pg)
            m = dry_program(opt, params, input, target)
            st = opt.__dict__.get('_dry_state')
            if m and m[0] == "pgo" and gs.prog.matches(*m[1:]) and not (st is not None and st.touched):
                with torch.no_grad():
                    return gs.finish(pg)
            gs.cancel()
            if st is not None and st.touched:
                opt.speculate = False
            if m:
                opt._dry_hint = (input, m)
            return None
    m = dry_program(opt, params, input, target)
    if not m:
        return None
    out = None
    if dev is not None:
        out = dev.checked_step(m, target, weight)
    if out is None and gs is not None and m[0] == "pgo" and gs.prog.matches(*m[1:]):
        w = opt.weight if weight is None else weight
        if gs.usable(pg, input, target, w, checked=True):
            with torch.no_grad():
                return gs.step(pg)
    if out is None:
        opt._dry_hint = (input, m)          # the general path re-uses this step's trace
    return out


def _sources(*tensors):
    """(tensor, address, version) of the operands a program was matched on"""
    return [(t, t.data_ptr(), t._version) for t in tensors]


def _same_input(a, b):
    if a is b:
        return True
    if isinstance(a, (tuple, list)) and isinstance(b, (tuple, list)) and len(a) == len(b):
        return all(x is y for x, y in zip(a, b))
    if isinstance(a, dict) and isinstance(b, dict) and a.keys() == b.keys():
        return all(a[k] is b[k] for k in a)
    return False


def _pgo_linearization(opt, prog, weight, P, trivial, s_dev=None):
    from . import posegraph as _pg
    from .corrector import fused_code
    from .kernel import robust_code
    from .optimizer import Trivial
    # a built-in robust kernel rides inside the linearisation kernel (csrc/robust.h): no corrector pass, no autograd graph
    c = opt.corrector[0]
    robust = fused_code(c) if (not trivial and opt.group is None) else None
    # One launch for the linearisation AND the edges' shares of the normal equations where nothing stands between the two: no weight,
    # no corrector pass left to run (Trivial, or a built-in kernel that rides in the linearisation), one GPU.  The outputs are then
    # allocated first, the linearisation object built around them (it only holds them) and asked for its block plan.
    fuse = (FUSE_PGO_ASSEMBLY and weight is None and opt.group is None and (robust is not None or isinstance(c, Trivial))
            and _C._test_backend is None and P.is_cuda and prog.E * 2 < (1 << 31))
    if fuse:
        r, J = prog.outputs()
    else:
        r, J = prog.linearize(robust)
    lin = _pg.build_graph_linearization(opt, weight, r, J, prog.idx, P, 7, 6, corrector=Trivial() if robust is not None else None)
    lin.kind = "fused:pgo"
    lin.robust = robust
    # r = Log(Z^-1 n_i^-1 n_j): d r / d n_i = -d r / d n_j (csrc/pgo_fused.hip), and a corrector scales both blocks of an edge alike
    lin.antisym = True
    if fuse:
        plan = lin.plan_blocks() if (lin.R is r and lin.J is J and lin.W is None and lin.group is None and lin._hip()) else None
        if plan is not None and lin.HB is not None:
            # the solve's workspace, if this optimizer has solved this system before: the launch then also clears its control block
            # (and, in a captured trial, fetches the damping factor from `s_dev`) -- what pplie_pcg_begin / a fill did in front of
            # every solve; the workspace is told through lin._begun
            wsp, begin = None, None
            solver = opt.solver
            if isinstance(solver, _pg.PCG) and getattr(solver, 'fused', True) and _pg.FusedPCG.fuse_prepare:
                wsp = (opt.__dict__.get('_pcg_workspaces') or {}).get(
                    (lin.E, lin.K, lin.dr, lin.m, lin.N, lin.J.dtype, lin.J.device, False, solver.check_every))
            if wsp is not None:
                begin = (wsp._ctl, s_dev, wsp.s_device)
            prog.linearize_lap(r, J, lin.incidence_slots(), lin.HB, plan['gg'], lin.HB_pack, robust, begin)
            plan['blocks_done'] = True
            if wsp is not None:
                # (valid only while nothing else has used the workspace in between: the workspace counts its solves)
                lin._begun = (wsp, s_dev, wsp.__dict__.get('_solves', 0))
        else:
            prog.linearize(robust, out=(r, J))

    def verify(ref, dmin, dmax, rtol=1e-3):
        """residuals and blocks against the autograd-derived pose-graph linearisation (whose edge ends are in
        gather order, ours in (inverted node, other node) order)"""
        if ref.J.shape != lin.J.shape:
            return False
        if torch.equal(ref.idx, lin.idx):
            Jr = ref.J
        elif torch.equal(ref.idx, lin.idx.flip(-1)):
            Jr = ref.J.flip(1)
        else:
            return False
        return bool((ref.R - lin.R).abs().max() <= rtol * ref.R.abs().max().clamp_min(1e-30)) \
            and bool((Jr - lin.J).abs().max() <= rtol * Jr.abs().max().clamp_min(1e-30))
    lin.verify = verify
    lin.reference_kind = "graph"
    loss_code = None if trivial else robust_code(opt.model.kernel[0]) if len(opt.model.kernel) == 1 else None
    if not trivial and loss_code is not None:
        lin.fast_loss = lambda: prog.loss(opt.group, loss_code)          # sum_e rho(|r_e|^2) in the residual kernel
    if trivial:
        lin.fast_loss = lambda: prog.loss(opt.group)
        if opt.group is None and lin._hip():
            def tail_buffers():
                from .pgograph import TrialTail
                pt = torch.Tensor.as_subclass(P, torch.Tensor).detach()
                tt = opt.__dict__.get('_trial_tail')
                if tt is None or tt.dtype != pt.dtype or tt.dev != pt.device:
                    tt = opt._trial_tail = TrialTail(pt.dtype, pt.device)
                bk = tt.__dict__.get('_backup')
                if bk is None or bk.shape != pt.shape or bk.dtype != pt.dtype or bk.device != pt.device:
                    bk = tt._backup = torch.empty_like(pt)
                return pt, tt, bk

            def tail_setup():
                """before the solve: a persistent solve may take the tail's first half (gain terms + retraction) into its epilogue
                (optim/posegraph.py FusedPCG.solve, pplie_pcg_ghost_tail) -- the same arrangement as the captured trial's"""
                pt, tt, bk = tail_buffers()
                if pt.is_contiguous() and lin.idx.data_ptr() == prog.idx.data_ptr() and lin.J.shape == (prog.E, 2, 6, 6):
                    lin._tail_in_solve = (pt, bk, tt.partial, tt.state)
            lin.tail_setup = tail_setup

            def trial_tail():
                """What the trial loop does between the solve and its decision (optimizer.py:669-673) as one C call and one wait on
                pinned memory (optim/pgograph.py TrialTail): the parameters retracted by the step the solve just returned, then
                (a, b, loss) as host floats and the loss as a device scalar; None if this trial cannot take the route."""
                Dn = lin.__dict__.pop('_last_Dn', None)
                pt, tt, bk = tail_buffers()
                if Dn is None or Dn.shape != (lin.N, 6) or not pt.is_contiguous() or lin.idx.data_ptr() != prog.idx.data_ptr() \
                        or lin.J.shape != (prog.E, 2, 6, 6) or Dn.dtype != pt.dtype:
                    if lin.__dict__.pop('_tail_done', None) is not None:
                        raise RuntimeError("pypose_amd: the solve's epilogue moved the parameters of a trial whose tail cannot finish")
                    lin.__dict__.pop('_tail_in_solve', None)
                    return None
                pend, lin.pending_info = lin.pending_info, None
                before = tt.seq
                slot = tt.advance()
                try:
                    # (the retraction keeps the old rows in `bk`: 28 B per node, against a trial that overwrites the parameters
                    #  with nothing to go back to if one of the later launches of the same C call fails)
                    tt.enqueue(pt, bk, prog, lin, Dn.contiguous(), None if pend is None else pend.info)
                except BaseException:
                    # some of the call's launches may have run (the retraction comes first): the device's own execution count says
                    # whether the result block was written; the parameters go back to what the retraction saved
                    torch.cuda.synchronize(pt.device)
                    tt.seq = before
                    tt.resync()
                    if tt.unfinished > 0:        # the retraction ran, the report did not: bk holds this trial's starting point
                        pt.copy_(bk)
                        tt.state[3] = tt.state[0]
                    raise
                _C.mark_written(P)
                a, b, loss_h, its, rr, bn2, flag = tt.wait()
                if pend is not None:             # the persistent solve's (iterations, flag) rode along: raises like an eager solve
                    lin._pending_solver.iterations = pend.resolve([its, rr, bn2, flag])
                return a, b, loss_h, tt.loss_views[slot]
            lin.trial_tail = trial_tail
    return lin
