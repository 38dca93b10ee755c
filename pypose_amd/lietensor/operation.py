"""The 32 Lie-group ``torch.autograd.Function`` classes, backed by the HIP library.

Host-side mirror of ``pypose/lietensor/operation.py:304-1113`` (same class names, same argument
meaning, same gradient convention: gradients w.r.t. a *group* element live in the left tangent
space, zero-padded to the embedding width).  Where the reference composes 20-150 eager aten
ops and materialises [B,3,3]..[B,7,7] temporaries per call, every forward and every backward
here is ONE kernel launch through the C ABI (``include/pplie.h``), reading/writing exactly the
algorithmic bytes.

Each class is generated from a small table (op kind x group); each has

* ``forward(*inputs)``      -> ``pplie_<g>_<op>_fwd``,
* ``setup_context``         saving what the backward kernel reads,
* ``backward``              -> a hidden, non-differentiable ``*_Bwd`` Function wrapping
                              ``pplie_<g>_<op>_bwd`` (so that the backward itself is vmappable),
* ``vmap``                  row-wise ops: the vmapped dim is folded into the row dimension,
                              which is what ``jacobian(vectorize=True)`` (optim/functional.py)
                              and ``torch.func.jacrev`` need (the reference relies on
                              ``generate_vmap_rule = True``).

There is no CPU implementation: tensors must live on a HIP device (see ``_C.row_op``).
"""
from __future__ import annotations

import torch

from .. import _C
from . import matrices as _mats
from .matrices import *            # noqa: F401,F403  so3_Jl ... Sim3_Act4_Jacobian (reference operation.py:7-301)


# ---- native autograd nodes (csrc_torch/pplie_autograd.cpp): the plain eager case of the 32 Functions below recorded as a C++
# node -- a Python Function costs ~35 us per backward node on the autograd engine's device thread (GIL, Python context), which is
# most of BASELINE configs[0]'s latency; everything else (transforms, tracers, broadcasting, host tensors) stays on the Python path
_native_state = {"mod": None, "tried": False, "ptr": {}, "rules": 0}
_ROW_OP = _C.row_op            # (a launcher somebody replaced -- tests that record launches, a stand-in backend -- is honoured)


def _native():
    st = _native_state
    if not st["tried"]:
        st["tried"] = True
        import importlib.util
        import os
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lib", "pplie_torch_ext.so")
        if os.environ.get("PPLIE_NATIVE_AUTOGRAD", "1") != "0" and os.path.exists(path) and _C._test_backend is None:
            try:
                spec = importlib.util.spec_from_file_location("pplie_torch_ext", path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                st["mod"] = mod
            except Exception as e:             # (built against another torch: the Python Functions carry the autograd)
                import warnings
                warnings.warn(f"pypose_amd: native autograd nodes unavailable ({e}); using the Python Functions")
    return st["mod"]


def _kernel_address(name, dtype):
    key = (name, dtype)
    hit = _native_state["ptr"].get(key)
    if hit is None:
        hit = _native_state["ptr"][key] = _C.library().address("pplie_" + name + ("_f32" if dtype == torch.float32 else "_f64"))
    return hit


def _native_ok(ins, widths):
    x0 = ins[0]
    if type(x0) is not torch.Tensor and type(x0) is not torch.nn.Parameter:
        return False
    if not x0.is_cuda or x0.dtype not in (torch.float32, torch.float64) or not x0.is_contiguous() or x0.dim() < 1 \
            or x0.shape[-1] != widths[0]:
        return False
    for t, w in zip(ins[1:], widths[1:]):
        if (type(t) is not torch.Tensor and type(t) is not torch.nn.Parameter) or t.dtype != x0.dtype or t.device != x0.device \
                or not t.is_contiguous() or t.shape[:-1] != x0.shape[:-1] or t.shape[-1] != w:
            return False
    return True

# (algebra width, group width)
_GROUPS = {"so3": (3, 4), "se3": (6, 7), "sim3": (7, 8), "rxso3": (4, 5)}
_CAP = {"so3": "SO3", "se3": "SE3", "sim3": "Sim3", "rxso3": "RxSO3"}


def _rows(t: torch.Tensor, width: int) -> torch.Tensor:
    if t.shape[-1] != width:
        raise ValueError(f"expected last dimension {width}, got shape {tuple(t.shape)}")
    return t.reshape(-1, width).contiguous()


_is_legacy_batched = torch._C._functorch.is_legacy_batchedtensor
_transforms_active = torch._C._are_functorch_transforms_active


def _legacy_level(t):
    """vmap level of a legacy-batched tensor.  The backward of a GPU graph runs on the autograd
    engine's device thread, where the (thread-local) vmap nesting counter is not visible, so the
    level is found by trial: removing the right level leaves a plain tensor."""
    for lvl in range(1, 9):
        if not _is_legacy_batched(torch._remove_batch_dim(t, lvl, 1, 0)):
            return lvl
    raise RuntimeError("pypose_amd: nested legacy vmap over a Lie op is not supported")


# optim/fused.py registers tracers here to recognise whole residual programs (kernel name, inputs, outputs)
_op_tracers = []


def _launch(name, ins, in_widths, out_widths, prm=None):
    """Broadcast leading dims, flatten to rows, launch, un-flatten (``prm``: launch-wide scalar of the
    conversion kernels, csrc/convert.hip)."""
    if any(_is_legacy_batched(t) for t in ins):
        # torch.autograd.grad(is_grads_batched=True) -- what jacobian(vectorize=True) uses -- runs
        # the backward under the *legacy* vmap, which does not consult Function.vmap: peel the
        # batch dim off by hand (row-wise op: it simply joins the leading dims) and put it back.
        lvl = _legacy_level(next(t for t in ins if _is_legacy_batched(t)))
        phys = [torch._remove_batch_dim(t, lvl, 1, 0) if _is_legacy_batched(t) else None for t in ins]
        bsz = next(p.shape[0] for p in phys if p is not None)
        outs = _launch_slices(name, ins, phys, bsz, in_widths, out_widths) if prm is None and not _op_tracers else None
        if outs is None:
            phys = [p if p is not None else t.unsqueeze(0).expand((bsz,) + tuple(t.shape)) for p, t in zip(phys, ins)]
            outs = _launch(name, phys, in_widths, out_widths, prm)
        return tuple(torch._add_batch_dim(o, 0, lvl) for o in outs)
    lead = ins[0].shape[:-1]
    if any(t.shape[:-1] != lead for t in ins[1:]):
        lead = torch.broadcast_shapes(*[t.shape[:-1] for t in ins])
    if _C.dry_tracing():
        # optim/fused.py DryTracer: the model's Python runs, nothing is launched; outputs are `meta` tensors -- shapes and
        # dtypes without storage, so anything that needs a VALUE downstream (a branch on a result, .item(), a copy) raises
        # instead of reading garbage, and the caller falls back to a real forward
        return _op_tracers[-1].dry_launch(name, ins, in_widths, out_widths, lead)
    flat = []
    for t, w in zip(ins, in_widths):
        if t.shape[:-1] != lead:
            t = t.expand(lead + (t.shape[-1],))
        flat.append(_rows(t, w))
    outs = _C.row_op(name, flat, out_widths) if prm is None else (_C.param_op(name, flat, out_widths[0], prm),)
    outs = tuple(o.view(lead + (w,)) for o, w in zip(outs, out_widths))
    for tr in _op_tracers:
        tr.note(name, ins, outs)
    return outs


_SLICE_ROWS = 1 << 16          # per-slice launches pay off once a slice is this many rows (a launch is ~10 us of host time)


def _launch_slices(name, ins, phys, bsz, in_widths, out_widths):
    """The batched backward of the block linearisations (d_res cotangents against the SAME saved operands): one launch per
    batch slice straight into the [bsz, rows, w] outputs, the un-batched operands read in place -- expanding them to the
    batch and flattening costs a bsz-fold copy per operand per kernel (3 x 170 MB per LM step at 10^6 problems).  None when
    it does not apply (everything batched, small slices, broadcasting between the operands)."""
    if bsz > 16 or all(p is not None for p in phys):
        return None
    lead = None
    for t, p in zip(ins, phys):
        shp = tuple((p if p is not None else t).shape[(1 if p is not None else 0):-1])
        if lead is None:
            lead = shp
        elif shp != lead:
            return None
    rows = 1
    for v in lead:
        rows *= v
    if rows < _SLICE_ROWS:
        return None
    flat = []
    for t, p, w in zip(ins, phys, in_widths):
        if p is None:
            flat.append(_rows(t, w))                                   # [rows, w], shared by every slice
        else:
            if p.shape[-1] != w:
                raise ValueError(f"expected last dimension {w}, got shape {tuple(p.shape)}")
            flat.append(p.reshape(bsz, rows, w).contiguous())          # [bsz, rows, w]
    x0 = flat[0]
    outs = tuple(torch.empty((bsz, rows, w), dtype=x0.dtype, device=x0.device) for w in out_widths)
    for b in range(bsz):
        _C.row_op(name, [f if f.dim() == 2 else f[b] for f in flat], out_widths, out=[o[b] for o in outs])
    return tuple(o.view((bsz,) + lead + (w,)) for o, w in zip(outs, out_widths))


def _fold_vmap(in_dims, args):
    """Move every vmapped dim to the front; expand un-batched args to the batch size."""
    bsz = next(a.shape[d] for a, d in zip(args, in_dims) if d is not None)
    out = []
    for a, d in zip(args, in_dims):
        if d is None:
            out.append(a.unsqueeze(0).expand((bsz,) + tuple(a.shape)))
        else:
            out.append(a.movedim(d, 0))
    return out


def _pad_group(t, dg):
    """a left-tangent gradient zero-padded to the group's embedding width (operation.py:336-337, 394-395, 849-851)"""
    return torch.cat([t, t.new_zeros(tuple(t.shape[:-1]) + (dg - t.shape[-1],))], dim=-1)


def _rowvec(g, Mx):
    return (g.unsqueeze(-2) @ Mx).squeeze(-2)


def _composed_rule(g, kind):
    """The backward pass of one Function written as ``cotangent @ matrix(saved)`` with the differentiable helpers of
    lietensor/matrices.py -- the reference's own formulation of its backward passes (operation.py:366-370, 389-395, 743-748,
    846-852, 945-949, 1039-1044, 535-543, 646-653).  Only evaluated when the backward ITSELF has to be differentiated
    (``_Bwd.backward``: create_graph=True, Hessians); first derivatives run the one-kernel backward."""
    da, dg = _GROUPS[g]
    G = _CAP[g]
    H = _mats.__dict__
    Jl, Jl_inv, little, Adj = H[g + "_Jl"], H[g + "_Jl_inv"], H[g + "_adj"], H[G + "_Adj"]
    Matrix, Matrix4, ActJ, Act4J = H[G + "_Matrix"], H[G + "_Matrix4x4"], H[G + "_Act_Jacobian"], H[G + "_Act4_Jacobian"]
    if kind == "exp":
        return lambda x, c: (_rowvec(c[..., :da], Jl(x)),)
    if kind == "log":
        return lambda y, c: (_pad_group(_rowvec(c, Jl_inv(y)), dg),)
    if kind == "inv":
        return lambda Y, c: (_pad_group(-_rowvec(c[..., :da], Adj(Y)), dg),)
    if kind == "mul":
        return lambda X, c: (_pad_group(c[..., :da], dg), _pad_group(_rowvec(c[..., :da], Adj(X)), dg))
    if kind == "act":
        return lambda X, out, c: (_pad_group(_rowvec(c, ActJ(out)), dg), _rowvec(c, Matrix(X)[..., :3, :3]))
    if kind == "act4":
        return lambda X, out, c: (_pad_group(_rowvec(c, Act4J(out)), dg), _rowvec(c, Matrix4(X)))
    if kind == "adj":
        return lambda X, out, c: (_pad_group(-_rowvec(c, little(out)), dg), _rowvec(c, Adj(X)))
    if kind == "adjt":
        def rule(X, a, c):
            ga = (Adj(X) @ c.unsqueeze(-1)).squeeze(-1)
            return _pad_group(-_rowvec(a, little(ga)), dg), ga
        return rule
    raise KeyError(kind)


def _grad_of(forward):
    """composed backward of an op given as a differentiable torch composition ``forward(*inputs)``: its vector-Jacobian
    product by autograd, itself differentiable"""
    def rule(*args):
        *ins, c = args
        with torch.enable_grad():
            ins = [t if t.requires_grad else t.detach().requires_grad_(True) for t in ins]
            out = forward(*ins)
            return torch.autograd.grad([out], ins, [c], create_graph=True, allow_unused=True)
    return rule


def _make_bwd(qualname, kernel, in_widths, out_widths, composed=None):
    """Hidden Function running one backward kernel (vmappable).  Differentiating it -- double backward through the Lie op
    it belongs to -- re-evaluates the same backward as a differentiable torch composition (``composed``) and takes autograd's
    vector-Jacobian product of that (the reference's backward passes ARE such compositions: operation.py:366-370 etc.)."""
    single = len(out_widths) == 1

    class _Bwd(torch.autograd.Function):
        @staticmethod
        def forward(*ins):
            outs = _launch(kernel, ins, in_widths, out_widths)
            return outs[0] if single else outs

        @staticmethod
        def setup_context(ctx, inputs, output):
            ctx.save_for_backward(*inputs)

        @staticmethod
        def backward(ctx, *grads):
            if composed is None:
                raise NotImplementedError(f"{qualname}: double backward is not implemented for this op")
            with torch.enable_grad():
                ins = [t.detach().requires_grad_(True) for t in ctx.saved_tensors]
                outs = composed(*ins)
                pairs = [(o, c) for o, c in zip(outs, grads) if c is not None and o is not None and o.requires_grad]
                if not pairs:
                    return tuple(None for _ in ins)
                got = torch.autograd.grad([o for o, _ in pairs], ins, [c for _, c in pairs], create_graph=torch.is_grad_enabled(),
                                          allow_unused=True)
            return tuple(got)

        @staticmethod
        def vmap(info, in_dims, *ins):
            res = _Bwd.apply(*_fold_vmap(in_dims, ins))
            return (res, 0) if single else (res, (0,) * len(out_widths))

    _Bwd.__name__ = _Bwd.__qualname__ = qualname
    return _Bwd


def _make_fwd(clsname, g, kind, doc):
    da, dg = _GROUPS[g]
    # kind -> (fwd input widths, fwd output width, bwd input widths, bwd output widths, what bwd reads)
    table = {
        "exp": ((da,), dg, (da, dg), (da,), "in0"),
        "log": ((dg,), da, (da, da), (dg,), "out"),
        "inv": ((dg,), dg, (dg, dg), (dg,), "out"),
        "mul": ((dg, dg), dg, (dg, dg), (dg, dg), "in0"),
        "act": ((dg, 3), 3, (dg, 3, 3), (dg, 3), "in0,out"),
        "act4": ((dg, 4), 4, (dg, 4, 4), (dg, 4), "in0,out"),
        "adj": ((dg, da), da, (dg, da, da), (dg, da), "in0,out"),
        "adjt": ((dg, da), da, (dg, da, da), (dg, da), "in0,in1"),
    }
    fin, fout, bin_, bout, reads = table[kind]
    fwd_kernel, bwd_kernel = f"{g}_{kind}_fwd", f"{g}_{kind}_bwd"
    saved_spec = [-1 if key == "out" else int(key[2:]) for key in reads.split(",")]     # (what the backward kernel reads, in its order)
    Bwd = _make_bwd(clsname + "_Bwd", bwd_kernel, bin_, bout, _composed_rule(g, kind))

    class _Fn(torch.autograd.Function):
        __doc__ = doc

        @classmethod
        def apply(cls, *ins):
            # nothing to record (no_grad, or no input requires grad) and no functorch transform:
            # launch directly -- Function.apply costs ~40 us of Python per call (signature binding,
            # context set-up), which is the whole budget of a small kernel inside LM.step
            if not _transforms_active():
                # the plain eager case through a PREPARED handle of the native extension (csrc_torch/pplie_autograd.cpp RowHandle):
                # operand checks, the no-gradient launch and the native autograd node behind ONE call; None -> the paths below
                # (the stand-in back end of the host tests is looked at on EVERY call, not when the handle is built, and tensor
                #  subclasses other than Parameter keep the Python route, as on the paths below: ADVICE r05)
                if not _op_tracers and _C.row_op is _ROW_OP and _C._test_backend is None and len(ins) <= 2 and _native() is not None \
                        and (type(ins[0]) is torch.Tensor or type(ins[0]) is torch.nn.Parameter) \
                        and (len(ins) == 1 or type(ins[1]) is torch.Tensor or type(ins[1]) is torch.nn.Parameter):
                    h = cls._handle(ins[0].dtype)
                    if h is not None and not _C.dry_tracing():
                        out = h(*ins)
                        if out is not None:
                            return out
                if not (torch.is_grad_enabled() and any(t.requires_grad for t in ins)):
                    return _launch(fwd_kernel, ins, fin, (fout,))[0]
                nat = _native()
                if nat is not None and len(ins) <= 2 and not _op_tracers and _C.row_op is _ROW_OP and not _C.dry_tracing() \
                        and _native_ok(ins, fin):
                    # the plain eager case: a native autograd node around the same two kernels (csrc_torch/pplie_autograd.cpp)
                    dt = ins[0].dtype
                    return nat.row_op(ins[0], ins[1] if len(ins) == 2 else None, _kernel_address(fwd_kernel, dt),
                                      _kernel_address(bwd_kernel, dt), fout, saved_spec, list(bout), cls._native_rule(nat),
                                      _kernel_address(bwd_kernel + "_gb", dt))       # (broadcast cotangent: rowmap.h GB)
                # recorded, but no functorch transform: the engine's own apply, past Function.apply's per-call
                # inspect.signature binding of forward()'s defaults (~15 us; there are none)
                return super(torch.autograd.Function, cls).apply(*ins)
            return super().apply(*ins)

        @classmethod
        def _handle(cls, dtype):
            """this operator's prepared handle for ``dtype`` (built on first use; None: no native extension / another dtype)"""
            hs = cls.__dict__.get('_handles')
            if hs is None:
                hs = cls._handles = {}
            if dtype in hs:
                return hs[dtype]
            h = None
            nat = _native()
            if nat is not None and hasattr(nat, "RowHandle") and dtype in (torch.float32, torch.float64) and len(fin) <= 2:
                try:
                    h = nat.RowHandle(_kernel_address(fwd_kernel, dtype), _kernel_address(bwd_kernel, dtype),
                                      _kernel_address(bwd_kernel + "_gb", dtype), fin[0], fin[1] if len(fin) == 2 else 0, fout,
                                      saved_spec, list(bout), cls._native_rule(nat), dtype == torch.float64)
                except Exception:
                    h = None
            hs[dtype] = h
            return h

        @classmethod
        def _native_rule(cls, nat):
            """key of this operator's differentiable backward rule in the extension (registered on first use)"""
            key = cls.__dict__.get('_rule_key')
            if key is None:
                _native_state["rules"] += 1
                key = cls._rule_key = _native_state["rules"]
                nat.set_rule(key, Bwd.apply)
            return key

        @staticmethod
        def forward(*ins):
            return _launch(fwd_kernel, ins, fin, (fout,))[0]

        @staticmethod
        def setup_context(ctx, inputs, output):
            saved = []
            for key in reads.split(","):
                saved.append(output if key == "out" else inputs[int(key[2:])])
            ctx.save_for_backward(*saved)

        @staticmethod
        def backward(ctx, grad_output):
            if not torch.is_grad_enabled() and not _transforms_active():
                # plain first-order backward: launch directly (Function.apply costs ~40 us of Python per call; the
                # legacy-vmap case of is_grads_batched is peeled inside _launch either way)
                res = _launch(bwd_kernel, (*ctx.saved_tensors, grad_output), bin_, bout)
                return res[0] if len(bout) == 1 else res
            res = Bwd.apply(*ctx.saved_tensors, grad_output)
            return res if isinstance(res, tuple) else res

        @staticmethod
        def vmap(info, in_dims, *ins):
            return _Fn.apply(*_fold_vmap(in_dims, ins)), 0

    _Fn.__name__ = _Fn.__qualname__ = clsname
    _Fn._bwd = Bwd
    _Fn._dry_kernel = (fwd_kernel, fin, fout)       # what a dry trace notes for this op (lietensor.LieType._dry)
    return _Fn


_DOC = {
    "exp": "{g}_Exp (reference operation.py: so3 :340, se3 :398, rxso3 :444, sim3 :492)",
    "log": "{G}_Log (reference operation.py: SO3 :304, SE3 :373, RxSO3 :421, Sim3 :467)",
    "inv": "{G}_Inv (reference operation.py:930-1021)",
    "mul": "{G}_Mul (reference operation.py:829-927)",
    "act": "{G}_Act (reference operation.py:516-620)",
    "act4": "{G}_Act4 (reference operation.py:623-722)",
    "adj": "{G}_AdjXa (reference operation.py:725-826)",
    "adjt": "{G}_AdjTXa (reference operation.py:1024-1113)",
}
_SUFFIX = {"log": "_Log", "inv": "_Inv", "mul": "_Mul", "act": "_Act", "act4": "_Act4", "adj": "_AdjXa", "adjt": "_AdjTXa"}

__all__ = []
for _g in _GROUPS:
    for _kind in _DOC:
        _name = f"{_g}_Exp" if _kind == "exp" else _CAP[_g] + _SUFFIX[_kind]
        globals()[_name] = _make_fwd(_name, _g, _kind, _DOC[_kind].format(g=_g, G=_CAP[_g]))
        __all__.append(_name)


# ---------------------------------------------------------------------------------------
# Jinvp: reference differentiates through so3_Jl_inv/calcQ with plain autograd
# (lietensor.py:257-264, 422-429, 556-563, 700-707).  Forward is one fused kernel.
# ---------------------------------------------------------------------------------------
def _make_jinvp(g):
    da, dg = _GROUPS[g]
    kernel = f"{g}_jinvp_fwd"
    # one kernel: forward-mode sweeps through Jl_inv(Log X) p, then <Group>_Log's backward (lie_math.h)
    def composed_forward(X, p):        # Jl_inv(Log X) p (lietensor.py:257-264 ...), through the differentiable Log Function
        log = globals()[_CAP[g] + "_Log"]
        return (_mats.__dict__[g + "_Jl_inv"](log.apply(X)) @ p.unsqueeze(-1)).squeeze(-1)
    Bwd = _make_bwd(_CAP[g] + "_Jinvp_Bwd", f"{g}_jinvp_bwd", (dg, da, da), (dg, da), _grad_of(composed_forward))

    class _Jinvp(torch.autograd.Function):
        @staticmethod
        def forward(X, p):
            return _launch(kernel, (X, p), (dg, da), (da,))[0]

        @staticmethod
        def setup_context(ctx, inputs, output):
            ctx.save_for_backward(*inputs)

        @staticmethod
        def backward(ctx, grad_output):
            return Bwd.apply(*ctx.saved_tensors, grad_output)

        @staticmethod
        def vmap(info, in_dims, *ins):
            return _Jinvp.apply(*_fold_vmap(in_dims, ins)), 0

    _Jinvp.__name__ = _Jinvp.__qualname__ = _CAP[g] + "_Jinvp"
    return _Jinvp


SO3_Jinvp, SE3_Jinvp, Sim3_Jinvp, RxSO3_Jinvp = (_make_jinvp(g) for g in ("so3", "se3", "sim3", "rxso3"))


_so3_Jr_Bwd = _make_bwd("so3_Jr_Bwd", "so3_jr_bwd", (3, 9), (3,), _grad_of(lambda x: _mats.so3_Jl(-x).flatten(-2)))


class so3_Jr(torch.autograd.Function):
    """Right Jacobian of so3 (reference lietensor.py:343-351): [...,3] -> [...,3,3]; the reference
    differentiates it with plain autograd, here the backward is one kernel (forward-mode sweeps)."""

    @staticmethod
    def forward(x):
        return _launch("so3_jr_fwd", (x,), (3,), (9,))[0].unflatten(-1, (3, 3))

    @staticmethod
    def setup_context(ctx, inputs, output):
        ctx.save_for_backward(inputs[0])

    @staticmethod
    def backward(ctx, grad_output):
        (x,) = ctx.saved_tensors
        return _so3_Jr_Bwd.apply(x, grad_output.flatten(-2))

    @staticmethod
    def vmap(info, in_dims, x):
        return so3_Jr.apply(*_fold_vmap(in_dims, (x,))), 0


__all_helpers__ = list(_mats.__all__)


def broadcast_inputs(x, y):
    """Reference operation.py:1116-1125: broadcast leading dims and flatten to rows."""
    if y is None:
        return (x.reshape(-1, x.shape[-1]).contiguous(),), tuple(x.shape[:-1])
    if x.dim() == 2 and y.dim() == 2 and x.shape[0] == y.shape[0] and x.is_contiguous() and y.is_contiguous():
        return (x, y), (x.shape[0],)                 # already rows: nothing to broadcast, flatten or copy
    if _C.dry_tracing():
        # a dry trace (optim/fused.py) records which tensors meet in which op: nothing is expanded or copied, the operands
        # stay the caller's own tensors (the recorded launch broadcasts the leading dimensions itself)
        shp = x.shape[:-1] if x.shape[:-1] == y.shape[:-1] else torch.broadcast_shapes(x.shape[:-1], y.shape[:-1])
        return (x, y), tuple(shp)
    # (equal shapes are the common case; torch.broadcast_shapes costs ~10 us of Python)
    out_shape = x.shape[:-1] if x.shape[:-1] == y.shape[:-1] else torch.broadcast_shapes(x.shape[:-1], y.shape[:-1])
    shape = out_shape if out_shape != torch.Size([]) else (1,)
    x = x.expand(tuple(shape) + (x.shape[-1],)).reshape(-1, x.shape[-1]).contiguous()
    y = y.expand(tuple(shape) + (y.shape[-1],)).reshape(-1, y.shape[-1]).contiguous()
    return (x, y), tuple(out_shape)
