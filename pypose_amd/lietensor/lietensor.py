"""``LieTensor`` / ``LieType`` / ``Parameter`` -- host-side mirror of the reference's tensor type.

Mirrors the public behaviour of ``pypose/lietensor/lietensor.py`` (LieType :37-194, the eight
concrete types :196-768, singletons :771-776, LieTensor :778-1233, Parameter :1236-1337,
retain_ltype :1339-1370): same class / attribute / method names, same argument meaning, same
error behaviour, so that code written against ``pypose`` runs unchanged against this package.

The design differs: there is ONE ``LieType`` implementation driven by a small per-type
descriptor (widths, layout of translation / quaternion / scale inside the embedding, the op
table) instead of eight hand-written classes; the eight reference class names are created from
it.  All arithmetic goes to ``operation.py`` (one HIP kernel per op).
"""
from __future__ import annotations

import importlib
import warnings
from collections.abc import Iterable, Sequence
from contextlib import contextmanager
from numbers import Number

import torch
from torch import Tensor, nn
from torch.utils._pytree import tree_flatten, tree_map

from .. import _C
from . import operation as _op
from .operation import broadcast_inputs

# torch functions whose Tensor results keep the ltype of their LieTensor argument
# (reference lietensor.py:26-35)
HANDLED_FUNCTIONS = frozenset("""
__getitem__ __setitem__ cpu cuda float double to detach view view_as squeeze unsqueeze cat stack split
hsplit dsplit vsplit tensor_split chunk concat column_stack dstack vstack hstack index_select
masked_select movedim moveaxis narrow permute reshape row_stack scatter scatter_add clone swapaxes
swapdims take take_along_dim tile copy transpose unbind gather repeat expand expand_as index_copy
index_copy_ select select_scatter index_put index_put_ copy_
""".split())


_gather_recorders = []      # active optim.posegraph.GatherRecorder instances
# torch functions that look at a tensor's metadata only
_VALUE_FREE = frozenset(("dim", "size", "stride", "numel", "is_contiguous", "data_ptr", "storage_offset", "element_size",
                         "ndimension", "is_floating_point", "is_complex", "type", "requires_grad_", "nelement", "get_device"))
# property getters reach __torch_function__ as the generic ``__get__`` of a getset descriptor: only the ones NAMED here read
# metadata; .data, .grad, .T, .mT, .H, .real, .imag ... hand out value-carrying aliases the tracer no longer sees
_VALUE_FREE_PROPERTIES = frozenset(("shape", "dtype", "device", "requires_grad", "ndim", "is_cuda", "is_cpu", "is_leaf", "layout",
                                    "is_sparse", "is_quantized", "is_meta", "names", "itemsize", "nbytes", "grad_fn", "output_nr",
                                    "_version", "is_mkldnn", "is_xpu", "is_mps", "is_nested", "is_sparse_csr", "retains_grad"))


def _value_free(func, name):
    if name == "__get__":
        return getattr(getattr(func, "__self__", None), "__name__", None) in _VALUE_FREE_PROPERTIES
    return name in _VALUE_FREE


def _raw(t):
    return t.tensor() if isinstance(t, LieTensor) else t


def _wrap(t, ltype):
    """an op's plain output as a LieTensor, past LieTensor.__init__'s shape assertion (``self.shape`` on a tensor subclass
    is a __torch_function__ round trip: ~5 us on the dispatch path of every op; the kernels fix the width)"""
    lt = Tensor.as_subclass(t, LieTensor)
    lt.ltype = ltype
    return lt


class LieType:
    """Descriptor + behaviour of one Lie type (group or algebra).

    dimension: width of the stored data; embedding: width of the group element it maps to;
    manifold: degrees of freedom (reference lietensor.py:39-58).
    """

    # per-type descriptor, filled by the subclasses created in ``_declare``
    _key = None          # 'so3' | 'se3' | 'sim3' | 'rxso3'
    _is_group = False
    _dims = (0, 0, 0)    # dimension, embedding, manifold
    _t = _q = _s = None  # slices of translation / quaternion / scale inside a GROUP element
    _sigma_n = 1

    def __init__(self):
        d, e, m = self._dims
        self._dimension, self._embedding, self._manifold = torch.Size([d]), torch.Size([e]), torch.Size([m])

    # -- descriptors ---------------------------------------------------------------------
    @property
    def dimension(self) -> torch.Size:
        return self._dimension

    @property
    def embedding(self) -> torch.Size:
        return self._embedding

    @property
    def manifold(self) -> torch.Size:
        return self._manifold

    @property
    def on_manifold(self) -> bool:
        return self.dimension == self.manifold

    @property
    def _group(self):
        return _GROUP_OF[self._key]

    @property
    def _algebra(self):
        return _ALGEBRA_OF[self._key]

    def _fn(self, kind):
        names = self.__dict__.get("_fn_names")
        if names is None:                  # (built once per type: this sits on the dispatch path of every op)
            cap = {"so3": "SO3", "se3": "SE3", "sim3": "Sim3", "rxso3": "RxSO3"}[self._key]
            names = self.__dict__["_fn_names"] = {
                "exp": self._key + "_Exp", "log": cap + "_Log", "inv": cap + "_Inv", "mul": cap + "_Mul", "act": cap + "_Act",
                "act4": cap + "_Act4", "adj": cap + "_AdjXa", "adjt": cap + "_AdjTXa", "jinvp": cap + "_Jinvp"}
        return getattr(_op, names[kind])   # resolved at call time (rebinding-friendly)

    def _dry(self, kind, xs, out_ltype):
        """During a dry trace (optim/fused.py DryTracer: the model's Python runs, nothing is launched) an op is a note in the
        trace and a storage-less result that the tracer keeps per trace position -- none of the Function / autograd /
        broadcasting machinery below is needed to produce it.  None unless a dry trace is active on this thread and the
        Function this op would call is still this package's own (someone who rebinds it gets the ordinary path)."""
        if not getattr(_C._tls, "dry", 0):
            return None
        fn = self._fn(kind)
        if getattr(fn, "_dry_kernel", None) is None:
            return None
        return _op._op_tracers[-1].dry_lie(fn, xs, out_ltype)

    # -- Exp / Log -----------------------------------------------------------------------
    def Exp(self, x):
        if self._is_group:
            raise AttributeError("Lie Group has no Exp attribute")
        out = self._dry("exp", (x,), self._group)
        return out if out is not None else _wrap(self._fn("exp").apply(_raw(x)), self._group)

    def Log(self, X):
        if not self._is_group:
            raise AttributeError("Lie Algebra has no Log attribute")
        out = self._dry("log", (X,), self._algebra)
        return out if out is not None else _wrap(self._fn("log").apply(_raw(X)), self._algebra)

    def Inv(self, X):
        if not self._is_group:
            return LieTensor(-X, ltype=self)
        out = self._dry("inv", (X,), self)
        return out if out is not None else _wrap(self._fn("inv").apply(_raw(X)), self)

    # -- binary ops ------------------------------------------------------------------------
    def _binary(self, kind, X, other, out_ltype):
        (x, y), out_shape = broadcast_inputs(_raw(X), _raw(other))
        out = self._fn(kind).apply(x, y)
        width = -1 if out.nelement() != 0 else y.shape[-1]
        out = out.view(tuple(out_shape) + (width,))
        return out if out_ltype is None else _wrap(out, out_ltype)

    def Act(self, X, p):
        if not self._is_group:
            raise NotImplementedError("Instance has no Act attribute.")
        assert isinstance(p, Tensor)
        assert p.shape[-1] == 3 or p.shape[-1] == 4, "Invalid Tensor Dimension"
        return self._binary("act" if p.shape[-1] == 3 else "act4", X, p, None)

    def Mul(self, X, Y):
        if self._is_group:
            if isinstance(Y, LieTensor) and not Y.ltype.on_manifold:      # transform o transform
                out = self._dry("mul", (X, Y), self)
                if out is not None:
                    return out
                (x, y), out_shape = broadcast_inputs(_raw(X), _raw(Y))
                out = self._fn("mul").apply(x, y)
                width = -1 if out.nelement() != 0 else x.shape[-1]
                return _wrap(out.view(tuple(out_shape) + (width,)), self)
            if isinstance(Y, Tensor) and not isinstance(Y, LieTensor):     # transform o points
                return self.Act(X, Y)
            raise NotImplementedError('Invalid __mul__ operation')
        return LieTensor(torch.mul(_raw(X), Y), ltype=self)               # (scalar | tensor) * algebra

    def Adj(self, X, a):
        if not self._is_group:
            raise NotImplementedError("Instance has no Adj attribute.")
        return self._binary("adj", X, a, self._algebra)

    def AdjT(self, X, a):
        if not self._is_group:
            raise NotImplementedError("Instance has no AdjT attribute.")
        return self._binary("adjt", X, a, self._algebra)

    def Jinvp(self, X, p):
        if not self._is_group:
            raise NotImplementedError("Instance has no Jinvp attribute.")
        return self._binary("jinvp", X, p, self._algebra)

    def Retr(self, X, a):
        if not self._is_group:
            raise AttributeError("Has no Retr attribute")
        return a.Exp() * X

    def Jr(self, X):
        if self._key != "so3":
            raise NotImplementedError("Instance has no Jr attribute")
        if self._is_group:
            return X.Log().Jr()
        return _op.so3_Jr.apply(_raw(X))

    # -- in-place update used by the optimizers (reference :60-65, 277-279, 442-444, ...) ---
    def add_(self, input, other):
        m = self.manifold[0]
        if not self._is_group:
            return input.copy_(Tensor.as_subclass(input, Tensor) + Tensor.as_subclass(other, Tensor)[..., :m])
        raw, x = Tensor.as_subclass(other, Tensor), Tensor.as_subclass(input, Tensor)
        if raw.shape == x.shape and not (torch.is_grad_enabled() and (raw.requires_grad or x.requires_grad)) \
                and raw.dtype == x.dtype and not _op._transforms_active():
            # the optimizer's update (step zero-padded to the group width): one fused kernel Exp(d[:m]) * p, written
            # in place when the storage allows it
            w = x.shape[-1]
            if x.is_contiguous() and raw.is_contiguous() and not _op._op_tracers:
                xr = x.detach().view(-1, w)
                _C.row_op(self._key + "_retract", [raw.detach().view(-1, w), xr], (w,), out=[xr])
                _C.mark_written(input)            # the kernel wrote through the raw pointer: autograd must see the in-place edit
                return input
            out = _op._launch(self._key + "_retract", (raw.detach(), x.detach()), (w,) * 2, (w,))[0]
            with torch.no_grad():
                return input.copy_(out)
        delta = LieTensor(raw[..., :m], ltype=self._algebra)
        return input.copy_(delta.Exp() * input)

    # -- views of the components -------------------------------------------------------------
    def matrix(self, input):
        X = input if self._is_group else input.Exp()
        k = 3 if self._key == "so3" else 4
        I = torch.eye(k, dtype=X.dtype, device=X.device).view([1] * (X.dim() - 1) + [k, k])
        return X.unsqueeze(-2).Act(I).transpose(-1, -2)

    def rotation(self, input):
        if not self._is_group:
            return input.Exp().rotation()
        if self._key == "so3":
            return input
        return LieTensor(input.tensor()[..., self._q], ltype=SO3_type)

    def translation(self, input):
        if self._t is None:
            warnings.warn("Instance has no translation. Zero vector(s) is returned.")
            return torch.zeros(input.lshape + (3,), dtype=input.dtype, device=input.device,
                               requires_grad=input.requires_grad)
        return input.tensor()[..., self._t] if self._is_group else input.Exp().translation()

    def scale(self, input):
        if self._s is None:
            warnings.warn("Instance has no scale. Scalar one(s) is returned.")
            return torch.ones(input.lshape + (1,), dtype=input.dtype, device=input.device,
                              requires_grad=input.requires_grad)
        return input.tensor()[..., self._s] if self._is_group else input.Exp().scale()

    # -- constructors ------------------------------------------------------------------------
    @classmethod
    def to_tuple(cls, input):
        out = tuple()
        for i in input:
            out += tuple(i) if isinstance(i, Iterable) else (i,)
        return out

    def identity(self, *size, **kwargs):
        grp = self._group
        if not self._is_group:      # Log(identity) is exactly zero for every group
            return LieTensor(torch.zeros(tuple(size) + tuple(self.dimension), **kwargs), ltype=self)
        vals = [0.] * grp.embedding[0]
        vals[grp._q.stop - 1] = 1.
        if grp._s is not None:
            vals[grp._s.start] = 1.
        data = torch.tensor(vals, **kwargs)
        return LieTensor(data.repeat(tuple(size) + (1,)), ltype=grp)

    def identity_like(self, *args, **kwargs):
        return self.identity(*args, **kwargs)

    def identity_(self, X):
        if not self._is_group:
            raise NotImplementedError("Instance has no identity_ method")
        X.fill_(0)
        idx = [self._q.stop - 1] + ([self._s.start] if self._s is not None else [])
        X.index_fill_(dim=-1, index=torch.tensor(idx, device=X.device), value=1)
        return X

    def randn_like(self, *args, sigma=1.0, **kwargs):
        return self.randn(*args, sigma=sigma, **kwargs)

    def randn(self, *size, sigma=1.0, requires_grad=False, **kwargs):
        """Same sampling recipe and RNG call order as the reference (:271-275, 323-331, 436-440,
        473-491, 579-583, 619-635, 720-724, 757-768) so that seeded draws coincide."""
        if self._is_group:
            data = self._algebra.Exp(self._algebra.randn(*size, sigma=sigma, **kwargs)).detach()
            return LieTensor(data, ltype=self).requires_grad_(requires_grad)
        size = self.to_tuple(size)
        key = self._key
        if key == "so3":
            assert isinstance(sigma, Number), 'Only accepts sigma as a single number'
            data = torch.randn(*(size + torch.Size([3])), **kwargs)
            dist = data.norm(dim=-1, keepdim=True)
            theta = sigma * torch.randn(*(size + torch.Size([1])), **kwargs)
            return LieTensor(data / dist * theta, ltype=self).requires_grad_(requires_grad)
        if key == "se3":
            if not isinstance(sigma, Sequence):
                sigma = (sigma,) * 4
            elif len(sigma) == 2:
                ts = sigma[0]
                sigma = (tuple(ts) if isinstance(ts, Sequence) else (ts,) * 3) + (sigma[-1],)
            else:
                assert len(sigma) == 4, 'Only accepts a tuple of sigma in size 1, 2, or 4.'
            rotation = so3_type.randn(*size, sigma=sigma[-1], **kwargs).tensor().detach()
            translation = torch.tensor(list(sigma[:3]), **kwargs) * torch.randn(*(size + torch.Size([3])), **kwargs)
            data = torch.cat([translation, rotation], dim=-1)
        elif key == "sim3":
            if not isinstance(sigma, Sequence):
                sigma = (sigma,) * 5
            elif len(sigma) == 3:
                ts = sigma[0]
                sigma = (tuple(ts) if isinstance(ts, Sequence) else (ts,) * 3) + (sigma[-2], sigma[-1])
            else:
                assert len(sigma) == 5, 'Only accepts a tuple of sigma in size 1, 3, or 5.'
            rotation = so3_type.randn(*size, sigma=sigma[-2], **kwargs).tensor().detach()
            scale = sigma[-1] * torch.randn(*(size + torch.Size([1])), **kwargs)
            translation = torch.tensor(list(sigma[:3]), **kwargs) * torch.randn(*(size + torch.Size([3])), **kwargs)
            data = torch.cat([translation, rotation, scale], dim=-1)
        else:  # rxso3
            if not isinstance(sigma, Sequence):
                sigma = (sigma, sigma)
            else:
                assert len(sigma) == 2, 'Only accepts a tuple of sigma in size 1 or 2.'
            rotation = so3_type.randn(*size, sigma=sigma[0], **kwargs).tensor()
            scale = sigma[1] * torch.randn(*(size + torch.Size([1])), **kwargs)
            data = torch.cat([rotation, scale], dim=-1)
        return LieTensor(data, ltype=self).requires_grad_(requires_grad)

    # -- scans (basics/ops.py) ----------------------------------------------------------------
    @classmethod
    def cumops(cls, X, dim, ops):
        from ..basics import cumops
        return cumops(X, dim, ops)

    @classmethod
    def cummul(cls, X, dim, left=True):
        from ..basics import cummul
        return cummul(X, dim, left)

    @classmethod
    def cumprod(cls, X, dim, left=True):
        from ..basics import cumprod
        return cumprod(X, dim, left)

    @classmethod
    def cumops_(cls, X, dim, ops):
        from ..basics import cumops_
        return cumops_(X, dim, ops)

    @classmethod
    def cummul_(cls, X, dim, left=True):
        from ..basics import cummul_
        return cummul_(X, dim, left)

    @classmethod
    def cumprod_(cls, X, dim, left=True):
        from ..basics import cumprod_
        return cumprod_(X, dim, left)


def _declare(clsname, key, is_group, dims, t=None, q=None, s=None):
    return type(clsname, (LieType,), dict(_key=key, _is_group=is_group, _dims=dims, _t=t, _q=q, _s=s))


SO3Type = _declare("SO3Type", "so3", True, (4, 4, 3), q=slice(0, 4))
so3Type = _declare("so3Type", "so3", False, (3, 4, 3))
SE3Type = _declare("SE3Type", "se3", True, (7, 7, 6), t=slice(0, 3), q=slice(3, 7))
se3Type = _declare("se3Type", "se3", False, (6, 7, 6), t=slice(0, 3))
Sim3Type = _declare("Sim3Type", "sim3", True, (8, 8, 7), t=slice(0, 3), q=slice(3, 7), s=slice(7, 8))
sim3Type = _declare("sim3Type", "sim3", False, (7, 8, 7), t=slice(0, 3), s=slice(6, 7))
RxSO3Type = _declare("RxSO3Type", "rxso3", True, (5, 5, 4), q=slice(0, 4), s=slice(4, 5))
rxso3Type = _declare("rxso3Type", "rxso3", False, (4, 5, 4), s=slice(3, 4))

SO3_type, so3_type = SO3Type(), so3Type()
SE3_type, se3_type = SE3Type(), se3Type()
Sim3_type, sim3_type = Sim3Type(), sim3Type()
RxSO3_type, rxso3_type = RxSO3Type(), rxso3Type()
_GROUP_OF = {"so3": SO3_type, "se3": SE3_type, "sim3": Sim3_type, "rxso3": RxSO3_type}
_ALGEBRA_OF = {"so3": so3_type, "se3": se3_type, "sim3": sim3_type, "rxso3": rxso3_type}
liegroup = [SO3_type, SE3_type, Sim3_type, RxSO3_type]
liealgebra = [so3_type, se3_type, sim3_type, rxso3_type]


class LieTensor(Tensor):
    """A ``torch.Tensor`` subclass tagged with a Lie type (reference lietensor.py:778-1233).

    ``LieTensor(data, ltype=pp.SE3_type)``; the last dimension must equal ``ltype.dimension``.
    ``lshape`` is ``shape[:-1]``.  Methods forward to the type (``x.Exp()``, ``X.Log()``,
    ``X.Inv()``, ``X @ Y``, ``X.Act(p)``, ``X.Adj(a)``, ...).
    """

    def __init__(self, *data, ltype: LieType):
        assert self.shape[-1:] == ltype.dimension, 'The last dimension of a LieTensor has to be ' \
            'corresponding to their LieType. More details go to {}. If this error happens in an ' \
            'optimization process, where LieType is not a necessary structure, we suggest to ' \
            'call .tensor() to convert a LieTensor to Tensor before passing it to an optimizer. ' \
            'If this still happens, create an issue on GitHub please.'.format(
                'https://pypose.org/docs/main/generated/pypose.LieTensor')
        self.ltype = ltype

    @staticmethod
    def __new__(cls, *data, ltype):
        tensor = data[0] if isinstance(data[0], Tensor) else Tensor(*data)
        return Tensor.as_subclass(tensor, LieTensor)

    def __repr__(self):
        if hasattr(self, 'ltype'):
            return f"{type(self.ltype).__name__} {type(self).__name__}:\n" + super().__repr__()
        return super().__repr__()

    def new_empty(self, size, *, dtype=None, layout=None, device=None, pin_memory=None, requires_grad=None):
        out = torch.empty(size, dtype=self.dtype if dtype is None else dtype,
                          layout=self.layout if layout is None else layout,
                          device=self.device if device is None else device, pin_memory=pin_memory,
                          requires_grad=self.requires_grad if requires_grad is None else requires_grad)
        out = Tensor.as_subclass(out, type(self))
        if hasattr(self, 'ltype'):
            out.ltype = self.ltype
        return out

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = {} if kwargs is None else kwargs
        plain = tuple(Tensor if issubclass(t, LieTensor) else t for t in types)
        name = getattr(func, '__name__', None)
        data = None
        if _C.dry_tracing():
            if _gather_recorders and name == '__getitem__' and len(args) == 2:
                # dry trace (optim/fused.py): a row gather on a tracked parameter is noted, not executed
                data = _op._op_tracers[-1].dry_gather(args[0], args[1], _gather_recorders)
            if data is None and not _value_free(func, name):
                # any other torch function on a LieTensor during a dry trace may read VALUES of a real tensor (a parameter):
                # remembered, so that nothing is run speculatively around such a model (fused.checked_shortcut)
                _op._op_tracers[-1].touched = True
        if data is None:
            data = Tensor.__torch_function__(func, plain, args, kwargs)
        if _gather_recorders and name == '__getitem__' and len(args) == 2:
            for rec in _gather_recorders:        # optim/posegraph.py: which rows feed which residual
                rec.note(args[0], args[1], data)
        if data is None or name not in HANDLED_FUNCTIONS:
            return data
        if type(data) is Tensor and args and isinstance(args[0], LieTensor):
            # fast path of the common case (method on a LieTensor returning one plain tensor): no pytree walk
            ltype = args[0].ltype
            lt = Tensor.as_subclass(data, LieTensor)
            lt.ltype = ltype
            if lt.shape[-1:] != ltype.dimension:
                warnings.warn('Tensor Shape Invalid by calling {}, go to {}'.format(
                    func, 'https://pypose.org/docs/main/generated/pypose.LieTensor'))
            return lt
        flat, _ = tree_flatten(args)
        ltype = next(a.ltype for a in flat if isinstance(a, LieTensor))

        def rewrap(t):
            if isinstance(t, Tensor) and not isinstance(t, cls):
                lt = Tensor.as_subclass(t, LieTensor)
                lt.ltype = ltype
                if lt.shape[-1:] != ltype.dimension:
                    warnings.warn('Tensor Shape Invalid by calling {}, go to {}'.format(
                        func, 'https://pypose.org/docs/main/generated/pypose.LieTensor'))
                return lt
            return t
        return tree_map(rewrap, data)

    @property
    def lshape(self) -> torch.Size:
        return self.shape[:-1]

    def lview(self, *shape):
        return self.view(*shape + self.ltype.dimension)

    def tensor(self) -> Tensor:
        pl = self.__dict__.get("_pl")          # (results of a dry trace carry their plain alias: optim/fused.py dry_lie)
        if pl is not None:
            return pl
        if _C.dry_tracing():                   # a plain alias of a REAL LieTensor escapes the tracer's view: see _VALUE_FREE
            _op._op_tracers[-1].touched = True
        return Tensor.as_subclass(self, Tensor)

    # arithmetic: all forwarded to the type
    def Exp(self):
        return self.ltype.Exp(self)

    def Log(self):
        return self.ltype.Log(self)

    def Inv(self):
        return self.ltype.Inv(self)

    def Act(self, p):
        return self.ltype.Act(self, p)

    def add(self, other, alpha=1):
        return self.clone().add_(other=alpha * other)

    def add_(self, other, alpha=1):
        return self.ltype.add_(self, other=other if alpha == 1 else alpha * other)      # (alpha * other is a launch)

    def __add__(self, other):
        return self.add(other=other)

    def __mul__(self, other):
        return self.ltype.Mul(self, other)

    def mul(self, other):
        return self.ltype.Mul(self, other)

    def __matmul__(self, other):
        if isinstance(other, LieTensor):
            return self.ltype.Mul(self, other)
        return self.Act(other)

    def Retr(self, a):
        return self.ltype.Retr(self, a)

    def Adj(self, a):
        return self.ltype.Adj(self, a)

    def AdjT(self, a):
        return self.ltype.AdjT(self, a)

    def Jinvp(self, p):
        return self.ltype.Jinvp(self, p)

    def Jr(self):
        return self.ltype.Jr(self)

    def matrix(self):
        return self.ltype.matrix(self)

    def translation(self):
        return self.ltype.translation(self)

    def rotation(self):
        return self.ltype.rotation(self)

    def scale(self):
        return self.ltype.scale(self)

    def euler(self, eps=2e-4):
        """roll/pitch/yaw of the rotation part (reference lietensor.py:1147-1173; kernel pplie_so3_euler)."""
        from .convert import so3_euler
        return so3_euler(self.rotation().tensor(), eps)

    def identity_(self):
        return self.ltype.identity_(self)

    def cumops(self, dim, ops):
        return self.ltype.cumops(self, dim, ops)

    def cummul(self, dim, left=True):
        return self.ltype.cummul(self, dim, left)

    def cumprod(self, dim, left=True):
        return self.ltype.cumprod(self, dim, left)

    def cumops_(self, dim, ops):
        return self.ltype.cumops_(self, dim, ops)

    def cummul_(self, dim, left=True):
        return self.ltype.cummul_(self, dim, left)

    def cumprod_(self, dim, left=True):
        return self.ltype.cumprod_(self, dim, left)


class Parameter(LieTensor, nn.Parameter):
    """``nn.Parameter`` that keeps its ``ltype`` (reference lietensor.py:1236-1337).

    ``sjac=True`` (the reference's request for its optional sparse-Jacobian plugin, ``bae``) is
    accepted and ignored: sparse structure is detected by ``pypose_amd.optim`` without tracing.
    """

    def __init__(self, data=None, requires_grad=True, sjac=False):
        if hasattr(data, 'ltype'):
            self.ltype = data.ltype

    def __new__(cls, data=None, requires_grad=True, sjac=False):
        if data is None:
            data = torch.tensor([])
        # sjac=True asks the reference's optional `bae` plugin to trace operations for sparse Jacobians.
        # Here structure is discovered by the optimizer itself (optim/blocks.py, optim/posegraph.py), so
        # the flag is accepted and needs no tracing tensor: the parameter is an ordinary Parameter.
        if isinstance(data, LieTensor):
            param = Tensor._make_subclass(cls, data.tensor(), requires_grad)
            param.ltype = data.ltype
            param._is_param = True
            return param
        if sjac and type(data) is Tensor:
            # the reference hands back nn.Parameter(TrackingTensor(data)), whose ``.tensor()`` its callers (and its own
            # tests/optim/test_sparse_lm.py:86) use to get at the plain values
            return Tensor._make_subclass(_TrackedParameter, data, requires_grad)
        return nn.Parameter(data, requires_grad)

    def __deepcopy__(self, memo):
        if id(self) not in memo:
            memo[id(self)] = type(self)(self.clone(memory_format=torch.preserve_format))
        return memo[id(self)]


class _TrackedParameter(nn.Parameter):
    """what ``Parameter(plain_tensor, sjac=True)`` returns: an ``nn.Parameter`` with the ``tensor()`` accessor of the
    reference's tracking wrapper (reference lietensor.py:1308-1323)"""

    def tensor(self):
        return Tensor.as_subclass(self, Tensor)

    def __deepcopy__(self, memo):
        if id(self) not in memo:
            memo[id(self)] = Tensor._make_subclass(type(self), self.data.clone(memory_format=torch.preserve_format),
                                                   self.requires_grad)
        return memo[id(self)]


@contextmanager
def retain_ltype():
    """Keep ``ltype`` on tensors wrapped by functorch / forward-AD inside the block
    (reference lietensor.py:1339-1370): patches the three torch entry points that re-wrap
    tensors (``make_dual``, ``_wrap_tensor_for_grad``, ``_add_batch_dim``)."""
    targets = [(torch.autograd.forward_ad, "make_dual"),
               (importlib.import_module("torch._functorch.eager_transforms"), "_wrap_tensor_for_grad"),
               (importlib.import_module("torch._functorch.vmap"), "_add_batch_dim")]
    originals = [(m, n, getattr(m, n)) for m, n in targets]

    def keep(fn):
        def inner(*args, **kwargs):
            ltype = args[0].ltype if isinstance(args[0], LieTensor) else None
            res = fn(*args, **kwargs)
            if ltype is not None:
                res = Tensor.as_subclass(res, LieTensor)
                res.ltype = ltype
            return res
        return inner
    try:
        for m, n, f in originals:
            setattr(m, n, keep(f))
        yield
    finally:
        for m, n, f in originals:
            setattr(m, n, f)
