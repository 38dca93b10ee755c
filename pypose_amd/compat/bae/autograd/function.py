"""``TrackingTensor`` and ``map_transform`` (what pypose.Parameter(sjac=True) and @psjac resolve to).

The reference's sparse LM step runs under ``torch.no_grad()`` and calls ``jacobian(R, params)`` afterwards
(pypose/optim/optimizer.py:498, 631-637), so the tracked values have to carry their own history:

* a TrackingTensor is CONTAGIOUS -- every torch function and every LieTensor method that touches one runs with grad mode
  on and returns TrackingTensors (``.tensor()`` drops the tracking, as the reference expects at optimizer.py:638-639);
* ROW GATHERS are remembered: ``param[index]`` (and the same on ``torch.cat`` of tracked and fixed rows along dim 0, the
  reference's chain-PGO pattern) appends (parameter, index, gathered rows) to a tape.  ``bae.autograd.graph.jacobian``
  differentiates the residual with respect to the gathered rows -- one batched backward sweep per residual component, so
  rows sharing a parameter row never collide -- and scatters the per-row blocks into sparse matrices.  A tracked
  parameter that is used directly (no gather) counts as gathered with ``index = arange``.

Known limit (documented in INTEGRATION.md): when the ONLY tracked operand of a LieTensor operation is a plain tensor on
the right-hand side of an untracked LieTensor (``fixed_pose @ tracked_points``), the reference calls its autograd
Function with grad mode off before this class sees anything; put the tracked value on the left or gather both operands.
"""
import operator
from functools import wraps

import torch

_TAPE = []          # (root parameter, row index or None for "all rows", gathered rows as they appear in the graph)
_TAPE_CAP = 32           # a model has a handful of gathers; jacobian() empties the tape, this bounds what forward-only use retains
_PRESERVE = {"detach", "requires_grad_", "clone", "to", "cuda", "cpu", "contiguous", "double", "float", "half", "type",
             "__deepcopy__", "share_memory_", "pin_memory"}
_INFO = {"dim", "size", "stride", "numel", "nelement", "ndimension", "data_ptr", "element_size", "storage_offset", "get_device",
         "__len__", "__repr__", "__str__", "__format__", "__hash__", "_is_view", "untyped_storage", "storage", "tolist", "item",
         "__bool__", "__int__", "__float__", "__index__", "__reduce_ex__", "_backward_hooks", "register_hook", "retain_grad",
         "backward", "_version", "_base", "__reversed__", "numpy", "__array__", "__dlpack__", "has_names", "dim_order"}
_dyn = {}


def _record(root, idx, out):
    _TAPE.append((root, idx, out))
    del _TAPE[:-_TAPE_CAP]


def _wrap(value, like=None):
    if isinstance(value, (tuple, list)):
        return type(value)(_wrap(v) for v in value)
    if not isinstance(value, torch.Tensor) or isinstance(value, TrackingTensor):
        return value
    base = type(value)
    klass = TrackingTensor if base in (torch.Tensor, torch.nn.Parameter) else _dyn_class(base)
    with torch.enable_grad(), torch._C.DisableTorchFunctionSubclass():
        out = torch.Tensor.as_subclass(value, klass)
    if hasattr(value, "ltype"):
        out.ltype = value.ltype
    return out


def _unwrap(a):
    if isinstance(a, TrackingTensor):
        return a._payload()
    if isinstance(a, (tuple, list)):
        return type(a)(_unwrap(x) for x in a)
    if isinstance(a, dict):
        return {k: _unwrap(v) for k, v in a.items()}
    return a


def _lift(fn):
    @wraps(fn)
    def method(self, *args, **kwargs):
        with torch.enable_grad():
            out = fn(_unwrap(self), *_unwrap(args), **_unwrap(kwargs))
        return _wrap(out)
    return method


def _reflected(op):
    def method(self, other):
        with torch.enable_grad():
            out = op(_unwrap(other), _unwrap(self))
        return _wrap(out)
    return method


def _dyn_class(base):
    """(TrackingTensor, base): still ``isinstance(p, pp.LieTensor)``; base's own methods run on the untracked value with
    grad mode on, reflected operators take precedence over the untracked left operand's (Python's subclass rule)."""
    klass = _dyn.get(base)
    if klass is not None:
        return klass
    body = {}
    own = {}
    for k in reversed(base.__mro__):
        if k is torch.Tensor or not issubclass(k, torch.Tensor) or k is object:
            continue
        own.update(k.__dict__)
    keep = {"__new__", "__init__", "__repr__", "__str__", "__torch_function__", "__deepcopy__", "__reduce_ex__", "__format__",
            "__class__", "__dict__", "__getattr__", "__setattr__", "new_empty", "__init_subclass__"}
    for name, attr in own.items():
        if name in keep or not callable(attr) or isinstance(attr, (staticmethod, classmethod, property, type)):
            continue
        if name.endswith("_") and not name.endswith("__"):
            continue                                   # in-place updates act on the parameter itself
        body[name] = _lift(attr)
    def tensor(self):                                  # leaves the tracked world (same graph)
        with torch.enable_grad():
            return torch.Tensor.as_subclass(self, torch.Tensor)
    body["tensor"] = tensor
    for nm, op in (("mul", operator.mul), ("matmul", operator.matmul), ("add", operator.add), ("sub", operator.sub),
                   ("truediv", operator.truediv)):
        body["__r%s__" % nm] = _reflected(op)
    klass = _dyn[base] = type("Tracking" + base.__name__, (TrackingTensor, base), body)
    return klass


class TrackingTensor(torch.Tensor):
    @staticmethod
    def __new__(cls, data):
        if isinstance(data, TrackingTensor):
            return data
        data = data if isinstance(data, torch.Tensor) else torch.as_tensor(data)
        base = type(data)
        klass = cls if base in (torch.Tensor, torch.nn.Parameter) else _dyn_class(base)
        t = torch.Tensor._make_subclass(klass, torch.Tensor.as_subclass(data, torch.Tensor).detach(), data.requires_grad)
        if hasattr(data, "ltype"):
            t.ltype = data.ltype
        return t

    def __init__(self, *a, **k):
        pass

    def tensor(self):
        with torch.enable_grad():
            return torch.Tensor.as_subclass(self, torch.Tensor)

    def _base(self):
        for klass in type(self).__mro__:
            if issubclass(klass, torch.Tensor) and not issubclass(klass, TrackingTensor):
                return klass
        return torch.Tensor

    def _payload(self):
        """the same storage and autograd history seen as the base class (a LieTensor keeps its ltype)"""
        with torch.enable_grad(), torch._C.DisableTorchFunctionSubclass():
            out = torch.Tensor.as_subclass(self, self._base())
        if "ltype" in self.__dict__:
            out.ltype = self.__dict__["ltype"]
        return out

    def _is_root(self):
        return getattr(self, "_is_param", False) or (self.is_leaf and "_bae_root" not in self.__dict__)

    def _rewrap(self, value):
        if not isinstance(value, torch.Tensor) or isinstance(value, TrackingTensor):
            return value
        out = torch.Tensor.as_subclass(value, type(self))
        if "ltype" in self.__dict__:
            out.ltype = self.__dict__["ltype"]
        return out

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        if name in ("__get__", "__set__", "__delete__") or name in _INFO or name.startswith("is_") \
                or (getattr(func, "__module__", None) or "").startswith("torch.autograd"):        # attribute access (requires_grad, grad, data, shape ...)
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        tracked = [a for a in args if isinstance(a, TrackingTensor)]
        # ---- metadata / in-place ops act on the parameter object itself ---------------------------------------------
        if tracked and (name in _PRESERVE or (name.endswith("_") and not name.endswith("__"))):
            with torch._C.DisableTorchFunctionSubclass():
                out = func(*args, **kwargs)
            if isinstance(out, torch.Tensor) and not isinstance(out, TrackingTensor):
                out = tracked[0]._rewrap(out)
            return out
        # ---- leaving the tracked world -----------------------------------------------------------------------------
        if name == "as_subclass" and len(args) > 1 and args[1] is torch.Tensor:
            with torch.enable_grad(), torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        # ---- row gathers -----------------------------------------------------------------------------------------
        if name == "__getitem__" and tracked and args[0] is tracked[0] and isinstance(args[1], torch.Tensor) \
                and args[1].dtype in (torch.int64, torch.int32) and args[0].dim() >= 2 and args[1].dim() == 1:
            src = args[0]
            root = src.__dict__.get("_bae_root", src if src._is_root() else None)
            if root is not None:
                with torch.enable_grad():
                    out = src._payload()[_unwrap(args[1])]
                rows = src.__dict__.get("_bae_rows")
                idx = args[1].reshape(-1).long()
                idx = torch.where(idx < 0, idx + src.shape[0], idx)
                _record(root, idx if rows is None else rows[idx], out)
                return _wrap(out)
        if func is torch.cat and isinstance(args[0], (tuple, list)) and kwargs.get("dim", args[1] if len(args) > 1 else 0) == 0:
            parts = list(args[0])
            roots = [p for p in parts if isinstance(p, TrackingTensor) and p._is_root()]
            if len(roots) == 1 and all(p is roots[0] or not isinstance(p, TrackingTensor) for p in parts):
                with torch.enable_grad():
                    out = torch.cat([_unwrap(p) for p in parts], 0)
                rows = [torch.arange(p.shape[0], device=out.device) if p is roots[0]
                        else torch.full((p.shape[0],), -1, dtype=torch.int64, device=out.device) for p in parts]
                res = _wrap(out)
                res.__dict__["_bae_root"], res.__dict__["_bae_rows"] = roots[0], torch.cat(rows)
                return res
        # ---- everything else: ordinary arithmetic on the underlying values, with grad mode on ----------------------
        def take(a):
            if isinstance(a, TrackingTensor):
                view = a._payload()
                if a._is_root() and a.requires_grad and view.dim() >= 1:
                    _record(a, None, view)              # used whole: gathered with index = arange
                return view
            if isinstance(a, (tuple, list)):
                return type(a)(take(x) for x in a)
            return a
        with torch.enable_grad():
            out = func(*[take(a) for a in args], **{k: take(v) for k, v in kwargs.items()})
        return _wrap(out)


def map_transform(function):
    """@psjac: the batched function's rows are independent -- which is what ``jacobian`` assumes of everything between a
    gather and the residual.  The function itself is left as it is (the reference: "doesn't change the function behavior")."""
    @wraps(function)
    def wrapped(*args, **kwargs):
        return function(*args, **kwargs)
    return wrapped
