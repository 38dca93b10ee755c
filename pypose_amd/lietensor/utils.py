"""Constructors and free-function aliases of the LieTensor API
(reference pypose/lietensor/utils.py: aliases :45-200, randn_* :226-915, identity_* :918-1342,
Exp/Log/Inv/... :1355-2660 -- all one-line forwards)."""
import functools

from .lietensor import (LieTensor, SE3_type, SO3_type, RxSO3_type, Sim3_type, rxso3_type, se3_type, sim3_type,
                        so3_type)

SO3 = functools.partial(LieTensor, ltype=SO3_type)
so3 = functools.partial(LieTensor, ltype=so3_type)
SE3 = functools.partial(LieTensor, ltype=SE3_type)
se3 = functools.partial(LieTensor, ltype=se3_type)
Sim3 = functools.partial(LieTensor, ltype=Sim3_type)
sim3 = functools.partial(LieTensor, ltype=sim3_type)
RxSO3 = functools.partial(LieTensor, ltype=RxSO3_type)
rxso3 = functools.partial(LieTensor, ltype=rxso3_type)


def _randn(ltype):
    def f(*lsize, sigma=1.0, **kwargs):
        return ltype.randn(*lsize, sigma=sigma, **kwargs)
    f.__doc__ = f"Random {type(ltype).__name__} LieTensor of lshape ``lsize`` (see reference randn_*)."
    return f


def _identity(ltype):
    def f(*lsize, **kwargs):
        return ltype.identity(*lsize, **kwargs)
    f.__doc__ = f"Identity {type(ltype).__name__} LieTensor of lshape ``lsize``."
    return f


randn_SO3, randn_so3 = _randn(SO3_type), _randn(so3_type)
randn_SE3, randn_se3 = _randn(SE3_type), _randn(se3_type)
randn_Sim3, randn_sim3 = _randn(Sim3_type), _randn(sim3_type)
randn_RxSO3, randn_rxso3 = _randn(RxSO3_type), _randn(rxso3_type)
identity_SO3, identity_so3 = _identity(SO3_type), _identity(so3_type)
identity_SE3, identity_se3 = _identity(SE3_type), _identity(se3_type)
identity_Sim3, identity_sim3 = _identity(Sim3_type), _identity(sim3_type)
identity_RxSO3, identity_rxso3 = _identity(RxSO3_type), _identity(rxso3_type)


def randn_like(input, sigma=1.0, **kwargs):
    return input.ltype.randn_like(*input.lshape, sigma=sigma, **kwargs)


def identity_like(liegroup, **kwargs):
    return liegroup.ltype.identity_like(*liegroup.lshape, **kwargs)


def _needs_lietensor(fn):
    """The free functions refuse a first argument that is not a LieTensor (reference utils.py:1345-1351)."""
    @functools.wraps(fn)
    def checked(*args, **kwargs):
        assert isinstance(args[0], LieTensor), "Invalid LieTensor Type."
        return fn(*args, **kwargs)
    return checked


@_needs_lietensor
def Exp(input):
    return input.Exp()


@_needs_lietensor
def Log(input):
    return input.Log()


@_needs_lietensor
def Inv(input):
    return input.Inv()


@_needs_lietensor
def Mul(input, other):
    return input * other


@_needs_lietensor
def Retr(X, a):
    return X.Retr(a)


@_needs_lietensor
def Act(X, p):
    return X.Act(p)


@_needs_lietensor
def Adj(input, p):
    return input.Adj(p)


@_needs_lietensor
def AdjT(input, p):
    return input.AdjT(p)


@_needs_lietensor
def Jinvp(input, p):
    return input.Jinvp(p)


@_needs_lietensor
def Jr(input):
    return input.Jr()


def tensor(input):
    return input.tensor()


def translation(input):
    return input.translation()


def rotation(input):
    return input.rotation()


def scale(input):
    return input.scale()


def matrix(input):
    return input.matrix()


def euler(input, eps=2e-4):
    return input.euler(eps=eps)
