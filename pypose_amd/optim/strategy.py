"""Damping policies of Levenberg-Marquardt (reference pypose/optim/strategy.py).

``update(pg, last, loss, J, D, R)`` mutates ``pg['damping']`` (and, for TrustRegion,
``pg['radius']`` / ``pg['down']``).  ``J`` only needs to support ``J @ D``: the structured fast
paths of :mod:`pypose_amd.optim.optimizer` pass a block operator instead of a dense matrix, so
the gain ratio costs one block mat-vec + two dot products instead of a dense ``[N_res, N_par]``
product.
"""
import torch


def _gain_ratio(last, loss, J, D, R):
    """(actual decrease) / (decrease predicted by the linear model), reference strategy.py:144/261."""
    JD = J @ D
    # (JD)^T (2R + JD) as a dot product: the reference's [1,n] @ [n,1] matmul is the same number but
    # lands on a GEMM kernel that takes milliseconds at n ~ 10^6
    return (last - loss) / -(JD * (2 * R + JD)).sum()


def _clip(x, lo, hi):
    return max(lo, min(x, hi))


class Constant(object):
    """Fixed damping (reference strategy.py:5-46)."""

    def __init__(self, damping=1e-6):
        assert damping > 0, ValueError("damping has to be positive: {}".format(damping))
        self.defaults = {'damping': damping}

    def update(self, pg, *args, **kwargs):
        pg['damping'] = pg['damping']


class Adaptive(object):
    """Damping scaled down / kept / up by the gain ratio (reference strategy.py:49-151)."""

    def __init__(self, damping=1e-6, high=0.5, low=1e-3, up=2., down=.5, min=1e-6, max=1e16):
        assert damping > 0, ValueError("damping has to be positive: {}".format(damping))
        assert high > 0, ValueError("high has to be positive: {}".format(high))
        assert low > 0, ValueError("low for decrease has to be positive: {}".format(low))
        assert 0 < down < 1, ValueError("down factor has to be smaller than 1: {}".format(down))
        assert 1 < up, ValueError("up factor has to be larger than 1: {}".format(up))
        self.defaults = {'damping': damping, 'high': high, 'low': low, 'up': up, 'down': down}
        self.min, self.max = min, max

    def update(self, pg, last, loss, J, D, R, *args, **kwargs):
        quality = _gain_ratio(last, loss, J, D, R)
        if quality > pg['high']:
            pg['damping'] = pg['damping'] * pg['down']
        elif quality > pg['low']:
            pg['damping'] = pg['damping']
        else:
            pg['damping'] = pg['damping'] * pg['up']
        pg['damping'] = _clip(pg['damping'], self.min, self.max)


class TrustRegion(object):
    """Trust-region radius policy, damping = 1/radius (reference strategy.py:154-274)."""

    def __init__(self, radius=1e6, high=.5, low=1e-3, up=2., down=.5, factor=.5, min=1e-6, max=1e16):
        assert radius > 0, ValueError("trust region radius has to be positive: {}".format(radius))
        assert high > 0, ValueError("high has to be positive: {}".format(high))
        assert low > 0, ValueError("low for decrease has to be positive: {}".format(low))
        assert 0 < down < 1, ValueError("down factor has to be smaller than 1: {}".format(down))
        assert 1 < up, ValueError("up factor has to be larger than 1: {}".format(up))
        assert 0 < factor < 1, ValueError("factor has to be smaller than 1: {}".format(factor))
        self.min, self.max, self.down = min, max, down
        self.defaults = {'radius': radius, 'damping': 1 / radius, 'high': high, 'low': low,
                         'up': up, 'down': down, 'factor': factor}

    def update(self, pg, last, loss, J, D, R, *args, **kwargs):
        quality = _gain_ratio(last, loss, J, D, R)
        pg['radius'] = 1. / pg['damping']
        if quality > pg['high']:
            pg['radius'], pg['down'] = pg['up'] * pg['radius'], self.down
        elif quality > pg['low']:
            pg['down'] = self.down
        else:
            pg['radius'], pg['down'] = pg['radius'] * pg['down'], pg['down'] * pg['factor']
        pg['down'] = _clip(pg['down'], self.min, self.max)
        pg['radius'] = _clip(pg['radius'], self.min, self.max)
        pg['damping'] = 1. / pg['radius']
