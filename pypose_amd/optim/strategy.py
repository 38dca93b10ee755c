"""Damping policies of Levenberg-Marquardt (API of pypose/optim/strategy.py).

``update(pg, last, loss, J, D, R)`` rewrites ``pg['damping']`` (TrustRegion also ``pg['radius']`` / ``pg['down']``).
``J`` only needs ``J @ D``: the structured paths of optim/optimizer.py pass a block operator (or, for graphs, an
equivalent 1 x 1 problem built from two device-side dot products), so the gain ratio never costs a dense
``[N_res, N_par]`` product.

Both adaptive policies grade a step by its gain ratio against two thresholds; what differs is what "good", "fair"
and "poor" do to the damping.
"""
GOOD, FAIR, POOR = 1, 0, -1


def gain_ratio(last, loss, J, D, R):
    """Actual decrease of the loss over the decrease the linear model predicted, ``-(J D)^T (2 R + J D)``
    (strategy.py:144, :261) -- as a dot product: the reference's [1,n] @ [n,1] matmul is the same number, but lands
    on a GEMM kernel that takes milliseconds at n ~ 10^6."""
    JD = J @ D
    return (last - loss) / -(JD * (2 * R + JD)).sum()


def _positive(value, text):
    assert value > 0, ValueError(text.format(value))


def _check_thresholds(high, low, up, down):
    _positive(high, "high has to be positive: {}")
    _positive(low, "low for decrease has to be positive: {}")
    assert 0 < down < 1, ValueError("down factor has to be smaller than 1: {}".format(down))
    assert 1 < up, ValueError("up factor has to be larger than 1: {}".format(up))


def _bounded(x, lo, hi):
    return lo if x < lo else (hi if x > hi else x)


def _grade(pg, quality):
    return GOOD if quality > pg['high'] else (FAIR if quality > pg['low'] else POOR)


def update_from_terms(strategy, pg, last, loss, a, b):
    """``strategy.update`` from the two dot products ``a = (J D).(J D)``, ``b = (J D).R`` a fused linearisation reads back with
    the loss: the equivalent 1 x 1 problem ``x^2 = a``, ``x r = b``.  The policies of this module take it as host floats (the same
    IEEE double operations the 1 x 1 tensors went through, without seven tensor ops on the step's critical path); any other
    strategy object -- the reference's own, a user's -- gets the 1 x 1 tensors its ``update`` expects."""
    x = max(a, 1e-300) ** 0.5
    if type(strategy) in (Constant, Adaptive, TrustRegion):
        r = b / x
        strategy.update_quality(pg, (last - loss) / -(x * (2 * r + x)))
        return
    import torch
    one = torch.ones((1, 1), dtype=torch.float64)
    strategy.update(pg, last=last, loss=loss, J=one, D=x * one, R=(b / x) * one)


class Constant(object):
    """The damping never changes (strategy.py:41-46)."""

    def __init__(self, damping=1e-6):
        _positive(damping, "damping has to be positive: {}")
        self.defaults = {'damping': damping}

    def update(self, pg, *args, **kwargs):
        pg['damping'] = pg['damping']

    def update_quality(self, pg, quality):
        pg['damping'] = pg['damping']


class Adaptive(object):
    """Good step: damping * down; fair: unchanged; poor: damping * up; kept inside [min, max] (strategy.py:134-151)."""

    def __init__(self, damping=1e-6, high=0.5, low=1e-3, up=2., down=.5, min=1e-6, max=1e16):
        _positive(damping, "damping has to be positive: {}")
        _check_thresholds(high, low, up, down)
        self.defaults = {'damping': damping, 'high': high, 'low': low, 'up': up, 'down': down}
        self.min, self.max = min, max

    def update(self, pg, last, loss, J, D, R, *args, **kwargs):
        self.update_quality(pg, gain_ratio(last, loss, J, D, R))

    def update_quality(self, pg, quality):
        factor = {GOOD: pg['down'], FAIR: 1, POOR: pg['up']}[_grade(pg, quality)]
        pg['damping'] = _bounded(pg['damping'] * factor, self.min, self.max)


class TrustRegion(object):
    """Damping = 1 / radius.  Good step: radius * up and the shrink factor resets; fair: only the reset; poor: radius
    * down and the shrink factor itself shrinks by ``factor``, so repeated failures contract the region ever faster
    (strategy.py:248-274)."""

    def __init__(self, radius=1e6, high=.5, low=1e-3, up=2., down=.5, factor=.5, min=1e-6, max=1e16):
        _positive(radius, "trust region radius has to be positive: {}")
        _check_thresholds(high, low, up, down)
        assert 0 < factor < 1, ValueError("factor has to be smaller than 1: {}".format(factor))
        self.min, self.max, self.down = min, max, down
        self.defaults = {'radius': radius, 'damping': 1 / radius, 'high': high, 'low': low,
                         'up': up, 'down': down, 'factor': factor}

    def update(self, pg, last, loss, J, D, R, *args, **kwargs):
        self.update_quality(pg, gain_ratio(last, loss, J, D, R))

    def update_quality(self, pg, quality):
        grade = _grade(pg, quality)
        radius = 1. / pg['damping']
        if grade == POOR:
            radius, shrink = radius * pg['down'], pg['down'] * pg['factor']
        else:
            radius, shrink = (radius * pg['up'] if grade == GOOD else radius), self.down
        pg['down'] = _bounded(shrink, self.min, self.max)
        pg['radius'] = _bounded(radius, self.min, self.max)
        pg['damping'] = 1. / pg['radius']
