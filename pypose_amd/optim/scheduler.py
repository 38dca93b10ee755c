"""Stopping logic around an optimizer (API of pypose/optim/scheduler.py).

``StopOnPlateau`` reads ``optimizer.last`` / ``optimizer.loss`` / ``optimizer.reject_count`` after every step and
decides whether another step is worth taking; ``optimize()`` is the loop ``while continual(): step``.
"""
import torch


class _ContinualFlag:
    """What ``scheduler.continual`` is: calling it asks the scheduler; testing the attribute itself for truth -- the
    pre-0.4 spelling -- is refused with the reference's message (scheduler.py:6-25)."""

    def __init__(self, ask):
        self.ask = ask

    def __call__(self, *args, **kwargs):
        return self.ask(*args, **kwargs)

    def __bool__(self):
        raise RuntimeError('Calling scheduler.continual is deprecated, '
                           'please call scheduler.continual() instead. '
                           'This error msg will be removed in a future release.')


class _Scheduler(object):
    def __init__(self, optimizer, max_steps, verbose=False):
        from .optimizer import _Optimizer
        if not isinstance(optimizer, _Optimizer):
            raise TypeError('{} is not an Optimizer'.format(type(optimizer).__name__))
        self.optimizer, self.verbose = optimizer, verbose
        self.max_steps, self.steps = max_steps, 0
        self._continual = True
        self.continual = _ContinualFlag(self.iscontinual)

    def iscontinual(self):
        return self._continual

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k not in ('optimizer', 'continual')}

    def load_state_dict(self, state_dict):
        self.__dict__.update(state_dict)

    def _quit(self, why):
        self._continual = False
        if self.verbose:
            print("{}: {}, Quitting..".format(type(self).__name__, why))


class StopOnPlateau(_Scheduler):
    """Stops after ``steps`` steps, after ``patience`` consecutive steps that lowered the loss by less than
    ``decreasing``, or as soon as the optimizer had to reject a step (scheduler.py:87-203)."""

    def __init__(self, optimizer, steps, patience=5, decreasing=1e-3, verbose=False):
        super().__init__(optimizer, steps, verbose)
        self.decreasing = decreasing
        self.patience, self.patience_count = patience, 0

    def step(self, loss):
        opt = self.optimizer
        assert opt.loss is not None, 'scheduler.step() should be called after optimizer.step()'
        self.steps += 1
        gained = opt.last - opt.loss
        if self.verbose:
            print('StopOnPlateau on step {} Loss {:.6e} --> Loss {:.6e} (reduction/loss: {:.4e}).'.format(
                self.steps, opt.last, opt.loss, gained / (opt.last + 1e-31)))
        self.patience_count = self.patience_count + 1 if gained < self.decreasing else 0
        if self.steps >= self.max_steps:
            self._quit("Maximum steps reached")
        if self.patience_count >= self.patience:
            self._quit("Maximum patience steps reached")
        if getattr(opt, 'reject_count', 0) > 0:
            self._quit("Maximum rejected steps reached")

    @torch.no_grad()
    def optimize(self, input, target=None, weight=None):
        while self.continual():
            self.step(self.optimizer.step(input, target, weight))
