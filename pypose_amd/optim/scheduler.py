"""Stopping logic around an optimizer (reference pypose/optim/scheduler.py)."""
import torch


class _Scheduler(object):
    class Continual:
        """``scheduler.continual()`` -> bool; using it as a bare bool raises (reference :6-25)."""

        def __init__(self, owner):
            self.owner = owner

        def __call__(self, *args, **kwargs):
            return self.owner.iscontinual(*args, **kwargs)

        def __bool__(self):
            raise RuntimeError('Calling scheduler.continual is deprecated, '
                               'please call scheduler.continual() instead. '
                               'This error msg will be removed in a future release.')

    def __init__(self, optimizer, max_steps, verbose=False):
        from .optimizer import _Optimizer
        if not isinstance(optimizer, _Optimizer):
            raise TypeError('{} is not an Optimizer'.format(type(optimizer).__name__))
        self.optimizer, self.verbose = optimizer, verbose
        self.max_steps, self.steps = max_steps, 0
        self.continual = self.Continual(self)
        self._continual = True

    def iscontinual(self):
        return self._continual

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != 'optimizer'}

    def load_state_dict(self, state_dict):
        self.__dict__.update(state_dict)


class StopOnPlateau(_Scheduler):
    """Stop on max steps, on ``patience`` steps with loss decrease < ``decreasing``, or when the
    optimizer exhausted its rejected-step budget (reference scheduler.py:64-203)."""

    def __init__(self, optimizer, steps, patience=5, decreasing=1e-3, verbose=False):
        super().__init__(optimizer, steps, verbose)
        self.decreasing = decreasing
        self.patience, self.patience_count = patience, 0

    def step(self, loss):
        opt = self.optimizer
        assert opt.loss is not None, 'scheduler.step() should be called after optimizer.step()'
        self.steps += 1
        if self.verbose:
            print('StopOnPlateau on step {} Loss {:.6e} --> Loss {:.6e} (reduction/loss: {:.4e}).'.format(
                self.steps, opt.last, opt.loss, (opt.last - opt.loss) / (opt.last + 1e-31)))
        if self.steps >= self.max_steps:
            self._continual = False
            if self.verbose:
                print("StopOnPlateau: Maximum steps reached, Quitting..")
        self.patience_count = self.patience_count + 1 if (opt.last - opt.loss) < self.decreasing else 0
        if self.patience_count >= self.patience:
            self._continual = False
            if self.verbose:
                print("StopOnPlateau: Maximum patience steps reached, Quitting..")
        if getattr(opt, 'reject_count', 0) > 0:
            self._continual = False
            if self.verbose:
                print("StopOnPlateau: Maximum rejected steps reached, Quitting..")

    @torch.no_grad()
    def optimize(self, input, target=None, weight=None):
        while self.continual():
            loss = self.optimizer.step(input, target, weight)
            self.step(loss)
