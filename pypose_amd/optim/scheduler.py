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
        """``while continual(): step`` (scheduler.py:162-203).  With ``verbose=False`` and an optimizer whose step is the
        device-resident one (optim/fused.py DeviceLM: loop state, accept / reject and the loss stay on the GPU) the stop rules are
        evaluated ON THE DEVICE as well: the remaining steps are enqueued without reading a loss back between them, the step that
        raises the stop flag is the last one that does anything (later launches return at once), and the counters come back in one
        read at the end -- the host loop costs a synchronisation per step, more than the step's two kernels."""
        while self.continual():
            if not self.verbose and self._optimize_on_device(input, target, weight):
                continue
            self.step(self.optimizer.step(input, target, weight))

    def _optimize_on_device(self, input, target, weight):
        """Run the rest of the steps with the stop rules on the device; False (nothing done) when that route does not apply now."""
        opt = self.optimizer
        dev = opt.__dict__.get('_device_lm')
        if dev is None or not hasattr(dev, 'set_plateau') or getattr(opt, 'structure', None) == "strict" or self.max_steps - self.steps <= 0:
            return False
        if not dev.set_plateau(self.decreasing, self.patience, self.max_steps, self.steps, self.patience_count):
            return False
        done = 0
        try:
            for _ in range(self.max_steps - self.steps):
                if dev.plateau_stopped():                # (pinned memory, no synchronisation: the GPU is at most a few steps behind)
                    break
                before = dev.cur
                opt.step(input, target, weight)
                if opt.__dict__.get('_device_lm') is not dev or dev.cur == before:
                    # this step did not take the device route (the program changed, a hook appeared ...): it ran on the general path
                    # with the host's own bookkeeping -- account for it like the plain loop and leave the rest to that loop
                    dev.clear_plateau()
                    steps, count, stopped = self._read_back(dev, done, with_stop=True)
                    self.steps, self.patience_count = steps, count
                    if stopped:                      # (the device had stopped the run before this step)
                        self._quit("Maximum patience steps reached" if count >= self.patience else "Maximum rejected steps reached")
                    else:
                        self.step(opt.loss)
                    return True
                done += 1
        finally:
            dev.clear_plateau()
        steps, count, stopped = self._read_back(dev, done, with_stop=True)
        self.steps, self.patience_count = steps, count
        if self.steps >= self.max_steps:
            self._quit("Maximum steps reached")
        elif stopped:
            self._quit("Maximum patience steps reached" if self.patience_count >= self.patience else "Maximum rejected steps reached")
        return True

    def _read_back(self, dev, enqueued, with_stop=False):
        """the device's counters after ``enqueued`` steps of this call (one synchronising read)"""
        if enqueued == 0:
            return (self.steps, self.patience_count, False) if with_stop else (self.steps, self.patience_count)
        dev.pending = True
        dev.flush()
        steps, count, stopped = dev.plateau
        return (steps, count, stopped) if with_stop else (steps, count)
