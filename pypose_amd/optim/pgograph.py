"""One Levenberg-Marquardt trial of the recognised pose-graph program as ONE hipGraph replay.

A 10k-pose LM step keeps the GPU busy for ~0.7 ms (0.54 ms of it the persistent PCG solve) but took 0.88 ms: between the
read-back that ends a trial and the launch of the next solve the GPU waited for ~0.2 ms of Python (program checks, a dozen
allocations, eight ctypes launches, LieTensor dispatch).  Everything between two read-backs is the same sequence of launches
on buffers that can stay where they are:

    linearise (residual + blocks per edge)  ->  assemble (block diagonal, gradient, off-diagonal blocks)  ->
    prepare (clamp + damp + block inverses; the damping factor is read from a DEVICE scalar)  ->  persistent PCG  ->
    retract the parameters  ->  loss at the candidate  ->  gain-ratio terms  ->  [gain, loss, solver info] in one vector

so it is captured once (torch.cuda.CUDAGraph over the very same Python code path the un-captured step runs) and a step
becomes: write the damping factor, replay, read the vector back, run the strategy / accept test on the host.  A rejected
trial (rare) continues in the ordinary trial loop on the captured linearisation.  The capture is used only while the
program, its operands, the weight, the parameter storage and the solver settings are what it was captured on: by default
the model's Python is run (dry, no launches) at the top of every step and the program re-matched from that trace
(fused.checked_shortcut); LM(static=True) takes the program on trust.
"""
from __future__ import annotations

import torch

from .. import _C
from . import fused as _fused


class PgoGraphStep:
    MIN_STREAK = 3          # ordinary steps on the same program before the capture is made

    def __init__(self, opt, pg, prog, input, weight, P, trivial):
        from .posegraph import PCG
        self.opt, self.prog, self.input, self.weight, self.P, self.trivial = opt, prog, input, weight, P, trivial
        self.weight_version = weight._version if isinstance(weight, torch.Tensor) else None
        self.ptr = P.data_ptr()
        self.clamp = (pg['min'], pg['max'])
        solver = opt.solver
        assert isinstance(solver, PCG)
        self.solver_key = (solver, solver.tol, solver.maxiter, solver.check_every)
        dev = P.device
        self.s_dev = torch.ones(1, dtype=torch.float64, device=dev)
        self.params = [p for p in pg['params'] if p.requires_grad]
        self.graph = None
        self.backup = torch.empty_like(torch.Tensor.as_subclass(P, torch.Tensor).detach())   # the parameters before the latest replay
        # capture on a side stream (torch.cuda.graph does that); the first replay-equivalent run happens during capture
        torch.cuda.synchronize(dev)
        saved = P.detach().clone()
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), _C.graph_capture(g):      # (no cyclic-GC pass inside the capture: see _C.graph_capture)
            self._trial(pg)
        self.graph = g
        with torch.no_grad():           # capture does not execute: nothing moved, but be explicit about the state we hand back
            torch.Tensor.as_subclass(P, torch.Tensor).detach().copy_(torch.Tensor.as_subclass(saved, torch.Tensor))

    def _trial(self, pg):
        opt = self.opt
        # (first node of the captured trial: a replay launched speculatively -- before this step's run of the model has
        #  confirmed the program, see fused.checked_shortcut -- is undone by copying this back)
        self.backup.copy_(torch.Tensor.as_subclass(self.P, torch.Tensor).detach())
        lin = _fused._pgo_linearization(opt, self.prog, self.weight, self.P, self.trivial)
        lin.build_normal_equations(*self.clamp)
        lin.s_dev = self.s_dev                     # prepare reads the damping factor from here (pplie_pcg_prepare_dev)
        opt._defer_solver_info = True
        try:
            D = lin.solve(opt.solver)
        finally:
            opt._defer_solver_info = False
        lin.s_dev = None                           # retries after a rejection run un-captured, with the host value
        opt.update_parameter(pg['params'], D)
        loss = lin.fast_loss()
        J, R = lin.strategy_args()
        ab = J.gain_terms(D)
        pend, lin.pending_info = lin.pending_info, None
        self.out = torch.cat([ab.reshape(-1).double(), loss.detach().reshape(1).double(), pend.info.double()])
        self.lin, self.D, self.loss, self.J, self.R = lin, D, loss, J, R

    # -- per step ------------------------------------------------------------------------------------
    def usable(self, pg, input, target, weight, checked=False):
        """``checked``: the caller has just matched this step's dry run of the model to ``self.prog`` (the default);
        otherwise (LM(static=True)) the program is taken on trust while the same input objects / operand storage are passed."""
        opt, P = self.opt, self.P
        cache = opt.__dict__.get('_structure_cache')
        if cache is None or cache.get("fused") is not True or target is not None:
            return False
        hit = cache.get("program")
        if hit is None or hit[3] is not self.prog or hit[1] is not P or not (checked or _fused._same_input(hit[0], input)):
            return False
        if weight is not self.weight or (isinstance(weight, torch.Tensor) and weight._version != self.weight_version):
            return False
        solver = opt.solver
        if (solver, solver.tol, solver.maxiter, solver.check_every) != self.solver_key or (pg['min'], pg['max']) != self.clamp:
            return False
        if P.data_ptr() != self.ptr or [p for p in pg['params'] if p.requires_grad] != self.params:
            return False
        if _fused._strategy_kind(opt.strategy) is None or opt.group is not None or len(opt.param_groups) != 1:
            return False
        from .optimizer import Trivial
        if (all(isinstance(c, Trivial) for c in opt.corrector) and all(isinstance(k, Trivial) for k in opt.model.kernel)) != self.trivial \
                or len(opt.corrector) != 1:
            return False
        from .optimizer import _REPROBE
        uses = cache['_uses'] = cache.get('_uses', 0) + 1
        if uses % _REPROBE == 0:                   # the ordinary path's periodic re-probing keeps its rhythm
            cache['_uses'] = uses - 1
            return False
        if not checked and not all(t.data_ptr() == ptr and t._version == ver for t, ptr, ver in hit[4]):
            return False                           # operands changed: the ordinary path re-derives the program
        return True

    def step(self, pg):
        self.launch(pg)
        return self.finish(pg)

    def launch(self, pg):
        """enqueue the trial (everything up to the read-back); the host returns while the GPU works"""
        opt = self.opt
        self._had_loss = hasattr(opt, 'loss')
        if not self._had_loss:                     # first step of a run: the loss at the starting point (optimizer.py:659)
            opt.loss = self.lin.fast_loss()
        self._prev_last = opt.__dict__.get('_last_view')
        opt.last = opt.loss
        self._last_h = opt._host(opt.loss)
        opt.reject_count = 0
        lin = self.lin
        lin.s = 1.0 + float(pg['damping'])         # (host mirror: a retry compounds from here)
        self.s_dev.fill_(lin.s)
        self.graph.replay()
        _C.mark_written(self.P)

    def cancel(self):
        """undo a speculative launch: wait for it, put the parameters back (nothing else it wrote is state)"""
        torch.cuda.synchronize(self.backup.device)
        with torch.no_grad():
            torch.Tensor.as_subclass(self.P, torch.Tensor).detach().copy_(self.backup)
        _C.mark_written(self.P)
        opt = self.opt
        if not self._had_loss:                     # (that starting loss was the CAPTURED program's: the ordinary path computes its own)
            del opt.loss
        opt.__dict__.pop('_host_loss', None)
        if self._prev_last is not None:
            opt.last = self._prev_last
        else:
            opt.__dict__.pop('_last_view', None)

    def finish(self, pg):
        opt, lin, last_h = self.opt, self.lin, self._last_h
        a, b, loss_h, its, rr, bn2, flag = self.out.tolist()          # the trial's one synchronisation
        opt.linearization = lin.kind
        opt._last_replicated = False
        if flag >= 2.0 or rr != rr:                # a failed solve returned a zero step: the parameters are where they were
            if flag == 3.0:                        # the persistent launch's workgroups were not all resident: drop the capture,
                from .posegraph import SolveFailed, _check_persist_flag      # take this step on the two-launch iteration
                try:
                    _check_persist_flag(flag, rr)
                except SolveFailed:
                    pass
                opt.__dict__.pop('_pgo_graph_step', None)
                opt.loss = opt.last
                return opt._step_general(self.input, None, self.weight)
            print('Linear solve produced NaN (matrix may not be positive-definite)', "\nLinear solver failed. Breaking optimization step...")
            opt.loss = opt.last
            return opt.loss
        opt.solver.iterations = int(its)
        x = max(a, 1e-300) ** 0.5
        one = torch.ones((1, 1), dtype=torch.float64)
        opt.strategy.update(pg, last=last_h, loss=loss_h, J=one, D=x * one, R=(b / x) * one)
        if last_h < loss_h and opt.reject_count < opt.reject:        # rejected: back, then the ordinary trial loop
            opt.update_parameter(params=pg['params'], step=-self.D)
            opt.loss, opt.reject_count, loss_h = opt.last, 1, last_h
            loss_h = opt._trial_loop(pg, lin, self.J, self.R, None, None, last_h, loss_h, defer=False)
        else:
            opt.loss = self.loss.clone()           # (the captured buffer is overwritten by the next replay)
        opt._host_loss = (opt.loss, loss_h)
        return opt.loss
