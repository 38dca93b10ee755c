"""One Levenberg-Marquardt trial of the recognised pose-graph program as ONE hipGraph replay.

A 10k-pose LM step keeps the GPU busy for ~0.7 ms (0.54 ms of it the persistent PCG solve) but took 0.88 ms: between the
read-back that ends a trial and the launch of the next solve the GPU waited for ~0.2 ms of Python (program checks, a dozen
allocations, eight ctypes launches, LieTensor dispatch).  Everything between two read-backs is the same sequence of launches
on buffers that can stay where they are:

    linearise (residual + blocks per edge)  ->  assemble (block diagonal, gradient, off-diagonal blocks)  ->
    prepare (clamp + damp + block inverses; the damping factor is read from a DEVICE scalar)  ->  persistent PCG  ->
    retract the parameters  ->  loss at the candidate  ->  gain-ratio terms  ->  [gain, loss, solver info] in one vector

so it is captured once and a step becomes: write the damping factor, replay, wait for the verdict, run the strategy / accept
test on the host.  Host and device talk through two small blocks of HOST-PINNED memory instead of copies and fills: the
prepare kernel reads the damping factor from one (system-scope load), the last kernel of the trial stores
{gain terms, loss, solver info, sequence number} into the other (system-scope stores, the sequence number last) and the host
polls that word -- no stream synchronisation, no device-to-host copy, no fill kernel on the critical path between two trials
(measured with rocprofv3: the step had ~105 us of host round trips and 15 small tensor kernels behind a 145-500 us solve).
Everything after the solve is four launches of one C entry point (`pplie_pgo_trial_tail`: retract, candidate loss, gain
terms, pack).  A rejected
trial (rare) continues in the ordinary trial loop on the captured linearisation.  The capture is used only while the
program, its operands, the weight, the parameter storage and the solver settings are what it was captured on: by default
the model's Python is run (dry, no launches) at the top of every step and the program re-matched from that trace
(fused.checked_shortcut); LM(static=True) takes the program on trust.
"""
from __future__ import annotations

import ctypes

import torch

from .. import _C
from . import fused as _fused
from . import strategy as _strategy

_TAIL_SIG = [ctypes.c_void_p] * 11 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
_TAIL_AFTER_SIG = [ctypes.c_void_p] * 7 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
_PARTIALS = 1024          # PPLIE_PGO_PARTIALS


_OPT_NAMES = None


def _optimizer_names():
    """(Trivial, _REPROBE) of optim/optimizer.py, which imports this module: resolved once (an import statement inside the
    per-step check costs a microsecond or two each)"""
    global _OPT_NAMES
    if _OPT_NAMES is None:
        from .optimizer import Trivial, _REPROBE
        _OPT_NAMES = (Trivial, _REPROBE)
    return _OPT_NAMES


class TrialTail:
    """``pplie_pgo_trial_tail`` with its buffers: everything between the linear solve and the host's decision in four launches of one
    C entry point, the result block in HOST-PINNED memory that :meth:`wait` polls, the loss in a ring of device scalars.  Used by the
    captured trial below (the launches are part of the graph) and by the ordinary trial loop of the optimizer (graphs the
    persistent solve does not hold, and the steps before a capture exists)."""

    RING = 1024

    def __init__(self, dtype, dev):
        self.dtype, self.dev = dtype, dev
        self.out = torch.zeros(8, dtype=torch.float64).pin_memory()       # {a, b, loss, iterations, |r|^2, |b|^2, flag, seq}
        self.out_np = self.out.numpy()
        self.seq = 0
        self.after_solve = 0                                               # trials whose first half ran in the solve's epilogue
        self.state = torch.zeros(8, dtype=torch.int64, device=dev)         # {seq (counts executions), loss ring address, its length,
                                                                           #  retractions (counts moved parameters), the second launch's
                                                                           #  arrival ticket (zero at rest), three reserved}
        self.partial = torch.empty(3 * _PARTIALS, dtype=dtype, device=dev)
        self.no_info = torch.zeros(4, dtype=dtype, device=dev)             # (a solve that reported to the host already)
        self.new_ring()
        self.fn = _C.library().symbol("pplie_pgo_trial_tail" + ("_f32" if dtype == torch.float32 else "_f64"), _TAIL_SIG)

    def new_ring(self):
        """a fresh ring when this one has gone round, so a loss handed out earlier is never overwritten"""
        self.ring = torch.zeros(self.RING, dtype=self.dtype, device=self.dev)
        self.loss_views = self.ring.unbind(0)
        self.state[1:3].copy_(torch.tensor([self.ring.data_ptr(), self.RING], dtype=torch.int64))    # (once per RING trials)

    def enqueue(self, pt, backup, prog, lin, Dn, info):
        """the tail's launches (nothing else: callable inside a stream capture)"""
        assert Dn.is_contiguous() and pt.is_contiguous() and lin.m == 6 and lin.K == 2 and lin.dr == 6 \
            and lin.idx.data_ptr() == prog.idx.data_ptr() and lin.E == prog.E and Dn.shape == (lin.N, 6)
        ngain = lin.__dict__.pop('_tail_done', None)
        lin.__dict__.pop('_tail_in_solve', None)
        if ngain is not None:
            # the solve's epilogue has moved the parameters and left the gain partials: the candidate loss and the report remain
            fn_after = _C.library().symbol("pplie_pgo_trial_tail_after_solve" + ("_f32" if self.dtype == torch.float32 else "_f64"), _TAIL_AFTER_SIG)
            with _C._on_device(pt.device):
                code = fn_after(pt.data_ptr(), prog.idx.data_ptr(), prog.Z.data_ptr(), (self.no_info if info is None else info).data_ptr(),
                                     self.partial.data_ptr(), self.state.data_ptr(), self.out.data_ptr(), lin.E, int(ngain),
                                     _C.stream_ptr(pt.device))
            _C.check(code, "pplie_pgo_trial_tail_after_solve")
            self.after_solve += 1
            return
        with _C._on_device(pt.device):
            code = self.fn(pt.data_ptr(), None if backup is None else backup.data_ptr(), prog.idx.data_ptr(), prog.Z.data_ptr(),
                           lin.J.data_ptr(), lin.R.data_ptr(), Dn.data_ptr(), (self.no_info if info is None else info).data_ptr(),
                           self.partial.data_ptr(), self.state.data_ptr(), self.out.data_ptr(), lin.N, lin.E, _C.stream_ptr(pt.device))
        _C.check(code, "pplie_pgo_trial_tail")

    def advance(self):
        """host mirror of the execution the device is about to count; returns the slot of the loss ring it will write"""
        self.seq += 1
        slot = self.seq % self.RING
        if slot == 0:
            self.new_ring()
        return slot

    SPIN_SECONDS = 0.05          # busy polling (the usual trial answers in well under a millisecond) before yielding the GIL
    WAIT_SECONDS = 30.0          # a trial that has not reported by then is given up on

    def wait(self, stream=None):
        """the trial's verdict: poll the pinned word the last kernel stores.  Bounded by WALL TIME, not by a poll count: after
        SPIN_SECONDS the loop yields the GIL between polls and asks the stream whether it is still working (an asynchronous HIP
        error on the stream surfaces there as an exception instead of an endless wait); a stream that went idle without the
        word arriving, or WAIT_SECONDS, ends it."""
        import time
        out, seq = self.out_np, float(self.seq)
        if out[7] == seq:
            return out[:7].tolist()
        t0 = time.perf_counter()
        st = torch.cuda.current_stream(self.dev) if stream is None else stream
        while out[7] != seq:
            dt = time.perf_counter() - t0
            if dt < self.SPIN_SECONDS:
                continue
            time.sleep(0)                                    # (other Python threads -- the autograd engine's among them -- get to run)
            idle = st.query()                                # raises on a sticky error of the device
            if out[7] == seq:
                break
            if idle or dt > self.WAIT_SECONDS:
                torch.cuda.synchronize(self.dev)
                if out[7] != seq:
                    self.resync()
                    raise RuntimeError("pypose_amd: the pose-graph trial finished without reporting its result"
                                       if idle else "pypose_amd: the pose-graph trial did not report within "
                                       f"{self.WAIT_SECONDS:.0f} s")
        return out[:7].tolist()

    def resync(self):
        """after a failed launch / an exception: the host's execution count is read back from the device's (a read-back, so a
        synchronisation -- error paths only).  Returns how many executions the device counted beyond the host's mirror."""
        dev_seq, moved = (int(v) for v in self.state[[0, 3]].tolist())
        self.state[4] = 0                            # (the tail's arrival ticket: at rest, whatever the failed call left)
        ahead = dev_seq - self.seq
        self.seq = dev_seq
        self.unfinished = moved - dev_seq            # retractions whose trial never reported (0 or 1)
        return ahead


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
        pt = torch.Tensor.as_subclass(P, torch.Tensor).detach()
        # host-pinned scalar the solve's first launch reads the damping factor from; the trial's tail and its result block
        self.ctl = torch.ones(2, dtype=torch.float64).pin_memory()
        self.ctl_f = self.ctl.numpy()
        self.tt = TrialTail(pt.dtype, dev)
        self.params = [p for p in pg['params'] if p.requires_grad]
        self.graph = None
        self.pt = pt                            # (keeps the captured storage alive whatever the caller does to the parameter)
        self.keep = list((opt.__dict__.get('_pcg_workspaces') or {}).values())       # (and the solve's buffers)
        self.backup = torch.empty_like(pt)      # the parameters before the latest replay
        # (the pointer table of the solve's epilogue tail is a host-to-device copy: made before the capture, not inside it)
        for w in (opt.__dict__.get('_pcg_workspaces') or {}).values():
            if getattr(w, 'm', None) == 6 and getattr(w, 'dtype', None) == pt.dtype and hasattr(w, '_tail_args'):
                w._tail_args((pt, self.backup, self.tt.partial, self.tt.state))
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
        pt = torch.Tensor.as_subclass(self.P, torch.Tensor).detach()
        # (a replay launched speculatively -- before this step's run of the model has confirmed the program, see
        #  fused.checked_shortcut -- is undone by copying `backup` back: the retraction kernel, the only writer of the
        #  parameters in the trial, saves the rows it overwrites)
        lin = _fused._pgo_linearization(opt, self.prog, self.weight, self.P, self.trivial, s_dev=self.ctl)
        lin.build_normal_equations(*self.clamp)
        lin.s_dev = self.ctl                       # prepare reads the damping factor from here (pplie_pcg_prepare_dev)
        # the persistent solve may take the tail's first half (gain terms + retraction) into its epilogue: it is told where the
        # parameters, their backup and the tail's buffers are, and says so in lin._tail_done if it did
        lin._tail_in_solve = (pt, self.backup, self.tt.partial, self.tt.state)
        opt._defer_solver_info = 'inplace'
        try:
            Dn = lin._pcg(opt.solver, lin.s, lin.dmin, lin.dmax, plain=False)       # [N, 6]: the solver workspace's x itself
        finally:
            opt._defer_solver_info = False
        lin.s_dev = None                           # retries after a rejection run un-captured, with the host value
        pend, lin.pending_info = lin.pending_info, None
        assert pend is not None
        self.tt.enqueue(pt, self.backup, self.prog, lin, Dn, pend.info)
        self.lin, self.Dn = lin, Dn

    # -- per step ------------------------------------------------------------------------------------
    def usable(self, pg, input, target, weight, checked=False):
        """``checked``: the caller has just matched this step's dry run of the model to ``self.prog`` (the default);
        otherwise (LM(static=True)) the program is taken on trust while the same input objects / operand storage are passed."""
        with _fused._no_tf():      # (attribute reads on a LieTensor parameter are __torch_function__ round trips: ~3 us each)
            return self._usable(pg, input, target, weight, checked)

    def quick(self, target):
        """the facts a SPECULATIVE replay needs (fused.checked_shortcut launches first and asks ``usable`` while the GPU works): the
        captured kernels still write where the parameters are -- a replay into storage the caller has swapped out could not be undone
        by copying ``backup`` back -- and this is not the step the ordinary path's periodic re-probe is due.  Everything else the
        replay touches is kept alive by this object; anything else that turns out to have changed is undone by ``cancel``."""
        cache = self.opt.__dict__.get('_structure_cache')
        return (target is None and cache is not None and (cache.get('_uses', 0) + 1) % _optimizer_names()[1] != 0
                and self.P.data_ptr() == self.ptr)

    def _usable(self, pg, input, target, weight, checked):
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
        Trivial, _REPROBE = _optimizer_names()
        if (all(isinstance(c, Trivial) for c in opt.corrector) and all(isinstance(k, Trivial) for k in opt.model.kernel)) != self.trivial \
                or len(opt.corrector) != 1:
            return False
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
            with torch.no_grad():
                opt.loss = self.lin.fast_loss()
        lin = self.lin
        lin.s = 1.0 + float(pg['damping'])         # (host mirror: a retry compounds from here)
        self.ctl_f[0] = lin.s
        self.slot = self.tt.advance()
        self.graph.replay()
        # (the GPU is working: the step's host-side bookkeeping happens behind the launch)
        self._prev_last = opt.__dict__.get('_last_view')
        opt.last = opt.loss
        # (the starting loss of a run is not on the host yet: reading it here would wait for the replay and keep the caller's
        #  check of the program -- the dry run -- from overlapping with it; finish() reads it after the trial's verdict)
        hit = opt.__dict__.get('_host_loss')
        self._last_h = hit[1] if hit is not None and hit[0] is opt.loss else None
        opt.reject_count = 0
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
        a, b, loss_h, its, rr, bn2, flag = self.tt.wait()             # the trial's one wait (no stream synchronisation)
        if last_h is None:
            last_h = opt._host(opt.last)
        opt.linearization = lin.kind
        opt._last_replicated = False
        if flag == 4.0:
            # a graph beyond the persistent solve (posegraph.FusedPCG.solve, defer='inplace'): the iterations the capture queues did not
            # reach the tolerance, and the trial's tail has moved the parameters by that unconverged step.  Put them back, drop this
            # capture (the next one is sized by the watched solves that follow) and take the step on the ordinary path.
            with torch.no_grad():
                torch.Tensor.as_subclass(self.P, torch.Tensor).detach().copy_(self.backup)
            _C.mark_written(self.P)
            opt.__dict__.pop('_pgo_graph_step', None)
            opt.__dict__['_pgo_streak'] = (None, 0)
            opt.loss = opt.last
            return opt._step_general(self.input, None, self.weight)
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
        _strategy.update_from_terms(opt.strategy, pg, last_h, loss_h, a, b)
        if last_h < loss_h and opt.reject_count < opt.reject:        # rejected: back, then the ordinary trial loop
            with torch.no_grad():
                opt.update_parameter(params=pg['params'], step=-lin.nodes_to_step(self.Dn))
                opt.loss, opt.reject_count, loss_h = opt.last, 1, last_h
                J, R = lin.strategy_args()
                loss_h = opt._trial_loop(pg, lin, J, R, None, None, last_h, loss_h, defer=False)
        else:
            opt.loss = self.tt.loss_views[self.slot]  # (a ring of device scalars: never overwritten while the view is alive)
        opt._host_loss = (opt.loss, loss_h)
        return opt.loss
