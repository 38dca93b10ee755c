"""Gauss-Newton and Levenberg-Marquardt (host-side mirror of pypose/optim/optimizer.py).

Same constructor and ``step(input, target=None, weight=None)`` contract as the reference
(GN :239-328, LM :459-679), same global semantics -- ONE scalar loss, ONE damping, ONE
accept/reject decision per trial step, damping applied multiplicatively to the clamped diagonal
and compounding over rejected retries, rejected steps undone by ``update_parameter(-D)``.

What differs is how a step is linearised.  Three linearisations share one driver loop:

``DenseLinearization``   the reference's algorithm: dense ``J`` from ``modjac``, ``A = J^T W J``,
                         any ``solver(A=, b=)``.  Used whenever no structure is found.
``BlockLinearization``   residual row n depends on parameter row n only (B independent problems):
                         per-row blocks from ``d_res`` batched backward sweeps, per-problem normal
                         equations and damped Cholesky in HIP kernels (optim/blocks.py).  Detected
                         automatically and verified by a random vector-Jacobian probe.
``GraphLinearization``   residual row e depends on gathered parameter rows idx_k[e] (pose graphs):
                         see optim/posegraph.py.

The block and graph paths produce the same iterates as the dense path up to rounding (the dense
``A`` is exactly their block / block-sparse matrix), which tests/test_optim_* check against
trajectories recorded from the reference.
"""
from __future__ import annotations

import torch
from torch import nn
from torch.optim import Optimizer
from .posegraph import SolveFailed as _SolveFailed
from torch.optim import optimizer as _torch_opt

from . import blocks as _blocks
from .corrector import FastTriggs
from .functional import modjac
from .solver import PINV, Cholesky
from .strategy import Adaptive, Constant, TrustRegion
from . import strategy as _strategy
from . import fused as _fused


class Trivial(nn.Module):
    """Identity-like module: returns its (single) argument, or all of them as a tuple
    (reference optimizer.py:51-61)."""

    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, *args, **kwargs):
        out = *args, *kwargs.values()
        return out[0] if len(out) == 1 else out


class RowWeights:
    """Block-diagonal weight of a stacked row system, kept as its blocks: per residual a [k, d, d] tensor tiled ``ni`` times.
    ``W @ M`` and ``M^T W`` (what LM / GN need, reference optimizer.py:318-322, 653-655) are batched d x d products."""

    def __init__(self, parts):
        self.parts = parts                                # [(blocks [k, d, d], ni)]

    def _each(self, M):
        off = 0
        for ws, ni in self.parts:
            k, d = ws.shape[0] * ni, ws.shape[-1]
            blk = ws.repeat(ni, 1, 1) if ni > 1 else ws
            yield blk, M[off:off + k * d].reshape(k, d, -1)
            off += k * d

    def matmul(self, M):
        """W @ M for M [N_res] or [N_res, c]"""
        vec = M.dim() == 1
        M2 = M.unsqueeze(-1) if vec else M
        out = torch.cat([(blk @ rows).reshape(-1, M2.shape[-1]) for blk, rows in self._each(M2)])
        return out.squeeze(-1) if vec else out

    def rmatmul_T(self, M):
        """M^T @ W for M [N_res, c]: ([W^T M])^T"""
        return torch.cat([(blk.mT @ rows).reshape(-1, M.shape[-1]) for blk, rows in self._each(M)]).T

    def dense(self):
        return torch.block_diag(*[b for ws, ni in self.parts for b in list(ws.unbind(0)) * ni])


class RobustModel(nn.Module):
    """Standardises a model into residual(s) and a robust loss (reference optimizer.py:64-125)."""

    def __init__(self, model, kernel=None, auto=False):
        super().__init__()
        self.model = model
        self.kernel = [Trivial()] if kernel is None else kernel

    @staticmethod
    def _weight_blocks(w, r):
        """per-row weight matrices [k, d, d] and the tiling factor covering all rows of r (a weight given for fewer rows than
        the residual has is repeated, a scalar-residual weight is a 1x1 block: the reference's conventions, optimizer.py:88-97)"""
        ni = r.numel() * w.shape[-1] / w.numel()
        w = w.view(*w.shape, 1, 1) if r.shape[-1] == 1 else w
        return w.reshape(-1, w.shape[-2], w.shape[-1]), int(ni)

    def stack_rows(self, R, weight, J, params):
        """All residuals of a model as ONE row system: (r [N_res], W or None, J [N_res, N_par]).

        J arrives per residual as a tuple over parameters of [*r.shape, *p.shape] derivatives; the parameters become the columns.
        W stays in its block-diagonal FORM (``RowWeights`` below): the reference materialises torch.block_diag of every row's
        d x d block -- an [N_res, N_res] matrix with d / N_res of its entries non-zero -- although everything
        that follows only ever multiplies by it."""
        rows = []
        for Jr in J:
            if isinstance(Jr, (tuple, list)):
                Jr = torch.cat([j.reshape(-1, p.numel()) for j, p in zip(Jr, params)], dim=1)
            rows.append(Jr)
        W = None
        if weight is not None:
            weight = weight if isinstance(weight, (tuple, list)) else [weight]
            assert len(R) == len(weight)
            W = RowWeights([self._weight_blocks(w, r) for w, r in zip(weight, R)])
        return torch.cat([r.reshape(-1) for r in R]), W, torch.cat(rows)

    def forward(self, input, target=None):
        return self.residuals(self.model_forward(input), target)

    def model_forward(self, input):
        if isinstance(input, dict):
            return self.model(**input)
        if isinstance(input, (tuple, list)):
            return self.model(*input)
        return self.model(input)

    def residual(self, output, target):
        return output if target is None else output - target

    def residuals(self, outputs, targets):
        if isinstance(outputs, (tuple, list)):
            targets = [None] * len(outputs) if targets is None else targets
            return tuple(self.residual(out, targets[i]) for i, out in enumerate(outputs))
        return (self.residual(outputs, targets),)

    def loss(self, input, target):
        residuals = self.residuals(self.model_forward(input), target)
        kernels = self.kernel if len(self.kernel) > 1 else [self.kernel[0]] * len(residuals)
        # (|r|^2 >= 0 by construction: built-in kernels skip the public call's sign check and its device round trip)
        from .kernel import RobustKernel        # (a user kernel that subclasses it and overrides forward keeps its forward)
        entry = lambda k: k.of_squared_norm if isinstance(k, RobustKernel) and type(k).forward is RobustKernel.forward else k
        return sum(entry(k)(r.square().sum(-1)).sum() for k, r in zip(kernels, residuals))


# ---------------------------------------------------------------------------------------------
# linearisations
# ---------------------------------------------------------------------------------------------
class DenseLinearization:
    """The reference's dense pipeline (optimizer.py:644-657 for LM, :310-324 for GN)."""

    kind = "dense"

    def __init__(self, opt, pg, input, target, weight):
        model = opt.model
        R = list(model(input, target))
        J = modjac(model, input=(input, target), flatten=False, **opt.jackwargs)
        values = tuple(dict(model.named_parameters()).values())
        J = [torch.cat([j.reshape(-1, p.numel()) for j, p in zip(Jr, values)], dim=1) if isinstance(Jr, (tuple, list)) else Jr
             for Jr in J]
        for i in range(len(R)):
            c = opt.corrector[0] if len(opt.corrector) == 1 else opt.corrector[i]
            R[i], J[i] = c(R=R[i], J=J[i])
        self.R, self.W, self.J = model.stack_rows(R, weight, J, values)

    # LM
    def build_normal_equations(self, dmin, dmax):
        self.J_T = self.W.rmatmul_T(self.J) if self.W is not None else self.J.T
        self.A = self.J_T @ self.J
        self.A.diagonal().clamp_(dmin, dmax)

    def damp(self, damping):
        self.A.diagonal().add_(self.A.diagonal() * damping)

    def solve(self, solver):
        return solver(A=self.A, b=-self.J_T @ self.R.view(-1, 1))

    # GN
    def solve_gauss_newton(self, solver):
        A, b = (self.J, -self.R) if self.W is None else (self.W.matmul(self.J), -self.W.matmul(self.R))
        return solver(A=A, b=b.view(-1, 1))

    def strategy_args(self):
        return self.J, self.R.view(-1, 1)


class BlockLinearization:
    """B independent problems: blocks [n, dr, dp] (see optim/blocks.py)."""

    kind = "block"

    def __init__(self, opt, pg, input, target, weight, R, params, Jb):
        model = opt.model
        n = Jb.shape[0]
        dims = [r.shape[-1] for r in R]
        # correctors act per residual row; split the stacked blocks back per residual
        Rs, Js, off = [], [], 0
        for i, r in enumerate(R):
            c = opt.corrector[0] if len(opt.corrector) == 1 else opt.corrector[i]
            # (FastTriggs / Triggs with a built-in kernel: one HIP launch, closed-form rho' -- optim/corrector.py)
            ri, ji = c(R=r.detach().reshape(n, dims[i]), J=Jb[:, off:off + dims[i], :])
            Rs.append(ri)
            Js.append(ji)
            off += dims[i]
        self.Rb = torch.cat(Rs, dim=-1)
        self.Jb = torch.cat(Js, dim=-2) if len(Js) > 1 else Js[0]
        self.Wb = None
        if weight is not None:
            weight = weight if isinstance(weight, (tuple, list)) else [weight]
            assert len(R) == len(weight)
            dr = sum(dims)
            self.Wb = torch.zeros((n, dr, dr), dtype=Jb.dtype, device=Jb.device)
            off = 0
            for w, r, d in zip(weight, R, dims):
                ws, ni = model._weight_blocks(w, r)
                self.Wb[:, off:off + d, off:off + d] = ws.repeat(ni, 1, 1)
                off += d
        self.op = _blocks.BlockJacobian(self.Jb, [p.shape[-1] for p in params], R=self.Rb)

    def build_normal_equations(self, dmin, dmax):
        # the blocks stay RAW; clamping and the (compounding) damping of the trial loop are a scale factor applied to the
        # diagonal -- inside the solve kernel where there is one (pplie_block_damped_chol_solve), else materialised on demand
        self.A_raw, self.g = _blocks.normal_equations(self.Jb, self.Rb, self.Wb)
        self.dmin, self.dmax, self.s, self._A = dmin, dmax, 1.0, None

    def damp(self, damping):
        self.s, self._A = self.s * (1.0 + damping), None

    @property
    def A(self):
        """the damped normal equations as a tensor (user solver objects, tests): diag <- clamp(diag) * s"""
        if self._A is None:
            A = self.A_raw.clone()
            d = A.diagonal(dim1=-2, dim2=-1)
            d.copy_(d.clamp(self.dmin, self.dmax) * self.s)
            self._A = A
        return self._A

    def solve(self, solver):
        from .posegraph import PCG
        if (isinstance(solver, Cholesky) and not solver.upper) or isinstance(solver, PCG):
            # (an iterative solver has nothing to iterate on for d_par x d_par blocks: factor them)
            Db = _blocks.damped_chol_solve(self.A_raw, self.g, self.s, self.dmin, self.dmax)
            if Db is None:
                Db = _blocks.chol_solve(self.A, self.g)
            assert not torch.any(torch.isnan(Db)), \
                'Cholesky decomposition failed. Check your matrix (may not be positive-definite)'
        else:   # any other solver object: batched call, block by block
            Db = solver(A=self.A, b=-self.g.unsqueeze(-1)).squeeze(-1)
        return self.op.blocks_to_step(Db)

    def solve_gauss_newton(self, solver):
        if self.Wb is None:
            A, b = self.Jb, -self.Rb.unsqueeze(-1)
        else:
            A, b = self.Wb @ self.Jb, -(self.Wb @ self.Rb.unsqueeze(-1))
        return self.op.blocks_to_step(solver(A=A, b=b).squeeze(-1))

    def strategy_args(self):
        return self.op, self.Rb.reshape(-1, 1)


_REPROBE = 64
_REVERIFY = 1024      # (a multiple of _REPROBE)


def _row_count(t):
    return t.numel() // t.shape[-1] if t.dim() >= 1 and t.shape[-1] > 0 else -1


def _linearize(opt, pg, input, target, weight, gauss_newton=False):
    """Pick the cheapest valid linearisation for this model (cached per shape signature).  Gauss-Newton solves
    the rectangular system ``W J d = -W R`` with the user's solver (pseudo-inverse by default, i.e. minimum-norm
    steps on gauge-free graphs): the block and dense linearisations reproduce that literally; single-parameter
    graphs too large for a dense J (or run with a PCG solver) get the same step from plain CG on the normal
    equations (posegraph.GraphLinearization.solve_gauss_newton)."""
    params = [p for p in pg['params'] if p.requires_grad]
    cache = opt.__dict__.setdefault('_structure_cache', {})
    # A structure verdict (block / graph / multi-parameter probe, fused-program cross-check) is keyed on shapes only; a
    # model whose row dependence changes with its input values at fixed shapes would keep a stale "yes".  Positive
    # verdicts therefore expire every _REPROBE linearisations and are re-established by a fresh probe (one extra
    # vector-Jacobian product); negative ones stay (the dense path is always correct).
    # The fused programs' verdict is the expensive one to re-establish (a whole autograd linearisation to compare with: 5 ms
    # on a 10k-pose graph, 8 % of the run if paid every 64 steps) and the least exposed -- a recognised program is re-derived
    # from a (dry) run of the model every step anyway -- so it expires every _REVERIFY linearisations instead.
    uses = cache['_uses'] = cache.get('_uses', 0) + 1
    if getattr(opt, 'structure', None) == "strict":
        # LM(structure="strict") / PPLIE_STRUCTURE=strict: NO positive verdict outlives the linearisation it was established for --
        # every step probes the model's row dependence again (and re-verifies a fused program against the autograd linearisation):
        # for models whose sparsity pattern depends on parameter VALUES at fixed shapes, at the price of the probe per step
        for k in [k for k, v in cache.items() if v is True]:
            cache[k] = None
    elif uses % _REPROBE == 0:
        for k in [k for k, v in cache.items() if v is True and (k != "fused" or uses % _REVERIFY == 0)]:
            cache[k] = None
    # Under torch.inference_mode nothing can be recorded for backward sweeps: only the reference's own
    # functional-jacobian linearisation (DenseLinearization) applies (tests/optim/test_optimizer.py:153-159).
    if getattr(opt, 'structured', True) and params and not torch.is_inference_mode_enabled():
        from . import posegraph as _pg
        if getattr(opt, 'fused', False) and not gauss_newton:
            from . import fused as _fused
            lin = _fused.try_fused(opt, pg, input, target, weight, cache)
            if lin is not None:
                if cache.get("fused") is None:       # first use: cross-check against the generic block path
                    cache["fused"] = False
                    ref = _linearize(opt, pg, input, target, weight)
                    cache["fused"] = ref.kind == lin.reference_kind and lin.verify(ref, pg['min'], pg['max'])
                if cache["fused"]:
                    return lin
        with torch.enable_grad():
            with _pg.GatherRecorder(params) as rec:
                R = list(opt.model(input, target))
            sig = (tuple(tuple(r.shape) for r in R), tuple(tuple(p.shape) for p in params), len(rec.events))
            n = _row_count(R[0])
            same_rows = n > 1 and all(r.dim() >= 2 and _row_count(r) == n for r in R)
            if same_rows and not rec.events and all(p.dim() >= 2 and _row_count(p) == n for p in params):
                verdict = cache.get(sig)
                if verdict is not False:
                    Jb = _blocks.jacobian_blocks(R, params)
                    if verdict is None:
                        verdict = cache[sig] = _blocks.probe_block_structure(R, params, Jb)
                    if verdict:
                        return BlockLinearization(opt, pg, input, target, weight, R, params, Jb)
            elif rec.events and (not gauss_newton or _pg.gauss_newton_on_graph(opt, params)):
                lin = None
                if (same_rows or len(R) > 1) and cache.get(sig) is not False:
                    lin = _pg.try_graph_linearization(opt, pg, input, target, weight, R, params, rec, cache, sig, gauss_newton)
                if lin is None and not gauss_newton and all(p.dim() == 2 for p in params) and R[0].dim() >= 2:
                    from . import multigraph as _mg       # several parameters / widths (bundle adjustment)
                    lin = _mg.try_multigraph_linearization(opt, pg, input, target, weight, R, params, rec, cache, sig)
                if lin is not None:
                    return lin
    return DenseLinearization(opt, pg, input, target, weight)


class _Optimizer(Optimizer):
    """Base class of the second-order optimizers (reference optimizer.py:128-140)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)

    def update_parameter(self, params, step):
        """``p.add_(d)`` parameter by parameter; for LieTensor parameters ``add_`` is the left
        retraction ``Exp(d[..., :dof]) * p`` (lietensor.py add_)."""
        steps = step.split([p.numel() for p in params if p.requires_grad])
        [p.add_(d.view(p.shape)) for p, d in zip(params, steps) if p.requires_grad]

    def _setup_correctors(self, kernel, corrector):
        if kernel is not None:
            kernel = [kernel] if not isinstance(kernel, (tuple, list)) else kernel
            kernel = [k if k is not None else Trivial() for k in kernel]
            self.corrector = [FastTriggs(k) for k in kernel] if corrector is None else corrector
        else:
            self.corrector = [Trivial()] if corrector is None else corrector
        self.corrector = [self.corrector] if not isinstance(self.corrector, (tuple, list)) else self.corrector
        self.corrector = [c if c is not None else Trivial() for c in self.corrector]
        return kernel


class GaussNewton(_Optimizer):
    """Gauss-Newton: solve ``W J d = -W R`` and update (reference optimizer.py:143-328)."""

    def __init__(self, model, solver=None, kernel=None, corrector=None, weight=None, vectorize=True):
        super().__init__(model.parameters(), defaults={})
        self.jackwargs = {'vectorize': vectorize}
        self.solver = PINV() if solver is None else solver
        self.weight = weight
        kernel = self._setup_correctors(kernel, corrector)
        self.model = RobustModel(model, kernel)

    @torch.no_grad()
    def step(self, input, target=None, weight=None):
        for pg in self.param_groups:
            weight = self.weight if weight is None else weight
            lin = _linearize(self, pg, input, target, weight, gauss_newton=True)
            D = lin.solve_gauss_newton(self.solver)
            self.last = self.loss if hasattr(self, 'loss') else self.model.loss(input, target)
            self.update_parameter(params=pg['params'], step=D)
            self.loss = self.model.loss(input, target)
        return self.loss


class LevenbergMarquardt(_Optimizer):
    """Levenberg-Marquardt (reference optimizer.py:331-679).

    ``sparse=True`` (the reference's switch to its optional ``bae`` plugin) is accepted: structured
    problems are handled by the automatic block / graph linearisations whether or not it is set.
    """

    def __init__(self, model, solver=None, strategy=None, kernel=None, corrector=None,
                 weight=None, reject=16, min=1e-6, max=1e32, vectorize=True, sparse=False, *, group=None, static=False, shard=None,
                 exchange=None, structure=None):
        assert min > 0, ValueError("min value has to be positive: {}".format(min))
        assert max > 0, ValueError("max value has to be positive: {}".format(max))
        # structure="strict" (or PPLIE_STRUCTURE=strict): the block / graph structure of the model's Jacobian is probed at EVERY
        # linearisation instead of being remembered per shape signature and re-probed every 64 (a model whose row dependence changes
        # with its parameter values at fixed shapes would otherwise run up to 63 steps on a stale verdict); the device-resident /
        # captured shortcuts of recognised programs are not taken.  None (default): cached verdicts.
        import os as _os
        structure = _os.environ.get("PPLIE_STRUCTURE") or None if structure is None else structure
        assert structure in (None, "strict"), ValueError("structure has to be 'strict' or None: {}".format(structure))
        self.structure = structure
        self.strategy = TrustRegion() if strategy is None else strategy
        defaults = {**{'min': min, 'max': max}, **self.strategy.defaults}
        super().__init__(model.parameters(), defaults=defaults)
        # sparse=True selects the reference's optional `bae` plugin (sparse J, CSR J^T J, PCG).  Here the
        # same problems are covered by the automatically detected block / pose-graph linearisations, so
        # the flag only records the intent; a model without detectable structure still runs densely.
        self.sparse = bool(sparse)
        self.fused = True          # whole-step kernels for recognised residual programs (optim/fused.py)
        # static=True: a promise (like capturing a hipGraph) that the model's residual program and its non-parameter
        # operands do not change between step() calls with the same `input` object; a recognised + verified program
        # is then evaluated directly.  The default (False) keeps the reference's semantics: the model's forward runs at
        # every step (as a dry trace that launches nothing, optim/fused.py DryTracer) and the program is re-derived from it
        self.static = bool(static)
        # torch.distributed process group: independent problems / graph edges are sharded over its
        # ranks (one process per GPU, RCCL); the loss, the gain ratio and -- for pose graphs -- the
        # normal-equation pieces are all-reduced so that every rank takes the same decisions.
        self.group = group
        # shard="nodes" (pose graphs, with group=): the linear solve is sharded by node rows -- each rank assembles and
        # iterates on the rows it owns (optim/nodeshard.py); "edges": edge shards with the solve replicated / all-reduced
        # (optim/posegraph.py).  None (default) decides per graph from facts every rank shares (posegraph.resolve_shard_mode):
        # one process per GPU over RCCL and a graph beyond what ONE GPU solves in a single persistent launch -> "nodes", the
        # mode built to scale; smaller graphs (latency-bound, already one launch per solve) and host groups -> "edges"
        assert shard in (None, "edges", "nodes"), ValueError("shard has to be 'edges', 'nodes' or None: {}".format(shard))
        self.shard = shard
        # exchange="p2p" (node shards): the solve runs as one persistent kernel per GPU that writes its p slices and partial sums
        # straight into the peers' memory (hipIpc-mapped, xGMI) -- no collective per PCG iteration; "rccl": one all-gather + one
        # all-reduce per iteration.  None (default): "p2p" where it applies (device group, <= 8 ranks, slice fits one launch),
        # else "rccl"; a failed peer exchange falls back to "rccl" on EVERY rank together (optim/nodeshard.py)
        assert exchange in (None, "rccl", "p2p"), ValueError("exchange has to be 'rccl', 'p2p' or None: {}".format(exchange))
        self.exchange = exchange
        self.jackwargs = {'vectorize': vectorize}
        self.solver = Cholesky() if solver is None else solver
        self.reject, self.reject_count = reject, 0
        self.weight = weight
        kernel = self._setup_correctors(kernel, corrector)
        self.model = RobustModel(model, kernel)

    def _loss(self, input, target):
        loss = self.model.loss(input, target)
        if self.group is not None:
            import torch.distributed as dist
            loss = loss.clone()
            dist.all_reduce(loss, group=self.group)
        return loss

    def _host(self, t):
        """Value of a scalar loss tensor on the host.  A read-back is a synchronisation, and the trial loop compares
        ``last`` and ``loss`` several times per trial: the value is remembered for the tensor object it came from."""
        hit = self.__dict__.get('_host_loss')
        if hit is not None and hit[0] is t:
            return hit[1]
        v = float(t)
        self._host_loss = (t, v)
        return v

    def _strategy_update(self, pg, J, D, R, last_h):
        """strategy.update(...) for this trial; returns the new loss as a host float (read back together with the
        gain-ratio terms where the linearisation provides them)."""
        builtin = hasattr(J, 'gain_terms') and type(self.strategy) in (Constant, Adaptive, TrustRegion)
        if builtin and type(self.strategy) is Constant and getattr(J, 'gain_is_costly', False):
            # Constant never looks at J, D, R (strategy.py:66-69): on the block path their gain terms are three passes over the
            # [n, dr, dp] blocks that nobody reads
            loss_h = float(self.loss)
            one = torch.ones((1, 1), dtype=torch.float64)
            self.strategy.update(pg, last=last_h, loss=loss_h, J=one, D=one, R=one)
            return loss_h
        ab = J.gain_terms(D) if builtin else None
        if ab is not None:
            # pose graphs: (J D).(J D) and (J D).R from one kernel; the built-in strategies only need the gain ratio,
            # which an equivalent 1x1 problem on the host reproduces (x^2 = a, x r = b) without device round trips
            if self.group is not None and not getattr(J, 'replicated', False):
                import torch.distributed as dist
                dist.all_reduce(ab, group=self.group)
            lin = getattr(J, 'lin', None)
            pend = getattr(lin, 'pending_info', None) if lin is not None else None
            parts = [ab.reshape(-1).double(), self.loss.detach().reshape(1).double()]
            if pend is not None:                 # the persistent PCG's (iterations, rr, bn2, flag) ride along: one read-back
                parts.append(pend.info.double())
            vals = torch.cat(parts).tolist()
            a, b, loss_h = vals[:3]
            if pend is not None:
                lin.pending_info = None
                lin._pending_solver.iterations = pend.resolve(vals[3:7])     # raises like an eager solve would have
            _strategy.update_from_terms(self.strategy, pg, last_h, loss_h, a, b)
            return loss_h
        if self.group is None or getattr(J, 'replicated', False):
            self.strategy.update(pg, last=self.last, loss=self.loss, J=J, D=D, R=R)
            return float(self.loss)
        # the gain ratio needs the GLOBAL (J D)^T (2 R + J D): all-reduce its two dot products and
        # hand the strategy an equivalent 1x1 problem x (2 r + x) with x^2 = a, x r = b
        import torch.distributed as dist
        JD = J @ D
        ab = torch.stack([(JD * JD).sum(), (JD * R).sum()])
        dist.all_reduce(ab, group=self.group)
        x = ab[0].sqrt().clamp_min(torch.finfo(ab.dtype).tiny)
        one = torch.ones((1, 1), dtype=ab.dtype, device=ab.device)
        self.strategy.update(pg, last=self.last, loss=self.loss, J=one, D=x * one, R=(ab[1] / x) * one)
        return float(self.loss)

    # ---- values a device-resident step leaves in device memory until somebody looks (optim/fused.py DeviceLM) ----
    @property
    def reject_count(self):
        dev = self.__dict__.get('_device_lm')
        if dev is not None and dev.pending:
            dev.flush()
        return self.__dict__.get('_reject_count', 0)

    @reject_count.setter
    def reject_count(self, value):
        self.__dict__['_reject_count'] = value

    @property
    def last(self):
        """the loss before the latest step (reference: ``self.last``, optimizer.py:659)"""
        d = self.__dict__
        if '_last_view' in d:
            return d['_last_view']
        raise AttributeError('last')

    @last.setter
    def last(self, value):
        self.__dict__['_last_view'] = value

    def step(self, input, target=None, weight=None):
        d = self.__dict__
        dev, gs = d.get('_device_lm'), d.get('_pgo_graph_step')
        if d.get('structure') == "strict":
            dev = gs = None                      # (every step goes through _linearize and its fresh probe)
        if (dev is not None or gs is not None) and not (self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks
                                                        or _torch_opt._global_optimizer_pre_hooks or _torch_opt._global_optimizer_post_hooks):
            # A verified fused program with its loop state on the device: the whole step is one or two launches.  Nothing
            # here touches autograd, and torch.optim's per-step profiler range + hook dispatch (a good 15 us) is only
            # entered when somebody registered a hook.
            if self.static:
                # the caller's promise that the residual program does not change: the model is not run
                if dev is not None:
                    out = dev.try_step(input, target, weight)
                    if out is not None:
                        return out
                if gs is not None:
                    pg = self.param_groups[0]
                    with _fused._no_tf():
                        early = gs.quick(target)
                    if early:                  # (replay first, the rest of the checks while the GPU works: fused.checked_shortcut)
                        gs.launch(pg)
                        if gs.usable(pg, input, target, self.weight if weight is None else weight):
                            return gs.finish(pg)
                        gs.cancel()
                    elif gs.usable(pg, input, target, self.weight if weight is None else weight):
                        with torch.no_grad():
                            return gs.step(pg)
            else:
                # default, the reference's semantics (optimizer.py:631, 646): the model runs every step -- as a dry trace that
                # launches nothing -- and the shortcut is taken only if this step's program is the one it was built for
                out = _fused.checked_shortcut(self, dev, gs, input, target, weight)
                if out is not None:
                    return out
        try:
            return self._step_general(input, target, weight)
        finally:
            d.pop('_dry_hint', None)         # (this step's trace: never carried into another step)
    step.hooked = True              # torch.optim.Optimizer.__init__ must not wrap it again: _step_general carries the wrapper

    @torch.no_grad()
    def _step_general(self, input, target=None, weight=None):
        dev = self.__dict__.get('_device_lm')
        if dev is not None and dev.pending:          # (a later flush must not overwrite what this step sets: reject_count, damping)
            dev.flush()
        for pg in self.param_groups:
            weight = self.weight if weight is None else weight
            lin = _linearize(self, pg, input, target, weight)
            if self.group is not None and lin.kind == "dense":
                raise RuntimeError("LM(group=...) needs a block or pose-graph structured model; the dense "
                                   "linearisation is not sharded")
            lin.build_normal_equations(pg['min'], pg['max'])
            self.linearization = lin.kind
            self._last_replicated = getattr(lin, 'replicated', False)
            if hasattr(lin, 'run_trials'):              # whole-step fused kernels (optim/fused.py)
                lin.run_trials(self, pg)
                continue
            self.last = self.loss = self.loss if hasattr(self, 'loss') else self._loss(input, target)
            self.reject_count = 0
            J, R = lin.strategy_args()
            # the loop's decisions are taken on host copies of last / loss (one read-back per trial instead of a
            # synchronising tensor comparison at every `<` / `<=`); self.last / self.loss stay tensors
            last_h = loss_h = self._host(self.loss)
            # pose graphs on the persistent PCG: the solve's own (iterations, flag) read-back is deferred into the one that
            # fetches the trial's loss and gain terms -- the update, loss and gain kernels queue up behind the solve instead of
            # waiting for the host to have looked at it (a failed solve returns a zero step)
            defer = (self.group is None and hasattr(J, 'gain_terms') and type(self.strategy) in (Constant, Adaptive, TrustRegion)
                     and getattr(getattr(J, 'lin', None), '_hip', lambda: False)())
            loss_h = self._trial_loop(pg, lin, J, R, input, target, last_h, loss_h, defer)
            self._consider_graph_step(pg, lin, input, target, weight, defer)
            self._host_loss = (self.loss, loss_h)
            if self.group is not None and getattr(lin, 'replicated', False):
                self._sync_replicas(pg['params'])
        return self.loss

    def _trial_loop(self, pg, lin, J, R, input, target, last_h, loss_h, defer):
        """damp / solve / update / loss / strategy / accept-or-reject until a trial is accepted (optimizer.py:662-678);
        returns the accepted loss as a host float"""
        use_tail = defer and getattr(lin, 'trial_tail', None) is not None and hasattr(lin, 'solve_nodes')
        while last_h <= loss_h:
            lin.damp(pg['damping'])
            self._defer_solver_info = defer
            D = Dn = None
            try:
                if use_tail:
                    if hasattr(lin, 'tail_setup'):
                        lin.tail_setup()
                    Dn = lin.solve_nodes(self.solver)          # (the padded step only if somebody needs it)
                else:
                    D = lin.solve(self.solver)
            except Exception as e:
                print(e, "\nLinear solver failed. Breaking optimization step...")
                break
            finally:
                self._defer_solver_info = False
            try:
                # the recognised pose-graph program: update, candidate loss, gain terms and their read-back are one C call and one
                # wait on pinned memory (optim/pgograph.py TrialTail)
                fused_tail = lin.trial_tail() if use_tail else None
                if fused_tail is not None:
                    a, b, loss_h, self.loss = fused_tail
                    _strategy.update_from_terms(self.strategy, pg, last_h, loss_h, a, b)
                else:
                    D = lin.nodes_to_step(Dn) if D is None else D
                    self.update_parameter(pg['params'], D)
                    self.loss = lin.fast_loss() if hasattr(lin, 'fast_loss') else self._loss(input, target)
                    loss_h = self._strategy_update(pg, J, D, R, last_h)
            except _SolveFailed as e:       # noticed after the fact; the step was zero, the parameters are where they were
                print(e, "\nLinear solver failed. Breaking optimization step...")
                self.loss = self.last
                break
            if getattr(lin, 'pending_info', None) is not None:     # (a strategy path that did not pick it up)
                pend, lin.pending_info = lin.pending_info, None
                lin._pending_solver.iterations = pend.resolve()
            if last_h < loss_h and self.reject_count < self.reject:           # reject the step
                self.update_parameter(params=pg['params'], step=-(lin.nodes_to_step(Dn) if D is None else D))
                self.loss, self.reject_count, loss_h = self.last, self.reject_count + 1, last_h
            else:
                break
        return loss_h

    def _consider_graph_step(self, pg, lin, input, target, weight, defer):
        """After PgoGraphStep.MIN_STREAK ordinary steps on the same verified pose-graph program: capture the trial as one
        hipGraph (optim/pgograph.py); LevenbergMarquardt.step replays it while nothing it was captured on changes."""
        from .pgograph import PgoGraphStep
        from .posegraph import PCG, PERSIST_NODES, FusedPCG
        d = self.__dict__
        # the captured trial's tail evaluates the PLAIN loss sum |r|^2 (pplie_pgo_trial_tail, RK_NONE): only the trivial kernel /
        # corrector configuration may be captured -- a robust `fast_loss` (fused.py) must not make a problem eligible
        trivial = all(isinstance(c, Trivial) for c in self.corrector) and all(isinstance(k, Trivial) for k in self.model.kernel)
        ok = (defer and trivial and lin.kind == "fused:pgo" and hasattr(lin, 'fast_loss') and getattr(lin, 'robust', None) is None
              and target is None and len(self.param_groups) == 1
              and isinstance(self.solver, PCG) and getattr(self.solver, 'fused', True)
              and FusedPCG.two_launch and getattr(self, 'graph_step', True)
              and not isinstance(weight, (tuple, list)))
        # graphs beyond the persistent solve: the packed two-launch iteration with the device-side stop test can run unwatched
        # (posegraph.FusedPCG.solve, defer='inplace'); the capture queues as many iterations as the longest watched solve of the
        # streak needed, rounded up to a multiple of eight -- launches behind the converging iteration cost ~5 us each, so a capture
        # sized far beyond what the solves take would give back what it saves in host latency
        unwatched = None
        if ok and not (lin.N <= PERSIST_NODES and FusedPCG.persist):
            # (the dict also holds the multigraph route's workspaces, which have none of these attributes: ADVICE r05)
            wsps = [w for w in (d.get('_pcg_workspaces') or {}).values()
                    if isinstance(w, FusedPCG) and w.N == lin.N and w.sym == 'pack' and w.iterations_seen > 0]
            ok = (FusedPCG.capture_large and FusedPCG.device_stop and bool(getattr(lin, 'HB_pack', False)) and len(wsps) == 1)
            if ok:
                unwatched = FusedPCG.unwatched_for(wsps[0].iterations_seen, self.solver.maxiter)
                ok = unwatched is not None
        cache = d.get('_structure_cache') or {}
        hit = cache.get("program")
        if not ok or hit is None or hit[2] != "pgo" or cache.get("fused") is not True:
            d['_pgo_streak'] = (None, 0)
            d.pop('_pgo_graph_step', None)
            return
        prog = hit[3]
        cur = d.get('_pgo_graph_step')
        if cur is not None and cur.prog is prog:
            return
        prev, n = d.get('_pgo_streak', (None, 0))
        n = n + 1 if prev is prog else 1
        d['_pgo_streak'] = (prog, n)
        if n >= PgoGraphStep.MIN_STREAK:
            params = [p for p in pg['params'] if p.requires_grad]
            try:
                if unwatched is not None:
                    wsps[0].unwatched_iterations = unwatched
                d['_pgo_graph_step'] = PgoGraphStep(self, pg, prog, input, weight, params[0], trivial)
            except Exception as e:                       # capture is an optimisation: the ordinary path stays correct
                d['_pgo_graph_step'] = None
                d['_pgo_streak'] = (None, -(1 << 30))
                import warnings
                warnings.warn(f"pypose_amd: the pose-graph trial could not be captured as a hipGraph ({e}); continuing un-captured")

    def _sync_replicas(self, params):
        """Every rank solved the same system by itself; atomic summation order may leave last-bit differences
        between the ranks' steps, so the first rank's parameters are broadcast (N * width floats per LM step)."""
        import torch.distributed as dist
        src = dist.get_global_rank(self.group, 0)
        for p in params:
            if p.requires_grad:
                data = p.data
                dist.broadcast(data.tensor() if hasattr(data, 'ltype') else data, src=src, group=self.group)


# the general path keeps torch.optim's step wrapper (profiler range, optimizer step pre / post hooks)
LevenbergMarquardt._step_general = Optimizer.profile_hook_step(LevenbergMarquardt._step_general)
