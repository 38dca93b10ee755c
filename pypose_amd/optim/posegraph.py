"""Gather-structured (pose-graph) linearisation for Levenberg-Marquardt.

The reference's pose-graph model (examples/module/pgo/pgo.py:15-25) is an ordinary nn.Module:

    node1, node2 = self.nodes[edges[..., 0]], self.nodes[edges[..., 1]]
    return (poses.Inv() @ node1.Inv() @ node2).Log().tensor()

Its dense LM path needs a [6E, 7N] Jacobian and cannot run beyond a few hundred nodes
(examples/module/pgo/readme.md:47-49); its scalable path is the un-vendored ``bae`` plugin.
This module gives the SAME unmodified model a scalable path on the GPU:

1. :class:`GatherRecorder` notes every integer-tensor ``__getitem__`` on an optimised
   ``pp.Parameter`` during the forward pass (index tensor + the gathered rows, which are nodes
   of the autograd graph).
2. Per-edge Jacobian blocks w.r.t. the *gathered rows* come from ``d_res`` batched backward
   sweeps (blocks.jacobian_blocks) -- block-diagonal by construction if each residual row only
   depends on its own gathered rows.  A random vector-Jacobian probe against the real model
   verifies that; otherwise the optimizer falls back to the dense path.
3. The normal equations stay block sparse and matrix-free (csrc/graph.hip): block diagonal +
   gradient by scatter-add, ``H p`` one lane per edge, block-Jacobi PCG; small graphs can instead
   be assembled densely and handed to the user's ``solver`` (bit-for-bit the reference's
   algebra, used by the parity tests).
4. With ``group=`` (torch.distributed) edges are sharded over ranks, nodes replicated: the
   block diagonal, the gradient, every ``H p`` and the loss are all-reduced (RCCL on GPUs).
"""
from __future__ import annotations

import ctypes
import warnings

import torch
from torch import nn

from .. import _C
from ..lietensor import lietensor as _lt
from . import blocks as _blocks

_SPMV_SIG = [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
_ASM_SIG = [ctypes.c_void_p] * 7 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
_ASMC_SIG = [ctypes.c_void_p] * 8 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
_ASML_SIG = [ctypes.c_void_p] * 9 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
_LAPD_SIG = [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
_PREP_LAP_SIG = ([ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 13 + [ctypes.c_double, ctypes.c_void_p, ctypes.c_double, ctypes.c_double,
                                                                                          ctypes.c_int64, ctypes.c_int, ctypes.c_void_p])
_PREP_SIG = [ctypes.c_void_p] * 10 + [ctypes.c_double] * 3 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
_BEGIN_SIG = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
_PREP_DEV_SIG = [ctypes.c_void_p] * 11 + [ctypes.c_double] * 2 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
_GAIN_SIG = [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 2 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
_GAIN_PARTIALS = 1024       # PPLIE_GAIN_PARTIALS
_BSR_SIG = [ctypes.c_void_p] * 8 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
_PCG_SCAL_ELEMS = 2 * 8 * 32 * 32          # PPLIE_PCG2_SCAL_ELEMS (covers PPLIE_PCG_SCAL_ELEMS): slot-spread scalars (csrc/graph.hip)
_PCG2_SPMV_SIG = [ctypes.c_void_p] * 11 + [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
_PCG2_SPMV_STOP_SIG = [ctypes.c_void_p] * 11 + [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
_PCG2_STEP_SIG = [ctypes.c_void_p] * 9 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
_PERSIST_SIG = [ctypes.c_void_p] * 15 + [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
_GHOST_SIG = [ctypes.c_void_p] * 15 + [ctypes.c_double] + [ctypes.c_int] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
_PERSIST_GRID_MAX = 256     # PPLIE_PCG_PERSIST_GRID
_PERSIST_SLOTS = 8          # PPLIE_PCG_PERSIST_SLOTS
_COARSE_SLOTS = 24          # PPLIE_PCG_COARSE_SLOTS: row of the partial-sum table of the two-level (gauge) variant
_PCG2_CS_ELEMS = 2 * 32 * 32  # PPLIE_PCG2_CS_ELEMS: coarse sums of the two-launch iteration
_GHOST_CZ_SIG = [ctypes.c_void_p] * 16 + [ctypes.c_double] + [ctypes.c_int] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
_GHOST_TAIL_SIG = [ctypes.c_void_p] * 16 + [ctypes.c_double] + [ctypes.c_int] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
_PGO_PARTIALS = 1024      # PPLIE_PGO_PARTIALS
_PREP_CZ_SIG = [ctypes.c_void_p] * 11 + [ctypes.c_double, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_int64, ctypes.c_int,
                                         ctypes.c_void_p]
_PCG2_SPMV_CZ_SIG = [ctypes.c_void_p] * 12 + [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
_PCG2_STEP_CZ_SIG = [ctypes.c_void_p] * 10 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
_CZ_INIT_SIG = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
_PCG2_REPORT_SIG = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
_PREP_CZ_DP_SIG = [ctypes.c_void_p] * 13 + [ctypes.c_double, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_int64, ctypes.c_int,
                                             ctypes.c_void_p]
import os as _os
# graphs up to this many nodes run the whole PCG solve in ONE persistent launch (csrc/pcg_persist.hip); larger ones need
# the whole chip's bandwidth per iteration and keep the two-launch hipGraph iteration
PERSIST_NODES = int(_os.environ.get("PPLIE_PCG_PERSIST_NODES", "32768"))
GHOST_GRIDS = (160, 192)    # grids the ghost-zone solve tries below PERSIST_GRID (tools/time_pcg_iter.py sweeps them)
PERSIST_GRID = int(_os.environ.get("PPLIE_PCG_PERSIST_GRID", "256"))   # one 1024-lane workgroup per CU: 64 -> 675, 128 -> 788, 256 -> 850 LM steps/s at 10 k nodes
_INV_SIG = [ctypes.c_void_p] * 2 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
_HIP_SHAPES = {(6, 6, 2), (7, 7, 2), (3, 3, 2), (6, 6, 1), (3, 3, 1)}
_STALL_CHECKS = 64          # plain CG on a singular system: give up after this many checks without a new best residual,
_DIVERGED = 100.0           # or as soon as the residual norm is this far above the best one; return the best iterate
DENSE_LIMIT = 4096          # assemble a dense A for the user's solver up to this many unknowns


_geo_mod = None


def _geometry():
    """pypose_amd.function.geometry, imported on first use (it imports this package's optimizers back)"""
    global _geo_mod
    if _geo_mod is None:
        from ..function import geometry as g
        _geo_mod = g
    return _geo_mod


class GatherRecorder:
    """Context manager recording ``param[index_tensor]`` gathers on the tracked parameters."""

    def __init__(self, params):
        self.ids = {id(p): p for p in params}
        # plain aliases of the tracked parameters: attribute reads on a LieTensor are __torch_function__ round trips
        self.plain = {id(p): torch.Tensor.as_subclass(p, torch.Tensor) for p in params}
        self.events = []      # (param, index LongTensor [E], gathered rows [E, w])
        self.closed = []      # (residual, [(input tensor, d residual / d input blocks)], blockers): ops that know their Jacobian

    def note_closed(self, residual, inputs, blockers=()):
        self.closed.append((residual, list(inputs), list(blockers)))

    def closed_blocks(self, r, mine):
        """``[E, dr, sum widths]`` blocks of residual ``r`` with respect to the gathered rows ``mine`` when ``r`` is the output
        of an op that handed over its own closed-form Jacobian (function/geometry.py reprojerr) and every input of that op is
        one of the gathers, row for row; None otherwise (the caller sweeps the autograd graph)."""
        for cr, inputs, blockers in self.closed:
            if blockers or cr.data_ptr() != r.data_ptr() or cr.numel() != r.numel() or cr.dtype != r.dtype:
                continue
            dr = r.shape[-1]
            E = r.numel() // dr
            cols, used = [], set()
            for src, ix, out in mine:
                hit = [k for k, (inp, blk) in enumerate(inputs) if inp.data_ptr() == out.data_ptr() and inp.numel() == out.numel()
                       and k not in used]
                if len(hit) != 1 or inputs[hit[0]][1].numel() % (E * dr) != 0:
                    return None
                used.add(hit[0])
                blk = inputs[hit[0]][1].reshape(E, dr, -1)
                w = src.shape[-1]
                if blk.shape[-1] > w:
                    return None
                if blk.shape[-1] < w:          # group parameters: tangent blocks, zero-padded to the stored width like their gradients
                    blk = torch.cat([blk, blk.new_zeros((E, dr, w - blk.shape[-1]))], -1)
                cols.append(blk)
            if any(inp.requires_grad for k, (inp, _) in enumerate(inputs) if k not in used):
                return None
            return torch.cat(cols, -1).contiguous()
        return None

    def __enter__(self):
        _geo = _geometry()
        _geo._closed_recorders.append(self)
        _lt._gather_recorders.append(self)
        # plain nn.Parameters (e.g. intrinsics / 3-D points of a bundle-adjustment model) do not pass through
        # LieTensor.__torch_function__: a TorchFunctionMode sees their ``param[index]`` too
        self._mode = _PlainGatherMode(self) if any(not isinstance(p, _lt.LieTensor) for p in self.ids.values()) else None
        if self._mode is not None:
            self._mode.__enter__()
        return self

    def __exit__(self, *exc):
        if self._mode is not None:
            self._mode.__exit__(*exc)
        _lt._gather_recorders.remove(self)
        _geometry()._closed_recorders.remove(self)

    def note(self, source, index, out):
        # gathers on a tracked parameter, or on a tensor derived from one (e.g. ``cat((root, nodes))`` in
        # the reference's chain example, tests/optim/test_sparse_lm.py:26-36)
        if isinstance(index, torch.Tensor) and index.dtype == torch.int64 and index.dim() >= 1 and isinstance(out, torch.Tensor) \
                and isinstance(source, torch.Tensor):
            src = self.plain.get(id(source))
            tracked = src is not None
            if (tracked or torch.Tensor.as_subclass(out, torch.Tensor).requires_grad) \
                    and (src if tracked else torch.Tensor.as_subclass(source, torch.Tensor)).dim() == 2:
                self.events.append((source, index, out))


class _PlainGatherMode(torch.overrides.TorchFunctionMode):
    def __init__(self, rec):
        super().__init__()
        self.rec = rec

    def __torch_function__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        if func is torch.Tensor.__getitem__ and len(args) == 2 and not isinstance(args[0], _lt.LieTensor) \
                and id(args[0]) in self.rec.ids:
            self.rec.note(args[0], args[1], out)
        return out


def _rows_of_parameter(src, param):
    """For a ``src`` [Ns, w] built from ``param`` [N, w] by pure row movement (cat with constants,
    slicing, identity): LongTensor [Ns] giving the parameter row behind each ``src`` row, -1 for rows that
    do not come from the parameter (fixed nodes).  None if ``src`` is not such a row copy."""
    if src is param:
        return torch.arange(param.shape[0], device=param.device)
    if src.shape[-1] != param.shape[-1] or src.grad_fn is None:
        return None
    ns, n = src.shape[0], param.shape[0]
    cot = torch.zeros_like(src)
    cot[:, 0] = torch.arange(1, ns + 1, device=src.device, dtype=src.dtype)
    (g,) = torch.autograd.grad([src], [param], [cot], retain_graph=True, allow_unused=True)
    if g is None or bool((g[:, 1:] != 0).any()):
        return None
    ids = g[:, 0].round().to(torch.int64)                      # 1-based src row of every parameter row (0: unused)
    used = ids > 0
    if bool((ids[used] > ns).any()) or ids[used].unique().numel() != int(used.sum()):
        return None
    back = torch.full((ns,), -1, dtype=torch.int64, device=src.device)
    back[ids[used] - 1] = torch.arange(n, device=src.device)[used]
    # a pure row copy passes values through unchanged
    chk = back >= 0
    if not torch.equal(torch.Tensor.as_subclass(src.detach(), torch.Tensor)[chk],
                       torch.Tensor.as_subclass(param.detach(), torch.Tensor)[back[chk]]):
        return None
    return back


class PCG(nn.Module):
    """Preconditioned conjugate gradient with a block-Jacobi preconditioner for the matrix-free
    pose-graph normal equations (the name the reference exposes for its plugin solver,
    pypose/optim/solver.py:343-364).  Stops when ||r|| <= tol * ||b|| or after ``maxiter``."""

    def __init__(self, maxiter=None, tol=1e-5, check_every=8, gauge=True):
        super().__init__()
        self.maxiter, self.tol, self.check_every = maxiter, tol, check_every
        # gauge: on relative-pose graphs (J[e, 0] = -J[e, 1]: a global motion is in J's null space) the block-Jacobi preconditioner
        # is completed by the exact coarse correction over the m gauge directions, M^-1 = blockdiag(A)^-1 + Z E^-1 Z^T
        # (csrc/pcg_persist.hip "CZ"): same stop test, same system, 17 / 35 / 105 -> ~18 / 20 / 26 iterations on the LM steps of a
        # 10 k-pose graph.  gauge=False: the plain block-Jacobi iteration of the reference's plugin solver.
        self.gauge = gauge

    def solve(self, matvec, b, precond, stall=None):
        """``stall`` (singular systems): keep the iterate of smallest residual and stop once the residual has
        grown ``_DIVERGED`` times above it or after ``stall`` checks without a new minimum -- past the rounding floor
        the null-space part of r dominates rho, alpha = rho / p.Ap explodes and x drifts along the null space."""
        x = torch.zeros_like(b)
        r = b.clone()
        bn = torch.linalg.norm(b)
        if bn == 0:
            return x
        maxiter = b.numel() * 10 if self.maxiter is None else self.maxiter
        z = precond(r)
        p = z.clone()
        rho = (r * z).sum()
        best, stalled = float('inf'), 0
        for it in range(maxiter):
            q = matvec(p)
            alpha = rho / (p * q).sum()
            x.add_(alpha * p)
            r.sub_(alpha * q)
            if (it + 1) % self.check_every == 0:
                rn = torch.linalg.norm(r)
                if rn <= self.tol * bn:
                    break
                if stall is not None:
                    if rn < best:
                        best, stalled, xbest = float(rn), 0, x.clone()
                    else:
                        stalled += 1
                        if stalled >= stall or rn > _DIVERGED * best:
                            x = xbest
                            break
            z = precond(r)
            rho_new = (r * z).sum()
            p.mul_(rho_new / rho).add_(z)
            rho = rho_new
        self.iterations = it + 1
        return x

    def forward(self, A, b, x=None, M=None):
        """Dense call shape ``solver(A=, b=)`` (Jacobi preconditioner) for API compatibility."""
        d = A.diagonal().clamp_min(torch.finfo(A.dtype).tiny).unsqueeze(-1)
        return self.solve(lambda v: A @ v, b, lambda v: v / d)


def _all_reduce(t, group):
    if group is not None:
        import torch.distributed as dist
        dist.all_reduce(t, group=group)
    return t


_PCG_SIG = [ctypes.c_int] + [ctypes.c_void_p] * 10 + [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]


class SolveFailed(AssertionError):
    """a linear solve whose failure was noticed after the fact (deferred read-back); the parameters were not touched"""


class _PendingInfo:
    """(iterations, |r|^2, |b|^2, flag) of a persistent solve, still on the device"""

    def __init__(self, info):
        self.info = info

    def resolve(self, values=None):
        its, rr, bn2, flag = self.info.tolist() if values is None else values
        _check_persist_flag(flag, rr)
        return int(its)


def _check_persist_flag(flag, rr):
    """(iterations, rr, bn2, flag) of pplie_pcg_persist: flag 2 = NaN, 3 = a workgroup never arrived at the exchange (the
    launch's workgroups were not all resident: CU mask, partition, a concurrent kernel) -- both handed back x = 0."""
    if flag == 3.0:
        if FusedPCG.persist:
            FusedPCG.persist = False          # permanently: the two-launch iteration needs no co-residency
            warnings.warn("pypose_amd: the persistent PCG solve timed out waiting for a workgroup (its workgroups were not all "
                          "resident on this device); falling back to the two-launch iteration for the rest of the process")
        raise SolveFailed('persistent PCG: grid exchange timed out; retrying with the two-launch iteration')
    if flag == 2.0 or rr != rr:
        raise SolveFailed('Linear solve produced NaN (matrix may not be positive-definite)')


class FusedPCG:
    """Device-resident block-Jacobi PCG for one graph shape (csrc/graph.hip).

    Two launches per iteration on one GPU (``pplie_pcg2_spmv`` streams the off-diagonal blocks in incidence order
    and reduces p.q, q.z, q.Binv q; ``pplie_pcg2_step`` does every vector update), three plus an all-reduce on edge
    shards (``pplie_graph_spmv``, ``pplie_pcg_stage`` 0-2); every scalar stays on the device.  ``check_every``
    iterations are captured into one hipGraph and replayed, so the loop costs neither Python dispatch nor a host
    sync per iteration.  Buffers (and the graph) are cached per shape and reused across LM steps.
    """

    # one off-diagonal block per edge on graphs beyond the persistent solve (pplie_pcg2_spmv_sym).  OFF: measured slower on
    # the BASELINE graph (10^5 nodes, 75 % uniformly random loop closures): 52 us per SpMV against 41 -- the per-incidence
    # blocks are one sequential stream per node, the per-edge blocks are 144-byte gathers (two 128-byte lines each) and the
    # second touch of a random closure's block is never a cache hit; it pays only on graphs whose edges are local in node order
    sym_blocks = False
    # symmetric off-diagonal blocks in packed form on graphs beyond the persistent solve, for linearisations that declare
    # J[e, 0] = -J[e, 1] (the relative-pose program): 84 instead of 144 bytes of matrix per incidence and iteration
    pack_blocks = True
    lap_assembly = True      # ... and their assembly in two launches without dependent chains (pplie_graph_assemble_lap)
    device_stop = True       # large graphs: the two-launch iteration tests convergence on the device (pplie_pcg2_*_stop)
    two_launch = True        # class-level switches (tools/ and tests compare the three-launch / graph-less variants)
    use_graph = True
    persist = True           # one persistent launch per solve on small graphs (csrc/pcg_persist.hip)
    ghost = True             # the ghost-zone form of the persistent solve (one grid-wide dependency per iteration)
    profile = False          # tools/time_pcg_iter.py: the persistent kernel leaves per-phase clock ticks in rr_hist[cap - 8:]
    # the "Laplacian" assembly's per-node sums inside the solve's set-up launch (pplie_pcg_prepare_lap); PPLIE_PCG_FUSE_PREPARE=0: off
    fuse_prepare = _os.environ.get("PPLIE_PCG_FUSE_PREPARE", "1") != "0"
    # the LM trial's gain terms and retraction in the persistent solve's epilogue (pplie_pcg_ghost_tail); PPLIE_PCG_TAIL_IN_SOLVE=0: off
    tail_in_solve = _os.environ.get("PPLIE_PCG_TAIL_IN_SOLVE", "1") != "0"
    # two-level preconditioner (block-Jacobi + gauge modes) where the linearisation allows it (PCG(gauge=)); PPLIE_PCG_GAUGE=0: off
    coarse = _os.environ.get("PPLIE_PCG_GAUGE", "1") != "0"
    # the whole LM trial of a graph BEYOND the persistent solve as one hipGraph replay too (optim/pgograph.py): the two-launch iterations
    # run unwatched, at most `unwatched_max` of them per capture (PPLIE_CAPTURE_LARGE=0: the watched chunks of eight only)
    capture_large = _os.environ.get("PPLIE_CAPTURE_LARGE", "1") != "0"
    # the two-level two-launch iteration reads D and Binv as packed upper triangles too (pplie_*_dp; PPLIE_PACK_DIAG=0: full blocks)
    pack_diag = _os.environ.get("PPLIE_PACK_DIAG", "1") != "0"
    unwatched_max = 48

    @staticmethod
    def unwatched_for(iterations_seen, maxiter=None):
        """How many iterations a captured trial queues after watched solves that took up to ``iterations_seen``: the next multiple of
        eight above it (at least 16), or None when that exceeds ``unwatched_max`` / the solver's ``maxiter`` -- every launch behind the
        converging iteration costs ~4.6 us, so a capture sized far beyond the solves gives back what it saves in host latency."""
        if iterations_seen <= 0:
            return None
        k = max(16, -(-(int(iterations_seen) + 1) // 8) * 8)
        return k if k <= min(FusedPCG.unwatched_max, maxiter or FusedPCG.unwatched_max) else None

    def __init__(self, E, K, dr, m, N, dtype, device, has_w, check_every):
        z = lambda *s: torch.zeros(s, dtype=dtype, device=device)
        self.key = (E, K, dr, m, N, dtype, device, has_w, check_every)
        self.E, self.K, self.dr, self.m, self.N, self.check_every = E, K, dr, m, N, check_every
        self.dtype, self.device, self.has_w = dtype, device, has_w
        self.J = self.W = self.idx = None                          # per-edge copies: only the matrix-free (sharded) path
        self.D, self.HB = z(N, m, m), None                         # damped diagonal blocks; off-diagonal blocks (bsr path)
        self.Binv, self.shift = z(N, m, m), z(N, m)
        self.x, self.r, self.q, self.z = (z(N, m) for _ in range(4))
        self.p = z(N + 2, m)[:N]                                   # (pplie_pcg2_spmv_pack reads up to m - 1 elements past a row)
        self.r2 = z(N, m)                                          # the two-launch iteration ping-pongs the residual
        # scal | part | it share ONE allocation: a solve clears them with a single fill instead of three
        esz = 4 if dtype == torch.float32 else 8
        nw = esz // 4
        # (part is sized for the wider rows of the two-level variant; cs, its coarse sums of the two-launch iteration, rides behind scal)
        # (+ 8 group rows per table: the two-level exchange of the gauge variant, csrc/pcg_persist.hip kHierGroups)
        nb_scal, nb_part, nb_it = (_PCG_SCAL_ELEMS + _PCG2_CS_ELEMS) * esz, 2 * (_PERSIST_GRID_MAX + 8) * _COARSE_SLOTS * nw * 8, 16
        # the persistent solve's hand-off table of p (tagged 64-bit words, double-buffered) sits in the same allocation
        nb_ptag = 2 * N * m * nw * 8 if (N <= PERSIST_NODES and m in (3, 6, 7)) else 0
        self._ctl = torch.zeros(nb_scal + nb_part + nb_it + nb_ptag, dtype=torch.uint8, device=device)
        self.ptag = self._ctl[nb_scal + nb_part + nb_it:].view(torch.int64) if nb_ptag else None
        self.no_persist = nb_ptag == 0                              # (also set when the device cannot hold the solve resident)
        self.scal = self._ctl[:_PCG_SCAL_ELEMS * esz].view(dtype)
        self.cs = self._ctl[_PCG_SCAL_ELEMS * esz:nb_scal].view(dtype)
        self.cz = False                                            # this solve runs the two-level variant
        self.cap = 1 << 16
        self.rr_hist = z(self.cap)
        self.part = self._ctl[nb_scal:nb_scal + nb_part].view(torch.int64)   # persistent solve: tagged partial sums
        self.it = self._ctl[nb_scal + nb_part:nb_scal + nb_part + nb_it].view(torch.int32)[:4]     # iterations done, scratch, stop flag, -
        self.info = z(4)
        self.s_device = torch.ones(1, dtype=torch.float64, device=device)      # the damping factor of a captured trial (pplie_pcg_begin)
        self.sfx = "_f32" if dtype == torch.float32 else "_f64"
        self.graph = None                                          # captured check_every iterations
        self.bsr = None                                            # which iteration the graph holds
        self.sym = False                                           # HB holds one block per edge
        self.stop_tol2 = None                                      # tol^2 when the captured iterations carry the device-side stop
        self.iterations_seen = 0                                   # the longest watched two-launch solve so far (sizes a captured trial)
        self.Dp = self.Bp = None                                   # packed upper triangles of D / Binv (the _dp iteration; allocated on first use)
        self.dp = False                                            # this solve's iteration reads them
        self.unwatched_iterations = 24                             # iterations a captured trial queues (PgoGraphStep sets it)
        self._csr_obj = None

    def _ghost_map(self, lin, grid):
        """What pplie_pcg_ghost needs of the incidence lists for `grid` workgroups: slot[c] = local index of incidence c's
        neighbour inside the workgroup that owns the incidence's node (its owned nodes first, then its ghosts = the neighbours it
        does not own, sorted), gptr / gids = the ghost lists, and the largest incidence / ghost count of any workgroup.  Cached
        with the incidence lists (rebuilt only when the edge list changes)."""
        hit = self.__dict__.get('_ghost')
        if hit is not None and hit[0] is self._csr_obj and hit[1] == grid:
            return hit[2]
        N, dev = self.N, self.ptr.device
        ptr, other = self.ptr.long(), self.other.long()
        wg = torch.arange(grid + 1, device=dev)
        bounds = (N * wg) // grid                                  # n0 of every workgroup (the kernel's own partition)
        deg = ptr[1:] - ptr[:-1]
        node = torch.repeat_interleave(torch.arange(N, device=dev), deg)
        w_node = torch.searchsorted(bounds, node, right=True) - 1
        w_other = torch.searchsorted(bounds, other, right=True) - 1
        ghost = w_node != w_other
        key = w_node[ghost] * N + other[ghost]
        uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
        g_wg, gids = uniq // N, (uniq % N).to(torch.int32)
        gcount = torch.bincount(g_wg, minlength=grid)
        gptr = torch.zeros(grid + 1, dtype=torch.int64, device=dev)
        gptr[1:] = torch.cumsum(gcount, 0)
        n_own = bounds[1:] - bounds[:-1]
        slot = other - bounds[w_node]                              # owned neighbours: their local node index
        slot[ghost] = n_own[w_node[ghost]] + (inv - gptr[w_node[ghost]])
        cnt = ptr[bounds[1:]] - ptr[bounds[:-1]]
        stats = torch.stack([cnt.max(), gcount.max()]).tolist()    # (one read-back per edge list)
        out = (slot.to(torch.int32).contiguous(), gptr.to(torch.int32).contiguous(), gids.contiguous(), int(stats[0]), int(stats[1]))
        self._ghost = (self._csr_obj, grid, out)
        return out

    def _persistent(self, plain):
        """this solve runs as ONE persistent launch (csrc/pcg_persist.hip)"""
        return self.two_launch and self.persist and not plain and not self.no_persist and self.N <= PERSIST_NODES

    def _csr(self, lin):
        """incidence lists sorted by node (shared with the assembly kernel, rebuilt only when the edge list changes)"""
        csr = lin.csr()
        if self._csr_obj is not csr:
            self._csr_obj, (self.ptr, self.blk, self.other) = csr, csr
            self.graph = None                                       # captured pointers are stale

    def _iteration(self, group):
        lib = _C.library()
        st = _C.stream_ptr(self.device)
        if self.bsr and self.two_launch:
            # q = A p with p.q, q.z, q.Binv q ; then every vector update in one launch (csrc/graph.hip, pcg2)
            # (stop: convergence test on the device -- a launch after the converging iteration returns at once)
            stop = self.stop_tol2 is not None and self.sym in (False, 'pack')
            if self.cz and stop and self.sym == 'pack':
                dp = "_dp" if self.dp else ""
                Dm, Bm = (self.Dp, self.Bp) if self.dp else (self.D, self.Binv)
                code = lib.symbol("pplie_pcg2_spmv_pack_coarse" + dp + self.sfx, _PCG2_SPMV_CZ_SIG)(
                    self.ptr.data_ptr(), self.other.data_ptr(), self.HB.data_ptr(), Dm.data_ptr(), Bm.data_ptr(),
                    self.p.data_ptr(), self.z.data_ptr(), self.q.data_ptr(), self.scal.data_ptr(), self.cs.data_ptr(),
                    self.rr_hist.data_ptr(), self.it.data_ptr(), self.cap, self.N, self.m, self.stop_tol2, st)
                _C.check(code, "pplie_pcg2_spmv_pack_coarse")
                code = lib.symbol("pplie_pcg2_step_coarse" + dp + self.sfx, _PCG2_STEP_CZ_SIG)(
                    self.x.data_ptr(), self.r.data_ptr(), self.r2.data_ptr(), self.p.data_ptr(), self.q.data_ptr(),
                    self.z.data_ptr(), Bm.data_ptr(), self.scal.data_ptr(), self.cs.data_ptr(), self.it.data_ptr(), self.N, self.m, st)
                _C.check(code, "pplie_pcg2_step_coarse")
                return
            if self.sym == 'pack':
                code = lib.symbol("pplie_pcg2_spmv_pack" + self.sfx, _PCG2_SPMV_STOP_SIG)(
                    self.ptr.data_ptr(), self.other.data_ptr(), self.HB.data_ptr(), self.D.data_ptr(), self.Binv.data_ptr(),
                    self.p.data_ptr(), self.z.data_ptr(), self.q.data_ptr(), self.scal.data_ptr(), self.rr_hist.data_ptr(),
                    self.it.data_ptr(), self.cap, self.N, self.m, self.stop_tol2 if stop else -1.0, st)
            elif stop:
                code = lib.symbol("pplie_pcg2_spmv_stop" + self.sfx, _PCG2_SPMV_STOP_SIG)(
                    self.ptr.data_ptr(), self.other.data_ptr(), self.HB.data_ptr(), self.D.data_ptr(), self.Binv.data_ptr(),
                    self.p.data_ptr(), self.z.data_ptr(), self.q.data_ptr(), self.scal.data_ptr(), self.rr_hist.data_ptr(),
                    self.it.data_ptr(), self.cap, self.N, self.m, self.stop_tol2, st)
            elif self.sym:
                code = lib.symbol("pplie_pcg2_spmv_sym" + self.sfx, [ctypes.c_void_p] + _PCG2_SPMV_SIG)(
                    self.ptr.data_ptr(), self.other.data_ptr(), self.blk.data_ptr(), self.HB.data_ptr(), self.D.data_ptr(),
                    self.Binv.data_ptr(), self.p.data_ptr(), self.z.data_ptr(), self.q.data_ptr(), self.scal.data_ptr(),
                    self.rr_hist.data_ptr(), self.it.data_ptr(), self.cap, self.N, self.m, st)
            else:
                code = lib.symbol("pplie_pcg2_spmv" + self.sfx, _PCG2_SPMV_SIG)(
                    self.ptr.data_ptr(), self.other.data_ptr(), self.HB.data_ptr(), self.D.data_ptr(), self.Binv.data_ptr(),
                    self.p.data_ptr(), self.z.data_ptr(), self.q.data_ptr(), self.scal.data_ptr(), self.rr_hist.data_ptr(),
                    self.it.data_ptr(), self.cap, self.N, self.m, st)
            _C.check(code, "pplie_pcg2_spmv")
            code = lib.symbol("pplie_pcg2_step" + ("_stop" if stop else "") + self.sfx, _PCG2_STEP_SIG)(
                self.x.data_ptr(), self.r.data_ptr(), self.r2.data_ptr(), self.p.data_ptr(), self.q.data_ptr(),
                self.z.data_ptr(), self.Binv.data_ptr(), self.scal.data_ptr(), self.it.data_ptr(), self.N, self.m, st)
            _C.check(code, "pplie_pcg2_step")
            return
        stage = lib.symbol("pplie_pcg_stage" + self.sfx, _PCG_SIG)
        if self.bsr:
            code = lib.symbol("pplie_graph_bsr_spmv" + self.sfx, _BSR_SIG)(
                self.ptr.data_ptr(), self.other.data_ptr(), self.HB.data_ptr(), self.D.data_ptr(),
                self.p.data_ptr(), self.q.data_ptr(), self.scal.data_ptr(), self.it.data_ptr(), self.N, self.m, st)
            _C.check(code, "pplie_graph_bsr_spmv")
            first = 1                                               # q = A p and p.q are already done
        else:
            self.q.zero_()
            code = lib.symbol("pplie_graph_spmv" + self.sfx, _SPMV_SIG)(
                self.J.data_ptr(), self.W.data_ptr() if self.W is not None else None, self.idx.data_ptr(), self.p.data_ptr(),
                self.q.data_ptr(), self.E, self.dr, self.m, self.K, st)
            _C.check(code, "pplie_graph_spmv")
            _all_reduce(self.q, group)
            first = 0
        for s in range(first, 3):
            code = stage(s, self.x.data_ptr(), self.r.data_ptr(), self.p.data_ptr(), self.q.data_ptr(), self.z.data_ptr(),
                         self.Binv.data_ptr(), self.shift.data_ptr(), self.scal.data_ptr(), self.rr_hist.data_ptr(),
                         self.it.data_ptr(), self.cap, self.N, self.m, st)
            _C.check(code, "pplie_pcg_stage")

    def _tail_args(self, tail):
        """the five pointers of csrc/pcg_persist.hip GhostTail in device memory (nodes, backup, gain partials, state, shift); kept per
        set of buffers: inside a captured graph the table must be the same memory at every replay"""
        pt, backup, partial, state = tail
        key = (pt.data_ptr(), 0 if backup is None else backup.data_ptr(), partial.data_ptr(), state.data_ptr(), self.shift.data_ptr())
        tabs = self.__dict__.setdefault('_tail_tabs', {})          # (every table stays alive: a captured graph may hold its address)
        tab = tabs.get(key)
        if tab is None:
            if torch.cuda.is_current_stream_capturing():           # (no host-to-device copy inside a capture: the two-launch tail runs)
                return None
            if len(tabs) >= 64:
                raise RuntimeError("pypose_amd: more than 64 sets of trial buffers on one solve workspace")
            gain = partial.data_ptr() + _PGO_PARTIALS * partial.element_size()       # (the tail's gain partials sit behind its loss partials)
            tab = tabs[key] = torch.tensor([key[0], key[1], gain, key[3], key[4]], dtype=torch.int64).to(self.device)
        return tab

    def _prepare_lap(self, lin, s, s_dev, dmin, dmax, coarse):
        """the set-up launch that also finishes the linearisation's assembly (its per-node sums are pending: pplie_pcg_prepare_lap);
        None when there is nothing pending -- the caller then runs the plain set-up on lin.B / lin.g"""
        pend = lin.__dict__.get('_diag_pending')
        if pend is None:
            return None
        lin._diag_pending = None
        ptr, gg = pend
        dp = self.dp and coarse
        return _C.library().symbol("pplie_pcg_prepare_lap" + self.sfx, _PREP_LAP_SIG)(
            ptr.data_ptr(), lin.HB.data_ptr(), gg.data_ptr(), 1 if lin.HB_pack else 0, lin._B.data_ptr(), lin._g.data_ptr(),
            self.D.data_ptr(), self.Binv.data_ptr(), self.Dp.data_ptr() if dp else None, self.Bp.data_ptr() if dp else None,
            self.shift.data_ptr(), self.x.data_ptr(), self.r.data_ptr(), self.z.data_ptr(), self.p.data_ptr(), self.scal.data_ptr(),
            self.cs.data_ptr() if coarse else None, float(s), s_dev, float(dmin), float(dmax), self.N, self.m, _C.stream_ptr(self.device))

    def _prepare_coarse(self, lin, s, s_dev, dmin, dmax):
        """pplie_pcg_prepare_coarse (+ the packed triangles of D / Binv when this solve's iteration reads those)"""
        code = self._prepare_lap(lin, s, s_dev, dmin, dmax, True)
        if code is not None:
            return code
        st = _C.stream_ptr(self.device)
        if self.dp:
            return _C.library().symbol("pplie_pcg_prepare_coarse_dp" + self.sfx, _PREP_CZ_DP_SIG)(
                lin.B.data_ptr(), lin.g.data_ptr(), self.D.data_ptr(), self.Binv.data_ptr(), self.Dp.data_ptr(), self.Bp.data_ptr(),
                self.shift.data_ptr(), self.x.data_ptr(), self.r.data_ptr(), self.z.data_ptr(), self.p.data_ptr(), self.scal.data_ptr(),
                self.cs.data_ptr(), float(s), s_dev, float(dmin), float(dmax), self.N, self.m, st)
        return _C.library().symbol("pplie_pcg_prepare_coarse" + self.sfx, _PREP_CZ_SIG)(
            lin.B.data_ptr(), lin.g.data_ptr(), self.D.data_ptr(), self.Binv.data_ptr(), self.shift.data_ptr(),
            self.x.data_ptr(), self.r.data_ptr(), self.z.data_ptr(), self.p.data_ptr(), self.scal.data_ptr(), self.cs.data_ptr(),
            float(s), s_dev, float(dmin), float(dmax), self.N, self.m, st)

    def solve(self, lin, s, dmin, dmax, tol, maxiter, group, plain=False, defer=False):
        """Solve (H + damping) x = -g for the linearisation ``lin`` (raw block diagonal ``lin.B``, gradient ``lin.g``)
        with the LM clamp [dmin, dmax] and compounded damping factor ``s`` folded in by ``pplie_pcg_prepare``.
        ``plain=True`` runs the same launches with the identity as preconditioner: from x = 0 plain CG stays in
        range(H) and converges to the minimum-norm solution of a singular H (Gauss-Newton's pseudo-inverse step)."""
        bsr = lin.HB is not None and group is None and self.m in (3, 6, 7)
        if bsr != self.bsr:
            self.graph = None                                       # the captured iteration differs
        self.bsr = bsr
        if bsr:
            self._csr(lin)
            sym = 'pack' if getattr(lin, "HB_pack", False) else bool(getattr(lin, "HB_sym", False))      # storage of the off-diagonal blocks
            if self._persistent(plain) and not sym:
                self.HB, self.sym = lin.HB, sym                     # the persistent solve reads the linearisation's blocks in place
            else:                                                   # captured iterations point at a buffer of the workspace
                own = self.__dict__.get('_own_HB')
                if own is None or own.shape != lin.HB.shape or own.dtype != lin.HB.dtype or sym != self.sym or self.HB is not own:
                    own = self._own_HB = (torch.empty((lin.HB.shape[0] + 1,) + tuple(lin.HB.shape[1:]), dtype=lin.HB.dtype, device=lin.HB.device)[:lin.HB.shape[0]]
                                             if own is None or own.shape != lin.HB.shape else own)      # (+1 record: see pplie_pcg2_spmv_pack)
                    self.HB, self.sym, self.graph = own, sym, None
                if lin.HB is not self.HB:                           # (assembled in place when the workspace already existed)
                    self.HB.copy_(lin.HB)                           # off-diagonal blocks in incidence (or edge) order
        else:
            if self.J is None:
                self.J, self.idx = torch.empty_like(lin.J), torch.empty_like(lin.idx)
                self.W = torch.empty_like(lin.W) if self.has_w else None
            self.J.copy_(lin.J)
            self.idx.copy_(lin.idx)
            if self.W is not None:
                self.W.copy_(lin.W)
        s_dev = getattr(lin, 's_dev', None)
        # (the fused pose-graph linearisation may have cleared the control block / fetched the damping factor in its own launch already)
        begun = lin.__dict__.pop('_begun', None)
        begun = begun is not None and begun[0] is self and begun[1] is s_dev and begun[2] == self.__dict__.get('_solves', 0)
        self._solves = self.__dict__.get('_solves', 0) + 1
        if s_dev is None and not begun:
            self._ctl.zero_()                                       # scal, cs, part (sequence tags restart at 1), it
        # the two-level preconditioner: relative-pose linearisations (J[e,0] = -J[e,1]) on the single-GPU block paths that have it
        # -- the ghost-zone persistent solve and the packed two-launch iteration with the device-side stop test
        persistent = bsr and self._persistent(plain) and not self.sym
        two = bsr and self.two_launch and not plain and group is None and self.sym == 'pack' and self.device_stop and not persistent
        cz = bool(FusedPCG.coarse and getattr(self, 'want_gauge', True) and getattr(lin, 'antisym', False) and not plain and group is None
                  and ((persistent and FusedPCG.ghost and not self.__dict__.get('_no_ghost')) or two))
        if cz != self.cz:
            self.cz, self.graph = cz, None                          # (the captured iterations differ)
        dp = bool(cz and two and FusedPCG.pack_diag)
        if dp and self.Dp is None:
            npd = self.m * (self.m + 1) // 2
            self.Dp, self.Bp = (torch.zeros((self.N + 1, npd), dtype=self.dtype, device=self.device)[:self.N] for _ in range(2))
        if dp != self.dp:
            self.dp, self.graph = dp, None
        with _C._on_device(self.device):
            # one launch: D = clamped + damped block diagonal, Binv = D^-1, shift, x = 0, r = -g, z = Binv r, p = z, r.z, |g|^2
            if s_dev is not None:
                # a captured trial: the damping factor of the day sits in a host-pinned scalar; the launch that clears the
                # control block also brings it into device memory (pplie_pcg_begin)
                if not begun:
                    code = _C.library().symbol("pplie_pcg_begin", _BEGIN_SIG)(
                        self._ctl.data_ptr(), self._ctl.numel(), s_dev.data_ptr(), self.s_device.data_ptr(), _C.stream_ptr(self.device))
                    _C.check(code, "pplie_pcg_begin")
                if self.cz and two:
                    # (a captured trial on a graph beyond the persistent solve: the two-launch iteration's coarse sums, the damping
                    #  factor from the device scalar pplie_pcg_begin has just filled)
                    code = self._prepare_coarse(lin, 1.0, self.s_device.data_ptr(), dmin, dmax)
                    _C.check(code, "pplie_pcg_prepare_coarse")
                    code = _C.library().symbol("pplie_pcg2_coarse_init" + self.sfx, _CZ_INIT_SIG)(
                        self.p.data_ptr(), self.cs.data_ptr(), self.N, self.m, _C.stream_ptr(self.device))
                else:
                    code = self._prepare_lap(lin, 1.0, self.s_device.data_ptr(), dmin, dmax, False)
                    if code is None:
                        code = _C.library().symbol("pplie_pcg_prepare_dev" + self.sfx, _PREP_DEV_SIG)(
                            lin.B.data_ptr(), lin.g.data_ptr(), self.D.data_ptr(), self.Binv.data_ptr(), self.shift.data_ptr(),
                            self.x.data_ptr(), self.r.data_ptr(), self.z.data_ptr(), self.p.data_ptr(), self.scal.data_ptr(),
                            self.s_device.data_ptr(), float(dmin), float(dmax), self.N, self.m, _C.stream_ptr(self.device))
            elif self.cz and two:
                # (the persistent solve sums E and Z^T r_0 itself, in its first exchange; the two-launch iteration gets them here)
                code = self._prepare_coarse(lin, float(s), None, dmin, dmax)
                _C.check(code, "pplie_pcg_prepare_coarse")
                code = _C.library().symbol("pplie_pcg2_coarse_init" + self.sfx, _CZ_INIT_SIG)(
                    self.p.data_ptr(), self.cs.data_ptr(), self.N, self.m, _C.stream_ptr(self.device))
            else:
                code = self._prepare_lap(lin, float(s), None, dmin, dmax, False)
                if code is None:
                    code = _C.library().symbol("pplie_pcg_prepare" + self.sfx, _PREP_SIG)(
                        lin.B.data_ptr(), lin.g.data_ptr(), self.D.data_ptr(), self.Binv.data_ptr(), self.shift.data_ptr(),
                        self.x.data_ptr(), self.r.data_ptr(), self.z.data_ptr(), self.p.data_ptr(), self.scal.data_ptr(),
                        float(s), float(dmin), float(dmax), self.N, self.m, _C.stream_ptr(self.device))
            _C.check(code, "pplie_pcg_prepare")
            if plain:                                               # z = r, p = r, rho = r.r = |b|^2, Binv = I
                self.Binv.copy_(torch.eye(self.m, dtype=self.Binv.dtype, device=self.Binv.device).expand_as(self.Binv))
                self.z.copy_(self.r)
                self.p.copy_(self.r)
                self.scal[0:1024].copy_(self.scal[3 * 1024:4 * 1024])
            if bsr and self._persistent(plain) and not self.sym:
                # the whole solve in one launch: iteration, reductions and the convergence test stay on the device
                maxit = min(maxiter, self.cap - 1)
                code = _C.ECAPACITY
                if FusedPCG.ghost and not self.__dict__.get('_no_ghost'):
                    # ghost-zone form: ONE grid-wide dependency per iteration (csrc/pcg_persist.hip); falls through to the
                    # two-dependency kernel when a workgroup's slice / ghost set does not fit.  Fewer, larger workgroups make the
                    # all-gather of the partial sums cheaper (5.6 us per iteration at 160-176 workgroups, 6.4 at 256, 10 k nodes):
                    # the smallest grid whose slices and ghost sets fit is kept for this workspace.
                    hit = self.__dict__.get('_ghost_grids')
                    if hit is None or hit[0] != (PERSIST_GRID, GHOST_GRIDS, self.cz) or hit[2] is not self._csr_obj:
                        # (the two-level variant's wide exchange goes in two levels from 224 workgroups on and is fastest there:
                        #  8.0 us per iteration at 256 against 8.4 at 160, profiles/r05/pcg_iter_gauge_pairs.json -- largest grid first)
                        cands = sorted({min(g, self.N) for g in GHOST_GRIDS if g < PERSIST_GRID} | {min(PERSIST_GRID, self.N)}, reverse=self.cz)
                        hit = self._ghost_grids = ((PERSIST_GRID, GHOST_GRIDS, self.cz), cands, self._csr_obj)
                    while hit[1]:
                        grid = hit[1][0]
                        slot, gptr, gids, max_cnt, max_ghost = self._ghost_map(lin, grid)
                        # (unweighted only: the gain terms are taken with the PLAIN J and R -- optimizer.py:670, strategy.py:128-140 -- and equal
                        #  d.(H d), -d.r_0 of the solve only while H = J^T J)
                        tail = lin.__dict__.get('_tail_in_solve') if self.m == 6 and not self.has_w and not FusedPCG.profile \
                            and FusedPCG.tail_in_solve else None
                        args = None if tail is None else self._tail_args(tail)
                        if args is not None:
                            # the LM trial's tail rides in the solve's epilogue (csrc/pcg_persist.hip GhostTail): gain terms + retraction
                            # by the workgroups that hold d and r; the trial then has ONE launch left (optim/pgograph.py TrialTail)
                            code = _C.library().symbol("pplie_pcg_ghost_tail" + self.sfx, _GHOST_TAIL_SIG)(
                                self.ptr.data_ptr(), slot.data_ptr(), self.HB.data_ptr(), self.D.data_ptr(), self.Binv.data_ptr(),
                                self.shift.data_ptr() if self.cz else None, self.x.data_ptr(), self.r.data_ptr(), self.z.data_ptr(),
                                gptr.data_ptr(), gids.data_ptr(), self.part.data_ptr(), self.ptag.data_ptr(), self.rr_hist.data_ptr(),
                                self.info.data_ptr(), self.it.data_ptr(), float(tol), int(maxit), self.cap, grid, max_cnt, max_ghost,
                                self.N, self.m, args.data_ptr(), _C.stream_ptr(self.device))
                            if code == 0:
                                lin._tail_done = grid                # (gain partials: one pair per workgroup of THIS grid)
                        elif self.cz:
                            code = _C.library().symbol("pplie_pcg_ghost_coarse" + self.sfx, _GHOST_CZ_SIG)(
                                self.ptr.data_ptr(), slot.data_ptr(), self.HB.data_ptr(), self.D.data_ptr(), self.Binv.data_ptr(),
                                self.shift.data_ptr(), self.x.data_ptr(), self.r.data_ptr(), self.z.data_ptr(), gptr.data_ptr(),
                                gids.data_ptr(), self.part.data_ptr(), self.ptag.data_ptr(), self.rr_hist.data_ptr(), self.info.data_ptr(),
                                self.it.data_ptr(), float(tol), int(maxit), -self.cap if FusedPCG.profile else self.cap, grid, max_cnt,
                                max_ghost, self.N, self.m, _C.stream_ptr(self.device))
                        else:
                            code = _C.library().symbol("pplie_pcg_ghost" + self.sfx, _GHOST_SIG)(
                                self.ptr.data_ptr(), slot.data_ptr(), self.HB.data_ptr(), self.D.data_ptr(), self.Binv.data_ptr(),
                                self.x.data_ptr(), self.r.data_ptr(), self.z.data_ptr(), gptr.data_ptr(), gids.data_ptr(), self.part.data_ptr(),
                                self.ptag.data_ptr(), self.rr_hist.data_ptr(), self.info.data_ptr(), self.it.data_ptr(), float(tol), int(maxit),
                                -self.cap if FusedPCG.profile else self.cap, grid, max_cnt, max_ghost, self.N, self.m, _C.stream_ptr(self.device))
                        if code != _C.ECAPACITY:
                            break
                        hit[1].pop(0)
                    if code == _C.ECAPACITY:
                        self._no_ghost = True
                if code == _C.ECAPACITY:
                    code = _C.library().symbol("pplie_pcg_persist" + self.sfx, _PERSIST_SIG)(
                    self.ptr.data_ptr(), self.other.data_ptr(), self.HB.data_ptr(), self.D.data_ptr(), self.Binv.data_ptr(),
                    self.x.data_ptr(), self.r.data_ptr(), self.p.data_ptr(), self.q.data_ptr(), self.z.data_ptr(),
                    self.part.data_ptr(), self.ptag.data_ptr(), self.rr_hist.data_ptr(), self.info.data_ptr(), self.it.data_ptr(),
                        float(tol), int(maxit), -self.cap if FusedPCG.profile else self.cap, PERSIST_GRID, self.N, self.m,
                        _C.stream_ptr(self.device))
                if code == _C.ECAPACITY:                            # this device cannot hold the solve resident: stream it instead
                    self.no_persist = True
                    return self.solve(lin, s, dmin, dmax, tol, maxiter, group, plain=plain, defer=defer)
                _C.check(code, "pplie_pcg_persist")
                if defer == 'inplace':                              # (a captured trial: the tail kernels read the workspace itself)
                    return self.x, _PendingInfo(self.info)
                if defer:
                    # the caller reads `info` together with the trial's loss and gain terms (ONE read-back per LM trial);
                    # a failed solve returns x = 0, so whatever was queued behind it left the parameters alone
                    return self.x.clone(), _PendingInfo(self.info.clone())
                its, rr, bn2, flag = self.info.tolist()             # the solve's one read-back
                if flag == 3.0:                                     # not all workgroups were resident: iterate with two launches
                    try:
                        _check_persist_flag(flag, rr)
                    except SolveFailed:
                        pass
                    return self.solve(lin, s, dmin, dmax, tol, maxiter, group, plain=plain, defer=False)
                assert flag != 2.0 and rr == rr, 'Linear solve produced NaN (matrix may not be positive-definite)'
                return self.x.clone(), int(its)
            bn2_slots = self.scal[3 * 1024:4 * 1024:32]             # |b|^2: set 0, quantity 3, 32 slots (csrc/graph.hip)
            bn2 = None
            maxiter = min(maxiter, self.cap - self.check_every)
            done, best, stalled, xbest = 0, float('inf'), 0, None
            if bsr and self.two_launch and not plain and group is None and self.sym in (False, 'pack') and self.device_stop:
                # The two-launch iteration with the convergence test ON THE DEVICE: chunks of `check_every` captured iterations
                # are queued `ahead` at a time per read-back of the 4-int control word; the launch that finds
                # |r|^2 <= tol^2 |b|^2 raises a flag and every later one returns at once, so the solve ends in the iteration
                # that converged (as the reference's CG does) however many chunks were queued.
                tol2 = float(tol) * float(tol)
                if self.stop_tol2 != tol2:
                    self.stop_tol2, self.graph = tol2, None        # (tol^2 is a launch argument of the captured chunk)
                if defer == 'inplace' and self.sym != 'pack':
                    # (a small graph whose persistent solve did not fit this device ends up here with full blocks and a capture sized
                    #  for nothing: no unwatched route -- the capture is abandoned once, as before there was one)
                    raise RuntimeError("the two-launch iteration on full blocks does not run unwatched")
                if defer == 'inplace':
                    # Inside the capture of a whole LM trial (optim/pgograph.py): nobody reads the control word between the iterations
                    # and the trial's tail.  `unwatched_iterations` launches of the pair are queued outright (those behind the converging
                    # one return at once), then pplie_pcg2_report tests the last one and leaves (iterations, |r|^2, |b|^2, flag) where
                    # the tail's pack kernel picks them up; flag 4 = not converged within what was queued: the host puts the parameters
                    # back and takes that step on the watched path below.
                    for _ in range(min(maxiter, self.unwatched_iterations)):
                        self._iteration(None)
                    code = _C.library().symbol("pplie_pcg2_report" + self.sfx, _PCG2_REPORT_SIG)(
                        self.scal.data_ptr(), self.rr_hist.data_ptr(), self.it.data_ptr(), self.cap, tol2, self.info.data_ptr(),
                        _C.stream_ptr(self.device))
                    _C.check(code, "pplie_pcg2_report")
                    return self.x, _PendingInfo(self.info)
                # (a schedule guessed from the previous solve's count loses more in no-op launches than it saves; with the two-level
                #  preconditioner a solve at the usual tolerances ends in 17 - 26 iterations: three chunks of eight, one read-back)
                ahead = 3 if self.cz else 2
                while done < maxiter:
                    for _ in range(ahead):
                        if self.graph is None and done > 0 and self.use_graph:
                            g = torch.cuda.CUDAGraph()
                            with _C.graph_capture(g):
                                for _ in range(self.check_every):
                                    self._iteration(None)
                            self.graph = g
                        if self.graph is not None:
                            self.graph.replay()
                        else:
                            for _ in range(self.check_every):
                                self._iteration(None)
                        done += self.check_every
                        if done >= maxiter:
                            break
                    its, _, flag, _ = self.it.tolist()
                    assert flag != 2, 'Linear solve produced NaN (matrix may not be positive-definite)'
                    if flag == 1:
                        self.iterations_seen = max(self.iterations_seen, int(its))
                        return self.x.clone(), int(its)
                # (ended at the iteration limit: a capture sized by a shorter solve would report "not converged" again and again --
                #  the count is recorded so that unwatched_for() sizes the next capture beyond it or declines: ADVICE r05)
                self.iterations_seen = max(self.iterations_seen, int(done))
                return self.x.clone(), done
            if self.stop_tol2 is not None:
                self.stop_tol2, self.graph = None, None
            while done < maxiter:
                if group is None and self.graph is None and done > 0 and self.use_graph:
                    g = torch.cuda.CUDAGraph()
                    with _C.graph_capture(g):
                        for _ in range(self.check_every):
                            self._iteration(None)
                    self.graph = g
                if self.graph is not None and group is None:
                    self.graph.replay()
                else:
                    for _ in range(self.check_every):
                        self._iteration(group)
                done += self.check_every
                if bsr and self.two_launch:     # |r|^2 of the last iteration still sits in its slot-spread accumulator
                    rr_src = self.scal.view(2, 8, 32, 32)[(done - 1) & 1, 2, :, 0]
                else:
                    rr_src = self.rr_hist[done - 1:done]
                if bn2 is None:                                     # first check: |b|^2 comes back with the residual norm
                    vals = torch.cat([rr_src, bn2_slots]).tolist()
                    k = rr_src.numel()
                    rr, bn2 = sum(vals[:k]), sum(vals[k:])
                    if bn2 == 0.0:
                        return self.x.clone(), 0
                else:
                    rr = sum(rr_src.tolist())
                assert rr == rr, 'Linear solve produced NaN (matrix may not be positive-definite)'
                if rr <= tol * tol * bn2:
                    return self.x.clone(), done
                if plain:               # singular H: past the rounding floor x drifts along the null space (PCG.solve)
                    if rr < best:
                        best, stalled, xbest = rr, 0, self.x.clone()
                    else:
                        stalled += 1
                        if stalled >= _STALL_CHECKS or rr > _DIVERGED ** 2 * best:
                            break
        return (xbest if plain and xbest is not None else self.x.clone()), done


class GraphOperator:
    """``J`` handed to ``strategy.update`` for the graph path: ``J @ D`` with D the flat step."""

    def __init__(self, lin):
        self.lin = lin
        self.replicated = lin.replicated

    def gain_terms(self, D):
        """device tensor [(J D).(J D), (J D).R] from one kernel (pplie_graph_gain_terms), or None off the HIP path"""
        lin = self.lin
        if not lin._hip():
            return None
        Dn = D.reshape(lin.N, lin.wfull)
        Dn = Dn if Dn.is_contiguous() else Dn.contiguous()
        part = torch.zeros((_GAIN_PARTIALS, 2), dtype=Dn.dtype, device=Dn.device)
        fn = _C.library().symbol("pplie_graph_gain_terms" + ("_f32" if Dn.dtype == torch.float32 else "_f64"), _GAIN_SIG)
        with _C._on_device(Dn.device):
            code = fn(lin.J.data_ptr(), lin.idx.data_ptr(), Dn.data_ptr(), lin.wfull, lin.R.data_ptr(), part.data_ptr(),
                      lin.E, lin.dr, lin.m, lin.K, _C.stream_ptr(Dn.device))
        _C.check(code, "pplie_graph_gain_terms")
        return part.sum(0)

    def __matmul__(self, D):
        lin = self.lin
        Dn = lin.step_to_nodes(D)                                  # [N, m]
        JD = torch.zeros((lin.E, lin.dr), dtype=Dn.dtype, device=Dn.device)
        for k in range(lin.K):
            JD += (lin.J[:, k] * Dn[lin.idx[:, k]].unsqueeze(-2)).sum(-1)     # (tiny batched GEMMs are slow)
        return JD.reshape(-1, 1)


class GraphLinearization:
    kind = "graph"

    def __init__(self, opt, weight, R, param, idx, J, wfull, m):
        """R [E,dr] (corrected), J [E,K,dr,m], idx [E,K], param the node Parameter [N,wfull]."""
        self.opt, self.param = opt, param
        self.R, self.J, self.idx = R.contiguous(), J.contiguous(), idx.contiguous()
        self.E, self.K, self.dr, self.m = J.shape
        self.N, self.wfull = param.shape[0], wfull
        self.W = weight
        self.group = getattr(opt, 'group', None)
        self.replicated = False  # True: the edges of ALL shards are held here, nothing below is a collective
        self.s = 1.0            # compounded damping factor prod(1 + lambda_i)
        self.HB = None
        self.pending_info = None  # a deferred read-back of the persistent PCG's (iterations, rr, bn2, flag)
        self.HB_sym = False     # True: HB holds one block per EDGE (symmetric weights, large graphs)
        self.HB_pack = False    # True: HB holds packed symmetric blocks per incidence (antisym linearisations, large graphs)
        self.antisym = False    # the builder's promise that J[e, 0] == -J[e, 1] (relative-pose residuals: optim/fused.py)
        self.node_group = None  # process group over which the SOLVE is sharded by node rows (optim/nodeshard.py)

    # -- index helpers -------------------------------------------------------------------------
    def step_to_nodes(self, D):
        return D.reshape(self.N, self.wfull)[:, :self.m]

    def nodes_to_step(self, Dn):
        if self.m < self.wfull:
            Dn = torch.cat([Dn, torch.zeros((self.N, self.wfull - self.m), dtype=Dn.dtype, device=Dn.device)], -1)
        return Dn.reshape(-1, 1)

    def csr(self):
        """(ptr [N+1], blk [K E] = K*edge + side, other [K E]) int32: the incidences of every node, sorted by
        node.  Cached on the optimizer; rebuilt only when the edge list changes."""
        cache = self.opt.__dict__.setdefault('_graph_csr', {})
        key = (self.E, self.K, self.N, self.idx.device)
        hit = cache.get(key)
        if hit is not None:
            if hit[2] is self.idx and hit[3] == self.idx._version:
                return hit[1]                                      # (same object, unmodified: no compare kernel / sync)
            if torch.equal(hit[0], self.idx):                      # an equal edge list in a new tensor: remember THAT object, so
                cache[key] = (hit[0], hit[1], self.idx, self.idx._version)     # that the next look-up is by identity again
                return hit[1]
        E, N, K, idx = self.E, self.N, self.K, self.idx
        flat = idx.t().reshape(-1)                                 # [side 0 entries | side 1 entries | ...]
        order = torch.argsort(flat, stable=True)
        ptr = torch.zeros(N + 1, dtype=torch.int32, device=idx.device)
        ptr[1:] = torch.cumsum(torch.bincount(flat, minlength=N), 0).to(torch.int32)
        e, side = order % E, order // E
        blk = (K * e + side).to(torch.int32)
        other = idx[e, 1 - side].to(torch.int32) if K == 2 else torch.zeros_like(blk)
        csr = (ptr, blk, other)
        cache[key] = (idx.clone(), csr, idx, idx._version)
        return csr

    def _weights_symmetric(self):
        """W_e == W_e^T for every edge (information matrices are; a user could pass anything): checked once per weight
        tensor version, one small reduction + read-back"""
        if self.W is None:
            return True
        cache = self.opt.__dict__.setdefault('_w_sym', {})
        src = getattr(self.W, '_pplie_src', None)
        if src is not None and isinstance(src[0], torch.Tensor):
            if src[2]:
                return True                                            # (Gauss-Newton: W^T W)
            if cache.get('src') is src[0] and cache.get('key') == (src[1], tuple(self.W.shape)):
                return cache['ok']
            cache['src'], cache['key'] = src[0], (src[1], tuple(self.W.shape))
            cache['ok'] = bool(torch.equal(self.W, self.W.mT))
            return cache['ok']
        key = (self.W.data_ptr(), self.W._version, tuple(self.W.shape))
        if cache.get('key') != key or cache.get('src') is not None:
            cache['src'], cache['key'], cache['ok'] = None, key, bool(torch.equal(self.W, self.W.mT))
        return cache['ok']

    def gauge_ok(self, solver=None):
        """The two-level (block-Jacobi + gauge) preconditioner applies: pairwise edges with J[e, 0] = -J[e, 1], i.e. a global
        motion Z = 1_N (x) I_m is in J's null space and E = Z^T A Z = diag(sum_n shift[n, :]).  Either the builder's promise
        (``antisym``: the recognised relative-pose program) or, for an autograd-derived J, a check of the blocks themselves
        (J_0 + J_1 = 0 up to rounding: any E > 0 keeps the preconditioner SPD, so rounding-level defects only cost iterations;
        a unary prior makes J Z != 0 and fails it).  Under edge shards the verdict is agreed over the group (MIN)."""
        if solver is not None and not getattr(solver, 'gauge', True):
            return False
        if not FusedPCG.coarse or self.K != 2:
            return False
        hit = self.__dict__.get('_gauge_ok')
        if hit is None:
            if self.antisym:
                hit = True
            else:
                J = self.J
                tol = float(torch.finfo(J.dtype).eps) ** 0.75
                hit = bool((J[:, 0] + J[:, 1]).abs().amax() <= tol * J.abs().amax()) if self.E else False
            if self.group is not None and not self.replicated:
                import torch.distributed as dist
                flag = torch.tensor([1.0 if hit else 0.0], device=self.J.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
                hit = bool(flag.item() > 0)
            self._gauge_ok = hit
        return hit

    # -- kernels ---------------------------------------------------------------------------------
    def _hip(self):
        return (_C._test_backend is None and self.J.is_cuda and (self.dr, self.m, self.K) in _HIP_SHAPES
                and self.J.dtype in (torch.float32, torch.float64))

    def incidence_slots(self):
        """inc [E, K] int32: the position of (edge, side) in the node-sorted incidence list of :meth:`csr` (the inverse permutation
        of its ``blk``); cached with it"""
        ptr, blk, _ = self.csr()
        cache = self.opt.__dict__.setdefault('_graph_inc', {})
        hit = cache.get('inc')
        if hit is None or hit[0] is not blk:
            inc = torch.empty_like(blk)
            inc[blk.long()] = torch.arange(blk.numel(), dtype=blk.dtype, device=blk.device)
            hit = cache['inc'] = (blk, inc.reshape(self.E, self.K).contiguous())
        return hit[1]

    def plan_blocks(self):
        """The storage decisions of the single-GPU block assembly, taken before any launch (idempotent): the layout of the off-diagonal
        blocks (``HB_pack`` / ``HB_sym``), the buffer ``HB`` they go to and -- returned -- the plan of the two-launch "Laplacian"
        assembly when this linearisation takes it ({'gg': the gradient shares' scratch}), else None."""
        plan = self.__dict__.get('_block_plan', False)
        if plan is not False:
            return plan
        N, m, dt, dev = self.N, self.m, self.J.dtype, self.J.device
        # off-diagonal blocks in incidence order feed the streaming node-parallel SpMV of the PCG; graphs too
        # large for the persistent solve keep ONE block per edge (H_ji = H_ij^T for symmetric weights):
        # half the bytes every SpMV streams
        large = self.K == 2 and FusedPCG.two_launch and not (FusedPCG.persist and N <= PERSIST_NODES)
        # J_0 = -J_1 and W symmetric: every off-diagonal block is -J_1^T W J_1, symmetric, the same for both incidences
        self.HB_pack = (large and self.antisym and FusedPCG.pack_blocks and self.dr == self.m
                        and self._weights_symmetric())
        self.HB_sym = large and not self.HB_pack and FusedPCG.sym_blocks and self._weights_symmetric()
        mode = 'pack' if self.HB_pack else self.HB_sym
        self.HB = None
        if self.K == 2:
            shape = (self.E * 2, m * (m + 1) // 2) if self.HB_pack else (self.E * (1 if self.HB_sym else 2), m, m)
            pad = 1 if self.HB_pack else 0      # (the packed SpMV reads up to m - 1 elements past a triangle's row)
            # large graphs: assemble straight into the PCG workspace's block buffer (the captured iterations point
            # at it) instead of into a fresh tensor that is then copied there -- 115 MB per LM step at 4e5 edges
            for w in (self.opt.__dict__.get('_pcg_workspaces') or {}).values():
                own = w.__dict__.get('_own_HB')
                if own is not None and own.shape == shape and own.dtype == dt and own.device == dev and w.sym == mode:
                    self.HB = own
                    break
            if self.HB is None:
                self.HB = torch.empty((shape[0] + pad,) + tuple(shape[1:]), dtype=dt, device=dev)[:shape[0]]
        plan = None
        if (self.K == 2 and self.antisym and self.dr == self.m and not self.HB_sym and FusedPCG.lap_assembly
                and self.m in (3, 6, 7) and self._weights_symmetric()):
            plan = {'gg': torch.empty((self.E * 2, m), dtype=dt, device=dev)}
        self._block_plan = plan
        return plan

    def _assemble(self):
        N, m = self.N, self.m
        dt, dev = self.J.dtype, self.J.device
        if self._hip():
            sfx = "_f32" if dt == torch.float32 else "_f64"
            lib, st = _C.library(), _C.stream_ptr(dev)
            wptr = self.W.data_ptr() if self.W is not None else None
            csr_ok = self.group is None and self.E * self.K < (1 << 31)
            with _C._on_device(dev):
                if csr_ok:      # node-parallel: every block / gradient row written once, no atomics, no zero-fill
                    B = torch.empty((N, m, m), dtype=dt, device=dev)
                    g = torch.empty((N, m), dtype=dt, device=dev)
                    ptr, blk, _ = self.csr()
                    lap = self.plan_blocks()
                    if lap is not None:
                        # J_0 = -J_1: incidence-parallel blocks, then per-node sums (csrc/graph.hip, pplie_graph_assemble_lap)
                        if lap.get('blocks_done') and FusedPCG.fuse_prepare and not self.__dict__.get('_eager_diag'):
                            # (the per-node sums ride in the solve's set-up launch, pplie_pcg_prepare_lap -- or run on the first
                            #  read of B / g by anybody else: _materialize_diag)
                            self._diag_pending = (ptr, lap['gg'])
                        elif lap.get('blocks_done'):     # (the fused linearisation left HB and gg already: optim/fused.py)
                            code = lib.symbol("pplie_graph_lap_diag" + sfx, _LAPD_SIG)(
                                ptr.data_ptr(), self.HB.data_ptr(), lap['gg'].data_ptr(), B.data_ptr(), g.data_ptr(), N, self.m,
                                1 if self.HB_pack else 0, st)
                            _C.check(code, "pplie_graph_lap_diag")
                        else:
                            code = lib.symbol("pplie_graph_assemble_lap" + sfx, _ASML_SIG)(
                                ptr.data_ptr(), blk.data_ptr(), self.J.data_ptr(), wptr, self.R.data_ptr(), B.data_ptr(), g.data_ptr(),
                                self.HB.data_ptr(), lap['gg'].data_ptr(), N, self.E * 2, self.m, 1 if self.HB_pack else 0, st)
                            _C.check(code, "pplie_graph_assemble_lap")
                    else:
                        code = lib.symbol("pplie_graph_assemble_csr" + ("_pack" if self.HB_pack else "_sym" if self.HB_sym else "") + sfx, _ASMC_SIG)(
                            ptr.data_ptr(), blk.data_ptr(), self.J.data_ptr(), wptr, self.R.data_ptr(), B.data_ptr(),
                            g.data_ptr(), self.HB.data_ptr() if self.HB is not None else None, N, self.dr, self.m, self.K, st)
                        _C.check(code, "pplie_graph_assemble_csr")
                else:           # edge shards (group=): scatter-add, then all-reduce
                    B = torch.zeros((N, m, m), dtype=dt, device=dev)
                    g = torch.zeros((N, m), dtype=dt, device=dev)
                    code = lib.symbol("pplie_graph_assemble" + sfx, _ASM_SIG)(
                        self.J.data_ptr(), wptr, self.R.data_ptr(), self.idx.data_ptr(), B.data_ptr(), g.data_ptr(),
                        None, self.E, self.dr, self.m, self.K, st)
                    _C.check(code, "pplie_graph_assemble")
        else:
            B = torch.zeros((N, m, m), dtype=dt, device=dev)
            g = torch.zeros((N, m), dtype=dt, device=dev)
            for k in range(self.K):
                Jk = self.J[:, k]
                JtW = Jk.mT if self.W is None else Jk.mT @ self.W
                B.index_add_(0, self.idx[:, k], JtW @ Jk)
                g.index_add_(0, self.idx[:, k], (JtW @ self.R.unsqueeze(-1)).squeeze(-1))
        return _all_reduce(B, self.group), _all_reduce(g, self.group)

    def _Hp(self, p):
        """(J^T W J) p for a node vector p [N, m] (all-reduced over edge shards)."""
        y = torch.zeros_like(p)
        if self._hip():
            sfx = "_f32" if p.dtype == torch.float32 else "_f64"
            fn = _C.library().symbol("pplie_graph_spmv" + sfx, _SPMV_SIG)
            with _C._on_device(p.device):
                code = fn(self.J.data_ptr(), self.W.data_ptr() if self.W is not None else None, self.idx.data_ptr(),
                          p.data_ptr(), y.data_ptr(), self.E, self.dr, self.m, self.K, _C.stream_ptr(p.device))
            _C.check(code, "pplie_graph_spmv")
        else:
            q = torch.zeros((self.E, self.dr), dtype=p.dtype, device=p.device)
            for k in range(self.K):
                q += (self.J[:, k] * p[self.idx[:, k]].unsqueeze(-2)).sum(-1)
            if self.W is not None:
                q = (self.W @ q.unsqueeze(-1)).squeeze(-1)
            for k in range(self.K):
                y.index_add_(0, self.idx[:, k], (self.J[:, k] * q.unsqueeze(-1)).sum(-2))
        return _all_reduce(y, self.group)

    # -- LM interface ------------------------------------------------------------------------------
    def build_normal_equations(self, dmin, dmax):
        self.dmin, self.dmax = float(dmin), float(dmax)
        self.s = 1.0
        if self.node_group is not None:
            from . import nodeshard as _ns
            cache = self.opt.__dict__.setdefault('_node_shards', {})
            csr = self.csr()
            hit = cache.get('shard')
            if hit is None or hit[0] is not csr:                    # (the csr object changes exactly when the edge list does)
                hit = cache['shard'] = (csr, _ns.NodeShard(self, self.node_group))
            self.ns = _ns.NodeShardedSystem(self, hit[1])
            self.ns.assemble()
            return
        self.B, self.g = self._assemble()
        # the parameter components outside the tangent space (7th of SE3, ...) have a structurally
        # zero Jacobian column: the reference clamps their diagonal to ``min`` and solves d = 0.
        self.s = 1.0

    # B (raw block diagonal [N, m, m]) and g (gradient [N, m]): attributes, except that the "Laplacian" assembly behind a fused
    # linearisation leaves their per-node sums to the solve's set-up launch -- whoever reads them first otherwise runs the sums
    @property
    def B(self):
        self._materialize_diag()
        return self._B

    @B.setter
    def B(self, v):
        self._B = v

    @property
    def g(self):
        self._materialize_diag()
        return self._g

    @g.setter
    def g(self, v):
        self._g = v

    def _materialize_diag(self):
        pend = self.__dict__.get('_diag_pending')
        if pend is None:
            return
        self._diag_pending = None
        ptr, gg = pend
        sfx = "_f32" if self.J.dtype == torch.float32 else "_f64"
        with _C._on_device(self.J.device):
            code = _C.library().symbol("pplie_graph_lap_diag" + sfx, _LAPD_SIG)(
                ptr.data_ptr(), self.HB.data_ptr(), gg.data_ptr(), self._B.data_ptr(), self._g.data_ptr(), self.N, self.m,
                1 if self.HB_pack else 0, _C.stream_ptr(self.J.device))
        _C.check(code, "pplie_graph_lap_diag")

    @property
    def diag_raw(self):
        return self.B.diagonal(dim1=-2, dim2=-1)

    @property
    def diag_clamped(self):
        return self.diag_raw.clamp(self.dmin, self.dmax)

    def damp(self, damping):
        self.s = self.s * (1.0 + damping)

    def solve(self, solver):
        return self.nodes_to_step(self.solve_nodes(solver))

    def solve_nodes(self, solver):
        """the step per node, [N, m] (``solve`` pads it to the parameter's width: a fill and a concatenation that only the
        generic update and a rejected trial's way back need)"""
        N, m = self.N, self.m
        if self.node_group is not None:
            pcg = solver if isinstance(solver, PCG) else PCG(tol=1e-10, maxiter=max(1000, 2 * N))
            maxiter = N * m * 10 if pcg.maxiter is None else pcg.maxiter
            Dn, its = self.ns.solve(self.s, self.dmin, self.dmax, pcg.tol, maxiter, pcg.check_every, gauge=self.gauge_ok(pcg))
            solver.iterations = its
            return Dn
        if not isinstance(solver, PCG) and N * m <= DENSE_LIMIT and self.group is None:
            shift = self.s * self.diag_clamped - self.diag_raw       # A = H + diag(shift)
            A = self.dense_matrix()
            A.diagonal().add_(shift.reshape(-1))
            Dn = solver(A=A, b=(-self.g).reshape(-1, 1)).reshape(N, m)
            assert not torch.any(torch.isnan(Dn)), 'Linear solve produced NaN (matrix may not be positive-definite)'
        else:
            Dn = self._pcg(solver, self.s, self.dmin, self.dmax, plain=False)     # (checks its residual norm for NaN)
        self._last_Dn = Dn                         # (what a fused trial tail takes)
        return Dn

    def _pcg(self, solver, s, dmin, dmax, plain):
        """(H + damping) d = -g by the matrix-free conjugate gradient (block-Jacobi preconditioned unless ``plain``)."""
        N, m = self.N, self.m
        if not isinstance(solver, PCG):
            if not getattr(self.opt, '_warned_pcg', False):
                warnings.warn(f"{type(solver).__name__} cannot factor a {N * m}-unknown pose graph densely; "
                              f"using the matrix-free {'conjugate gradient' if plain else 'block-Jacobi PCG'} (tol 1e-10) instead.")
                self.opt._warned_pcg = True
            solver = PCG(tol=1e-10, maxiter=max(1000, 2 * N))
        maxiter = N * m * 10 if solver.maxiter is None else solver.maxiter
        # (edge shards with an all-reduced H p -- LM(group=) with replicate_solve off: the fused launches of that route carry the
        #  block-Jacobi preconditioner only; with the gauge correction asked for and applicable the iteration below runs it)
        sharded_gauge = self.group is not None and not self.replicated and not plain and self.gauge_ok(solver)
        if self._hip() and getattr(solver, 'fused', True) and m in (3, 6, 7) and not sharded_gauge:
            cache = self.opt.__dict__.setdefault('_pcg_workspaces', {})
            key = (self.E, self.K, self.dr, self.m, self.N, self.J.dtype, self.J.device, self.W is not None,
                   solver.check_every)
            wsp = cache.get(key)
            if wsp is None:
                wsp = cache[key] = FusedPCG(*key)
            defer = getattr(self.opt, '_defer_solver_info', False)
            defer = False if plain else (defer if defer == 'inplace' else bool(defer))
            wsp.want_gauge = bool(getattr(solver, 'gauge', True))
            Dn, its = wsp.solve(self, s, dmin, dmax, solver.tol, maxiter, self.group, plain=plain, defer=defer)
            if isinstance(its, _PendingInfo):
                self.pending_info, self._pending_solver = its, solver
            else:
                solver.iterations = its
            return Dn
        clamped = self.diag_raw.clamp(dmin, dmax)
        shift = s * clamped - self.diag_raw
        if plain:
            precond = lambda r: r.clone()
        else:
            Bd = self.B.clone()
            Bd.diagonal(dim1=-2, dim2=-1).copy_(s * clamped)
            Binv = torch.linalg.inv(Bd)
            precond = lambda r: (Binv * r.unsqueeze(-2)).sum(-1)
            if self.gauge_ok(solver):
                # + Z E^-1 Z^T (csrc/pcg_persist.hip "CZ"): r is whole on every rank here (edge shards all-reduce H p), no collective
                E = shift.sum(0)
                Einv = torch.where(E > 0, 1.0 / E.clamp_min(torch.finfo(E.dtype).tiny), torch.zeros_like(E))
                local = precond
                precond = lambda r: local(r) + r.sum(0) * Einv
        Dn = solver.solve(lambda p: self._Hp(p) + shift * p, -self.g, precond, stall=_STALL_CHECKS if plain else None)
        assert not torch.any(torch.isnan(Dn)), 'Linear solve produced NaN (matrix may not be positive-definite)'
        return Dn

    def dense_matrix(self):
        """H = J^T W J as a dense [N m, N m] matrix (small graphs / parity tests)."""
        N, m = self.N, self.m
        A = torch.zeros((N * N, m, m), dtype=self.J.dtype, device=self.J.device)
        for k in range(self.K):
            JtW = self.J[:, k].mT if self.W is None else self.J[:, k].mT @ self.W
            for l in range(self.K):
                A.index_add_(0, self.idx[:, k] * N + self.idx[:, l], JtW @ self.J[:, l])
        return A.view(N, N, m, m).permute(0, 2, 1, 3).reshape(N * m, N * m).contiguous()

    def solve_gauss_newton(self, solver):
        """The reference solves the rectangular ``W J d = -W R`` with its solver (pseudo-inverse by default,
        optimizer.py:318-326): the least-squares step of minimum norm.  That is the minimum-norm solution of the
        normal equations ``J^T (W^T W) J d = -J^T (W^T W) R`` (this linearisation was built with ``W^T W``), which
        unpreconditioned CG started at zero converges to -- also on gauge-free graphs, where H is singular."""
        self.B, self.g = self._assemble()
        inf = float('inf')
        Dn = self._pcg(solver, 1.0, -inf, inf, plain=True)
        assert not torch.any(torch.isnan(Dn)), 'Linear solve produced NaN'
        return self.nodes_to_step(Dn)

    def strategy_args(self):
        return GraphOperator(self), self.R.reshape(-1, 1)


def gauss_newton_on_graph(opt, params):
    """Gauss-Newton takes the graph path when the user asks for an iterative solver or a dense J is out of reach."""
    return len(params) == 1 and (isinstance(opt.solver, PCG) or params[0].numel() > DENSE_LIMIT)


def try_graph_linearization(opt, pg, input, target, weight, R, params, rec, cache, sig, gauss_newton=False):
    """Build a GraphLinearization if the recorded gathers explain the whole Jacobian.

    One residual: every gather feeds it.  Several residuals (``model`` returns a tuple -- e.g. odometry edges, loop
    closures under their own robust kernel, unary priors; reference optimizer.py:644-654 handles them as stacked
    rows of one dense J): each is attributed the gathers it depends on, gets its own corrector / weight
    (``corrector[i]``, ``weight[i]``), and the per-edge terms are stacked with zero padding up to the largest
    residual width and gather count (a zero row / zero block adds nothing to J^T W J or J^T W r)."""
    if len(params) != 1 or params[0].dim() != 2 or any(r.dim() < 2 for r in R) \
            or len(opt.corrector) not in (1, len(R)):
        cache[sig] = False
        return None
    if weight is not None and len(R) > 1 and not (isinstance(weight, (tuple, list)) and len(weight) == len(R)):
        cache[sig] = False
        return None
    param = params[0]
    wfull = param.shape[-1]
    events = list(rec.events)
    if not events or any(ev[0] is not events[0][0] or ev[2].numel() != ev[1].numel() * wfull for ev in events):
        cache[sig] = False
        return None
    outs_all = [out for _, _, out in events]
    parts = []
    with torch.enable_grad():
        back = _rows_of_parameter(events[0][0], param)
        if back is None:
            cache[sig] = False
            return None
        for r in R:
            E, dr = r.numel() // r.shape[-1], r.shape[-1]
            if len(R) == 1:
                mine = events
            else:       # which gathers does this residual see?
                seen = torch.autograd.grad(r.sum(), outs_all, retain_graph=True, allow_unused=True)
                mine = [ev for ev, g in zip(events, seen) if g is not None]
            if not mine or any(ix.numel() != E for _, ix, _ in mine):
                cache[sig] = False
                return None
            node_idx = [back[ix.reshape(-1)] for _, ix, _ in mine]       # parameter row per edge end (-1: fixed)
            Jcat = _blocks.jacobian_blocks([r], [out for _, _, out in mine])     # [E, dr, K*wfull]
            for k, ni in enumerate(node_idx):                            # a fixed end contributes no unknowns
                Jcat[:, :, k * wfull:(k + 1) * wfull] *= (ni >= 0).to(Jcat.dtype).view(E, 1, 1)
            parts.append((r, E, dr, Jcat, [ni.clamp_min(0) for ni in node_idx]))
        if cache.get(sig) is None:
            # probe: u^T dR/dnodes by one real backward == scatter-add of the per-edge blocks
            us = [torch.randn_like(r) for r in R]
            true = torch.autograd.grad(list(R), [param], us, retain_graph=True)[0]
            got = torch.zeros_like(true)
            for u, (r, E, dr, Jcat, node_idx) in zip(us, parts):
                contrib = (u.reshape(E, dr).unsqueeze(-1) * Jcat).sum(-2)
                for k, ni in enumerate(node_idx):
                    got.index_add_(0, ni, contrib[:, k * wfull:(k + 1) * wfull])
            scale = true.abs().max().clamp_min(torch.finfo(true.dtype).tiny)
            cache[sig] = bool((got - true).abs().max() <= 1e-3 * scale)
    if not cache[sig]:
        return None
    # tangent width: gradients of LieTensor group parameters are zero-padded to the embedding
    m = _blocks.lie_manifold_width(param) or wfull
    terms = []
    for i, (r, E, dr, Jcat, node_idx) in enumerate(parts):
        K = len(node_idx)
        J = Jcat.reshape(E, dr, K, wfull)[..., :m].permute(0, 2, 1, 3)    # [E, K, dr, m]
        corrector = opt.corrector[0] if len(opt.corrector) == 1 else opt.corrector[i]
        w = weight[i] if isinstance(weight, (tuple, list)) else weight
        terms.append(_edge_terms(opt, corrector, w, r.detach().reshape(E, dr), J, gauss_newton) + (torch.stack(node_idx, dim=-1),))
    Rc, Jc, Wb, idx = terms[0] if len(terms) == 1 else _stack_edge_terms(terms)
    return _finish_graph_linearization(opt, Rc, Jc, idx, Wb, param, wfull, m)


def _stack_edge_terms(terms):
    """Residuals of different width / gather count as one edge list: zero-pad to (max dr, max K)."""
    dr = max(t[0].shape[-1] for t in terms)
    K = max(t[1].shape[1] for t in terms)
    weighted = any(t[2] is not None for t in terms)
    Rs, Js, Ws, Is = [], [], [], []
    for Rc, Jc, Wb, idx in terms:
        E, k, d, m = Jc.shape
        R2 = Rc.new_zeros((E, dr))
        R2[:, :d] = Rc
        J2 = Jc.new_zeros((E, K, dr, m))
        J2[:, :k, :d] = Jc
        I2 = idx.new_zeros((E, K))
        I2[:, :k] = idx
        Rs.append(R2), Js.append(J2), Is.append(I2)
        if weighted:
            W2 = torch.eye(dr, dtype=Jc.dtype, device=Jc.device).repeat(E, 1, 1)
            if Wb is not None:
                W2[:, :d, :d] = Wb
            Ws.append(W2)
    return torch.cat(Rs), torch.cat(Js), (torch.cat(Ws) if weighted else None), torch.cat(Is)


REPLICATE_LIMIT = 8 << 30   # bytes of gathered Jacobian blocks up to which edge shards solve on every rank (below)


def _gather_edge_shards(group, tensors, limit):
    """All-gather the per-edge tensors of every rank's shard (first dimension = local edge count, padded with zero
    rows to the largest shard: a zero block / zero residual contributes nothing anywhere).  Returns None when the
    gathered blocks would exceed ``limit`` bytes -- decided from the exchanged counts, so identically on all ranks."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    ref = next(t for t in tensors if t is not None)
    count = torch.tensor([ref.shape[0]], dtype=torch.int64, device=ref.device)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    most = int(torch.cat(counts).max())
    if world * most * sum(t[0].numel() * t.element_size() for t in tensors if t is not None) > limit:
        return None
    out = []
    for t in tensors:
        if t is None:
            out.append(None)
            continue
        mine = t.new_zeros((most,) + tuple(t.shape[1:]))
        mine[:t.shape[0]] = t
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        out.append(torch.cat(parts, 0))
    return out


def Trivial_type():
    from .optimizer import Trivial
    return Trivial


def _fused_rows(corrector, r, J):
    """FastTriggs / Triggs with a built-in kernel: sqrt(rho') is one scalar per edge, so the blocks are scaled where they lie
    (pplie_robust_scale_rows, csrc/robust.hip); None -> the corrector's own formulation"""
    from .corrector import fused_code, fused_scale_rows
    if fused_code(corrector) is None or not J.is_contiguous() or J.requires_grad:
        return None
    return fused_scale_rows(corrector.kernel, r, J, inplace=True)


def _edge_terms(opt, corrector, weight, r, J, gauss_newton=False):
    """Corrector and weight of ONE residual applied to its per-edge residuals r [E,dr] and blocks J [E,K,dr,m]:
    (Rc, Jc, Wb or None).  Gauss-Newton weights both sides of its rectangular system with W (optimizer.py:318-322),
    so its normal equations carry ``W^T W`` where Levenberg-Marquardt's ``J^T W J`` carries ``W``."""
    E, K, dr, m = J.shape
    from .optimizer import Trivial
    if isinstance(corrector, Trivial):          # no kernel: skip the [E, dr, K*m] round trip (two copies of J)
        Rc, Jc = r, J
    elif (fused := _fused_rows(corrector, r, J)) is not None:
        Rc, Jc = fused                          # built-in kernel: one launch, J scaled in place in its own layout
    else:
        Rc, Jc = corrector(R=r, J=J.permute(0, 2, 1, 3).reshape(E, dr, K * m))       # row-local: acts on [E, dr, K*m]
        Jc = Jc.reshape(E, dr, K, m).permute(0, 2, 1, 3)
    Wb = None
    if weight is not None:
        ws, ni = opt.model._weight_blocks(weight, r)
        Wb = ws.repeat(ni, 1, 1).contiguous()
        if gauss_newton:
            Wb = (Wb.mT @ Wb).contiguous()
        # (what the symmetry check of GraphLinearization._weights_symmetric is remembered by: Wb is a new tensor every step)
        Wb._pplie_src = (weight, weight._version if isinstance(weight, torch.Tensor) else None, bool(gauss_newton))
    return Rc, Jc, Wb


def build_graph_linearization(opt, weight, r, J, idx, param, wfull, m, gauss_newton=False, corrector=None):
    """Single-residual entry (also used by the fused pose-graph program): r [E,dr], J [E,K,dr,m] -> GraphLinearization.
    ``corrector``: overrides opt.corrector[0] (the fused program passes Trivial when its kernel already applied the weighting)."""
    w = weight[0] if isinstance(weight, (tuple, list)) else weight
    Rc, Jc, Wb = _edge_terms(opt, opt.corrector[0] if corrector is None else corrector, w, r, J, gauss_newton)
    return _finish_graph_linearization(opt, Rc, Jc, idx, Wb, param, wfull, m)


def resolve_shard_mode(opt, group, n_nodes, pairwise):
    """("edges" | "nodes", "rccl" | "p2p") for one pose graph under LM(group=...): the caller's explicit choice, or -- the default --
    a decision from group-uniform facts only (backend, world size, node count), so that every rank issues the same collectives.
    Node rows are sharded when the ranks are GPUs (RCCL) and the graph is beyond the capacity of ONE GPU's persistent solve
    (PERSIST_NODES: measured, 5.6 us / iteration up to there; 32 us two-launch iterations beyond): each rank's slice then fits a
    persistent launch again.  Below that size a solve is one launch on every rank already and sharding it only adds exchanges."""
    import torch.distributed as dist
    shard, exchange = getattr(opt, 'shard', None), getattr(opt, 'exchange', None)
    device_group = dist.get_backend(group) == "nccl"
    if shard is None:
        shard = "nodes" if (device_group and pairwise and dist.get_world_size(group) > 1 and n_nodes > PERSIST_NODES) else "edges"
    if exchange is None:
        # Node shards exchange over RCCL collectives by default (one all-gather + one all-reduce per PCG iteration): the route whose
        # every collective is a stock RCCL call.  The in-kernel peer exchange ("p2p": hipIpc-mapped tables, system-scope stores over
        # xGMI, no collective per iteration) is the faster design on paper but has only ever run with its ranks as processes on ONE
        # GPU (no multi-GPU box was available to build it on): it is opt-in -- LM(exchange="p2p") or PPLIE_EXCHANGE=p2p -- until
        # it has crossed a real link.  (Round 6; rounds 3-5 defaulted to it.)
        env = _os.environ.get("PPLIE_EXCHANGE")
        exchange = env if (env in ("rccl", "p2p") and shard == "nodes" and device_group) else "rccl"
    opt._shard_eff, opt._exchange_eff = shard, exchange
    return shard, exchange


def _finish_graph_linearization(opt, Rc, Jc, idx, Wb, param, wfull, m):
    # Edge shards (LM(group=...)).  The linear solve dominates a pose-graph step and is latency-bound when every PCG
    # iteration carries an all-reduce, so while the blocks of all shards fit on one GPU they are gathered ONCE per
    # LM step -- the J^T J / J^T r accumulators of every rank are then built from the same blocks -- and each rank
    # runs the un-sharded solve (streaming SpMV, graph-captured iterations, no collective).  Larger problems keep
    # the blocks distributed and all-reduce diag / gradient once per step and H p once per iteration.
    group = getattr(opt, 'group', None)
    nodes = group is not None and resolve_shard_mode(opt, group, param.shape[0], Jc.shape[1] == 2)[0] == 'nodes' and Jc.shape[1] == 2
    if group is not None and (getattr(opt, 'replicate_solve', True) or nodes):
        gathered = _gather_edge_shards(group, [Rc.contiguous(), Jc.contiguous(), idx.contiguous(), Wb], REPLICATE_LIMIT)
        if gathered is not None:
            lin = GraphLinearization(opt, gathered[3], gathered[0], param, gathered[2], gathered[1], wfull, m)
            lin.group, lin.replicated = None, True
            # shard="nodes": every rank holds the blocks of all edges (gathered once per LM step, above) but assembles
            # and solves only the node rows it owns -- optim/nodeshard.py
            lin.node_group = group if nodes else None
            opt._last_shard_mode = f"node-sharded solve ({opt._exchange_eff} exchange)" if nodes else "replicated solve"
            return lin
    if group is not None:
        opt._last_shard_mode = "edge-sharded (all-reduce per PCG iteration)"
    return GraphLinearization(opt, Wb, Rc, param, idx, Jc, wfull, m)
