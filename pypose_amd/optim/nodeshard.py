"""Pose-graph LM with the LINEAR SOLVE sharded over the ranks by node rows (``LM(group=..., shard="nodes")``).

SURVEY.md section 8(e) row 3 / BASELINE configs[3] ("J^T J sharded 8 GPUs via RCCL/xGMI").  The edge-sharded modes of
optim/posegraph.py either replicate the solve on every rank (no speed-up of the dominant cost) or all-reduce a whole
node vector per PCG iteration (every rank still touches all N rows).  Here each rank OWNS a contiguous range of node rows
of the normal equations:

  once per LM step   every rank linearises its edge shard; the per-edge residuals / Jacobian blocks are all-gathered
                     (E (6 + 72) floats + indices; each rank then keeps the incidences of its own rows) -- the only
                     volume-bound collective, 131 MB at configs[3], ~0.1 ms over 7 xGMI links;
  assembly           diagonal blocks, gradient and off-diagonal blocks of the OWNED rows only: complete locally, no
                     collective (an edge cut by the partition is simply seen by both owners);
  PCG iteration      q_own = (D + H_offdiag) p over owned rows -- needs p of the owned rows and of their HALO (the remote
                     neighbours); one all-gather of the owned p slices (N m floats in total: 2.4 MB at configs[3], 300 KB
                     per rank) refreshes the halo, one all-reduce of three scalars carries p.q and, one iteration late,
                     r.z and r.r.  Everything else (x, r, z, preconditioner) is owner-local.

Per iteration a rank therefore streams 1/R of the blocks and takes part in two latency-bound collectives; DESIGN.md
section 6 carries the byte / latency model.  This module is the device-agnostic formulation (torch ops + the node-parallel
HIP kernels for assembly and SpMV where they apply); it is what the world-size-2 / 4 gloo tests run, and what RCCL runs
one process per GPU.  Trajectories equal the single-process ones up to the summation order of the dot products.
"""
from __future__ import annotations

import ctypes

import torch

from .. import _C


_P2P_SIG = [ctypes.c_void_p] * 9 + [ctypes.POINTER(ctypes.c_void_p)] * 2 + [ctypes.c_void_p] * 3 + \
           [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
_P2P_CZ_SIG = [ctypes.c_void_p] * 10 + [ctypes.POINTER(ctypes.c_void_p)] * 2 + [ctypes.c_void_p] * 3 + \
              [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
_P2P_CAP = 1 << 16
_ROW_SLOTS = 24          # PPLIE_PCG_COARSE_SLOTS: the tables are sized for the wide rows of the two-level variant


class P2PRank:
    """One rank's buffers of the multi-GPU persistent PCG (csrc/pcg_persist.hip, ``pplie_pcg_persist_p2p``): its copy of the
    hand-off table of p (all nodes of the graph), its rank-level sum table, the workgroup-level table and the outputs.
    The two exchange tables are what the peers map (hipIpc) and write into."""

    def __init__(self, n_global, m, dtype, device):
        nw = 1 if dtype == torch.float32 else 2
        z64 = lambda k: torch.zeros(k, dtype=torch.int64, device=device)
        self.ptag = z64(2 * n_global * m * nw)                         # zeroed once: tags carry the solve's epoch
        self.rpart = z64(2 * 8 * _ROW_SLOTS * nw)
        self.part = z64(2 * 256 * _ROW_SLOTS * nw)
        self.rr_hist = torch.zeros(_P2P_CAP, dtype=dtype, device=device)
        self.info = torch.zeros(4, dtype=dtype, device=device)
        self.it = torch.zeros(4, dtype=torch.int32, device=device)


def persist_p2p_launch(rk, ptag_ptrs, rpart_ptrs, ptr, other, HB, D, Binv, x, r, z, tol, maxiter, grid, row0, n_global, world, rank,
                       epoch, m, shift=None):
    """enqueue one rank's kernel on the current stream.  ``ptag_ptrs`` / ``rpart_ptrs``: the ``world`` table addresses as this
    process sees them (its own allocation + the peers' mapped ones).  ``shift`` (the owned rows' damping shift): the two-level
    (block-Jacobi + gauge) preconditioner, ``pplie_pcg_persist_p2p_coarse`` -- every rank must pass it or none."""
    n_own = D.shape[0]
    sfx = "_f32" if D.dtype == torch.float32 else "_f64"
    arr = ctypes.c_void_p * world
    rk.part.zero_()                                                  # (workgroup-level table: local, cleared per solve)
    tail = (rk.part.data_ptr(), arr(*ptag_ptrs), arr(*rpart_ptrs), rk.rr_hist.data_ptr(), rk.info.data_ptr(),
            rk.it.data_ptr(), float(tol), int(min(maxiter, 65534)), _P2P_CAP, int(grid), n_own, int(row0), int(n_global),
            int(world), int(rank), int(epoch), int(m), _C.stream_ptr(D.device))
    with _C._on_device(D.device):
        if shift is not None:
            fn = _C.library().symbol("pplie_pcg_persist_p2p_coarse" + sfx, _P2P_CZ_SIG)
            code = fn(ptr.data_ptr(), other.data_ptr(), HB.data_ptr(), D.data_ptr(), Binv.data_ptr(), shift.data_ptr(), x.data_ptr(),
                      r.data_ptr(), z.data_ptr(), *tail)
        else:
            fn = _C.library().symbol("pplie_pcg_persist_p2p" + sfx, _P2P_SIG)
            code = fn(ptr.data_ptr(), other.data_ptr(), HB.data_ptr(), D.data_ptr(), Binv.data_ptr(), x.data_ptr(), r.data_ptr(),
                      z.data_ptr(), *tail)
    return code


def _bounds(N, world, rank):
    chunk = -(-N // world)
    a = min(N, rank * chunk)
    return chunk, a, min(N, a + chunk)


class NodeShard:
    """Ownership, local incidence lists and halo map of one rank for one edge list (cached on the optimizer)."""

    def __init__(self, lin, group):
        import torch.distributed as dist
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        N, K = lin.N, lin.K
        assert K == 2, "node-sharded solve: pairwise edges"
        self.N, self.m = N, lin.m
        self.chunk, self.a, self.b = _bounds(N, self.world, self.rank)
        self.n_own = self.b - self.a
        ptr, blk, other = lin.csr()
        lo, hi = (int(v) for v in ptr[[self.a, self.b]].tolist())          # (once per edge list)
        dev = ptr.device
        self.ptr = (ptr[self.a:self.b + 1] - lo).to(torch.int32).contiguous()
        self.blk = blk[lo:hi].contiguous()
        oth = other[lo:hi].long()
        own = (oth >= self.a) & (oth < self.b)
        self.halo = torch.unique(oth[~own])                                  # sorted global ids of the remote neighbours
        slot = torch.searchsorted(self.halo, oth.clamp_max(max(N - 1, 0))) if self.halo.numel() else torch.zeros_like(oth)
        self.other = torch.where(own, oth - self.a, self.n_own + slot).to(torch.int32).contiguous()
        self.other_global = oth.to(torch.int32).contiguous()            # (the p2p exchange addresses the full hand-off table)
        self.p2p = None                                                  # P2PRank + the peers' mapped tables, set up on first use
        counts = (self.ptr[1:] - self.ptr[:-1]).long()
        self.row = torch.repeat_interleave(torch.arange(self.n_own, device=dev), counts)    # owned row of every incidence
        self.edge, self.side = (self.blk // K).long(), (self.blk % K).long()
        # halo rows inside the all-gathered [world * chunk, m] buffer: global id g sits at row g (chunks are contiguous)
        self.halo_rows = self.halo

    # ---- collectives -----------------------------------------------------------------------------------------------
    def gather_rows(self, own_rows):
        """[n_own, m] owned rows of every rank -> the full [N, m] vector (one all-gather of equal-size padded slices)."""
        import torch.distributed as dist
        m = own_rows.shape[-1]
        mine = own_rows.new_zeros((self.chunk, m))
        mine[:self.n_own] = own_rows
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        return torch.cat(parts, 0)[:self.N]

    def sum_scalars(self, t):
        import torch.distributed as dist
        dist.all_reduce(t, group=self.group)
        return t

    # ---- peer-to-peer exchange (exchange="p2p") ------------------------------------------------------------------------
    def p2p_setup(self, dtype, device):
        """this rank's exchange tables + the peers' mapped ones (collective, once per edge list and dtype)"""
        if self.p2p is not None and self.p2p['dtype'] == dtype:
            return self.p2p
        import torch.distributed as dist
        rk, ptag, rpart, err = None, None, None, None
        try:
            rk = P2PRank(self.N, self.m, dtype, device)
        except Exception as e:                          # (allocation failure on this rank only)
            err = e
        # the table exchange is a sequence of collectives: whether to enter it is agreed first, and so is its outcome -- a rank
        # that failed alone would leave the others waiting inside a collective it never joins
        bad = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=device)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=self.group)
        if int(bad) == 0:
            try:
                ptag, rpart = p2p_exchange_tables(rk, self.group)
            except Exception as e:
                err = e
            bad = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=device)
            dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=self.group)
        ok = int(bad) == 0
        self.p2p = dict(dtype=dtype, rk=rk, ptag=ptag, rpart=rpart, epoch=0, ok=ok, error=err)
        return self.p2p


def p2p_exchange_tables(rk, group):
    """Publish this rank's two exchange tables to the peers as IPC handles (hipIpcGetMemHandle through torch.multiprocessing's
    CUDA-tensor reduction; one process per GPU) and map theirs (hipIpcOpenMemHandle).  Collective over ``group``; returns the
    ``world`` hand-off tables and the ``world`` rank-sum tables as tensors addressable from this process (own ones included)."""
    import torch.distributed as dist
    from torch.multiprocessing.reductions import reduce_tensor
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    torch.cuda.synchronize(rk.ptag.device)                               # (zero-filled before anybody can write into them)
    mine = [reduce_tensor(rk.ptag), reduce_tensor(rk.rpart)]
    everyone = [None] * world
    dist.all_gather_object(everyone, mine, group=group)
    ptag, rpart, err = [], [], None
    for k, handles in enumerate(everyone):
        if k == rank:
            ptag.append(rk.ptag)
            rpart.append(rk.rpart)
        else:
            try:
                (f0, a0), (f1, a1) = handles
                ptag.append(f0(*a0))                                     # the peer's table, addressable from this device
                rpart.append(f1(*a1))
            except Exception as e:                                       # (peer memory not mappable from this process)
                err = err or e
    # the kernel of THIS rank's GPU stores into the peers' tables: peer access from our device to every device that owns one
    mine_dev = rk.ptag.device.index
    fn = _C.library().symbol("pplie_enable_peer_access", [ctypes.c_int, ctypes.c_int])
    for t in ptag:
        if err is None and t.device.index != mine_dev:
            code = fn(mine_dev, t.device.index)
            if code != 0:
                err = RuntimeError(f"no peer access from cuda:{mine_dev} to cuda:{t.device.index} (status {code})")
    dist.barrier(group=group)                                            # every table exists and is mapped everywhere
    if err is not None:
        raise err
    return ptag, rpart


class NodeShardedSystem:
    """Owned rows of (J^T W J + damping) and their PCG (see the module docstring)."""

    def __init__(self, lin, shard):
        self.lin, self.sh = lin, shard
        self.B = self.g = self.HB = None
        self.gauge = False            # this solve's preconditioner carries the gauge correction (set by solve())

    def _hip(self):
        return self.lin._hip() and self.lin.m in (3, 6, 7)

    def group_is_device(self):
        import torch.distributed as dist
        return dist.get_backend(self.sh.group) == "nccl"

    def assemble(self):
        lin, sh = self.lin, self.sh
        m, dt, dev = lin.m, lin.J.dtype, lin.J.device
        n, C = sh.n_own, sh.blk.numel()
        if self._hip() and n > 0:
            from . import posegraph as _pg
            sfx = "_f32" if dt == torch.float32 else "_f64"
            self.B = torch.empty((n, m, m), dtype=dt, device=dev)
            self.g = torch.empty((n, m), dtype=dt, device=dev)
            self.HB = torch.empty((max(C, 1), m, m), dtype=dt, device=dev)
            with _C._on_device(dev):
                code = _C.library().symbol("pplie_graph_assemble_csr" + sfx, _pg._ASMC_SIG)(
                    sh.ptr.data_ptr(), sh.blk.data_ptr(), lin.J.data_ptr(), lin.W.data_ptr() if lin.W is not None else None,
                    lin.R.data_ptr(), self.B.data_ptr(), self.g.data_ptr(), self.HB.data_ptr(), n, lin.dr, m, lin.K,
                    _C.stream_ptr(dev))
            _C.check(code, "pplie_graph_assemble_csr")
            return
        Jn = lin.J[sh.edge, sh.side]                                   # [C, dr, m] block of the owned end
        Jf = lin.J[sh.edge, 1 - sh.side]                               # block of the far end
        JtW = Jn.mT if lin.W is None else Jn.mT @ lin.W[sh.edge]
        self.B = torch.zeros((n, m, m), dtype=dt, device=dev).index_add_(0, sh.row, JtW @ Jn)
        self.g = torch.zeros((n, m), dtype=dt, device=dev).index_add_(0, sh.row, (JtW @ lin.R[sh.edge].unsqueeze(-1)).squeeze(-1))
        self.HB = JtW @ Jf

    def matvec(self, D, p_loc):
        """q_own = D p_own + sum over the owned rows' incidences of HB[c] p_loc[other[c]]"""
        sh, m = self.sh, self.lin.m
        n = sh.n_own
        if self._hip() and n > 0:
            from . import posegraph as _pg
            q = torch.empty((n, m), dtype=p_loc.dtype, device=p_loc.device)
            if not hasattr(self, "_scal"):
                self._scal = torch.zeros(_pg._PCG_SCAL_ELEMS, dtype=p_loc.dtype, device=p_loc.device)
                self._it = torch.zeros(2, dtype=torch.int32, device=p_loc.device)
            sfx = "_f32" if p_loc.dtype == torch.float32 else "_f64"
            with _C._on_device(p_loc.device):
                code = _C.library().symbol("pplie_graph_bsr_spmv" + sfx, _pg._BSR_SIG)(
                    sh.ptr.data_ptr(), sh.other.data_ptr(), self.HB.data_ptr(), D.data_ptr(), p_loc.data_ptr(), q.data_ptr(),
                    self._scal.data_ptr(), self._it.data_ptr(), n, m, _C.stream_ptr(p_loc.device))
            _C.check(code, "pplie_graph_bsr_spmv")
            return q
        q = (D @ p_loc[:n].unsqueeze(-1)).squeeze(-1)
        if sh.blk.numel():
            q = q.index_add(0, sh.row, (self.HB @ p_loc[sh.other.long()].unsqueeze(-1)).squeeze(-1))
        return q

    def _solve_hip(self, s, dmin, dmax, tol, maxiter, check_every):
        """The same iteration on the fused kernels of the single-GPU path (csrc/graph.hip): pplie_pcg_prepare once, then per
        iteration pplie_pcg2_spmv and pplie_pcg2_step on the OWNED rows (local incidence lists, p laid out as
        [owned rows padded to the chunk | halo rows]), with the two exchanges between them: the slot-spread partial sums
        { r.z, p.q, q.z, q.Binv q } are totalled, all-reduced (one 4-float collective) and written back, and the owned p
        slices are all-gathered to refresh the halo.  No host synchronisation except the convergence test every
        `check_every` iterations."""
        import torch.distributed as dist
        from . import posegraph as _pg
        sh, m, lin = self.sh, self.lin.m, self.lin
        n, chunk, dev, dt = sh.n_own, sh.chunk, self.g.device, self.g.dtype
        sfx = "_f32" if dt == torch.float32 else "_f64"
        lib, st = _C.library(), _C.stream_ptr(dev)
        z = lambda *shape: torch.zeros(shape, dtype=dt, device=dev)
        gauge = self.gauge
        w = self.__dict__.get('_ws')
        if w is None or w['key'] != (n, chunk, sh.halo.numel(), dt):
            w = self._ws = dict(key=(n, chunk, sh.halo.numel(), dt), D=z(n, m, m), Binv=z(n, m, m), shift=z(n, m), x=z(n, m), r=z(n, m),
                                r2=z(n, m), q=z(n, m), z=z(n, m), p=z(chunk + sh.halo.numel(), m), full=z(sh.world * chunk, m),
                                scal=z(_pg._PCG_SCAL_ELEMS), cs=z(_pg._PCG2_CS_ELEMS), rr_hist=z(1 << 16),
                                it=torch.zeros(4, dtype=torch.int32, device=dev),
                                other=torch.where(sh.other >= n, sh.other + (chunk - n), sh.other).contiguous(),
                                qsel=torch.tensor([0, 1, 4, 5], device=dev), tot=z(8 + 16))
        w['scal'].zero_()
        w['cs'].zero_()
        w['it'].zero_()
        w['p'].zero_()
        S = w['scal'].view(2, 8, 32, 32)
        CS = w['cs'].view(2, 32, 32)                                   # [set][slot][E 0..7 | Z^T q 8..15 | Z^T r 16..23 | -] (csrc/graph.hip)
        with _C._on_device(dev):
            if gauge:
                # two-level preconditioner: the rank's partial E and Z^T r_0 land in set 0 of cs; their totals over the ranks go back
                # into slot 0 (the kernels sum the 32 slots of a set), then p_0 += Z (Z^T r_0 / E)
                _C.check(lib.symbol("pplie_pcg_prepare_coarse" + sfx, _pg._PREP_CZ_SIG)(
                    self.B.data_ptr(), self.g.data_ptr(), w['D'].data_ptr(), w['Binv'].data_ptr(), w['shift'].data_ptr(), w['x'].data_ptr(),
                    w['r'].data_ptr(), w['z'].data_ptr(), w['p'].data_ptr(), w['scal'].data_ptr(), w['cs'].data_ptr(), float(s), None,
                    float(dmin), float(dmax), n, m, st), "pplie_pcg_prepare_coarse")
                head = torch.cat([S[0, 3, :, 0].sum().reshape(1), CS[0].sum(0)[:24]])
                dist.all_reduce(head, group=sh.group)
                bn2 = float(head[0])
                CS[0].zero_()
                CS[0, 0, :24] = head[1:]
                _C.check(lib.symbol("pplie_pcg2_coarse_init" + sfx, _pg._CZ_INIT_SIG)(w['p'].data_ptr(), w['cs'].data_ptr(), n, m, st),
                         "pplie_pcg2_coarse_init")
            else:
                _C.check(lib.symbol("pplie_pcg_prepare" + sfx, _pg._PREP_SIG)(
                    self.B.data_ptr(), self.g.data_ptr(), w['D'].data_ptr(), w['Binv'].data_ptr(), w['shift'].data_ptr(), w['x'].data_ptr(),
                    w['r'].data_ptr(), w['z'].data_ptr(), w['p'].data_ptr(), w['scal'].data_ptr(), float(s), float(dmin), float(dmax), n, m, st),
                    "pplie_pcg_prepare")
                bn2 = S[0, 3, :, 0].sum().reshape(1)
                dist.all_reduce(bn2, group=sh.group)
                bn2 = float(bn2)
            if bn2 == 0.0:
                return sh.gather_rows(w['x']), 0
            if gauge:
                spmv_cz = lib.symbol("pplie_pcg2_spmv_coarse" + sfx, _pg._PCG2_SPMV_CZ_SIG)
                step_cz = lib.symbol("pplie_pcg2_step_coarse" + sfx, _pg._PCG2_STEP_CZ_SIG)
            else:
                spmv = lib.symbol("pplie_pcg2_spmv" + sfx, _pg._PCG2_SPMV_SIG)
                step = lib.symbol("pplie_pcg2_step" + sfx, _pg._PCG2_STEP_SIG)
            done, thresh = 0, tol * tol * bn2
            maxiter = min(maxiter, (1 << 16) - check_every)
            tot = w['tot']
            while done < maxiter:
                a = done & 1
                dist.all_gather_into_tensor(w['full'], w['p'][:chunk], group=sh.group)       # halo refresh
                if sh.halo.numel():
                    torch.index_select(w['full'], 0, sh.halo_rows, out=w['p'][chunk:])
                if gauge:
                    # (tol2 = -1: the device-side stop test compares THIS rank's |r|^2 and never fires; the host tests the global one)
                    _C.check(spmv_cz(sh.ptr.data_ptr(), w['other'].data_ptr(), self.HB.data_ptr(), w['D'].data_ptr(), w['Binv'].data_ptr(),
                                     w['p'].data_ptr(), w['z'].data_ptr(), w['q'].data_ptr(), w['scal'].data_ptr(), w['cs'].data_ptr(),
                                     w['rr_hist'].data_ptr(), w['it'].data_ptr(), 1 << 16, n, m, -1.0, st), "pplie_pcg2_spmv_coarse")
                    # the 8 scalars and this iteration's Z^T q / Z^T r (set a of cs, elements 8..23) in ONE all-reduce
                    torch.sum(S[a, :, :, 0], -1, out=tot[:8])
                    torch.sum(CS[a, :, 8:24], 0, out=tot[8:])
                    dist.all_reduce(tot, group=sh.group)
                    S[a, w['qsel'], :, 0] = 0
                    S[a, w['qsel'], 0, 0] = tot[w['qsel']]
                    CS[a, :, 8:24] = 0
                    CS[a, 0, 8:24] = tot[8:]
                    _C.check(step_cz(w['x'].data_ptr(), w['r'].data_ptr(), w['r2'].data_ptr(), w['p'].data_ptr(), w['q'].data_ptr(),
                                     w['z'].data_ptr(), w['Binv'].data_ptr(), w['scal'].data_ptr(), w['cs'].data_ptr(), w['it'].data_ptr(),
                                     n, m, st), "pplie_pcg2_step_coarse")
                else:
                    _C.check(spmv(sh.ptr.data_ptr(), w['other'].data_ptr(), self.HB.data_ptr(), w['D'].data_ptr(), w['Binv'].data_ptr(),
                                  w['p'].data_ptr(), w['z'].data_ptr(), w['q'].data_ptr(), w['scal'].data_ptr(), w['rr_hist'].data_ptr(),
                                  w['it'].data_ptr(), 1 << 16, n, m, st), "pplie_pcg2_spmv")
                    t8 = S[a, :, :, 0].sum(-1)                                                # [8] totals of this rank
                    dist.all_reduce(t8, group=sh.group)
                    S[a, w['qsel'], :, 0] = 0
                    S[a, w['qsel'], 0, 0] = t8[w['qsel']]
                    _C.check(step(w['x'].data_ptr(), w['r'].data_ptr(), w['r2'].data_ptr(), w['p'].data_ptr(), w['q'].data_ptr(), w['z'].data_ptr(),
                                  w['Binv'].data_ptr(), w['scal'].data_ptr(), w['it'].data_ptr(), n, m, st), "pplie_pcg2_step")
                done += 1
                if done % check_every == 0 or done >= maxiter:
                    rr = S[a, 2, :, 0].sum().reshape(1)
                    dist.all_reduce(rr, group=sh.group)
                    rr = float(rr)
                    assert rr == rr, 'Linear solve produced NaN (matrix may not be positive-definite)'
                    if rr <= thresh:
                        break
        return sh.gather_rows(w['x']), done

    def _solve_p2p(self, s, dmin, dmax, tol, maxiter):
        """The whole solve as ONE persistent launch per rank (pplie_pcg_persist_p2p): p and the dot products cross GPUs as tagged
        words written straight into the peers' tables from inside the kernel -- no collective, no launch and no host round trip
        per iteration.  One RCCL all-gather of the solution at the end."""
        import torch.distributed as dist
        from . import posegraph as _pg
        sh, m = self.sh, self.lin.m
        n, chunk, dev, dt = sh.n_own, sh.chunk, self.g.device, self.g.dtype
        sfx = "_f32" if dt == torch.float32 else "_f64"
        z = lambda *shape: torch.zeros(shape, dtype=dt, device=dev)
        pp_ = sh.p2p_setup(dt, dev)
        if not pp_['ok']:                               # (agreed by all ranks inside p2p_setup)
            raise _pg.SolveFailed(f"p2p exchange tables could not be set up on every rank ({pp_.get('error')}); using RCCL collectives")
        w = self.__dict__.get('_wp')
        if w is None or w['key'] != (n, chunk, dt):
            w = self._wp = dict(key=(n, chunk, dt), D=z(n, m, m), Binv=z(n, m, m), shift=z(n, m), x=z(chunk, m), r=z(n, m), z=z(n, m),
                                p=z(n, m), scal=z(_pg._PCG_SCAL_ELEMS), full=z(sh.world * chunk, m))
        pp_['epoch'] += 1                               # (every rank makes this call for every solve: the epochs stay in step)
        local_err = None
        try:
            with _C._on_device(dev):
                _C.check(_C.library().symbol("pplie_pcg_prepare" + sfx, _pg._PREP_SIG)(
                    self.B.data_ptr(), self.g.data_ptr(), w['D'].data_ptr(), w['Binv'].data_ptr(), w['shift'].data_ptr(), w['x'].data_ptr(),
                    w['r'].data_ptr(), w['z'].data_ptr(), w['p'].data_ptr(), w['scal'].data_ptr(), float(s), float(dmin), float(dmax), n, m,
                    _C.stream_ptr(dev)), "pplie_pcg_prepare")
                code = persist_p2p_launch(pp_['rk'], [t.data_ptr() for t in pp_['ptag']], [t.data_ptr() for t in pp_['rpart']], sh.ptr,
                                          sh.other_global, self.HB, w['D'], w['Binv'], w['x'][:n], w['r'], w['z'], tol, maxiter,
                                          _pg.PERSIST_GRID, sh.a, sh.N, sh.world, sh.rank, pp_['epoch'], m,
                                          shift=w['shift'] if self.gauge else None)
            _C.check(code, "pplie_pcg_persist_p2p")
        except Exception as e:                          # this rank could not launch: the peers will time out at the exchange
            local_err = e
        # The verdict of the solve is COLLECTIVE: a spin time-out (flag 3) is local to the rank that hit it -- a peer may have finished
        # its last iteration normally -- and the next solve must take the same route (peer stores or RCCL collectives) on every
        # rank, or the ranks wait for each other in different collectives for ever.  MAX over {flag, 4 if the launch itself failed}.
        verdict = pp_['rk'].info[3:4].clone() if local_err is None else torch.full((1,), 4.0, dtype=dt, device=dev)
        dist.all_reduce(verdict, op=dist.ReduceOp.MAX, group=sh.group)
        dist.all_gather_into_tensor(w['full'], w['x'], group=sh.group)
        its, rr, bn2, flag = pp_['rk'].info.tolist()
        worst = float(verdict)
        if worst >= 3.0:
            pp_['ok'] = False                           # on every rank: from here on the RCCL iteration
            raise _pg.SolveFailed('p2p persistent PCG: ' + ('a rank could not launch its kernel' if worst >= 4.0 else
                                  'a peer never arrived at an exchange (kernels not co-resident / peer memory not visible)')
                                  + '; every rank falls back to RCCL collectives') from local_err
        assert flag != 2.0 and rr == rr, 'Linear solve produced NaN (matrix may not be positive-definite)'
        full = w['full'].view(sh.world, chunk, m)
        return torch.cat([full[k, :(_bounds(sh.N, sh.world, k)[2] - _bounds(sh.N, sh.world, k)[1])] for k in range(sh.world)], 0), int(its)

    def p2p_applicable(self):
        """group-uniform facts only (every rank must take the same path): device backend, HIP shapes, rows per rank within what
        one persistent launch holds, and the `ok` flag -- which is only ever cleared by the collective verdicts of p2p_setup /
        _solve_p2p, i.e. on all ranks at once"""
        sh, m = self.sh, self.lin.m
        per_wg = 16 * (64 // m)
        return (getattr(self.lin.opt, '_exchange_eff', getattr(self.lin.opt, 'exchange', None) or 'rccl') == 'p2p' and self.lin._hip()
                and m in (3, 6, 7) and self.group_is_device()
                and sh.world <= 8 and sh.chunk <= 256 * per_wg and sh.chunk > 0 and sh.N % 1 == 0
                and (sh.p2p is None or sh.p2p.get('ok', True)))

    def solve(self, s, dmin, dmax, tol, maxiter, check_every=8, gauge=True):
        """(H + damping) d = -g over all ranks; returns the FULL step [N, m] (all-gathered) and the iteration count."""
        sh, m = self.sh, self.lin.m
        n = sh.n_own
        self.gauge = bool(gauge)      # (the caller's verdict, from the gathered blocks every rank holds alike: group-uniform)
        if self.p2p_applicable() and (sh.world - 1) * sh.chunk < sh.N:       # (every rank owns at least one row)
            from . import posegraph as _pg
            try:
                return self._solve_p2p(s, dmin, dmax, tol, maxiter)
            except _pg.SolveFailed as e:
                # the verdict behind this exception was agreed by all ranks (see _solve_p2p): every rank is here, and every rank
                # now runs the same RCCL iteration for this solve and the following ones
                import warnings
                warnings.warn(f"pypose_amd: {e}")
        # the HIP and the torch formulation issue different collective sequences: the choice is made from facts every rank
        # agrees on (backend, shapes, every rank owning at least one row), never from this rank's own row count
        if self._hip() and self.group_is_device() and (sh.world - 1) * sh.chunk < sh.N:
            return self._solve_hip(s, dmin, dmax, tol, maxiter, check_every)
        diag = self.B.diagonal(dim1=-2, dim2=-1)
        D = self.B.clone()
        D.diagonal(dim1=-2, dim2=-1).copy_(s * diag.clamp(dmin, dmax))        # optimizer.py:656-657, :666
        Binv = torch.linalg.inv(D) if n else D
        apply_Binv = lambda v: (Binv @ v.unsqueeze(-1)).squeeze(-1)
        # two-level preconditioner (block-Jacobi + gauge modes, csrc/pcg_persist.hip "CZ"):  z = Binv r + Z (Z^T r / E) with
        # E = sum over ALL nodes of the damping shift: m more sums in the scalar all-reduce of every iteration, one more at the start
        gauge = self.gauge
        x = torch.zeros_like(self.g)
        r = -self.g
        z = apply_Binv(r)
        head = [(r * z).sum().reshape(1), (r * r).sum().reshape(1)]
        if gauge:
            head += [r.sum(0), (s * diag.clamp(dmin, dmax) - diag).sum(0)]
        sums = sh.sum_scalars(torch.cat(head))
        rho, bn2 = sums[0], sums[1]
        if gauge:
            E = sums[2 + m:2 + 2 * m]
            Einv = torch.where(E > 0, 1.0 / E.clamp_min(torch.finfo(E.dtype).tiny), torch.zeros_like(E))
            c = sums[2:2 + m] * Einv
            rho = rho + (sums[2:2 + m] * c).sum()
            z = z + c
        p = z.clone()
        if float(bn2) == 0.0:
            return sh.gather_rows(x), 0
        thresh = tol * tol * float(bn2)
        done = 0
        while done < maxiter:
            p_full = sh.gather_rows(p)                                        # halo refresh (one all-gather)
            p_loc = torch.cat([p, p_full[sh.halo_rows]], 0) if sh.halo.numel() else p
            q = self.matvec(D, p_loc)
            pq = sh.sum_scalars((p * q).sum().reshape(1))[0]
            alpha = torch.where(pq != 0, rho / pq, torch.zeros_like(pq))
            x = x + alpha * p
            r = r - alpha * q
            z = apply_Binv(r)
            tail = [(r * z).sum().reshape(1), (r * r).sum().reshape(1)]
            if gauge:
                tail.append(r.sum(0))
            sums = sh.sum_scalars(torch.cat(tail))
            rho_new, rr = sums[0], sums[1]
            if gauge:
                c = sums[2:2 + m] * Einv
                rho_new = rho_new + (sums[2:2 + m] * c).sum()
                z = z + c
            done += 1
            if done % check_every == 0 or done >= maxiter:
                rr_h = float(rr)
                assert rr_h == rr_h, 'Linear solve produced NaN (matrix may not be positive-definite)'
                if rr_h <= thresh:
                    break
            beta = torch.where(rho != 0, rho_new / rho, torch.zeros_like(rho))
            p = z + beta * p
            rho = rho_new
        return sh.gather_rows(x), done
