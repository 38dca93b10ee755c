"""Gather-structured linearisation over SEVERAL parameters of different widths (bundle adjustment).

The reference's BA model (examples/module/ba/bundle_adjustment.py:16-43) optimises intrinsics ``K [Nc,3]``, camera
poses ``C [Nc,7]`` (SE3) and points ``P [Np,3]``; residual row ``e`` (one observation) reads ``K[cidx[e]]``,
``C[cidx[e]]`` and ``P[pidx[e]]``.  Its dense LM path needs a ``[2E, 3Nc + 7Nc + 3Np]`` Jacobian; its scalable
path is the un-vendored ``bae`` plugin.  Here the SAME unmodified model is linearised per observation:

* every ``param[index]`` gather on an optimised parameter is a *slot* ``(parameter, index [E], block width)``;
  the per-observation blocks ``J_s [E, d_res, m_s]`` come from ``d_res`` batched backward sweeps w.r.t. the gathered
  rows (blocks.jacobian_blocks) and a random vector-Jacobian probe against the real model verifies the structure;
* the normal equations are never formed: block diagonals + gradient by scatter-add, ``H p`` matrix-free, block-Jacobi
  PCG on the concatenated unknowns (small problems are assembled densely for the user's ``solver``: bit-for-bit
  the reference's algebra, used by the parity tests).

This is the device-agnostic formulation (gathers, broadcast-multiply-sums and ``index_add_`` on whatever device
the model lives on; the Lie arithmetic inside the model is the HIP kernels).  Single-parameter SE3 / Sim3 / SO3
graphs take the HIP kernels of optim/posegraph.py instead.
"""
from __future__ import annotations

import ctypes
import warnings

import torch

from .. import _C

from ..lietensor import lietensor as _lt
from . import blocks as _blocks
from .posegraph import DENSE_LIMIT, PCG, _all_reduce

SCHUR_LIMIT = 12288        # largest reduced (non-eliminated) system assembled densely for the user's solver


_SEG_SIG = [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
_MGJ_SIG = [ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
_MGS_SIG = [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
_BMV_SIG = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
_STAGE_SIG = [ctypes.c_int] + [ctypes.c_void_p] * 10 + [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]   # pplie_pcg_stage
_FLAT_SIG = [ctypes.c_int] + [ctypes.c_void_p] * 8 + [ctypes.c_int, ctypes.c_int64, ctypes.c_void_p]                    # pplie_pcg_flat
_MG3_JT_SIG = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] + [ctypes.c_void_p] * 7 + [ctypes.c_int64] * 3 + [ctypes.c_void_p] * 10 \
    + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]                                                                   # pplie_mg3_jt
_MG3_STEP_SIG = [ctypes.c_int] + [ctypes.c_void_p] * 11 + [ctypes.c_void_p]                                           # pplie_mg3_step
MG3_CHUNK = int(__import__('os').environ.get('PPLIE_MG3_CHUNK', '128'))     # PPLIE_MG3_CHUNK: incidences of one work item of pplie_mg3_jt


class _Scatter:
    """``out.index_add_(0, idx, vals)`` into N rows for a FIXED index vector: on a HIP device a segmented sum over
    the incidence lists (``pplie_segment_sum``: deterministic, no atomics -- a camera row of a BA problem receives
    ~10^3 contributions); elsewhere ``index_add_``.  Rows with long lists are split into chunks of ``CHUNK`` incidences
    summed by one wavefront each, then a second segmented sum adds a row's chunk partials (one wavefront looping
    over 10^3 incidences of a camera row was the critical path of the whole matvec)."""
    CHUNK = 128

    def __init__(self, idx, N):
        self.idx, self.N = idx, N
        self.hip = idx.is_cuda and _C._test_backend is None and idx.numel() < (1 << 31)
        self.two_level = False
        if self.hip:
            order = torch.argsort(idx, stable=True)
            self.perm = order.to(torch.int32)
            counts = torch.bincount(idx, minlength=N)
            ptr64 = torch.zeros(N + 1, dtype=torch.int64, device=idx.device)
            ptr64[1:] = torch.cumsum(counts, 0)
            self.ptr = ptr64.to(torch.int32)
            if int(counts.max()) > 2 * self.CHUNK:
                C = self.CHUNK
                per = (counts + C - 1) // C                                  # chunks per row
                nptr = torch.zeros(N + 1, dtype=torch.int64, device=idx.device)
                nptr[1:] = torch.cumsum(per, 0)
                nch = int(nptr[-1])
                row = torch.repeat_interleave(torch.arange(N, device=idx.device), per)
                k = torch.arange(nch, device=idx.device) - nptr[row]
                starts = ptr64[row] + k * C
                self.cptr = torch.cat([starts, ptr64[-1:]]).to(torch.int32)  # incidence range of every chunk
                self.nptr = nptr.to(torch.int32)                             # chunk range of every row
                self.ident = torch.arange(nch, dtype=torch.int32, device=idx.device)
                self.nch, self.two_level = nch, True

    def _segsum(self, vals, perm, ptr, n, w):
        out = torch.empty((n, w), dtype=vals.dtype, device=vals.device)
        fn = _C.library().symbol("pplie_segment_sum" + ("_f32" if vals.dtype == torch.float32 else "_f64"), _SEG_SIG)
        with _C._on_device(vals.device):
            _C.check(fn(vals.data_ptr(), perm.data_ptr(), ptr.data_ptr(), out.data_ptr(), n, w, _C.stream_ptr(vals.device)),
                     "pplie_segment_sum")
        return out

    def __call__(self, vals):
        """vals [E, ...] -> [N, ...] (trailing dims flattened for the kernel, at most 64 values per row)"""
        tail = vals.shape[1:]
        w = 1
        for t in tail:
            w *= t
        if self.hip and w <= 64 and vals.dtype in (torch.float32, torch.float64):
            vals = vals.reshape(vals.shape[0], w).contiguous()
            if self.two_level:
                out = self._segsum(self._segsum(vals, self.perm, self.cptr, self.nch, w), self.ident, self.nptr, self.N, w)
            else:
                out = self._segsum(vals, self.perm, self.ptr, self.N, w)
            return out.view((self.N,) + tuple(tail))
        out = torch.zeros((self.N,) + tuple(tail), dtype=vals.dtype, device=vals.device)
        return out.index_add_(0, self.idx, vals)

    def jt_q(self, J, q, out, accumulate):
        """out [N, m] (+)= scatter-add over this slot of J[e]^T q[e]  (pplie_mg_jt_segsum, chunked like __call__)"""
        dr, m = J.shape[-2], J.shape[-1]
        sfx = "_f32" if J.dtype == torch.float32 else "_f64"
        fn = _C.library().symbol("pplie_mg_jt_segsum" + sfx, _MGS_SIG)
        st = _C.stream_ptr(J.device)
        with _C._on_device(J.device):
            if self.two_level:
                part = torch.empty((self.nch, m), dtype=J.dtype, device=J.device)
                _C.check(fn(J.data_ptr(), q.data_ptr(), self.perm.data_ptr(), self.cptr.data_ptr(), part.data_ptr(), self.nch, dr, m,
                            0, st), "pplie_mg_jt_segsum")
                if accumulate:
                    out += self._segsum(part, self.ident, self.nptr, self.N, m)
                else:
                    fs = _C.library().symbol("pplie_segment_sum" + sfx, _SEG_SIG)
                    _C.check(fs(part.data_ptr(), self.ident.data_ptr(), self.nptr.data_ptr(), out.data_ptr(), self.N, m, st),
                             "pplie_segment_sum")
            else:
                _C.check(fn(J.data_ptr(), q.data_ptr(), self.perm.data_ptr(), self.ptr.data_ptr(), out.data_ptr(), self.N, dr, m,
                            1 if accumulate else 0, st), "pplie_mg_jt_segsum")


def _tangent_width(p):
    m = _blocks.lie_manifold_width(p)
    return p.shape[-1] if m is None else m


class MultiGraphOperator:
    """``J`` handed to ``strategy.update``: ``J @ D`` with D the flat step in optimizer parameter order."""

    def __init__(self, lin):
        self.lin = lin

    def __matmul__(self, D):
        return self.lin._J_times(self.lin.step_to_nodes(D)).reshape(-1, 1)


class MultiGraphLinearization:
    kind = "multigraph"

    def __init__(self, opt, W, R, params, slots):
        """R [E,dr] (corrected), W [E,dr,dr] | None, params: the optimised parameters in optimizer order,
        slots: list of (parameter position, idx [E] int64, J [E,dr,m])."""
        self.opt, self.params, self.slots = opt, list(params), slots
        self.R, self.W = R.contiguous(), W
        self.E, self.dr = R.shape
        self.m = [_tangent_width(p) for p in self.params]
        self.N = [p.shape[0] for p in self.params]
        self.group = getattr(opt, 'group', None)
        self.s = 1.0
        self._scatters = None

    def scatters(self):
        """one deterministic scatter-add per slot (incidence lists cached on the optimizer per index vector)"""
        if self._scatters is None:
            cache = self.opt.__dict__.setdefault('_multigraph_scatter', {}) if hasattr(self.opt, '__dict__') else {}
            out = []
            for k, (pi, idx, _) in enumerate(self.slots):
                hit = cache.get((k, self.N[pi]))
                if hit is None or hit.idx.shape != idx.shape or not torch.equal(hit.idx, idx):
                    hit = cache[(k, self.N[pi])] = _Scatter(idx.clone(), self.N[pi])
                out.append(hit)
            self._scatters = out
        return self._scatters

    # -- layouts ------------------------------------------------------------------------------------
    def step_to_nodes(self, D):
        """flat [sum N_p w_p, 1] -> per-parameter tangent rows [N_p, m_p]"""
        out, off = [], 0
        flat = D.reshape(-1)
        for p, m in zip(self.params, self.m):
            n = p.numel()
            out.append(flat[off:off + n].view(p.shape[0], p.shape[-1])[:, :m])
            off += n
        return out

    def nodes_to_step(self, Dn):
        cols = []
        for p, m, d in zip(self.params, self.m, Dn):
            w = p.shape[-1]
            if m < w:
                d = torch.cat([d, torch.zeros((d.shape[0], w - m), dtype=d.dtype, device=d.device)], -1)
            cols.append(d.reshape(-1))
        return torch.cat(cols).view(-1, 1)

    def _split(self, x):
        out, off = [], 0
        for n, m in zip(self.N, self.m):
            out.append(x[off:off + n * m].view(n, m))
            off += n * m
        return out

    @staticmethod
    def _cat(xs):
        return torch.cat([x.reshape(-1) for x in xs])

    # -- HIP products on flat vectors (csrc/graph.hip: pplie_mg_jtimes / pplie_mg_jt_segsum / pplie_block_matvec) ------
    def hip_ok(self):
        return (self.R.is_cuda and _C._test_backend is None and self.group is None and self.dr <= 8 and len(self.slots) <= 4
                and max(self.m) <= 8 and self.R.dtype in (torch.float32, torch.float64)
                and all(sc.hip for sc in self.scatters()))

    def _sfx(self):
        return "_f32" if self.R.dtype == torch.float32 else "_f64"

    def matvec_flat(self, v, shift_flat):
        """(H + diag(shift)) v on the concatenated unknowns: one edge-parallel launch for q = W J v, one segmented
        J^T q per slot written straight into the output vector."""
        lib, st = _C.library(), _C.stream_ptr(v.device)
        xs = self._split(v)
        S = len(self.slots)
        arr = lambda ts: (ctypes.c_void_p * S)(*[t.data_ptr() for t in ts])
        q = torch.empty((self.E, self.dr), dtype=v.dtype, device=v.device)
        ms = (ctypes.c_int * S)(*[J.shape[-1] for _, _, J in self.slots])
        with _C._on_device(v.device):
            code = lib.symbol("pplie_mg_jtimes" + self._sfx(), _MGJ_SIG)(
                S, arr([J for _, _, J in self.slots]), arr([i for _, i, _ in self.slots]), arr([xs[pi] for pi, _, _ in self.slots]),
                ms, self.W.data_ptr() if self.W is not None else None, q.data_ptr(), self.E, self.dr, st)
            _C.check(code, "pplie_mg_jtimes")
            out = torch.empty_like(v)
            outs, seen = self._split(out), set()
            for (pi, _, J), sc in zip(self.slots, self.scatters()):
                sc.jt_q(J, q, outs[pi], pi in seen)
                seen.add(pi)
        return out if shift_flat is None else out.addcmul_(shift_flat, v)

    def precond_flat(self, v, Binv):
        lib, st = _C.library(), _C.stream_ptr(v.device)
        out = torch.empty_like(v)
        fn = lib.symbol("pplie_block_matvec" + self._sfx(), _BMV_SIG)
        with _C._on_device(v.device):
            for Bi, x, y, n, m in zip(Binv, self._split(v), self._split(out), self.N, self.m):
                _C.check(fn(Bi.data_ptr(), x.data_ptr(), y.data_ptr(), n, m, st), "pplie_block_matvec")
        return out

    # -- products -----------------------------------------------------------------------------------
    def _WJ(self, J):
        return J if self.W is None else (self.W.unsqueeze(-1) * J.unsqueeze(-3)).sum(-2)     # [E,dr,dr] x [E,dr,m]

    def _J_times(self, nodes):
        q = torch.zeros((self.E, self.dr), dtype=self.R.dtype, device=self.R.device)
        for pi, idx, J in self.slots:
            q += (J * nodes[pi][idx].unsqueeze(-2)).sum(-1)           # (tiny batched GEMMs are slow: multiply-sum)
        return q

    def _Hp(self, nodes):
        q = self._J_times(nodes)
        if self.W is not None:
            q = (self.W * q.unsqueeze(-2)).sum(-1)
        ys = [None] * len(nodes)
        for (pi, idx, J), sc in zip(self.slots, self.scatters()):
            y = sc((J * q.unsqueeze(-1)).sum(-2))
            ys[pi] = y if ys[pi] is None else ys[pi] + y
        return [_all_reduce(y, self.group) for y in ys]

    def _assemble(self):
        dt, dev = self.R.dtype, self.R.device
        B = [torch.zeros((n, m, m), dtype=dt, device=dev) for n, m in zip(self.N, self.m)]
        g = [torch.zeros((n, m), dtype=dt, device=dev) for n, m in zip(self.N, self.m)]
        Wr = self.R if self.W is None else (self.W * self.R.unsqueeze(-2)).sum(-1)
        for (pi, idx, J), sc in zip(self.slots, self.scatters()):
            WJ = self._WJ(J)
            B[pi] += sc((J.unsqueeze(-1) * WJ.unsqueeze(-2)).sum(-3))                        # J^T W J  [E,m,m]
            g[pi] += sc((J * Wr.unsqueeze(-1)).sum(-2))
        # two slots of the SAME parameter meeting in the same row (a self loop) put their cross term on the
        # diagonal block as well
        for a, (pa, ia, Ja) in enumerate(self.slots):
            for pb, ib, Jb in self.slots[a + 1:]:
                if pa == pb:
                    same = ia == ib
                    if bool(same.any()):
                        blk = (Ja[same].unsqueeze(-1) * self._WJ(Jb)[same].unsqueeze(-2)).sum(-3)
                        B[pa].index_add_(0, ia[same], blk + blk.mT)
        return [_all_reduce(b, self.group) for b in B], [_all_reduce(x, self.group) for x in g]

    # -- LM interface ---------------------------------------------------------------------------------
    def build_normal_equations(self, dmin, dmax):
        self.B, self.g = self._assemble()
        self.diag_raw = [b.diagonal(dim1=-2, dim2=-1).clone() for b in self.B]
        self.diag_clamped = [d.clamp(dmin, dmax) for d in self.diag_raw]
        self.s = 1.0

    def damp(self, damping):
        self.s = self.s * (1.0 + damping)

    def dense_matrix(self):
        """H = J^T W J as a dense matrix over the concatenated tangent unknowns (small problems / parity)."""
        offs, tot = [], 0
        for n, m in zip(self.N, self.m):
            offs.append(tot)
            tot += n * m
        A = torch.zeros((tot, tot), dtype=self.R.dtype, device=self.R.device)
        for pa, ia, Ja in self.slots:
            WJa = self._WJ(Ja)
            ra = offs[pa] + ia.unsqueeze(-1) * self.m[pa] + torch.arange(self.m[pa], device=ia.device)       # [E, ma]
            for pb, ib, Jb in self.slots:
                rb = offs[pb] + ib.unsqueeze(-1) * self.m[pb] + torch.arange(self.m[pb], device=ib.device)   # [E, mb]
                blk = (WJa.unsqueeze(-1) * Jb.unsqueeze(-2)).sum(-3)                                         # [E, ma, mb]
                A.index_put_((ra.unsqueeze(-1).expand_as(blk), rb.unsqueeze(-2).expand_as(blk)), blk, accumulate=True)
        return A

    def solve(self, solver):
        shift = [self.s * c - r for c, r in zip(self.diag_clamped, self.diag_raw)]               # A = H + diag(shift)
        b = [-x for x in self.g]
        tot = sum(n * m for n, m in zip(self.N, self.m))
        if not isinstance(solver, PCG) and tot <= DENSE_LIMIT and self.group is None:
            A = self.dense_matrix()
            A.diagonal().add_(self._cat(shift))
            Dn = self._split(solver(A=A, b=self._cat(b).view(-1, 1)).reshape(-1))
        elif not isinstance(solver, PCG) and self.group is None and self.schur_plan() is not None:
            # a direct solver was asked for and the problem is bipartite (bundle adjustment): eliminate the large,
            # mutually independent set of rows exactly and hand the small reduced system to the user's solver
            Dn = self._schur_solve(solver, shift, b)
        else:
            if not isinstance(solver, PCG):
                if not getattr(self.opt, '_warned_pcg', False):
                    warnings.warn(f"{type(solver).__name__} cannot factor a {tot}-unknown problem densely; "
                                  f"using the matrix-free block-Jacobi PCG (tol 1e-10) instead.")
                    self.opt._warned_pcg = True
                solver = PCG(tol=1e-10, maxiter=max(1000, 2 * tot))
            Binv = []
            for B, c in zip(self.B, self.diag_clamped):
                Bd = B.clone()
                Bd.diagonal(dim1=-2, dim2=-1).copy_(self.s * c)
                Binv.append(torch.linalg.inv(Bd))
            if self.R.is_cuda and self.group is None and getattr(solver, 'fused', True):
                cache = self.opt.__dict__.setdefault('_pcg_workspaces', {})
                key = ("multigraph", self.E, self.dr, tuple(self.N), tuple(self.m), tuple((pi, J.shape[-1]) for pi, _, J in self.slots),
                       self.R.dtype, self.R.device, self.W is not None, solver.check_every)
                wsp = cache.get(key)
                if wsp is None:
                    wsp = cache[key] = _GraphedPCG(self, solver.check_every)
                maxiter = tot * 10 if solver.maxiter is None else solver.maxiter
                x, solver.iterations = wsp.solve(self, shift, Binv, b, solver.tol, maxiter)
                Dn = self._split(x)
            else:
                matvec = lambda v: self._cat([y + sh * x for y, sh, x in zip(self._Hp(self._split(v)), shift, self._split(v))])
                precond = lambda v: self._cat([(Bi * x.unsqueeze(-2)).sum(-1) for Bi, x in zip(Binv, self._split(v))])
                Dn = self._split(solver.solve(matvec, self._cat(b), precond))
        assert not any(bool(torch.isnan(d).any()) for d in Dn), 'Linear solve produced NaN (matrix may not be positive-definite)'
        return self.nodes_to_step(Dn)

    # -- Schur complement (bipartite problems) -----------------------------------------------------------
    def schur_plan(self):
        """Bipartite structure, or None.  Slots sharing one index vector form a group (BA: {K, C} by camera index, {P} by
        point index); with exactly two groups, each parameter in one slot, the group with more rows is eliminated:
        its rows couple only through the other group, so its block of the normal equations is block diagonal.  The
        plan (cached on the optimizer per index structure) lists, for every pair of observations of the same
        eliminated row, the two observation ids -- the fill of the reduced system."""
        if hasattr(self, '_plan'):
            return self._plan
        self._plan = None
        groups = []
        for k, (pi, idx, _) in enumerate(self.slots):
            for g in groups:
                if idx.shape == g[0].shape and (idx.data_ptr() == g[0].data_ptr() or torch.equal(idx, g[0])):
                    g[1].append(k)
                    break
            else:
                groups.append((idx, [k]))
        used = [pi for pi, _, _ in self.slots]
        if len(groups) != 2 or len(set(used)) != len(used):
            return None
        rows = [self.N[self.slots[g[1][0]][0]] for g in groups]
        if any(self.N[self.slots[k][0]] != r for g, r in zip(groups, rows) for k in g[1]):
            return None
        a, b = (0, 1) if rows[0] >= rows[1] else (1, 0)
        mb = sum(self.m[self.slots[k][0]] for k in groups[b][1])
        if rows[b] * mb > SCHUR_LIMIT:
            return None
        cache = self.opt.__dict__.setdefault('_schur_plans', {}) if hasattr(self.opt, '__dict__') else {}
        ia, ib = groups[a][0], groups[b][0]
        hit = cache.get((self.E, rows[a], rows[b]))
        if hit is None or not (torch.equal(hit["ia"], ia) and torch.equal(hit["ib"], ib)):
            perm = torch.argsort(ia, stable=True)
            k = torch.bincount(ia, minlength=rows[a])
            ptr = torch.zeros(rows[a] + 1, dtype=torch.int64, device=ia.device)
            ptr[1:] = torch.cumsum(k, 0)
            kc = k[ia[perm]]                                           # partners of every sorted incidence
            left = torch.repeat_interleave(torch.arange(self.E, device=ia.device), kc)
            start = torch.cumsum(kc, 0) - kc
            right = ptr[ia[perm][left]] + (torch.arange(left.numel(), device=ia.device) - start[left])
            e1, e2 = perm[left], perm[right]
            hit = {"ia": ia.clone(), "ib": ib.clone(), "e1": e1, "e2": e2, "key": ib[e1] * rows[b] + ib[e2]}
            cache[(self.E, rows[a], rows[b])] = hit
        self._plan = {"a": groups[a][1], "b": groups[b][1], "Na": rows[a], "Nb": rows[b], **hit}
        return self._plan

    def _schur_solve(self, solver, shift, b):
        P = self._plan
        dt, dev = self.R.dtype, self.R.device
        cat_J = lambda ks: torch.cat([self.slots[k][2] for k in ks], dim=-1)          # [E, dr, m_group]
        cat_p = lambda xs, ks: torch.cat([xs[self.slots[k][0]] for k in ks], dim=-1)    # per-parameter rows -> group rows
        Ja, Jb, ia, ib, Na, Nb = cat_J(P["a"]), cat_J(P["b"]), P["ia"], P["ib"], P["Na"], P["Nb"]
        ma, mb = Ja.shape[-1], Jb.shape[-1]
        WJa = self._WJ(Ja)
        Wr = self.R if self.W is None else (self.W * self.R.unsqueeze(-2)).sum(-1)
        mm = lambda A, B: (A.unsqueeze(-1) * B.unsqueeze(-2)).sum(-3)                 # A^T B over d_res: [E, ., .]
        Va = torch.zeros((Na, ma, ma), dtype=dt, device=dev).index_add_(0, ia, mm(Ja, WJa))
        Vb = torch.zeros((Nb, mb, mb), dtype=dt, device=dev).index_add_(0, ib, mm(Jb, self._WJ(Jb)))
        ga = torch.zeros((Na, ma), dtype=dt, device=dev).index_add_(0, ia, (Ja * Wr.unsqueeze(-1)).sum(-2))
        gb = torch.zeros((Nb, mb), dtype=dt, device=dev).index_add_(0, ib, (Jb * Wr.unsqueeze(-1)).sum(-2))
        Va.diagonal(dim1=-2, dim2=-1).add_(cat_p(shift, P["a"]))                      # LM clamp + damping
        Vb.diagonal(dim1=-2, dim2=-1).add_(cat_p(shift, P["b"]))
        Vinv = torch.linalg.inv(Va)
        Eba = mm(Jb, WJa)                                                             # [E, mb, ma] coupling blocks
        Y = (Eba.unsqueeze(-1) * Vinv[ia].unsqueeze(-3)).sum(-2)                      # E_e Vinv_p(e)
        # reduced (camera) system: S = Vb - sum over pairs (e, e') of one eliminated row of Y_e E_e'^T
        blocks = (Y[P["e1"]].unsqueeze(-2) * Eba[P["e2"]].unsqueeze(-3)).sum(-1)      # [pairs, mb, mb]
        S = torch.zeros((Nb * Nb, mb, mb), dtype=dt, device=dev).index_add_(0, P["key"], -blocks)
        S = S.view(Nb, Nb, mb, mb)
        S[torch.arange(Nb, device=dev), torch.arange(Nb, device=dev)] += Vb
        S = S.permute(0, 2, 1, 3).reshape(Nb * mb, Nb * mb)
        rhs = gb - torch.zeros((Nb, mb), dtype=dt, device=dev).index_add_(0, ib, (Y * ga[ia].unsqueeze(-2)).sum(-1))
        xb = solver(A=S, b=-rhs.reshape(-1, 1)).reshape(Nb, mb)
        t = torch.zeros((Na, ma), dtype=dt, device=dev).index_add_(0, ia, (Eba * xb[ib].unsqueeze(-1)).sum(-2))
        xa = -(Vinv * (ga + t).unsqueeze(-2)).sum(-1)
        out = [None] * len(self.params)
        for x, ks in ((xa, P["a"]), (xb, P["b"])):
            off = 0
            for k in ks:
                pi = self.slots[k][0]
                out[pi] = x[:, off:off + self.m[pi]].contiguous()
                off += self.m[pi]
        return out

    def solve_gauss_newton(self, solver):
        # Gauss-Newton's pseudo-inverse step is not reproduced here: the normal equations of bundle adjustment are too
        # ill-conditioned (focal lengths ~ 10^3 next to unit-scale points, plus the 7-dof gauge) for CG to reach the
        # exact minimum-norm solution the reference's SVD returns -- tried, the iterates differ by O(1).  GN on
        # several parameters therefore stays on the dense linearisation.
        raise NotImplementedError

    def strategy_args(self):
        return MultiGraphOperator(self), self.R.reshape(-1, 1)


class _GraphedPCG:
    """Block-Jacobi PCG of one problem shape with every operand in persistent device buffers, ``check_every``
    iterations captured in a hipGraph and replayed (the loop is launch-bound: ~40 small kernels per iteration);
    one host sync per replay for the stopping test.  Reused across LM steps and trial steps."""

    def __init__(self, lin, check_every):
        dt, dev = lin.R.dtype, lin.R.device
        self.lin_like = MultiGraphLinearization.__new__(MultiGraphLinearization)
        L = self.lin_like
        L.opt, L.params, L.group, L.E, L.dr, L.N, L.m = lin.opt, lin.params, None, lin.E, lin.dr, lin.N, lin.m
        L.R = torch.empty_like(lin.R)
        L.W = torch.empty_like(lin.W) if lin.W is not None else None
        L.slots = [(pi, torch.empty_like(idx), torch.empty_like(J)) for pi, idx, J in lin.slots]
        L._scatters = None
        z = lambda *s: torch.zeros(s, dtype=dt, device=dev)
        tot = sum(n * m for n, m in zip(lin.N, lin.m))
        self.shift_flat = z(tot)
        self.shift = lin._split(self.shift_flat)
        self.Binv = [z(n, m, m) for n, m in zip(lin.N, lin.m)]
        self.hip = lin.hip_ok()
        self.x, self.r, self.p = z(tot), z(tot), z(tot)
        self.rho = torch.zeros((), dtype=dt, device=dev)
        self.check_every = check_every
        self.rr = z(check_every)
        self.graph = None
        # HIP iteration: every scalar on the device (csrc/graph.hip, pplie_pcg_stage(0) + pplie_pcg_flat)
        self.z = z(tot)
        self.scal = z(2 * 4 * 32 * 32)
        self.cap = 1 << 16
        self.rr_hist = z(self.cap)
        self.it = torch.zeros(2, dtype=torch.int32, device=dev)
        self.tot = tot
        self.scal2 = z(2 * 8 * 32 * 32)            # PPLIE_PCG2_SCAL_ELEMS: the three-launch iteration's scalars
        self.use_mg3 = self.hip and len(lin.N) <= 4 and all(sc.hip for sc in lin.scatters())

    # the three-launch iteration (csrc/graph.hip pplie_mg3_*); False / PPLIE_MG3=0: the eleven-launch formulation below
    mg3 = __import__("os").environ.get("PPLIE_MG3", "1") != "0"

    def _mg3_plan(self):
        """work items of pplie_mg3_jt for the current incidence lists (rebuilt when the edge list changes): every (slot, row) list cut
        into chunks of MG3_CHUNK incidences, the items of a row contiguous and in (slot, chunk) order; rows without any incidence get
        one empty item (their y is shift o p)"""
        L = self.lin_like
        hit = self.__dict__.get('_mg3')
        if hit is not None and hit[0] is L._scatters:
            return hit[1]
        dev = self.p.device
        scs = L._scatters
        rows0, g_all, s_all, b_all, e_all = [], [], [], [], []
        base = 0
        for k, n in enumerate(L.N):
            rows0.append(base)
            covered = torch.zeros(n, dtype=torch.bool, device=dev)
            for si, ((pi, _, _), sc) in enumerate(zip(L.slots, scs)):
                if pi != k:
                    continue
                ptr = sc.ptr.long()
                cnt = ptr[1:] - ptr[:-1]
                per = (cnt + MG3_CHUNK - 1) // MG3_CHUNK
                row = torch.repeat_interleave(torch.arange(n, device=dev), per)
                first = torch.cumsum(per, 0) - per
                ch = torch.arange(row.numel(), device=dev) - first[row]
                beg = ptr[row] + ch * MG3_CHUNK
                g_all.append(row + base), s_all.append(torch.full_like(row, si)), b_all.append(beg)
                e_all.append(torch.minimum(beg + MG3_CHUNK, ptr[row + 1]))
                covered |= cnt > 0
            empty = (~covered).nonzero().reshape(-1)
            if empty.numel():
                g_all.append(empty + base), s_all.append(torch.full_like(empty, -1)), b_all.append(torch.zeros_like(empty))
                e_all.append(torch.zeros_like(empty))
            base += n
        g, sl, bg, en = (torch.cat(t) for t in (g_all, s_all, b_all, e_all))
        # order: SHORT rows first (one item of <= 16 incidences: four of them share a wavefront), then the rest; within each class by
        # (row, slot, chunk) so that a row's items are contiguous
        row_items = torch.bincount(g, minlength=base)
        short = (row_items[g] == 1) & (en - bg <= 16)
        bounds = torch.tensor([0] + list(L.N), device=dev).cumsum(0)
        width = torch.tensor(list(L.m), device=dev)[torch.searchsorted(bounds, g, right=True) - 1]
        tiny = short & (en - bg <= 8) & (width <= 4)                   # eight of these share a wavefront
        key = (g * (len(L.slots) + 1) + (sl + 1)) * (1 << 22) + (bg // MG3_CHUNK) % (1 << 22)
        klass = torch.where(tiny, 0, torch.where(short, 1, 2))
        order = torch.argsort(key + klass * (1 << 61), stable=True)
        g, sl, bg, en = g[order], sl[order], bg[order], en[order]
        nshort, ntiny = int(short.sum()), int(tiny.sum())
        items = torch.stack([g, sl, bg, en], 1).to(torch.int32).contiguous()
        first = torch.full((base,), -1, dtype=torch.int64, device=dev)
        pos = torch.arange(g.numel(), device=dev)
        first.scatter_reduce_(0, g, pos, reduce="amin", include_self=False)       # index of a row's first item (its items are contiguous)
        row_first = first.to(torch.int32)
        row_items = row_items.to(torch.int32)
        P = len(L.N)
        S = len(L.slots)
        offs, o = [], 0
        for n, m in zip(L.N, L.m):
            offs.append(o)
            o += n * m
        plan = dict(
            N=(ctypes.c_int64 * P)(*L.N), off=(ctypes.c_int64 * P)(*offs), m=(ctypes.c_int * P)(*L.m),
            Binv=(ctypes.c_void_p * P)(*[b.data_ptr() for b in self.Binv]),
            slot_param=(ctypes.c_int * S)(*[pi for pi, _, _ in L.slots]), J=(ctypes.c_void_p * S)(*[J.data_ptr() for _, _, J in L.slots]),
            perm=(ctypes.c_void_p * S)(*[sc.perm.data_ptr() for sc in scs]), ptr=(ctypes.c_void_p * S)(*[sc.ptr.data_ptr() for sc in scs]),
            items=items, row_first=row_first, row_items=row_items, nitems=int(items.shape[0]), nshort=nshort, ntiny=ntiny,
            # (tagged 64-bit words: 8 per item in fp32, 16 in fp64; zeroed per solve -- iteration numbers are the tags)
            part=torch.zeros(items.shape[0] * 8 * (1 if self.p.dtype == torch.float32 else 2), dtype=torch.int64, device=dev),
            cnt=torch.zeros(base, dtype=torch.int32, device=dev),
            y=torch.empty_like(self.p), P=P, S=S)
        self._mg3 = (L._scatters, plan)
        return plan

    def _iteration_mg3(self):
        """3 launches: q_e = W J p (edge-parallel), y = J^T q + shift o p with p.y / y.z / y.Binv y (row-parallel work items),
        the whole vector update with alpha and beta from the reduced scalars (csrc/graph.hip pplie_mg3_*)"""
        L, lib, st = self.lin_like, _C.library(), _C.stream_ptr(self.p.device)
        sfx = L._sfx()
        pl = self._mg3_plan()
        xs = L._split(self.p)
        S = pl["S"]
        arr = lambda ts: (ctypes.c_void_p * S)(*[t.data_ptr() for t in ts])
        q = self.__dict__.get('_mg3_q')
        if q is None or q.shape != (L.E, L.dr) or q.dtype != self.p.dtype:
            q = self._mg3_q = torch.empty((L.E, L.dr), dtype=self.p.dtype, device=self.p.device)
        ms = (ctypes.c_int * S)(*[J.shape[-1] for _, _, J in L.slots])
        with _C._on_device(self.p.device):
            _C.check(lib.symbol("pplie_mg_jtimes" + sfx, _MGJ_SIG)(
                S, arr([J for _, _, J in L.slots]), arr([i for _, i, _ in L.slots]), arr([xs[pi] for pi, _, _ in L.slots]),
                ms, L.W.data_ptr() if L.W is not None else None, q.data_ptr(), L.E, L.dr, st), "pplie_mg_jtimes")
            _C.check(lib.symbol("pplie_mg3_jt" + sfx, _MG3_JT_SIG)(
                pl["P"], pl["N"], pl["off"], pl["m"], pl["Binv"], S, pl["slot_param"], pl["J"], pl["perm"], pl["ptr"],
                pl["items"].data_ptr(), pl["row_first"].data_ptr(), pl["row_items"].data_ptr(), pl["nitems"], pl["nshort"], pl["ntiny"], pl["part"].data_ptr(),
                pl["cnt"].data_ptr(), q.data_ptr(), self.p.data_ptr(), self.z.data_ptr(), self.shift_flat.data_ptr(), pl["y"].data_ptr(),
                self.scal2.data_ptr(), self.rr_hist.data_ptr(), self.it.data_ptr(), self.cap, L.dr, st), "pplie_mg3_jt")
            _C.check(lib.symbol("pplie_mg3_step" + sfx, _MG3_STEP_SIG)(
                pl["P"], pl["N"], pl["off"], pl["m"], pl["Binv"], self.x.data_ptr(), self.r.data_ptr(), self.p.data_ptr(),
                pl["y"].data_ptr(), self.z.data_ptr(), self.scal2.data_ptr(), self.it.data_ptr(), st), "pplie_mg3_step")

    def _iteration_hip(self):
        if self.mg3 and self.use_mg3:
            return self._iteration_mg3()
        """11 launches: q = H p (1 + one per slot), q += shift o p with p.q, x / r update with |r|^2, z = Binv r per parameter,
        r.z, p = z + beta p -- scalars stay in the slot-spread device sets of the pose-graph PCG."""
        L, lib, st = self.lin_like, _C.library(), _C.stream_ptr(self.p.device)
        sfx = L._sfx()
        q = L.matvec_flat(self.p, None)
        stage, flat = lib.symbol("pplie_pcg_stage" + sfx, _STAGE_SIG), lib.symbol("pplie_pcg_flat" + sfx, _FLAT_SIG)
        with _C._on_device(self.p.device):
            _C.check(stage(0, None, None, self.p.data_ptr(), q.data_ptr(), None, None, self.shift_flat.data_ptr(), self.scal.data_ptr(),
                           None, self.it.data_ptr(), self.cap, self.tot, 1, st), "pplie_pcg_stage")
            _C.check(flat(0, self.x.data_ptr(), self.r.data_ptr(), self.p.data_ptr(), q.data_ptr(), None, self.scal.data_ptr(),
                          self.rr_hist.data_ptr(), self.it.data_ptr(), self.cap, self.tot, st), "pplie_pcg_flat")
            fn = lib.symbol("pplie_block_matvec" + sfx, _BMV_SIG)
            for Bi, x, y, n, m in zip(self.Binv, L._split(self.r), L._split(self.z), L.N, L.m):
                _C.check(fn(Bi.data_ptr(), x.data_ptr(), y.data_ptr(), n, m, st), "pplie_block_matvec")
            for s_ in (1, 2):
                _C.check(flat(s_, None, self.r.data_ptr(), self.p.data_ptr(), None, self.z.data_ptr(), self.scal.data_ptr(),
                              self.rr_hist.data_ptr(), self.it.data_ptr(), self.cap, self.tot, st), "pplie_pcg_flat")

    def _iteration(self, k):
        L = self.lin_like
        if self.hip:
            return self._iteration_hip()
        ps = L._split(self.p)
        q = L._cat([y + sh * x for y, sh, x in zip(L._Hp(ps), self.shift, ps)])
        pq = (self.p * q).sum()
        alpha = torch.where(pq != 0, self.rho / pq, torch.zeros_like(pq))       # p.q = 0 only once r = 0
        self.x.add_(alpha * self.p)
        self.r.sub_(alpha * q)
        zv = L._cat([(Bi * x.unsqueeze(-2)).sum(-1) for Bi, x in zip(self.Binv, L._split(self.r))])
        rho_new = (self.r * zv).sum()
        beta = torch.where(self.rho != 0, rho_new / self.rho, torch.zeros_like(rho_new))
        self.p.mul_(beta).add_(zv)
        self.rho.copy_(rho_new)
        self.rr[k] = (self.r * self.r).sum()

    def solve(self, lin, shift, Binv, b, tol, maxiter):
        L = self.lin_like
        for (_, di, dJ), (_, si, sJ) in zip(L.slots, lin.slots):
            di.copy_(si)
            dJ.copy_(sJ)
        new = lin.scatters()
        if L._scatters is None or any(a is not b for a, b in zip(L._scatters, new)):
            L._scatters, self.graph = new, None                     # new edge list: the captured incidence lists are stale
        if L.W is not None:
            L.W.copy_(lin.W)
        for d, s_ in zip(self.shift, shift):
            d.copy_(s_)
        for d, s_ in zip(self.Binv, Binv):
            d.copy_(s_)
        bv = lin._cat(b)
        self.x.zero_()
        self.r.copy_(bv)
        zv = L._cat([(Bi * x.unsqueeze(-2)).sum(-1) for Bi, x in zip(self.Binv, L._split(self.r))])
        self.p.copy_(zv)
        self.rho.copy_((self.r * zv).sum())
        if self.hip:
            self.scal.zero_()
            self.scal[0] = self.rho                                 # set 0, rho, slot 0
            self.it.zero_()
            if self.mg3 and self.use_mg3:
                self.z.copy_(zv)
                self.scal2.zero_()
                self.scal2[0] = self.rho
                pl = self._mg3_plan()
                pl["part"].zero_()                                  # (the partials' tags are iteration numbers of THIS solve)
                pl["cnt"].zero_()
        bn2 = float((bv * bv).sum())
        if bn2 == 0.0:
            return self.x.clone(), 0
        done = 0
        maxiter = min(maxiter, self.cap - self.check_every)
        while done < maxiter:
            if self.graph is None and done > 0:                   # first block runs eagerly (warm-up), then capture
                g = torch.cuda.CUDAGraph()
                with _C.graph_capture(g):
                    for k in range(self.check_every):
                        self._iteration(k)
                self.graph = g
            if self.graph is not None:
                self.graph.replay()
            else:
                for k in range(self.check_every):
                    self._iteration(k)
            done += self.check_every
            if self.hip and self.mg3 and self.use_mg3:              # |r|^2 of the last iteration still sits in its slot-spread accumulator
                rr = float(self.scal2.view(2, 8, 32, 32)[(done - 1) & 1, 2, :, 0].sum())
            else:
                rr = float(self.rr_hist[done - 1]) if self.hip else float(self.rr[-1])
            if rr <= tol * tol * bn2:
                break
        return self.x.clone(), done


def try_multigraph_linearization(opt, pg, input, target, weight, R, params, rec, cache, sig):
    """A MultiGraphLinearization if every optimised parameter enters the residual(s) only through recorded gathers.

    Several residuals (reprojection errors + priors on some cameras, ...; reference optimizer.py:644-654 stacks them
    as rows of one dense J) are stacked as rows here too: residual i owns rows [o_i, o_i + E_i), is attributed the
    gathers it depends on, gets ``corrector[i]`` / ``weight[i]``, and is zero-padded to the widest residual.  The
    j-th gather of parameter k in each residual shares ONE slot (row ranges are disjoint), so a prior on the cameras
    adds rows to the camera slot instead of a mostly-zero slot of its own."""
    key = (sig, "multi")
    if cache.get(key) is False or len(opt.corrector) not in (1, len(R)) or any(r.dim() < 2 for r in R):
        return None
    if weight is not None and len(R) > 1 and not (isinstance(weight, (tuple, list)) and len(weight) == len(R)):
        return None
    pos = {id(p): k for k, p in enumerate(params)}
    # (gather indices may be negative, as in torch indexing and the reference's dense path: every scatter / segment sum /
    #  kernel below addresses rows directly, so they are normalised here, once)
    events = [(src, torch.where(ix < 0, ix + src.shape[0], ix), out) for src, ix, out in rec.events if id(src) in pos and src.dim() == 2
              and out.numel() == ix.numel() * src.shape[-1]]
    if not events or len(events) != len(rec.events) or {id(s) for s, _, _ in events} != set(pos):
        cache[key] = False
        return None
    outs_all = [out for _, _, out in events]
    parts = []
    closed = True
    with torch.enable_grad():
        for r in R:
            dr = r.shape[-1]
            E = r.numel() // dr
            if len(R) == 1:
                mine = events
            else:
                seen = torch.autograd.grad(r.sum(), outs_all, retain_graph=True, allow_unused=True)
                mine = [ev for ev, g in zip(events, seen) if g is not None]
            if not mine or any(ix.numel() != E for _, ix, _ in mine):
                cache[key] = False
                return None
            Jcat = rec.closed_blocks(r, mine) if getattr(opt, "closed_form", True) else None   # [E, dr, sum widths]
            closed = closed and Jcat is not None
            if Jcat is None:
                Jcat = _blocks.jacobian_blocks([r], [out for _, _, out in mine])
            parts.append((r, E, dr, mine, Jcat))
        if cache.get(key) is None:
            # probe: u^T dR/dparams by one real backward == scatter-add of the per-observation blocks
            us = [torch.randn_like(r) for r in R]
            true = torch.autograd.grad(list(R), params, us, retain_graph=True, allow_unused=True)
            got = [torch.zeros_like(p) for p in params]
            for u, (r, E, dr, mine, Jcat) in zip(us, parts):
                contrib = (u.reshape(E, dr).unsqueeze(-1) * Jcat).sum(-2)
                off = 0
                for src, ix, _ in mine:
                    w = src.shape[-1]
                    got[pos[id(src)]].index_add_(0, ix.reshape(-1), contrib[:, off:off + w])
                    off += w
            ok = True
            for t, gt in zip(true, got):
                t = torch.zeros_like(gt) if t is None else t
                ok = ok and bool((gt - t).abs().max() <= 1e-3 * t.abs().max().clamp_min(torch.finfo(t.dtype).tiny))
            cache[key] = ok
    if not cache[key]:
        return None
    dr_max = max(p[2] for p in parts)
    E_tot = sum(p[1] for p in parts)
    dt, dev = parts[0][0].dtype, parts[0][0].device
    weighted = weight is not None
    Rs, Ws, rows0 = [], [], 0
    slot_of = {}                     # (parameter, ordinal of its gather inside a residual) -> [idx [E_tot], J [E_tot, dr_max, m]]
    for i, (r, E, dr, mine, Jcat) in enumerate(parts):
        ms = [_tangent_width(src) for src, _, _ in mine]
        cols, off = [], 0
        for (src, _, _), m in zip(mine, ms):
            cols.append(Jcat[:, :, off:off + m])
            off += src.shape[-1]
        corrector = opt.corrector[0] if len(opt.corrector) == 1 else opt.corrector[i]
        from .posegraph import Trivial_type, _fused_rows
        rr, Jc0 = r.detach().reshape(E, dr), torch.cat(cols, -1)             # row-local on the concatenated tangent blocks
        fused = None if isinstance(corrector, Trivial_type()) else _fused_rows(corrector, rr, Jc0)   # built-in kernel: one launch
        Rc, Jc = fused if fused is not None else corrector(R=rr, J=Jc0)
        Rp = torch.zeros((E, dr_max), dtype=dt, device=dev)
        Rp[:, :dr] = Rc
        Rs.append(Rp)
        if weighted:
            Wp = torch.eye(dr_max, dtype=dt, device=dev).repeat(E, 1, 1)
            w = weight[i] if isinstance(weight, (tuple, list)) else weight
            if w is not None:
                ws, ni = opt.model._weight_blocks(w, r)
                Wp[:, :dr, :dr] = ws.repeat(ni, 1, 1)
            Ws.append(Wp)
        ordinal, off = {}, 0
        for (src, ix, _), m in zip(mine, ms):
            k = pos[id(src)]
            j = ordinal.get(k, 0)
            ordinal[k] = j + 1
            if (k, j) not in slot_of:
                slot_of[(k, j)] = [torch.zeros(E_tot, dtype=torch.int64, device=dev), torch.zeros((E_tot, dr_max, m), dtype=dt, device=dev)]
            sl = slot_of[(k, j)]
            sl[0][rows0:rows0 + E] = ix.reshape(-1)
            sl[1][rows0:rows0 + E, :dr] = Jc[:, :, off:off + m]
            off += m
        rows0 += E
    slots = [(k, ix, J) for (k, _), (ix, J) in sorted(slot_of.items())]
    opt._last_blocks = "closed-form" if closed else "autograd"
    return MultiGraphLinearization(opt, torch.cat(Ws) if weighted else None, torch.cat(Rs), params, slots)
