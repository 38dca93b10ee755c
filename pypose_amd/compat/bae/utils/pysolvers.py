"""``PCG``: Jacobi-preconditioned conjugate gradient on a sparse (CSR) system -- what ``pypose.optim.solver.PCG`` resolves
to (solver.py:358-364); call shape of the reference's solvers, ``solver(A=, b=) -> x`` (solver.py:204)."""
import ctypes

import torch
from torch import nn

_P = ctypes.c_void_p
_PREP_SIG = [_P] * 10 + [ctypes.c_int64, ctypes.c_int, _P]
_SPMV_SIG = [_P] * 9 + [ctypes.c_int64, ctypes.c_double, ctypes.c_int, _P]
_STEP_SIG = [_P] * 8 + [ctypes.c_int64, _P]
_SCAL_ELEMS = 2 * 8 * 32 * 32          # PPLIE_CSR_PCG_SCAL_ELEMS (csrc/csr_pcg.hip)


def _hip_pcg(A, b, tol, maxiter, check):
    """The same iteration on the HIP kernels of csrc/csr_pcg.hip (two launches per iteration, the stop test on the device), or
    None when this call is not theirs: (x [n], iterations)."""
    from pypose_amd import _C          # (absolute: this namespace is imported as the top-level package `bae`)
    if A.layout != torch.sparse_csr or not b.is_cuda or b.dtype not in (torch.float32, torch.float64) or A.dtype != b.dtype \
            or _C._test_backend is not None or A.dim() != 2 or A.shape[0] != A.shape[1] or A.shape[0] != b.numel():
        return None
    crow, col, val = A.crow_indices(), A.col_indices(), A.values()
    if crow.dtype != col.dtype or crow.dtype not in (torch.int64, torch.int32) or not (crow.is_contiguous() and col.is_contiguous()
                                                                                       and val.is_contiguous()):
        return None
    n, dev, dt = b.numel(), b.device, b.dtype
    sfx = "_f32" if dt == torch.float32 else "_f64"
    lib = _C.library()
    idx64 = 1 if crow.dtype == torch.int64 else 0
    bc = b.contiguous()
    z = lambda: torch.empty(n, dtype=dt, device=dev)
    minv, x, r, zz, p, q = z(), z(), z(), z(), z(), z()
    scal = torch.zeros(_SCAL_ELEMS, dtype=dt, device=dev)
    it = torch.zeros(4, dtype=torch.int32, device=dev)
    st = _C.stream_ptr(dev)
    P = lambda t: t.data_ptr()
    with _C._on_device(dev):
        _C.check(lib.symbol("pplie_csr_pcg_prepare" + sfx, _PREP_SIG)(P(crow), P(col), P(val), P(bc), P(minv), P(x), P(r), P(zz), P(p),
                                                                     P(scal), n, idx64, st), "pplie_csr_pcg_prepare")
        spmv, step = lib.symbol("pplie_csr_pcg_spmv" + sfx, _SPMV_SIG), lib.symbol("pplie_csr_pcg_step" + sfx, _STEP_SIG)
        tol2 = float(tol) * float(tol)
        done = 0
        while True:
            for _ in range(check):
                # (one launch beyond maxiter is the stop test of the last iteration: its q is never applied)
                _C.check(spmv(P(crow), P(col), P(val), P(p), P(zz), P(minv), P(q), P(scal), P(it), n, tol2, idx64, st), "pplie_csr_pcg_spmv")
                if done >= maxiter:
                    break
                _C.check(step(P(x), P(r), P(p), P(q), P(zz), P(minv), P(scal), P(it), n, st), "pplie_csr_pcg_step")
                done += 1
            its, _, flag, _ = it.tolist()                      # the chunk's one read-back
            if flag == 2:
                raise AssertionError('Conjugate gradient produced NaN. Check your matrix (may not be positive-definite)')
            if flag == 1 or done >= maxiter:
                return x, int(its)


class PCG(nn.Module):
    def __init__(self, maxiter=None, tol=1e-5, check_every=8):
        super().__init__()
        self.maxiter, self.tol, self.check_every = maxiter, tol, max(1, int(check_every))
        self.iterations = 0
        self.route = None                      # "hip" / "torch": which implementation the last call took

    def forward(self, A, b, x=None, M=None):
        b = b.reshape(-1)
        n = b.numel()
        maxiter = self.maxiter if self.maxiter is not None else 10 * n
        if x is None and M is None and n > 0:
            hip = _hip_pcg(A, b, self.tol, maxiter, self.check_every)
            if hip is not None:
                self.route = "hip"
                self.iterations = hip[1]
                assert not torch.any(torch.isnan(hip[0])), 'Conjugate gradient produced NaN. Check your matrix (may not be positive-definite)'
                return hip[0].unsqueeze(-1)
        self.route = "torch"
        if A.layout == torch.sparse_csr:
            crow, col, val = A.crow_indices(), A.col_indices(), A.values()
            row = torch.repeat_interleave(torch.arange(n, device=col.device), crow[1:] - crow[:-1])
            on_diag = row == col
            diag = torch.zeros(n, dtype=val.dtype, device=val.device).index_add_(0, row[on_diag], val[on_diag])
        else:
            diag = A.diagonal() if not A.is_sparse else A.to_dense().diagonal()
        Minv = torch.where(diag != 0, 1.0 / diag, torch.ones_like(diag)) if M is None else None
        prec = (lambda r: Minv * r) if M is None else (lambda r: (M @ r.unsqueeze(-1)).squeeze(-1))
        mv = lambda v: (A @ v.unsqueeze(-1)).squeeze(-1)
        x = torch.zeros_like(b) if x is None else x.reshape(-1).clone()
        r = b - mv(x)
        z = prec(r)
        p = z.clone()
        rz = torch.dot(r, z)
        bnorm = b.norm()
        self.iterations = 0
        if float(bnorm) == 0.0:
            return x.unsqueeze(-1)
        # The stop test `|r| <= tol |b|` runs on the DEVICE every iteration and the host looks at it every `check` iterations: the
        # iteration that meets it is the last one that moves x and r (`live` multiplies the step from then on), so the result and
        # the iteration count are those of a loop that tests on the host every time -- without a synchronisation per iteration.
        check = self.check_every
        thresh = self.tol * bnorm
        live = torch.ones((), dtype=b.dtype, device=b.device)
        zero = torch.zeros((), dtype=b.dtype, device=b.device)
        count = torch.zeros((), dtype=torch.int64, device=b.device)
        for it in range(maxiter):
            Ap = mv(p)
            alpha = torch.where(live > 0, rz / torch.dot(p, Ap), zero)      # (a select: whatever the idle iterations divide by)
            x = x + alpha * p
            r = r - alpha * Ap
            count = count + live.to(torch.int64)
            live = live * (r.norm() > thresh).to(b.dtype)
            if (it + 1) % check == 0 and float(live) == 0.0:
                break
            z = prec(r)
            rz_new = torch.dot(r, z)
            on = live > 0                                            # (idle iterations leave p and rz where they were: all finite)
            p = torch.where(on, z + (rz_new / rz) * p, p)
            rz = torch.where(on, rz_new, rz)
        self.iterations = int(count)
        assert not torch.any(torch.isnan(x)), 'Conjugate gradient produced NaN. Check your matrix (may not be positive-definite)'
        return x.unsqueeze(-1)
