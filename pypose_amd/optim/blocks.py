"""Block-structured linear algebra behind the LM / GN fast paths.

A model whose residual row ``n`` depends only on parameter row ``n`` (B independent problems,
e.g. the reference's ``InvNet`` README example) has a block-diagonal Jacobian.  The reference
still builds the dense ``[sum N_res, sum N_param]`` matrix with ``N_res`` vmapped backward sweeps
(optimizer.py:647) -- O(B^2) memory, impossible beyond B ~ 10^3.  Here:

* :func:`jacobian_blocks`   ``d_res`` batched backward sweeps (one batched ``autograd.grad``) give
                            the blocks ``[B, d_res, d_par]``;
* :func:`probe_block_structure`  one extra backward with a random cotangent verifies the
                            independence assumption (otherwise the caller uses the dense path);
* :func:`normal_equations`  ``A_b = J_b^T W_b J_b``, ``g_b = J_b^T W_b r_b``  (HIP: pplie_block_normal_eq);
* :func:`chol_solve`        ``x_b = A_b^-1 (-g_b)``                           (HIP: pplie_block_chol_solve);
* :class:`BlockJacobian`    the ``J`` handed to ``strategy.update`` -- supports ``J @ D``.
"""
from __future__ import annotations

import ctypes

import torch

from .. import _C

_NE_SIG = [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
_CH_SIG = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
_DCH_SIG = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
_HIP_DR, _HIP_DP = (3, 4, 6, 7), (3, 4, 5, 6, 7, 8)


def lie_manifold_width(p):
    """Tangent width of a Lie-GROUP parameter (whose gradients are zero-padded to the stored width), else None.
    Duck-typed on ``.ltype`` so that LieTensors of an activated reference ``pypose`` count as well."""
    lt = getattr(p, "ltype", None)
    if lt is None or not hasattr(lt, "manifold") or not hasattr(lt, "dimension"):
        return None
    return int(lt.manifold[0]) if tuple(lt.dimension) != tuple(lt.manifold) else None


def _suffix(t):
    return {torch.float32: "_f32", torch.float64: "_f64"}.get(t.dtype)


_GRAM_SIG = [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]


def normal_equations(J, R, W=None):
    """J [n,dr,dp], R [n,dr], W [n,dr,dr] | None  ->  A [n,dp,dp] = J^T W J,  g [n,dp] = J^T W R."""
    n, dr, dp = J.shape
    if _C._test_backend is not None:
        return _C._test_backend("block_normal_eq", [J, R] + ([W] if W is not None else []), None)
    if J.is_cuda and _suffix(J) and dr in _HIP_DR and dp in _HIP_DP and n > 0:
        J, R = J.contiguous(), R.contiguous()
        W = W.contiguous() if W is not None else None
        A = torch.empty((n, dp, dp), dtype=J.dtype, device=J.device)
        g = torch.empty((n, dp), dtype=J.dtype, device=J.device)
        fn = _C.library().symbol("pplie_block_normal_eq" + _suffix(J), _NE_SIG)
        with _C._on_device(J.device):
            code = fn(J.data_ptr(), R.data_ptr(), W.data_ptr() if W is not None else None, A.data_ptr(), g.data_ptr(),
                      n, dr, dp, _C.stream_ptr(J.device))
        _C.check(code, "pplie_block_normal_eq")
        return A, g
    if J.is_cuda and _suffix(J) and dp <= 15 and n > 0 and W is None:
        # large residual stacks: the Gram product [J | R]^T [J | R] of every problem on the matrix cores, one wavefront per
        # problem (csrc/gram_mfma.hip)
        J, R = J.contiguous(), R.contiguous()
        A = torch.empty((n, dp, dp), dtype=J.dtype, device=J.device)
        g = torch.empty((n, dp), dtype=J.dtype, device=J.device)
        fn = _C.library().symbol("pplie_block_gram_mfma" + _suffix(J), _GRAM_SIG)
        with _C._on_device(J.device):
            code = fn(J.data_ptr(), R.data_ptr(), A.data_ptr(), g.data_ptr(), None, n, dr, dp, _C.stream_ptr(J.device))
        _C.check(code, "pplie_block_gram_mfma")
        return A, g
    # anything else (a weight W: J^T W J is not the Gram product of one matrix; host tensors): batched torch ops
    JtW = J.mT if W is None else J.mT @ W
    return JtW @ J, (JtW @ R.unsqueeze(-1)).squeeze(-1)


def chol_solve(A, g):
    """x [n,dp] with A_b x_b = -g_b, A_b SPD (lower Cholesky). NaN rows signal a failed factorisation."""
    n, dp, _ = A.shape
    if _C._test_backend is not None:
        return _C._test_backend("block_chol_solve", [A, g], None)
    if A.is_cuda and _suffix(A) and dp in _HIP_DP and n > 0:
        A, g = A.contiguous(), g.contiguous()
        x = torch.empty((n, dp), dtype=A.dtype, device=A.device)
        fn = _C.library().symbol("pplie_block_chol_solve" + _suffix(A), _CH_SIG)
        with _C._on_device(A.device):
            code = fn(A.data_ptr(), g.data_ptr(), x.data_ptr(), n, dp, _C.stream_ptr(A.device))
        _C.check(code, "pplie_block_chol_solve")
        return x
    L, _ = torch.linalg.cholesky_ex(A)
    return torch.cholesky_solve(-g.unsqueeze(-1), L).squeeze(-1)


def damped_chol_solve(A, g, s, dmin, dmax):
    """x [n,dp] with (A, diagonal clamped to [dmin, dmax] and scaled by s) x = -g, or None where only the two-pass route applies
    (off the HIP path, block sizes outside the kernel table).  A is NOT modified: the LM trial loop damps by growing ``s``."""
    n, dp, _ = A.shape
    if _C._test_backend is not None or not (A.is_cuda and _suffix(A) and dp in _HIP_DP and n > 0):
        return None
    A, g = A.contiguous(), g.contiguous()
    x = torch.empty((n, dp), dtype=A.dtype, device=A.device)
    fn = _C.library().symbol("pplie_block_damped_chol_solve" + _suffix(A), _DCH_SIG)
    with _C._on_device(A.device):
        code = fn(A.data_ptr(), g.data_ptr(), x.data_ptr(), n, dp, float(s), float(dmin), float(dmax), _C.stream_ptr(A.device))
    _C.check(code, "pplie_block_damped_chol_solve")
    return x


_COT_CACHE = {}
_COT_CACHE_BYTES = 1 << 28


def _one_hot_cotangents(d, r):
    """[d, *r.shape]: cotangent k is one-hot in residual component k, for every row.  Constant per (shape, dtype, device):
    large ones are kept MATERIALISED (the first backward kernel needs contiguous rows; as an expanded view the d-fold copy
    is made again on every linearisation -- 144 MB at 10^6 problems), one entry, up to 256 MB."""
    eye = torch.eye(d, dtype=r.dtype, device=r.device)
    view = eye.view((d,) + (1,) * (r.dim() - 1) + (d,)).expand((d,) + tuple(r.shape))
    nbytes = view.numel() * r.element_size()
    if not r.is_cuda or nbytes < (1 << 22) or nbytes > _COT_CACHE_BYTES:
        return view
    key = (d, tuple(r.shape), r.dtype, r.device)
    hit = _COT_CACHE.get(key)
    if hit is None:
        _COT_CACHE.clear()
        hit = _COT_CACHE[key] = view.contiguous()
    return hit


def jacobian_blocks(residuals, params):
    """Per-row Jacobian blocks under the row-independence hypothesis.

    residuals: list of tensors ``lead + (d_i,)`` (graph attached); params: list of tensors
    ``lead_j + (w_j,)`` with the same number of rows n.  Returns ``J [n, sum d_i, sum w_j]``:
    ``d_i`` cotangents (one-hot in the residual component, broadcast over rows) are pushed through
    ONE batched backward; under row independence the gradient landing in parameter row n is
    exactly d r[n, i] / d p[n, :].
    """
    n = residuals[0].numel() // residuals[0].shape[-1]
    rows = []
    for k, r in enumerate(residuals):
        d = r.shape[-1]
        cot = _one_hot_cotangents(d, r)
        grads = torch.autograd.grad(r, params, cot, is_grads_batched=True, retain_graph=True, allow_unused=True)
        cols = []
        for p, gr in zip(params, grads):
            w = p.shape[-1]
            if gr is None:
                cols.append(torch.zeros((n, d, w), dtype=r.dtype, device=r.device))
            else:
                cols.append(gr.reshape(d, n, w).permute(1, 0, 2))
        rows.append(cols[0] if len(cols) == 1 else torch.cat(cols, dim=-1))      # (cat of one tensor is still a copy)
    # one residual, one parameter: the [d, n, w] sweep output is transposed exactly once
    return (rows[0] if len(rows) == 1 else torch.cat(rows, dim=-2)).contiguous()


def probe_block_structure(residuals, params, J, rtol=1e-3):
    """True iff ``J`` (from :func:`jacobian_blocks`) reproduces a random vector-Jacobian product
    of the real model: u^T (dR/dp) computed by one ordinary backward must equal the block
    contraction.  Any cross-row dependence makes the two differ by O(1)."""
    n = J.shape[0]
    us = [torch.randn_like(r) for r in residuals]
    true = torch.autograd.grad(residuals, params, us, retain_graph=True, allow_unused=True)
    u = torch.cat([x.reshape(n, -1) for x in us], dim=-1)
    got = (u.unsqueeze(-1) * J).sum(-2)
    col = 0
    for p, t in zip(params, true):
        w = p.shape[-1]
        t = torch.zeros((n, w), dtype=J.dtype, device=J.device) if t is None else t.reshape(n, w)
        g = got[:, col:col + w]
        col += w
        scale = t.abs().max().clamp_min(torch.finfo(J.dtype).tiny)
        if not bool(((g - t).abs().max() <= rtol * scale)):
            return False
    return True


class BlockJacobian:
    """``J`` of a block-diagonal problem, for ``strategy.update(..., J=, D=, R=)``.

    blocks ``[n, dr, dp]``; ``D`` arrives as the flat ``[sum N_param, 1]`` step in the optimizer's
    parameter order (parameter by parameter); ``J @ D`` returns ``[n * dr, 1]`` in row-block order
    (the order of :attr:`R` built by the same linearisation)."""

    gain_is_costly = True       # (tensor passes over the blocks; the pose-graph operator has a kernel for it)

    def __init__(self, blocks, param_widths, R=None):
        self.blocks, self.param_widths, self.R = blocks, list(param_widths), R

    def gain_terms(self, D):
        """device tensor [(J D).(J D), (J D).R] -- all the built-in damping strategies need of J, D and R (strategy.py:144,
        261); the optimizer reads it back together with the trial's loss (one synchronisation instead of the strategy's own
        tensor comparisons)"""
        if self.R is None:
            return None
        JD = (self.blocks * self.step_to_blocks(D).unsqueeze(-2)).sum(-1)
        return torch.stack([(JD * JD).sum(), (JD * self.R.reshape(JD.shape)).sum()])

    def step_to_blocks(self, D):
        n = self.blocks.shape[0]
        flat, out, off = D.reshape(-1), [], 0
        for w in self.param_widths:
            out.append(flat[off:off + n * w].view(n, w))
            off += n * w
        return torch.cat(out, dim=-1)

    def blocks_to_step(self, Db):
        cols, off = [], 0
        for w in self.param_widths:
            cols.append(Db[:, off:off + w].reshape(-1))
            off += w
        return torch.cat(cols).view(-1, 1)

    def __matmul__(self, D):
        # broadcast-multiply-sum: a batched GEMM of 10^6 6x7 blocks through hipBLASLt takes milliseconds
        return (self.blocks * self.step_to_blocks(D).unsqueeze(-2)).sum(-1).reshape(-1, 1)

    @property
    def shape(self):
        n, dr, dp = self.blocks.shape
        return torch.Size([n * dr, n * dp])
