"""ORACLE (test infrastructure only) -- numpy restatement of the per-block linear algebra that
pypose/optim/optimizer.py:655-668 + solver.py:213-216 perform on one dense matrix:
A = J^T W J, g = J^T W r per block, and x = A^-1 (-g) by Cholesky."""
import numpy as np


def block_normal_eq(J, R, W=None):
    JtW = np.swapaxes(J, -1, -2) if W is None else np.swapaxes(J, -1, -2) @ W
    return (JtW @ J).astype(J.dtype), (JtW @ R[..., None])[..., 0].astype(J.dtype)


def block_chol_solve(A, g):
    x = np.full(g.shape, np.nan, dtype=g.dtype)
    for i in range(A.shape[0]):
        try:
            L = np.linalg.cholesky(A[i].astype(np.float64) if A.dtype == np.float64 else A[i])
        except np.linalg.LinAlgError:
            continue
        y = np.linalg.solve(L, -g[i])
        x[i] = np.linalg.solve(L.T, y)
    return (x,)
