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


def lm_se3inv_trial(R, P, X, scale, dmin, dmax):
    """One LM trial step of B independent problems r = Log(P X) (reference README.md:120-129 run through
    optimizer.py:644-679): J = [se3_Jl_inv(r) | 0] (operation.py:385-395 SE3_Log backward, :905-908 SE3_Mul
    backward), A = J^T J with the diagonal clamped to [dmin, dmax] and scaled by ``scale`` = prod(1 + damping),
    d = -A^-1 J^T r (Cholesky), P' = Exp(d) P.  Returns P', d (zero-padded to 7), and the four sums
    [ |Log(P' X)|^2, |r|^2, (J d).(J d), (J d).r ]."""
    from oracle import lie_np
    J = lie_np.se3_Jl_inv(R)                                     # [n,6,6]
    A = np.swapaxes(J, -1, -2) @ J
    g = (np.swapaxes(J, -1, -2) @ R[..., None])[..., 0]
    idx = np.arange(6)
    A[:, idx, idx] = np.clip(A[:, idx, idx], dmin, dmax) * scale
    (d,) = block_chol_solve(A, g)
    Pn = lie_np.se3_mul_fwd(lie_np.se3_exp_fwd(d)[0], P)[0]
    rn = lie_np.se3_log_fwd(lie_np.se3_mul_fwd(Pn, X)[0])[0]
    Jd = (J @ d[..., None])[..., 0]
    sums = np.array([(rn * rn).sum(), (R * R).sum(), (Jd * Jd).sum(), (Jd * R).sum()])
    return Pn, np.concatenate([d, np.zeros_like(d[:, :1])], -1), sums


def pgo_linearize(nodes, idx, Z):
    """Per-edge residuals and Jacobian blocks of the reference's PoseGraph model (examples/module/pgo/pgo.py:15-25),
    r = Log(Z^-1 n_i^-1 n_j), by the chain of backward rules the reference's autograd applies:
    SE3_Log.backward g @ se3_Jl_inv(r) (operation.py:385-395), SE3_Mul.backward Y_grad = g @ Adj(X) (:905-908),
    SE3_Inv.backward X_grad = -g @ Adj(Y) (:992-998).  Returns R [E,6], J [E,2,6,6] (left tangents)."""
    from oracle import lie_np
    n1, n2 = nodes[idx[:, 0]], nodes[idx[:, 1]]
    a = lie_np.se3_inv_fwd(Z)[0]
    b = lie_np.se3_inv_fwd(n1)[0]
    c = lie_np.se3_mul_fwd(a, b)[0]
    u = lie_np.se3_mul_fwd(c, n2)[0]
    R = lie_np.se3_log_fwd(u)[0]
    Ji = lie_np.se3_Jl_inv(R)                                  # d r / d u
    J2 = Ji @ lie_np.SE3_Adj(c)                                # through u = c * n2 to n2
    J1 = -(Ji @ lie_np.SE3_Adj(a)) @ lie_np.SE3_Adj(b)         # through c = a * b to b, through b = n1^-1 to n1
    return R, np.stack([J1, J2], axis=1)
