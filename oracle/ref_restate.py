"""ORACLE (test infrastructure only) -- full-size CPU restatement of the reference's Levenberg-Marquardt loop for the
BASELINE configs the reference itself cannot execute ("CPU restatement needed?").

The reference's dense LM (pypose/optim/optimizer.py:644-679) builds a dense ``[N_res, N_par]`` Jacobian with ``modjac``:
InvNet at 10^6 problems would need a ``[6e6, 7e6]`` matrix, a 10 k-pose graph a ``[2.4e5, 7e4]`` one.  This file runs the
SAME loop -- ``A = J^T J``, ``A.diagonal().clamp_(min, max)`` (:657), the compounding ``A.diagonal() *= 1 + damping``
(:666), ``solver(A, b = -J^T R)`` (:668), ``update_parameter`` (:135-140), ``strategy.update`` (:672), accept / reject
(:673-677) -- with J kept in the block form its structure allows, and every piece of arithmetic taken from the
reference package itself (imported from oracle/_ref through ``ref_loader``; nothing here re-derives a formula):

* residuals from the reference's LieTensor ops (``Inv``, ``@``, ``Log``: operation.py:373-395, 829-927, 930-1021);
* Jacobian blocks from ``se3_Jl_inv`` (operation.py:68-75) and ``SE3_Adj`` (:202-210), composed by the backward rules of
  ``SE3_Log`` (:385-395), ``SE3_Mul`` (:905-908) and ``SE3_Inv`` (:992-998);
* InvNet (configs[2]): ``[B,7,7]`` blocks, ``torch.linalg.cholesky_ex`` + ``cholesky_solve`` exactly as the reference's
  ``Cholesky`` solver does on its one matrix (solver.py:204-216) -- the dense A of B independent problems IS block diagonal;
* pose graph (metric / configs[3]): J as ``torch.sparse_csr`` ``[6E, 7N]``, ``A = J^T J`` in CSR (what the reference's own
  ``sparse=True`` branch does, optimizer.py:640-643), solved by the reference's ``CG`` (solver.py:276-340, accepts CSR and
  a preconditioner matrix ``M`` applied by matmul) with the block-Jacobi ``M`` the bae plugin's PCG uses;
* the damping policy is the reference's own strategy object (strategy.py:41-46, 134-151, 248-274), fed J / D / R.

Pinned (tests/test_ref_restate.py, CPU): the per-step loss / damping / reject sequences and final poses of this loop equal
those recorded from the REAL reference ``pp.optim.LM`` (tests/golden/lm_golden2.npz: InvNet B = 64 / 1024, pose graphs
N = 50 / 200) to 1e-9.  Used by: tests/ (-m gpu parity at the full BASELINE sizes) and bench.py's per-leg ``cpu_baseline``.
"""
from __future__ import annotations

import time

import torch

from oracle import ref_loader


def _rpp():
    return ref_loader.load()


def _strategy(rpp, name, **kw):
    S = rpp.optim.strategy
    return {"constant": S.Constant, "adaptive": S.Adaptive, "trustregion": S.TrustRegion}[name.lower()](**kw)


class _BlockJ:
    """``J`` of B independent problems, blocks [B, dr, dp]: ``J @ D`` for strategy.update (strategy.py:143, 260)."""

    def __init__(self, Jb):
        self.Jb = Jb

    def __matmul__(self, D):
        B, dr, dp = self.Jb.shape
        return (self.Jb @ D.view(B, dp, 1)).reshape(-1, 1)


# ---------------------------------------------------------------------------------------------------------------------
# configs[2]: LM on InvNet, B independent SE3 problems (README.md:120-129 through optimizer.py:644-679)
# ---------------------------------------------------------------------------------------------------------------------
def invnet_lm(init, inp, steps, strategy="constant", strategy_kw=None, dmin=1e-6, dmax=1e32, reject=16, sample=None):
    """``steps`` LM steps on ``pose <- argmin |Log(pose @ inp)|^2`` from ``init`` [B,7] / ``inp`` [B,7] (torch CPU tensors).
    Returns {"loss", "damping", "reject", "final"} like tests/golden/make_lm_golden.py:run; ``sample`` (an index / slice):
    also "poses" = the poses ``P[sample]`` after every step."""
    rpp = _rpp()
    op = rpp.lietensor.operation
    strat = _strategy(rpp, strategy, **(strategy_kw or {}))
    pg = {"min": dmin, "max": dmax, **strat.defaults}
    P = rpp.SE3(init.clone())
    X = rpp.SE3(inp)
    B = P.shape[0]
    rec = {"loss": [], "damping": [], "reject": [], "step_seconds": [], "poses": []}

    def residual(P):
        return (P @ X).Log().tensor()

    loss = None
    for _ in range(steps):
        t0 = time.perf_counter()
        R = residual(P)                                                   # [B,6]
        J6 = op.se3_Jl_inv(R)                                             # d Log / d (left perturbation of P X) = d / d P
        Jb = torch.cat([J6, J6.new_zeros(B, 6, 1)], -1)                   # 7th embedding column: structurally zero
        A = Jb.mT @ Jb                                                    # [B,7,7] = the diagonal blocks of J^T J
        g = (Jb.mT @ R.unsqueeze(-1))                                     # J^T R
        A.diagonal(dim1=-2, dim2=-1).clamp_(pg["min"], pg["max"])
        last = loss = loss if loss is not None else R.square().sum()
        rejects = 0
        while last <= loss:
            d = A.diagonal(dim1=-2, dim2=-1)
            d.add_(d * pg["damping"])
            L, info = torch.linalg.cholesky_ex(A)
            if bool((info != 0).any()) or bool(torch.isnan(L).any()):
                print("Cholesky decomposition failed.\nLinear solver failed. Breaking optimization step...")
                break
            D = torch.cholesky_solve(-g, L).squeeze(-1)                   # [B,7]
            Pn = rpp.se3(D[:, :6]).Exp() @ P                              # LieTensor.add_ (lietensor.py:60-65)
            loss = residual(Pn).square().sum()
            strat.update(pg, last=last, loss=loss, J=_BlockJ(Jb), D=D.reshape(-1, 1), R=R.reshape(-1, 1))
            if last < loss and rejects < reject:
                loss, rejects = last, rejects + 1                         # (P untouched: the reference returns by Exp(-D))
            else:
                P = Pn
                break
        rec["loss"].append(float(loss))
        rec["damping"].append(float(pg["damping"]))
        rec["reject"].append(rejects)
        rec["step_seconds"].append(time.perf_counter() - t0)
        if sample is not None:
            rec["poses"].append(P.tensor()[sample].clone())
    rec["final"] = P.tensor().clone()
    return rec


# ---------------------------------------------------------------------------------------------------------------------
# metric / configs[3]: pose-graph LM (examples/module/pgo/pgo.py:15-25 through optimizer.py:630-679)
# ---------------------------------------------------------------------------------------------------------------------
def pgo_blocks(nodes, edges, poses):
    """r [E,6] and the two [E,6,6] blocks d r / d (left perturbation of nodes[i]), nodes[j] of r = Log(Z^-1 n_i^-1 n_j)."""
    rpp = _rpp()
    op = rpp.lietensor.operation
    n1, n2 = rpp.SE3(nodes[edges[:, 0]]), rpp.SE3(nodes[edges[:, 1]])
    a = rpp.SE3(poses).Inv()
    b = n1.Inv()
    c = a @ b
    r = (c @ n2).Log().tensor()
    Ji = op.se3_Jl_inv(r)
    J2 = Ji @ op.SE3_Adj(c.tensor())
    J1 = -(Ji @ op.SE3_Adj(a.tensor())) @ op.SE3_Adj(b.tensor())
    return r, J1, J2


def _csr_jacobian(J1, J2, edges, N):
    """the [6E, 7N] Jacobian as torch.sparse_csr (7th column of every node structurally empty)"""
    E = edges.shape[0]
    rows = torch.arange(6 * E).view(E, 6, 1).expand(E, 6, 6)
    cols = torch.arange(6).view(1, 1, 6)
    i0 = rows.reshape(-1)
    c1 = (edges[:, 0].view(E, 1, 1) * 7 + cols).expand(E, 6, 6).reshape(-1)
    c2 = (edges[:, 1].view(E, 1, 1) * 7 + cols).expand(E, 6, 6).reshape(-1)
    J = torch.sparse_coo_tensor(torch.stack([torch.cat([i0, i0]), torch.cat([c1, c2])]),
                                torch.cat([J1.reshape(-1), J2.reshape(-1)]), (6 * E, 7 * N)).coalesce()
    return J.to_sparse_csr()


def _with_diagonal(A_off, diag):
    n = diag.numel()
    i = torch.arange(n)
    return (A_off + torch.sparse_coo_tensor(torch.stack([i, i]), diag, (n, n))).coalesce().to_sparse_csr()


def _block_jacobi(A_coo, diag, N):
    """M = blockdiag(A_nn)^-1 over the 7x7 node blocks, as CSR (the preconditioner matrix solver.py:276-340 multiplies by)"""
    ij, v = A_coo.indices(), A_coo.values()
    same = (ij[0] // 7) == (ij[1] // 7)
    blocks = torch.zeros(N, 7, 7, dtype=v.dtype)
    blocks.index_put_((ij[0][same] // 7, ij[0][same] % 7, ij[1][same] % 7), v[same], accumulate=True)
    blocks.diagonal(dim1=-2, dim2=-1).copy_(diag.view(N, 7))
    Minv = torch.linalg.inv(blocks)
    r = (torch.arange(N).view(N, 1, 1) * 7 + torch.arange(7).view(1, 7, 1)).expand(N, 7, 7).reshape(-1)
    c = (torch.arange(N).view(N, 1, 1) * 7 + torch.arange(7).view(1, 1, 7)).expand(N, 7, 7).reshape(-1)
    return torch.sparse_coo_tensor(torch.stack([r, c]), Minv.reshape(-1), (7 * N, 7 * N)).coalesce().to_sparse_csr()


def pgo_lm(init, edges, poses, steps, radius=1e4, tol=1e-4, maxiter=250, dmin=1e-6, dmax=1e32, reject=16,
           precondition=True):
    """``steps`` LM steps of the reference's PoseGraph model from ``init`` [N,7] with TrustRegion(radius) and the
    reference's CG (tol, maxiter) on the CSR normal equations.  Returns {"loss","damping","reject","final","cg_iterations"}."""
    rpp = _rpp()
    strat = _strategy(rpp, "trustregion", radius=radius)
    pg = {"min": dmin, "max": dmax, **strat.defaults}
    solver = rpp.optim.solver.CG(maxiter=maxiter, tol=tol)
    nodes = init.clone()
    N = nodes.shape[0]
    rec = {"loss": [], "damping": [], "reject": [], "step_seconds": [], "solve_seconds": []}
    loss = None
    for _ in range(steps):
        t0 = time.perf_counter()
        r, J1, J2 = pgo_blocks(nodes, edges, poses)
        J = _csr_jacobian(J1, J2, edges, N)
        R = r.reshape(-1, 1)
        J_T = J.mT.to_sparse_csr()
        A = (J_T @ J).to_sparse_coo().coalesce()                           # optimizer.py:642
        b = -(J_T @ R)
        ij = A.indices()
        on = ij[0] == ij[1]
        diag = torch.zeros(7 * N, dtype=r.dtype)
        diag[ij[0][on]] = A.values()[on]
        diag.clamp_(pg["min"], pg["max"])                                 # diagonal_op_(A, clamp_) :643
        A_off = torch.sparse_coo_tensor(ij[:, ~on], A.values()[~on], A.shape)
        last = loss = loss if loss is not None else r.square().sum()
        rejects, t_solve = 0, 0.0
        while last <= loss:
            diag = diag * (1 + pg["damping"])                             # diagonal_op_(A, mul 1 + damping) :664
            Ad = _with_diagonal(A_off, diag)
            M = _block_jacobi(A, diag, N) if precondition else None
            ts = time.perf_counter()
            D = solver(Ad, b, M=M)
            t_solve += time.perf_counter() - ts
            Dn = D.view(N, 7)
            new = (rpp.se3(Dn[:, :6]).Exp() @ rpp.SE3(nodes)).tensor()
            loss = pgo_blocks(new, edges, poses)[0].square().sum()
            strat.update(pg, last=last, loss=loss, J=J, D=D, R=R)
            if last < loss and rejects < reject:
                loss, rejects = last, rejects + 1
            else:
                nodes = new
                break
        rec["loss"].append(float(loss))
        rec["damping"].append(float(pg["damping"]))
        rec["reject"].append(rejects)
        rec["step_seconds"].append(time.perf_counter() - t0)
        rec["solve_seconds"].append(t_solve)
    rec["final"] = nodes
    return rec


def pose_graph_problem(N, E, seed=0, dtype=torch.float64):
    """SURVEY.md section 8(d) C4 generator on the CPU with the reference's own ops: chain + uniform random closures,
    sigma 0.01 edge noise, sigma 0.05 initial error (same construction as bench.py:_pose_graph_problem)."""
    rpp = _rpp()
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    gt = rpp.cumprod(rpp.randn_SE3(N, sigma=0.3, dtype=dtype), dim=0, left=False)
    chain = torch.stack([torch.arange(N - 1), torch.arange(1, N)], -1)
    extra = torch.randint(0, N, (E - (N - 1), 2), generator=g)
    extra[:, 1] = torch.where(extra[:, 0] == extra[:, 1], (extra[:, 1] + 1) % N, extra[:, 1])
    e = torch.cat([chain, extra], 0)
    rel = gt[e[:, 0]].Inv() @ gt[e[:, 1]] @ rpp.randn_SE3(E, sigma=0.01, dtype=dtype)
    init = gt @ rpp.randn_SE3(N, sigma=0.05, dtype=dtype)
    return e, rel.tensor().contiguous(), init.tensor().contiguous()


# ---------------------------------------------------------------------------------------------------------------------
# configs[4]: the reference's IMUPreintegrator itself (module/imu_preintegrator.py:128-465) -- runs at B <= 512 on a host
# ---------------------------------------------------------------------------------------------------------------------
def imu_forward(dt, gyro, acc, prop_cov=True):
    rpp = _rpp()
    integ = rpp.module.IMUPreintegrator(prop_cov=prop_cov, reset=True)
    with torch.no_grad():
        return integ(dt=dt, gyro=gyro, acc=acc)
