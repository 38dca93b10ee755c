"""LM / GN on the MI355X through the HIP kernels: reference trajectories (fp64), kernel-level
parity of the block / graph linear algebra, and the BASELINE-size runs (configs[2], configs[3])."""
import numpy as np
import pytest
import torch

import pypose_amd as pp
from tests.optim_models import InvNet, PoseGraph, T, compare_trajectory, invnet_cases, load_lm_golden, run_steps

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def G():
    return load_lm_golden()


FUSABLE = {"constant", "adaptive", "trustregion", "far"}     # Trivial kernel, no weight, no target: optim/fused.py


@pytest.mark.parametrize("mode", ["dense", "block", "fused"])
@pytest.mark.parametrize("case", ["constant", "adaptive", "trustregion", "huber_weight", "cauchy_target", "gn", "far"])
def test_invnet_trajectory_matches_reference(G, case, mode):
    from pypose_amd import _C
    assert _C._test_backend is None
    mk, init, args, kwargs, n = invnet_cases(G, DEV)[case]
    net = InvNet(init)
    opt = mk(net)
    opt.structured, opt.fused = mode != "dense", mode == "fused"
    rec = run_steps(opt, args, kwargs, n)
    want = {"dense": "dense", "block": "block", "fused": "fused:se3inv" if case in FUSABLE else "block"}[mode]
    assert set(rec["kind"]) <= {want, "?"}, rec["kind"]
    compare_trajectory(rec, G, "invnet/" + case)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 3e-5)])
@pytest.mark.parametrize("n", [1, 255, 256, 70_001])
def test_fused_se3inv_trial_kernel_vs_oracle(dtype, tol, n):
    """pplie_lm_se3inv_trial through the C ABI against oracle/optim_np.lm_se3inv_trial (fp64 anchor)."""
    from oracle import optim_np
    from pypose_amd.optim import fused
    torch.manual_seed(n)
    P = pp.Parameter(pp.randn_SE3(n, dtype=dtype, device=DEV))
    X = pp.randn_SE3(n, dtype=dtype, device=DEV)
    r = (P @ X).Log().tensor().detach()
    P0 = P.detach().tensor().clone()

    class Opt:
        group = None
    lin = fused.Se3InvLinearization(Opt(), P, X, r)
    lin.build_normal_equations(1e-6, 1e32)
    out = torch.empty((n, 7), dtype=dtype, device=DEV)
    D, sums = lin._trial(out, 1.37)
    f64 = lambda t: t.detach().double().cpu().numpy()
    Pn, d, s = optim_np.lm_se3inv_trial(f64(r), f64(P.tensor()), f64(X.tensor()), 1.37, 1e-6, 1e32)
    assert np.abs(f64(D) - d).max() <= tol * max(1.0, np.abs(d).max())
    assert np.abs(f64(out) - Pn).max() <= tol * max(1.0, np.abs(Pn).max())
    got = f64(sums)
    assert np.all(np.abs(got[1:] - s[1:]) <= 20 * tol * np.abs(s[1:]).max()), (got, s)
    # the new loss is a squared distance to the optimum: compare on the scale of the old one
    assert abs(got[0] - s[0]) <= tol * s[1]
    # in place: P_new written over P_cur gives the same result
    D2, sums2 = lin._trial(None, 1.37)
    assert torch.equal(P.detach().tensor(), out) and torch.equal(D2, D) and torch.equal(sums2, sums)
    # R = NULL: the kernel computes the residual at P_cur itself and leaves it for the retries of the step
    P3 = pp.Parameter(pp.SE3(P0))
    lin3 = fused.Se3InvLinearization(Opt(), P3, X, None)
    lin3.build_normal_equations(1e-6, 1e32)
    out3 = torch.empty((n, 7), dtype=dtype, device=DEV)
    D3, sums3 = lin3._trial(out3, 1.37)
    assert (D3 - D).abs().max().item() <= tol * max(1.0, D.abs().max().item()) and (out3 - out).abs().max().item() <= tol
    assert lin3.R is not None and (lin3.R - r).abs().max().item() <= tol


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 3e-5)])
@pytest.mark.parametrize("E", [1, 63, 64, 20_001])
def test_fused_pgo_kernels_vs_oracle(dtype, tol, E):
    """pplie_pgo_linearize / pplie_pgo_residual through the C ABI against oracle/optim_np.pgo_linearize (fp64)."""
    from oracle import optim_np
    from pypose_amd.optim import fused
    torch.manual_seed(E)
    N = max(2, E // 3)
    nodes = pp.Parameter(pp.randn_SE3(N, dtype=dtype, device=DEV))
    idx = torch.randint(0, N, (E, 2), device=DEV)
    Z = pp.randn_SE3(E, dtype=dtype, device=DEV)
    prog = fused.PgoProgram(nodes, idx[:, 0], idx[:, 1], Z.tensor())
    R, J = prog.linearize()
    f64 = lambda t: t.detach().double().cpu().numpy()
    Rw, Jw = optim_np.pgo_linearize(f64(nodes.tensor()), idx.cpu().numpy(), f64(Z.tensor()))
    assert np.abs(f64(R) - Rw).max() <= tol * max(1.0, np.abs(Rw).max())
    assert np.abs(f64(J) - Jw).max() <= tol * max(1.0, np.abs(Jw).max())
    assert abs(float(prog.loss()) - (Rw * Rw).sum()) <= 10 * tol * (Rw * Rw).sum()
    # and against the autograd route of the same model (six batched backward sweeps)
    r = (Z.Inv() @ nodes[idx[:, 0]].Inv() @ nodes[idx[:, 1]]).Log().tensor()
    assert (r.detach() - R).abs().max().item() <= 10 * tol


def test_fused_pgo_is_recognised_only_when_exact(G):
    edges, poses = T(G["pgo40/edges"], DEV), pp.SE3(T(G["pgo40/poses"], DEV))
    init = pp.SE3(T(G["pgo40/init"], DEV))

    class Reversed(PoseGraph):                       # a different program: node2^-1 * node1
        def forward(self, edges, poses):
            n1, n2 = self.nodes[edges[..., 0]], self.nodes[edges[..., 1]]
            return (poses.Inv() @ n2.Inv() @ n1).Log().tensor()

    class RightAssociated(PoseGraph):                 # the same residual, multiplied in the other order
        def forward(self, edges, poses):
            n1, n2 = self.nodes[edges[..., 0]], self.nodes[edges[..., 1]]
            return (poses.Inv() @ (n1.Inv() @ n2)).Log().tensor()

    class NoMeasurementInverse(PoseGraph):            # a different program: Z instead of Z^-1
        def forward(self, edges, poses):
            n1, n2 = self.nodes[edges[..., 0]], self.nodes[edges[..., 1]]
            return (poses @ n1.Inv() @ n2).Log().tensor()

    class Halved(PoseGraph):
        def forward(self, edges, poses):
            return 0.5 * super().forward(edges, poses)

    for cls, kw, want in ((PoseGraph, {}, "fused:pgo"), (PoseGraph, {"kernel": pp.optim.kernel.Huber()}, "fused:pgo"),
                          (Reversed, {}, "fused:pgo"), (RightAssociated, {}, "fused:pgo"), (NoMeasurementInverse, {}, "graph"),
                          (Halved, {}, "graph")):
        graph = cls(init.clone())
        opt = pp.optim.LM(graph, strategy=pp.optim.strategy.TrustRegion(radius=1e4), **kw)
        l0 = float(opt.model.loss((edges, poses), None).detach())
        assert float(opt.step((edges, poses))) < l0
        assert opt.linearization == want, (cls.__name__, kw, opt.linearization)


def test_fused_program_is_recognised_only_when_exact(G):
    """Log(P @ X) takes the hand-derived program, other product chains over one P the normal-form kernel (fused:lpr); a robust
    kernel, a non-Cholesky solver or a residual that is not such a chain stays on the generic paths."""
    torch.manual_seed(0)
    inp = pp.randn_SE3(64, device=DEV)

    class Scaled(InvNet):
        def forward(self, input):
            return 2 * super().forward(input)

    class Swapped(InvNet):
        def forward(self, input):
            return (input @ self.pose).Log().tensor()

    for cls, kw, want in ((InvNet, {}, "fused:se3inv"), (Scaled, {}, "block"), (Swapped, {}, "fused:lpr"),
                          (InvNet, {"kernel": pp.optim.kernel.Huber()}, "block"),
                          (InvNet, {"solver": pp.optim.solver.PINV()}, "block")):
        net = cls(pp.randn_SE3(64, device=DEV))
        opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(1e-6), **kw)
        l0 = float(net(inp).detach().square().sum())
        assert float(opt.step(inp)) < l0
        assert opt.linearization == want, (cls.__name__, kw, opt.linearization)
    net = InvNet(pp.randn_SE3(64, device=DEV))
    opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(1e-6))
    opt.step(inp, target=torch.zeros(64, 6, device=DEV))
    assert opt.linearization == "fused:lpr"          # (a target is part of the normal form: r = Log(P X) - b)


@pytest.mark.parametrize("mode", ["dense", "graph", "fused:pgo"])
@pytest.mark.parametrize("tag,wname", [("pgo12", "noweight"), ("pgo40", "infos")])
def test_posegraph_trajectory_matches_reference(G, tag, wname, mode):
    edges, poses = T(G[f"{tag}/edges"], DEV), pp.SE3(T(G[f"{tag}/poses"], DEV))
    graph = PoseGraph(pp.SE3(T(G[f"{tag}/init"], DEV)))
    opt = pp.optim.LM(graph, solver=pp.optim.solver.Cholesky(), strategy=pp.optim.strategy.TrustRegion(radius=1e4), min=1e-6)
    opt.structured, opt.fused = mode != "dense", mode == "fused:pgo"
    w = T(G[f"{tag}/infos"], DEV) if wname == "infos" else None
    rec = run_steps(opt, ((edges, poses),), {"weight": w}, 5)
    assert set(rec["kind"]) == {mode}, rec["kind"]
    compare_trajectory(rec, G, f"{tag}/{wname}", floor=1e-12, rtol=1e-8)
    # (the graph has no fixed node: the damped gauge directions amplify rounding differences of the linearisation)
    torch.testing.assert_close(graph.nodes.detach().tensor().cpu(), T(G[f"{tag}/{wname}/final"]), rtol=0, atol=1e-7)


@pytest.mark.parametrize("two_launch,persist", [(True, True), (True, False), (False, True)])
def test_posegraph_pcg_matrix_free_on_gpu(G, two_launch, persist, monkeypatch):
    """Reference trajectory through the device-resident PCG: the persistent one-launch solve, the two-launch iteration with
    the convergence test on the device (pplie_pcg2_*_stop: what graphs beyond 32k nodes take) and the three-launch one
    (pplie_graph_bsr_spmv + pplie_pcg_stage, also the form the edge-sharded path uses)."""
    from pypose_amd.optim import posegraph
    monkeypatch.setattr(posegraph.FusedPCG, "two_launch", two_launch, raising=False)
    monkeypatch.setattr(posegraph.FusedPCG, "persist", persist, raising=False)
    edges, poses = T(G["pgo40/edges"], DEV), pp.SE3(T(G["pgo40/poses"], DEV))
    graph = PoseGraph(pp.SE3(T(G["pgo40/init"], DEV)))
    opt = pp.optim.LM(graph, solver=pp.optim.solver.PCG(tol=1e-13, maxiter=2000, check_every=1),
                      strategy=pp.optim.strategy.TrustRegion(radius=1e4), min=1e-6)
    rec = run_steps(opt, ((edges, poses),), {"weight": T(G["pgo40/infos"], DEV)}, 5)
    assert set(rec["kind"]) == {"fused:pgo"}
    assert all(w.two_launch == two_launch for w in opt._pcg_workspaces.values())
    compare_trajectory(rec, G, "pgo40/infos", floor=1e-12, rtol=1e-7)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("weighted", [False, True])
def test_posegraph_captured_trial_equals_the_uncaptured_step(G, dtype, weighted):
    """optim/pgograph.py: after three ordinary steps the LM trial of the recognised pose-graph program is replayed as one
    hipGraph.  Same kernels on the same numbers: losses, damping, reject counts and parameters must be IDENTICAL to the
    un-captured path -- through the descent, at the rounding floor (where trials are rejected and the captured trial hands
    over to the ordinary retry loop), after the caller rewrites the parameters in place, and when the input objects change."""
    edges, poses = T(G["pgo40/edges"], DEV), pp.SE3(T(G["pgo40/poses"], DEV).to(dtype))
    kw = {"weight": T(G["pgo40/infos"], DEV).to(dtype)} if weighted else {}
    runs = {}
    for captured in (True, False):
        graph = PoseGraph(pp.SE3(T(G["pgo40/init"], DEV).to(dtype)))
        opt = pp.optim.LM(graph, solver=pp.optim.solver.PCG(tol=1e-6, maxiter=500), strategy=pp.optim.strategy.TrustRegion(radius=1e4),
                          min=1e-6)
        opt.graph_step = captured
        rec = run_steps(opt, ((edges, poses),), kw, 14)
        assert (opt.__dict__.get('_pgo_graph_step') is not None) == captured
        with torch.no_grad():                                  # the caller moves the parameters: the replay linearises where they are
            graph.nodes.copy_(pp.SE3(T(G["pgo40/init"], DEV).to(dtype)))
        del opt.loss
        rec2 = run_steps(opt, ((edges, poses),), kw, 5)
        edges2 = edges.clone()                                 # new input objects: the capture does not apply, a new one is made
        rec3 = run_steps(opt, ((edges2, poses),), kw, 5)
        runs[captured] = (rec, rec2, rec3, graph.nodes.detach().tensor().clone(), opt.solver.iterations)
    for a, b in zip(runs[True][:3], runs[False][:3]):
        assert a["loss"] == b["loss"] and a["damping"] == b["damping"] and a["reject"] == b["reject"], (a, b)
        assert set(a["kind"]) == {"fused:pgo"}
    assert sum(runs[True][0]["reject"]) > 0                   # the floor was reached: rejected trials went through the hand-over
    assert torch.equal(runs[True][3], runs[False][3]) and runs[True][4] == runs[False][4]
    if dtype == torch.float64:
        compare_trajectory({k: v[:5] for k, v in runs[True][0].items()}, G, "pgo40/infos" if weighted else "pgo40/noweight",
                           floor=1e-12, rtol=1e-5)


def test_two_launch_pcg_stops_in_the_converging_iteration(G, monkeypatch):
    """pplie_pcg2_*_stop: chunks of 8 captured iterations are queued two at a time, yet the solve ends in the iteration that
    meets the tolerance -- the same count as the persistent solve, which tests every iteration on the device too"""
    from pypose_amd.optim import posegraph
    edges, poses = T(G["pgo40/edges"], DEV), pp.SE3(T(G["pgo40/poses"], DEV))
    seen = {}
    for persist in (True, False):
        monkeypatch.setattr(posegraph.FusedPCG, "persist", persist, raising=False)
        graph = PoseGraph(pp.SE3(T(G["pgo40/init"], DEV)))
        solver = pp.optim.solver.PCG(tol=1e-6, maxiter=400, check_every=8)
        opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4), min=1e-6)
        opt.graph_step = False
        its, losses = [], []
        for _ in range(4):
            losses.append(float(opt.step((edges, poses))))
            its.append(solver.iterations)
        seen[persist] = (its, losses)
    assert seen[True][0] == seen[False][0], seen
    assert any(i % 8 for i in seen[False][0])                  # not a multiple of the chunk: the stop came from the device
    np.testing.assert_allclose(seen[True][1], seen[False][1], rtol=1e-9)


def test_batched_sweeps_by_slices_equal_the_expanded_launch(monkeypatch):
    """lietensor/operation._launch_slices: from 64k rows per cotangent slice the batched backward of the block linearisations
    launches once per slice against the saved operands in place instead of expanding them d-fold -- same blocks, bit for bit"""
    from pypose_amd.lietensor import operation as _op
    from pypose_amd.optim import blocks as _blocks
    n = 70_001
    torch.manual_seed(3)
    P = pp.Parameter(pp.randn_SE3(n, device=DEV))
    X = pp.randn_SE3(n, device=DEV)
    a = pp.randn_se3(n, device=DEV).tensor()

    def blocks():
        r1 = (P @ X).Log().tensor()
        r2 = P.Inv().Adj(pp.se3(a)).tensor() + P.Act(a[:, :3]).repeat(1, 2)
        return _blocks.jacobian_blocks([r1, r2], [P])

    calls = []
    real = _op._launch_slices
    monkeypatch.setattr(_op, "_launch_slices", lambda *a_, **k: (calls.append(1), real(*a_, **k))[1])
    J_slices = blocks()
    assert len(calls) >= 4
    monkeypatch.setattr(_op, "_SLICE_ROWS", 1 << 40)
    _blocks._COT_CACHE.clear()
    J_expand = blocks()
    assert torch.equal(J_slices, J_expand)
    assert J_slices.shape == (n, 12, 7) and float(J_slices.abs().max()) > 0


def test_posegraph_pcg_one_block_per_edge(G, monkeypatch):
    """the opt-in symmetric storage (pplie_graph_assemble_csr_sym + pplie_pcg2_spmv_sym: H_ji read as H_ij^T) walks the
    reference's trajectory too, weighted (symmetric information matrices) and unweighted"""
    from pypose_amd.optim import posegraph
    monkeypatch.setattr(posegraph.FusedPCG, "sym_blocks", True, raising=False)
    monkeypatch.setattr(posegraph.FusedPCG, "persist", False, raising=False)
    edges, poses = T(G["pgo40/edges"], DEV), pp.SE3(T(G["pgo40/poses"], DEV))
    for tag, kw in (("pgo40/infos", {"weight": T(G["pgo40/infos"], DEV)}), ("pgo40/noweight", {})):
        graph = PoseGraph(pp.SE3(T(G["pgo40/init"], DEV)))
        opt = pp.optim.LM(graph, solver=pp.optim.solver.PCG(tol=1e-13, maxiter=2000, check_every=1),
                          strategy=pp.optim.strategy.TrustRegion(radius=1e4), min=1e-6)
        rec = run_steps(opt, ((edges, poses),), kw, 5)
        assert set(rec["kind"]) == {"fused:pgo"}
        assert all(w.sym for w in opt._pcg_workspaces.values())
        compare_trajectory(rec, G, tag, floor=1e-12, rtol=1e-7)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 2e-5)])
@pytest.mark.parametrize("dr,dp,has_w", [(6, 7, False), (6, 7, True), (3, 4, False), (7, 8, True), (4, 5, False), (6, 6, True)])
def test_block_kernels_vs_oracle(dtype, tol, dr, dp, has_w):
    from oracle import optim_np
    from pypose_amd.optim import blocks
    n = 20_011
    g = torch.Generator().manual_seed(dr * 10 + dp)
    J = torch.randn(n, dr, dp, generator=g, dtype=dtype)
    R = torch.randn(n, dr, generator=g, dtype=dtype)
    W = None
    if has_w:
        W = torch.randn(n, dr, dr, generator=g, dtype=dtype) * 0.3
        W = W @ W.mT + torch.eye(dr, dtype=dtype)
    A, gr = blocks.normal_equations(J.to(DEV), R.to(DEV), W.to(DEV) if has_w else None)
    Ar, gr_ref = optim_np.block_normal_eq(J.double().numpy(), R.double().numpy(), W.double().numpy() if has_w else None)
    assert np.abs(A.cpu().double().numpy() - Ar).max() <= tol * np.abs(Ar).max()
    assert np.abs(gr.cpu().double().numpy() - gr_ref).max() <= tol * np.abs(gr_ref).max()
    # SPD system: damp the diagonal like LM does, then solve
    A.diagonal(dim1=-2, dim2=-1).add_(1.0)
    x = blocks.chol_solve(A, gr)
    xr = np.linalg.solve(A.cpu().double().numpy(), -gr.cpu().double().numpy()[..., None])[..., 0]
    assert np.abs(x.cpu().double().numpy() - xr).max() <= 50 * tol * max(1.0, np.abs(xr).max())
    # an indefinite block yields NaNs (-> the reference's "Cholesky decomposition failed")
    bad = A.clone()
    bad[0] = -bad[0]
    assert torch.isnan(blocks.chol_solve(bad, gr)[0]).any()


def _synthetic_graph(N, E, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    steps = pp.randn_SE3(N, sigma=0.3, dtype=dtype, device=DEV)
    gt = pp.cumprod(steps, dim=0, left=False)
    chain = torch.stack([torch.arange(N - 1), torch.arange(1, N)], -1)
    extra = torch.randint(0, N, (E - (N - 1), 2), generator=g)
    extra[:, 1] = torch.where(extra[:, 0] == extra[:, 1], (extra[:, 1] + 1) % N, extra[:, 1])
    edges = torch.cat([chain, extra], 0).to(DEV)
    rel = gt[edges[:, 0]].Inv() @ gt[edges[:, 1]] @ pp.randn_SE3(E, sigma=0.01, dtype=dtype, device=DEV)
    init = gt @ pp.randn_SE3(N, sigma=0.05, dtype=dtype, device=DEV)
    return edges, rel, init


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 5e-4)])
def test_graph_kernels_vs_torch_reference(dtype, tol):
    """pplie_graph_assemble / pplie_graph_spmv against plain torch index_add_ formulations."""
    from pypose_amd.optim import posegraph
    N, E = 3000, 12_001
    g = torch.Generator().manual_seed(1)
    J = torch.randn(E, 2, 6, 6, generator=g, dtype=dtype).to(DEV)
    R = torch.randn(E, 6, generator=g, dtype=dtype).to(DEV)
    W = torch.randn(E, 6, 6, generator=g, dtype=dtype)
    W = (W @ W.mT + torch.eye(6, dtype=dtype)).to(DEV)
    idx = torch.randint(0, N, (E, 2), generator=g).to(DEV)
    p = torch.randn(N, 6, generator=g, dtype=dtype).to(DEV)
    for Wm in (None, W):
        class Opt:
            pass
        lin = posegraph.GraphLinearization(Opt(), Wm, R, torch.zeros(N, 7, dtype=dtype, device=DEV), idx, J, 7, 6)
        assert lin._hip()
        B, gr = lin._assemble()                 # node-parallel (CSR) kernel + H12
        y = lin._Hp(p)
        # the edge-parallel scatter-add kernel (used when edges are sharded over ranks), through the C ABI
        from pypose_amd import _C
        Ba, ga = torch.zeros_like(B), torch.zeros_like(gr)
        fn = _C.library().symbol("pplie_graph_assemble" + ("_f32" if dtype == torch.float32 else "_f64"), posegraph._ASM_SIG)
        _C.check(fn(J.data_ptr(), Wm.data_ptr() if Wm is not None else None, R.data_ptr(), idx.data_ptr(), Ba.data_ptr(),
                    ga.data_ptr(), None, E, 6, 6, 2, _C.stream_ptr(J.device)), "pplie_graph_assemble")
        assert (Ba - B).abs().max().item() <= tol * B.abs().max().item()
        assert (ga - gr).abs().max().item() <= tol * gr.abs().max().item()
        HB, (ptr, blk, other) = lin.HB, lin.csr()
        lin._hip = lambda: False               # torch formulation on the same device
        B2, gr2 = lin._assemble()
        y2 = lin._Hp(p)
        # off-diagonal blocks in incidence order: HB[c] = J[e, side]^T W J[e, 1 - side] with blk[c] = 2 e + side
        e, side = (blk // 2).long(), (blk % 2).long()
        Jn, Jf = J[e, side], J[e, 1 - side]
        want = (Jn.mT if Wm is None else Jn.mT @ Wm[e]) @ Jf
        assert torch.equal(other.long(), idx[e, 1 - side])
        for a, b in ((B, B2), (gr, gr2), (y, y2), (HB, want)):
            assert (a - b).abs().max().item() <= tol * b.abs().max().item()


def test_c3_invnet_one_million_problems():
    """BASELINE configs[2]: LM on InvNet SE3, B = 1M independent problems, fp32."""
    torch.manual_seed(0)
    B = 1_000_000
    net = InvNet(pp.randn_SE3(B, device=DEV))
    torch.manual_seed(1)
    inp = pp.randn_SE3(B, device=DEV)
    opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(damping=1e-4))
    l0 = float(net.forward(inp).square().sum())
    losses = [float(opt.step(inp)) for _ in range(3)]
    assert opt.linearization == "fused:se3inv"
    assert losses[0] < 1e-3 * l0 and losses[-1] < 1e-6 * l0, (l0, losses)
    # the optimum is pose = input^-1: pose * input == identity
    I = (pp.SE3(net.pose.detach().tensor()) * inp).Log().tensor()
    assert I.abs().max().item() < 1e-3


def test_c4_pose_graph_10k_and_100k():
    """BASELINE metric + configs[3] sizes on one GPU: the unmodified PoseGraph model takes the graph path,
    the loss decreases monotonically and the solution approaches the generating trajectory."""
    for N, E in ((10_000, 40_000), (100_000, 400_000)):
        edges, rel, init = _synthetic_graph(N, E, torch.float32)
        graph = PoseGraph(init)
        opt = pp.optim.LM(graph, solver=pp.optim.solver.PCG(tol=1e-4, maxiter=250),
                          strategy=pp.optim.strategy.TrustRegion(radius=1e4))
        l_init = float(graph(edges, rel).square().sum())
        losses = [float(opt.step((edges, rel))) for _ in range(4)]
        assert opt.linearization == "fused:pgo"
        assert all(b <= a * (1 + 1e-6) for a, b in zip([l_init] + losses, losses)), (l_init, losses)
        assert losses[-1] < 0.2 * l_init, (N, l_init, losses)


@pytest.mark.parametrize("group,prior", [("SO3", False), ("Sim3", False), ("SE3", True), ("SO3", True)])
def test_graph_kernels_other_groups_match_dense_path(group, prior):
    """The (3,3,2) / (7,7,2) pose-graph and the (6,6,1) / (3,3,1) prior instantiations of the graph kernels: LM on the
    HIP graph path == LM on the dense reference algorithm (fp64), for SO3 / Sim3 graphs and unary priors."""
    torch.manual_seed(5)
    rand = getattr(pp, "randn_" + group)
    N, E = 30, 80
    gt = rand(N, sigma=0.4, device=DEV, dtype=torch.float64)
    e = torch.stack([torch.randint(0, N, (E,)), torch.randint(0, N, (E,))], -1)
    e[:, 1] = torch.where(e[:, 0] == e[:, 1], (e[:, 1] + 1) % N, e[:, 1])
    e = e.to(DEV)
    init = gt @ rand(N, sigma=0.05, device=DEV, dtype=torch.float64)

    class Graph(torch.nn.Module):
        def __init__(self, nodes):
            super().__init__()
            self.nodes = pp.Parameter(nodes)

        def forward(self, edges, rel):
            n1, n2 = self.nodes[edges[..., 0]], self.nodes[edges[..., 1]]
            return (rel.Inv() @ n1.Inv() @ n2).Log().tensor()

    class Prior(torch.nn.Module):                      # unary factors: one gather per residual row
        def __init__(self, nodes):
            super().__init__()
            self.nodes = pp.Parameter(nodes)

        def forward(self, idx, meas):
            return (meas.Inv() @ self.nodes[idx]).Log().tensor()

    if prior:
        idx = torch.cat([torch.arange(N), torch.randint(0, N, (E - N,))]).to(DEV)
        args = (idx, gt[idx] @ rand(E, sigma=0.01, device=DEV, dtype=torch.float64))
        make = lambda: Prior(init.clone())
    else:
        args = (e, gt[e[:, 0]].Inv() @ gt[e[:, 1]] @ rand(E, sigma=0.01, device=DEV, dtype=torch.float64))
        make = lambda: Graph(init.clone())
    res = {}
    for mode in ("dense", "graph"):
        model = make()
        opt = pp.optim.LM(model, solver=pp.optim.solver.Cholesky(), strategy=pp.optim.strategy.TrustRegion(radius=1e4))
        opt.structured, opt.fused = mode != "dense", False
        losses = [float(opt.step(args)) for _ in range(4)]
        assert opt.linearization == mode
        res[mode] = (losses, model.nodes.detach().tensor().clone())
    for a, b in zip(res["dense"][0], res["graph"][0]):
        assert abs(a - b) <= 1e-8 * max(abs(a), 1e-30) + 1e-20, res
    assert (res["dense"][1] - res["graph"][1]).abs().max().item() < 1e-7
    # and the matrix-free PCG on the same kernels (bsr spmv for m in {3, 6, 7})
    model = make()
    opt = pp.optim.LM(model, solver=pp.optim.solver.PCG(tol=1e-13, maxiter=3000, check_every=1), strategy=pp.optim.strategy.TrustRegion(radius=1e4))
    opt.fused = False
    losses = [float(opt.step(args)) for _ in range(4)]
    for a, b in zip(res["dense"][0], losses):
        assert abs(a - b) <= 1e-6 * max(abs(a), 1e-30) + 1e-18, (res["dense"][0], losses)


@pytest.mark.parametrize("problem,fused", [("invnet", True), ("invnet", False), ("pgo", True), ("pgo", False)])
def test_lm_step_fp32_within_1e5_of_fp64(G, problem, fused):
    """BASELINE north_star: 'LM-step numerics within 1e-5 of reference'.  One LM step in fp32 against the same step
    in fp64 (whose trajectories are pinned to the reference's by the tests above), on every structured path."""
    out = {}
    for dtype in (torch.float64, torch.float32):
        if problem == "invnet":
            torch.manual_seed(3)
            init = pp.randn_SE3(512, dtype=torch.float64).to(dtype).to(DEV)
            inp = pp.randn_SE3(512, dtype=torch.float64).to(dtype).to(DEV)
            model, args, kw = InvNet(init), (inp,), {"strategy": pp.optim.strategy.Constant(damping=1e-4)}
        else:
            edges = T(G["pgo40/edges"], DEV)
            poses = pp.SE3(T(G["pgo40/poses"], DEV).to(dtype))
            model = PoseGraph(pp.SE3(T(G["pgo40/init"], DEV).to(dtype)))
            args, kw = ((edges, poses),), {"solver": pp.optim.solver.Cholesky(), "strategy": pp.optim.strategy.TrustRegion(radius=1e4)}
        opt = pp.optim.LM(model, **kw)
        opt.fused = fused
        loss = float(opt.step(*args))
        P = next(model.parameters()).detach().double().clone()
        if problem == "pgo":
            # the graph has no fixed node: a global rigid motion is determined by the damping alone (1e-4 of the
            # diagonal) and amplifies fp32 rounding by 1/damping -- compare the gauge-invariant poses node_0^-1 node_k
            P = (pp.SE3(P[:1]).Inv() @ pp.SE3(P)).tensor()
        out[dtype] = (loss, P, opt.linearization)
    assert out[torch.float32][2] == out[torch.float64][2]
    p64, p32 = out[torch.float64][1], out[torch.float32][1]
    assert (p64 - p32).abs().max().item() <= 1e-5 * max(1.0, p64.abs().max().item()), (p64 - p32).abs().max()


@pytest.mark.parametrize("problem", ["invnet", "pgo"])
def test_static_option_gives_identical_steps(G, problem):
    """LM(static=True) evaluates the verified program directly instead of re-tracing the model every step: same
    iterates as the default; a different `input` object falls back to tracing."""
    runs = {}
    for static in (False, True):
        torch.manual_seed(9)
        if problem == "invnet":
            model = InvNet(pp.randn_SE3(300, device=DEV, dtype=torch.float64))
            args = (pp.randn_SE3(300, device=DEV, dtype=torch.float64),)
            opt = pp.optim.LM(model, strategy=pp.optim.strategy.Adaptive(damping=1e-6), static=static)
            call = lambda: opt.step(args[0])
        else:
            edges, poses = T(G["pgo40/edges"], DEV), pp.SE3(T(G["pgo40/poses"], DEV))
            model = PoseGraph(pp.SE3(T(G["pgo40/init"], DEV)))
            inp = (edges, poses)
            opt = pp.optim.LM(model, solver=pp.optim.solver.PCG(tol=1e-12, maxiter=2000), strategy=pp.optim.strategy.TrustRegion(radius=1e4),
                              static=static)
            call = lambda: opt.step(inp)
        losses = [float(call()) for _ in range(4)]
        if static:
            assert opt.__dict__["_structure_cache"].get("program") is not None
            # a new input object (equal values) is not trusted: traced again, same result path
            if problem == "invnet":
                losses.append(float(opt.step(args[0].clone())))
            else:
                losses.append(float(opt.step((edges.clone(), poses))))
        else:
            losses.append(float(call()))
        runs[static] = (losses, next(model.parameters()).detach().clone(), opt.linearization)
    assert runs[True][2] == runs[False][2] and runs[True][2].startswith("fused:")
    for a, b in zip(runs[False][0], runs[True][0]):
        assert abs(a - b) <= 1e-9 * max(abs(a), 1e-30) + 1e-24, (runs[False][0], runs[True][0])
    torch.testing.assert_close(runs[True][1], runs[False][1], rtol=0, atol=1e-11)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 3e-5)])
def test_pcg_prepare_gain_terms_segment_sum_vs_tensor_formulation(dtype, tol):
    """pplie_pcg_prepare, pplie_graph_gain_terms and pplie_segment_sum (one- and two-level) through the C ABI against
    the tensor formulations they replace."""
    import ctypes
    from pypose_amd import _C
    from pypose_amd.optim import posegraph, multigraph
    sfx = "_f32" if dtype == torch.float32 else "_f64"
    torch.manual_seed(2)
    N, m, E = 2001, 6, 7003
    A = torch.randn(N, m, m, dtype=dtype, device=DEV)
    B = A @ A.mT + 0.5 * torch.eye(m, dtype=dtype, device=DEV)
    B[::7, 2, :] = 0                                               # a structurally (almost) zero Jacobian column:
    B[::7, :, 2] = 0                                               # its diagonal entry gets clamped up to dmin
    B[::7, 2, 2] = 1e-9
    g = torch.randn(N, m, dtype=dtype, device=DEV)
    s, dmin, dmax = 1.37, 1e-6, 1e32
    z = lambda *sh: torch.empty(sh, dtype=dtype, device=DEV)
    D, Binv, shift, x, r, zz, p = z(N, m, m), z(N, m, m), z(N, m), z(N, m), z(N, m), z(N, m), z(N, m)
    scal = torch.zeros(2 * 8 * 32 * 32, dtype=dtype, device=DEV)
    fn = _C.library().symbol("pplie_pcg_prepare" + sfx, posegraph._PREP_SIG)
    _C.check(fn(B.data_ptr(), g.data_ptr(), D.data_ptr(), Binv.data_ptr(), shift.data_ptr(), x.data_ptr(), r.data_ptr(), zz.data_ptr(),
                p.data_ptr(), scal.data_ptr(), s, dmin, dmax, N, m, _C.stream_ptr(B.device)), "pplie_pcg_prepare")
    diag = B.diagonal(dim1=-2, dim2=-1)
    Dw = B.clone()
    Dw.diagonal(dim1=-2, dim2=-1).copy_(s * diag.clamp(dmin, dmax))
    close = lambda a, b, k=1.0: (a - b).abs().max().item() <= k * tol * max(1.0, b.abs().max().item())
    assert close(D, Dw) and close(shift, s * diag.clamp(dmin, dmax) - diag) and close(Binv @ Dw, torch.eye(m, dtype=dtype, device=DEV).expand(N, m, m), 100)
    zw = (torch.linalg.inv(Dw) @ (-g).unsqueeze(-1)).squeeze(-1)
    assert close(r, -g) and close(zz, zw, 100) and close(p, zw, 100) and float(x.abs().max()) == 0.0
    sv = scal.view(2, 8, 32, 32)                                   # (4-quantity layout: set 0 occupies the same offsets)
    rho, bn2 = scal[0:1024:32].sum(), scal[3 * 1024:4 * 1024:32].sum()
    assert abs(float(rho) - float((-g * zw).sum())) <= 100 * tol * float((g * zw).abs().sum())
    assert abs(float(bn2) - float((g * g).sum())) <= 10 * tol * float((g * g).sum())
    # gain terms
    J = torch.randn(E, 2, 6, 6, dtype=dtype, device=DEV)
    idx = torch.randint(0, N, (E, 2), device=DEV)
    R = torch.randn(E, 6, dtype=dtype, device=DEV)

    class Opt:
        pass
    lin = posegraph.GraphLinearization(Opt(), None, R, torch.zeros(N, 7, dtype=dtype, device=DEV), idx, J, 7, 6)
    Dstep = torch.randn(N * 7, 1, dtype=dtype, device=DEV)
    op = posegraph.GraphOperator(lin)
    ab = op.gain_terms(Dstep)
    JD = op @ Dstep
    want = torch.stack([(JD * JD).sum(), (JD * R.reshape(-1, 1)).sum()])
    assert (ab - want).abs().max().item() <= 10 * tol * want.abs().max().item()
    # segmented sums: short lists (one level) and a few very long ones (two levels)
    for idxv in (torch.randint(0, 500, (E,), device=DEV), torch.cat([torch.randint(0, 3, (E - 200,)), torch.randint(3, 500, (200,))]).to(DEV)):
        sc = multigraph._Scatter(idxv, 500)
        vals = torch.randn(E, 3, 3, dtype=dtype, device=DEV)
        want = torch.zeros(500, 3, 3, dtype=dtype, device=DEV).index_add_(0, idxv, vals)
        assert sc.hip and (sc(vals) - want).abs().max().item() <= 50 * tol * want.abs().max().item()
    assert sc.two_level


def test_edge_list_written_in_place_between_steps_is_seen(G):
    """The stacked edge list and its incidence lists are cached across steps (keyed on the index tensors' storage and
    version counter): rewiring ``edges`` IN PLACE must invalidate both.  (Default mode; ``LM(static=True)`` is the
    caller's promise that operands do NOT change between steps.)"""
    static = False
    edges, poses = T(G["pgo40/edges"], DEV).clone(), pp.SE3(T(G["pgo40/poses"], DEV))
    mk = lambda g: pp.optim.LM(g, solver=pp.optim.solver.PCG(tol=1e-12, maxiter=2000),
                               strategy=pp.optim.strategy.TrustRegion(radius=1e4), static=static)
    graph = PoseGraph(pp.SE3(T(G["pgo40/init"], DEV)))
    opt = mk(graph)
    for _ in range(2):
        opt.step((edges, poses))
    assert opt.linearization == "fused:pgo"
    edges[3, 1] = (edges[3, 1] + 7) % 40                   # same tensor object, new graph
    edges[10, 0] = (edges[10, 0] + 11) % 40
    before = graph.nodes.detach().tensor().clone()
    del opt.loss                                           # (like the reference, LM carries the previous step's loss over)
    loss = float(opt.step((edges, poses)))
    true = float(graph(edges, poses).detach().square().sum())
    assert abs(loss - true) <= 1e-9 * true                 # the fused loss kernel reads the rewired edge list
    fresh = PoseGraph(pp.SE3(before.clone()))
    ref = mk(fresh)
    for _ in range(8):                                     # both reach the optimum of the NEW graph
        opt.step((edges, poses))
        ref.step((edges.clone(), poses))
    assert abs(float(opt.loss) - float(ref.loss)) <= 1e-6 * float(ref.loss)
    np.testing.assert_allclose(float(ref.loss), float(fresh(edges, poses).detach().square().sum()), rtol=1e-9)


def test_negative_edge_indices_on_the_fused_pose_graph_path(G):
    """`nodes[edges[..., 0]]` with -1 for the last node: the fused program addresses node rows directly in its kernels,
    so the indices are normalised once per edge list; same steps as with the equivalent non-negative indices."""
    edges, poses = T(G["pgo40/edges"], DEV), pp.SE3(T(G["pgo40/poses"], DEV))
    out = []
    for negative in (False, True):
        e = edges.clone()
        if negative:
            e = torch.where(e == 39, torch.full_like(e, -1), e)
            assert (e < 0).any()
        graph = PoseGraph(pp.SE3(T(G["pgo40/init"], DEV)))
        opt = pp.optim.LM(graph, solver=pp.optim.solver.Cholesky(), strategy=pp.optim.strategy.TrustRegion(radius=1e4))
        losses = [float(opt.step((e, poses))) for _ in range(3)]
        assert opt.linearization == "fused:pgo"
        out.append((losses, graph.nodes.detach().tensor().clone()))
    assert out[0][0] == pytest.approx(out[1][0], rel=1e-12)
    torch.testing.assert_close(out[0][1], out[1][1], rtol=0, atol=1e-12)


@pytest.mark.parametrize("dtype,tol,atol", [(torch.float32, 1e-5, 1e-4), (torch.float64, 1e-11, 1e-9)])
def test_ghost_zone_solve_equals_the_two_dependency_kernel(dtype, tol, atol, monkeypatch):
    """pplie_pcg_ghost (one grid-wide dependency per iteration: ghosts advanced locally) against pplie_pcg_persist: same
    iteration count, same solution, on a graph large enough for every workgroup to have ghosts in several layers"""
    from pypose_amd.optim import fused as F, posegraph
    # (fp64 blocks are twice the size: fewer nodes per workgroup so that the slice + ghost state still fit in LDS)
    edges, rel, init = _synthetic_graph(*((9000, 36000) if dtype == torch.float32 else (5000, 20000)), dtype)
    graph = PoseGraph(init.clone())
    solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250, gauge=False)      # (the two-dependency kernel has block-Jacobi only)
    opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4))
    opt.step((edges, rel))
    prog = opt._structure_cache["program"][3]
    with torch.no_grad():
        lin = F._pgo_linearization(opt, prog, None, graph.nodes, True)
        lin.build_normal_equations(1e-6, 1e32)
        lin.damp(1e-4)
        wsp = next(iter(opt._pcg_workspaces.values()))
        assert wsp.want_gauge is False
        res = {}
        for ghost in (True, False):
            monkeypatch.setattr(posegraph.FusedPCG, "ghost", ghost, raising=False)
            res[ghost] = wsp.solve(lin, lin.s, lin.dmin, lin.dmax, tol, 3000, None)
    assert not wsp.__dict__.get("_no_ghost", False), "the ghost-zone kernel did not take this graph"
    assert abs(res[True][1] - res[False][1]) <= 2, (res[True][1], res[False][1])
    scale = float(res[False][0].abs().max())
    assert float((res[True][0] - res[False][0]).abs().max()) <= atol * max(1.0, scale)
