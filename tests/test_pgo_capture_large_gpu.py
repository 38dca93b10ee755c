"""The whole LM trial of a pose graph BEYOND the persistent solve as one hipGraph replay (optim/pgograph.py + the unwatched
two-launch solve of optim/posegraph.py, csrc/graph.hip pplie_pcg2_report).

Reference semantics: the trial of optimizer.py:662-678 around the CG of solver.py:276-340 (which tests every iteration).  Checked on
a graph small enough for seconds (the two-launch iteration forced as beyond 32 k nodes):
  * the captured steps are taken (the trial's kernels come from a replay) and losses / damping / rejects / poses equal the run with
    PPLIE_CAPTURE_LARGE off -- same kernels on the same data, so the comparison is exact up to what the watched chunks' early exit
    changes (nothing: launches behind the converging iteration return at once in both);
  * a capture that queues FEWER iterations than a solve needs reports flag 4, the parameters go back, the step is taken on the watched
    path and the trajectory is still the uncaptured one;
  * pplie_pcg2_report alone: the last queued iteration's stop test and the four-value record.
"""
import ctypes

import pytest
import torch

import pypose_amd as pp
from pypose_amd import _C
from pypose_amd.optim import posegraph, pgograph
from tests.optim_models import PoseGraph, run_steps

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")


def _graph(N, E, dtype, seed=3):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    truth = pp.cumprod(pp.randn_SE3(N, sigma=0.3, dtype=dtype, device=DEV), dim=0)
    i = torch.cat([torch.arange(N - 1), torch.randint(0, N, (E - N + 1,), generator=g)])
    j = torch.cat([torch.arange(1, N), torch.randint(0, N, (E - N + 1,), generator=g)])
    keep = i != j
    i, j = i[keep].to(DEV), j[keep].to(DEV)
    rel = truth[i].Inv() @ truth[j] @ pp.randn_SE3(len(i), sigma=0.01, dtype=dtype, device=DEV)
    init = truth @ pp.randn_SE3(N, sigma=0.05, dtype=dtype, device=DEV)
    return torch.stack([i, j], 1), rel.tensor(), init.tensor()


def _run(edges, rel, init, capture, monkeypatch, steps, tol, spoil=None):
    monkeypatch.setattr(posegraph, "PERSIST_NODES", 64, raising=False)          # the two-launch iteration, as beyond 32 k nodes
    monkeypatch.setattr(posegraph.FusedPCG, "capture_large", capture, raising=False)
    graph = PoseGraph(pp.SE3(init.clone()))
    opt = pp.optim.LM(graph, solver=pp.optim.solver.PCG(tol=tol, maxiter=400), strategy=pp.optim.strategy.TrustRegion(radius=1e4))
    rec = {"loss": [], "damping": [], "reject": [], "captured": [], "its": []}
    for k in range(steps):
        if spoil is not None:
            spoil(opt, k)
        had = opt.__dict__.get('_pgo_graph_step') is not None
        loss = opt.step((edges, pp.SE3(rel)))
        rec["loss"].append(float(loss))
        rec["damping"].append(float(opt.param_groups[0].get("damping", 0.0)))
        rec["reject"].append(int(getattr(opt, "reject_count", 0)))
        # (a step counts as captured when the capture existed before it and survived it)
        rec["captured"].append(had and opt.__dict__.get('_pgo_graph_step') is not None)
        rec["its"].append(int(opt.solver.iterations))
    return rec, graph.nodes.detach().tensor().clone(), opt


def _same_trajectory(a, b, dtype, pa, pb):
    # The two-launch iteration adds its partial sums with atomics (32 slots per quantity): the last bits of a solve differ from run
    # to run, so two runs of the SAME path agree like this too -- losses to rounding, the decisions exactly, iteration counts to a
    # few while the step still moves the loss (near the noise floor the relative stop test flips on rounding)
    rtol = 2e-5 if dtype == torch.float32 else 1e-9
    torch.testing.assert_close(torch.tensor(a["loss"]), torch.tensor(b["loss"]), rtol=rtol, atol=0)
    assert a["damping"] == b["damping"] and a["reject"] == b["reject"]
    assert all(abs(x - y) <= 2 for x, y in zip(a["its"], b["its"])), (a["its"], b["its"])
    err = (pp.SE3(pa).Inv() @ pp.SE3(pb)).Log().tensor().abs().max().item()
    assert err <= (2e-5 if dtype == torch.float32 else 1e-9), err


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.float64, 1e-8)])
def test_captured_trial_beyond_the_persistent_solve_equals_the_watched_run(dtype, tol, monkeypatch):
    edges, rel, init = _graph(1500, 6000, dtype)
    a, pa, opt = _run(edges, rel, init, True, monkeypatch, 6, tol)
    b, pb, _ = _run(edges, rel, init, False, monkeypatch, 6, tol)
    assert a["captured"] == [False, False, False, True, True, True] and not any(b["captured"]), (a["captured"], b["captured"])
    assert {w.sym for w in opt._pcg_workspaces.values()} == {"pack"}
    _same_trajectory(a, b, dtype, pa, pb)


def test_a_capture_with_too_few_iterations_hands_the_step_to_the_watched_path(monkeypatch):
    dtype, tol = torch.float64, 1e-8
    edges, rel, init = _graph(1500, 6000, dtype)

    # the capture is sized from the watched solves of the streak; here it is made to queue 16 iterations where the solves at this
    # tolerance need more (set between the optimizer's sizing and the capture itself)
    init0 = pgograph.PgoGraphStep.__init__

    def short_init(self, opt, *a, **kw):
        for w in (opt.__dict__.get('_pcg_workspaces') or {}).values():
            w.unwatched_iterations = 16
        init0(self, opt, *a, **kw)
    monkeypatch.setattr(pgograph.PgoGraphStep, "__init__", short_init)
    seen = []
    orig = pgograph.PgoGraphStep.finish

    def finish(self, pg):
        out = orig(self, pg)
        seen.append(float(self.tt.out_np[6]))
        return out
    monkeypatch.setattr(pgograph.PgoGraphStep, "finish", finish)
    a, pa, opt = _run(edges, rel, init, True, monkeypatch, 6, tol)
    monkeypatch.setattr(pgograph.PgoGraphStep, "finish", orig)
    monkeypatch.setattr(pgograph.PgoGraphStep, "__init__", init0)
    b, pb, _ = _run(edges, rel, init, False, monkeypatch, 6, tol)
    assert min(b["its"][3:]) > 16, b["its"]                   # (else the instance does not exercise the case)
    assert seen and seen[0] == 4.0, seen                      # the short capture reported "unfinished" ...
    assert a["captured"][3] is False                          # ... and was dropped by the step that found out
    _same_trajectory(a, b, dtype, pa, pb)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_report_kernel_runs_the_last_stop_test(dtype):
    sfx = "_f32" if dtype == torch.float32 else "_f64"
    sig = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
    fn = _C.library().symbol("pplie_pcg2_report" + sfx, sig)
    st = _C.stream_ptr(DEV)

    def call(done, flag, rr_slots, bn2_slots, hist, tol2):
        scal = torch.zeros(2, 8, 32, 32, dtype=dtype, device=DEV)
        scal[(done & 1) ^ 1, 2, :, 0] = torch.tensor(rr_slots, dtype=dtype, device=DEV)      # Q2_RR of the set the last step kernel wrote
        scal[0, 3, :, 0] = torch.tensor(bn2_slots, dtype=dtype, device=DEV)                  # Q2_BN2
        rr_hist = torch.tensor(hist, dtype=dtype, device=DEV)
        it = torch.tensor([done, done, flag, 0], dtype=torch.int32, device=DEV)
        info = torch.full((4,), -1.0, dtype=dtype, device=DEV)
        with _C._on_device(DEV):
            _C.check(fn(scal.data_ptr(), rr_hist.data_ptr(), it.data_ptr(), rr_hist.numel(), float(tol2), info.data_ptr(), st), "report")
        torch.cuda.synchronize()
        return info.tolist(), it.tolist(), rr_hist.tolist()

    one = [1.0 / 32] * 32
    # still running, the last iteration converged: flag 0, it[2] raised, |r|^2 recorded
    info, it, hist = call(5, 0, [x * 1e-10 for x in one], one, [9.0] * 8, 1e-8)
    assert info[0] == 5 and info[3] == 0 and it[2] == 1 and abs(info[1] - 1e-10) < 1e-15 and abs(info[2] - 1.0) < 1e-6
    assert abs(hist[4] - 1e-10) < 1e-15 and hist[3] == 9.0
    # still running, not converged: 4
    info, it, _ = call(6, 0, [x * 1e-3 for x in one], one, [9.0] * 8, 1e-8)
    assert info[0] == 6 and info[3] == 4 and it[2] == 0
    # NaN residual: 2
    info, it, _ = call(6, 0, [float("nan")] + [0.0] * 31, one, [9.0] * 8, 1e-8)
    assert info[3] == 2 and it[2] == 2
    # already stopped by an spmv launch (slots cleared since): |r|^2 from the history
    info, it, _ = call(4, 1, [0.0] * 32, one, [9.0, 8.0, 7.0, 6.5, 5.0], 1e-8)
    assert info[0] == 4 and info[3] == 0 and info[1] == 6.5 and it[2] == 1
