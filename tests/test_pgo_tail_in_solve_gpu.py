"""pplie_pcg_ghost_tail + pplie_pgo_trial_tail_after_solve (round 6): the LM trial's gain terms and retraction in the persistent
solve's epilogue, against the two-launch tail they replace (pplie_pgo_trial_tail: retract, then J d per edge).  The gain terms are the
same numbers by another route -- d.(H d) = d.(r_0 - r - shift d) from the solve's own residual instead of sum |J d|^2, d.J^T R = -d.r_0
-- so they agree to the solve's rounding, not bit for bit; the retraction is the same arithmetic on the same step.
Reference semantics: pypose/optim/optimizer.py:662-678 (the trial), pypose/optim/strategy.py:128-140 (the gain ratio)."""
import pytest
import torch

import pypose_amd as pp
from pypose_amd.optim import posegraph as G, strategy as S
from tests.optim_models import PoseGraph
from tests.test_optim_gpu import _synthetic_graph

pytestmark = pytest.mark.gpu


def _run(N, E, dtype, captured, in_solve, monkeypatch, steps=7):
    monkeypatch.setattr(G.FusedPCG, "tail_in_solve", in_solve)
    edges, rel, init = _synthetic_graph(N, E, dtype)
    graph = PoseGraph(init.clone())
    opt = pp.optim.LM(graph, solver=pp.optim.solver.PCG(tol=1e-4 if dtype == torch.float32 else 1e-8, maxiter=250),
                      strategy=pp.optim.strategy.TrustRegion(radius=1e4))
    opt.graph_step = captured
    terms, real = [], S.update_from_terms

    def spy(strategy, pg, last, loss, a, b):
        terms.append((a, b, loss))
        return real(strategy, pg, last, loss, a, b)
    import contextlib, io
    buf = io.StringIO()
    with monkeypatch.context() as mp, contextlib.redirect_stdout(buf):
        mp.setattr(S, "update_from_terms", spy)
        losses = [float(opt.step((edges, rel))) for _ in range(steps)]
    tts = [t for t in (opt.__dict__.get('_trial_tail'), getattr(opt.__dict__.get('_pgo_graph_step'), 'tt', None)) if t is not None]
    return dict(losses=losses, terms=terms, nodes=graph.nodes.detach().tensor().clone(), its=opt.solver.iterations,
                after=sum(t.after_solve for t in tts), damping=opt.param_groups[0]['damping'], failed=buf.getvalue().count("solver failed"))


@pytest.mark.parametrize("captured", [False, True], ids=["stepwise", "captured"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-3), (torch.float64, 1e-7)], ids=["fp32", "fp64"])
def test_epilogue_tail_equals_two_launch_tail(dtype, tol, captured, monkeypatch):
    N, E = 3000, 12_001
    a = _run(N, E, dtype, captured, True, monkeypatch)
    b = _run(N, E, dtype, captured, False, monkeypatch)
    # (a captured trial is enqueued once and replayed: the counter counts enqueues)
    assert a["after"] >= (1 if captured else len(a["terms"])) and b["after"] == 0, (a["after"], b["after"], len(a["terms"]))
    print("trials", len(a["terms"]), len(b["terms"]), "after", a["after"], "failed solves", a["failed"], b["failed"])
    assert a["failed"] == b["failed"]
    assert len(a["terms"]) == len(b["terms"])
    # first trial: identical state in both runs -- the gain terms by the two routes, the candidate loss of the same retraction
    for k, (ta, tb) in enumerate(zip(a["terms"], b["terms"])):
        for x, y, name in zip(ta, tb, "abL"):
            assert abs(x - y) <= (tol if k == 0 else 20 * tol) * max(abs(y), 1e-30), (k, name, x, y)
    print("max relative difference of (a, b, loss):", [max(abs(ta[i] - tb[i]) / max(abs(tb[i]), 1e-30) for ta, tb in zip(a["terms"], b["terms"]))
                                                      for i in range(3)])
    # what the strategy divides by, -(a + 2 b) (strategy.py:128-140), is at least d.r_0 in size while a's rounding is eps x |d.r_0|
    # whatever the damping: the denominator agrees to a few ulp even where a itself (tiny at the floor) shows 1e-3
    # (the first trial starts from the same state in both runs; later ones from states that agree to the losses' tolerance only)
    for k, (ta, tb) in enumerate(zip(a["terms"], b["terms"])):
        da, db = ta[0] + 2 * ta[1], tb[0] + 2 * tb[1]
        assert abs(da - db) <= (2e-5 if dtype == torch.float32 else 1e-9) * (1 if k == 0 else 1e3) * abs(db), (k, da, db)
    assert a["its"] == b["its"] or captured
    ltol = 1e-4 if dtype == torch.float32 else 1e-9
    for x, y in zip(a["losses"], b["losses"]):
        assert abs(x - y) <= ltol * abs(y), (a["losses"], b["losses"])
    assert a["damping"] == b["damping"]
    err = (a["nodes"] - b["nodes"]).abs().max()
    assert float(err) <= (2e-4 if dtype == torch.float32 else 1e-9), float(err)


def test_epilogue_tail_leaves_a_failed_solves_parameters_alone(monkeypatch):
    """a solve that ends with flag >= 2 (breakdown / NaN) returns the zero step: the epilogue must not move the parameters"""
    N, E = 3000, 12_001
    edges, rel, init = _synthetic_graph(N, E, torch.float32)
    graph = PoseGraph(init.clone())
    opt = pp.optim.LM(graph, solver=pp.optim.solver.PCG(tol=1e-4, maxiter=250), strategy=pp.optim.strategy.TrustRegion(radius=1e4))
    opt.step((edges, rel))
    before = graph.nodes.detach().tensor().clone()
    bad = rel.tensor().clone()
    bad[5] = float("nan")
    try:
        opt.step((edges, pp.SE3(bad)))
    except Exception:
        pass
    after = graph.nodes.detach().tensor()
    assert torch.isfinite(after).all() and torch.equal(after, before)


def test_speculative_replay_is_undone_when_the_steps_checks_fail():
    """fused.checked_shortcut replays the captured trial BEFORE the step's checks (PgoGraphStep.quick: storage address and re-probe rhythm
    only).  Whatever the checks then find -- other solver settings, a weight, the parameter's storage swapped -- the step must be the
    ordinary path's step from the parameters as the caller left them: same losses and poses as an optimizer that never captured."""
    N, E = 3000, 12_001
    dtype = torch.float64
    edges, rel, init = _synthetic_graph(N, E, dtype)
    runs = {}
    for captured in (True, False):
        graph = PoseGraph(init.clone())
        opt = pp.optim.LM(graph, solver=pp.optim.solver.PCG(tol=1e-8, maxiter=250), strategy=pp.optim.strategy.TrustRegion(radius=1e4))
        opt.graph_step = captured
        losses = [float(opt.step((edges, rel))) for _ in range(5)]
        assert (opt.__dict__.get('_pgo_graph_step') is not None) == captured
        opt.solver.tol = 1e-3                                    # (the capture holds the old tolerance)
        with torch.no_grad():
            graph.nodes.data.copy_(init.tensor())
        del opt.loss
        losses += [float(opt.step((edges, rel))) for _ in range(2)]
        w = torch.eye(6, dtype=dtype, device=init.device).mul(2.0).expand(E, 6, 6).contiguous()
        losses += [float(opt.step((edges, rel), weight=w)) for _ in range(2)]
        with torch.no_grad():                                    # the parameter's storage is swapped: nothing may be replayed into the old one
            graph.nodes.data = init.tensor().clone()
        del opt.loss
        losses += [float(opt.step((edges, rel))) for _ in range(4)]
        runs[captured] = (losses, graph.nodes.detach().tensor().clone(), opt.solver.iterations)
    a, b = runs[True], runs[False]
    for x, y in zip(a[0], b[0]):
        assert abs(x - y) <= 1e-9 * abs(y), (a[0], b[0])
    assert float((a[1] - b[1]).abs().max()) <= 1e-9
