"""The device-resident LM step (csrc/lm_step.hip, optim/fused.py DeviceLM): kernel arithmetic against the oracle,
the in-kernel decision against the reference's strategy / accept-reject logic, retries through the grid-barrier
kernel, lazy host mirrors."""
import ctypes

import numpy as np
import pytest
import torch

import pypose_amd as pp
from pypose_amd import _C
from pypose_amd.optim import fused
from tests.optim_models import InvNet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
f64 = lambda t: t.detach().double().cpu().numpy()


def _cfg(strategy=0, damping=1e-4, reject=16, flags=3, **kw):
    c = fused._LmCfg()
    c.high, c.low, c.up, c.factor, c.smin, c.smax, c.sdown = 0.5, 1e-3, 2.0, 0.5, 1e-6, 1e16, 0.5
    c.dmin, c.dmax, c.host_damping, c.host_down = 1e-6, 1e32, damping, 0.5
    c.strategy, c.reject, c.flags = strategy, reject, flags
    for k, v in kw.items():
        setattr(c, k, v)
    return c


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 3e-5)])
@pytest.mark.parametrize("n", [1, 255, 256, 257, 70_001, 1_200_000])
def test_step_kernel_vs_oracle(dtype, tol, n):
    """pplie_lm_se3inv_step (first trial, in-kernel decision) through the C ABI against oracle/optim_np.lm_se3inv_trial."""
    from oracle import lie_np, optim_np
    torch.manual_seed(n)
    P = pp.randn_SE3(n, dtype=dtype, device=DEV).tensor().contiguous()
    X = pp.randn_SE3(n, dtype=dtype, device=DEV).tensor().contiguous()
    P0 = P.clone()
    sfx = "_f32" if dtype == torch.float32 else "_f64"
    fn = _C.library().symbol("pplie_lm_se3inv_step" + sfx, fused._STEP_SIG)
    save, part = torch.empty_like(P), torch.empty((4096, 4), dtype=dtype, device=DEV)
    state = torch.zeros((2, 16), dtype=torch.float64, device=DEV)
    sync = torch.zeros(8, dtype=torch.int32, device=DEV)
    out = torch.zeros(2, dtype=dtype, device=DEV)
    cfg = _cfg(damping=0.37)
    item = P.element_size()
    code = fn(P.data_ptr(), X.data_ptr(), save.data_ptr(), part.data_ptr(), state[0].data_ptr(), state[1].data_ptr(), sync.data_ptr(),
              cfg, n, out.data_ptr(), out.data_ptr() + item, _C.stream_ptr(torch.device(DEV)))
    assert code == 0
    torch.cuda.synchronize()
    m = min(n, 20_000)                        # the oracle is numpy: bounded sample of rows, all sums from the state
    r = lie_np.se3_log_fwd(lie_np.se3_mul_fwd(f64(P0[:m]), f64(X[:m]))[0])[0]
    Pn, d, s = optim_np.lm_se3inv_trial(r, f64(P0[:m]), f64(X[:m]), 1.37, 1e-6, 1e32)
    assert np.abs(f64(P[:m]) - Pn).max() <= tol * max(1.0, np.abs(Pn).max())
    assert torch.equal(save, P0)
    st = state[1].tolist()
    assert st[7] == 1.0 and st[8] == 0.0 and st[6] == 0.0 and st[3] == pytest.approx(1.37)
    if m == n:
        assert abs(st[4] - s[1]) <= 20 * tol * s[1]                 # last = |r|^2 (no previous loss)
        assert abs(st[5] - s[0]) <= tol * s[1]
        assert abs(float(out[0]) - s[0]) <= tol * s[1] and abs(float(out[1]) - s[1]) <= 20 * tol * s[1]
    assert sync.tolist() == [0] * 8                                 # the arrival counter and the barrier are back at rest


def _reference_decision(strategy, pg, last, new, jj, jr, rejects, reject):
    """optimizer.py:673-678 + strategy.update on the equivalent 1x1 problem (x^2 = jj, x r = jr)."""
    x = max(jj, 1e-300) ** 0.5
    one = torch.ones((1, 1), dtype=torch.float64)
    strategy.update(pg, last=last, loss=new, J=one, D=x * one, R=(jr / x) * one)
    if last < new and rejects < reject:
        return last, rejects + 1, 0
    return new, rejects, 1


@pytest.mark.parametrize("kind", ["constant", "adaptive", "trustregion"])
def test_device_decision_equals_reference_strategies(kind):
    """pplie_lm_decide against the reference's strategy.update + accept/reject on random trial outcomes, chained so
    that the state (damping, radius, shrinking `down`, reject count) carries over as in a real retry loop."""
    S = pp.optim.strategy
    strat = {"constant": S.Constant(damping=1e-4), "adaptive": S.Adaptive(damping=1e-3), "trustregion": S.TrustRegion(radius=1e3)}[kind]
    pg = dict(strat.defaults)
    fn = _C.library().symbol("pplie_lm_decide_f64", fused._DECIDE_SIG)
    state = torch.zeros(16, dtype=torch.float64, device=DEV)
    out = torch.zeros(2, dtype=torch.float64, device=DEV)
    rng = np.random.default_rng(5)
    stream = _C.stream_ptr(torch.device(DEV))
    last, rejects, first = 10.0, 0, 1
    state[5] = last
    for it in range(200):
        jj = float(rng.uniform(0.1, 4.0))
        jr = float(-rng.uniform(0.1, 4.0))
        new = float(last * rng.choice([0.2, 0.9, 0.9999, 1.0, 1.3, 5.0]))
        sums = torch.tensor([new, 123.0, jj, jr], dtype=torch.float64, device=DEV)
        cfg = _cfg(strategy={"constant": 0, "adaptive": 1, "trustregion": 2}[kind], reject=3, flags=1 if it == 0 else 0,
                   damping=pg['damping'], host_down=pg.get('down', 0.5), high=pg.get('high', .5), low=pg.get('low', 1e-3),
                   up=pg.get('up', 2.0), factor=pg.get('factor', .5), smin=getattr(strat, 'min', 0.0), smax=getattr(strat, 'max', 0.0),
                   sdown=getattr(strat, 'down', 0.5))
        scale_want = (1 + pg['damping']) if first else scale_want * (1 + pg['damping'])
        assert fn(state.data_ptr(), state.data_ptr(), cfg, first, sums.data_ptr(), out.data_ptr(), out.data_ptr() + 8, stream) == 0
        loss, rejects, done = _reference_decision(strat, pg, last, new, jj, jr, rejects, 3)
        st = state.tolist()
        assert st[0] == pytest.approx(pg['damping'], rel=1e-15), (it, st, pg)
        if kind == "trustregion":
            assert st[1] == pytest.approx(pg['radius'], rel=1e-15) and st[2] == pytest.approx(pg['down'], rel=1e-15)
        assert st[3] == pytest.approx(scale_want, rel=1e-14)
        assert (st[5], int(st[6]), int(st[7])) == (loss, rejects, done), (it, st, loss, rejects, done)
        assert out.tolist() == [loss, last]
        if done:                       # the next trial opens a new step
            last, rejects, first = loss, 0, 1
        else:
            first = 0


def _far_problem(B, dtype, seed=0):
    torch.manual_seed(seed)
    inp = pp.randn_SE3(B, dtype=dtype, device=DEV)
    init = pp.SE3(inp.Inv().tensor()) @ pp.randn_SE3(B, sigma=2.5, dtype=dtype, device=DEV)
    return init, inp


@pytest.mark.parametrize("B", [300, 200_000])
@pytest.mark.parametrize("strategy", ["trustregion", "adaptive", "constant"])
def test_device_steps_equal_host_driven_block_path(B, strategy):
    """Whole trajectories (loss, damping, reject count, last per step) of the device-resident step against the
    host-driven block path of the same optimizer (whose trajectories are pinned to the reference's goldens).  Heavy
    initial damping keeps the descent going for many steps, so that every decision is taken well above rounding."""
    S = pp.optim.strategy
    mk = {"trustregion": lambda: S.TrustRegion(radius=0.5), "adaptive": lambda: S.Adaptive(damping=2.0),
          "constant": lambda: S.Constant(damping=1.0)}[strategy]
    rec = {}
    for fused_on in (False, True):
        init, inp = _far_problem(B, torch.float64)
        net = InvNet(init.clone())
        opt = pp.optim.LM(net, strategy=mk())
        opt.fused = fused_on
        rows = []
        for _ in range(8):
            loss = opt.step(inp)
            rows.append((float(loss), opt.param_groups[0]['damping'], opt.reject_count, float(opt.last)))
        assert opt.linearization == ("fused:se3inv" if fused_on else "block")
        rec[fused_on] = (rows, net.pose.detach().tensor().clone())
    assert rec[True][0][-1][0] > 1e-20 * rec[True][0][0][0], "still descending at the last compared step"
    for a, b in zip(rec[False][0], rec[True][0]):
        assert a[2] == b[2] and a[1] == pytest.approx(b[1], rel=1e-12), (rec[False][0], rec[True][0])
        assert a[0] == pytest.approx(b[0], rel=1e-9) and a[3] == pytest.approx(b[3], rel=1e-9)
    assert (rec[False][1] - rec[True][1]).abs().max().item() < 1e-9


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-4)])
@pytest.mark.parametrize("n", [100, 150_000])
@pytest.mark.parametrize("kind", ["constant", "adaptive", "trustregion"])
def test_rejected_trials_through_the_finish_kernel(kind, n, dtype, tol):
    """A previous loss far below anything reachable makes every trial a rejection until the reject budget is used up
    (optimizer.py:674-678): the finish kernel must run `reject` retries behind its grid barrier with the damping
    compounding (optimizer.py:666) and the strategy updating between them, and end in the accepted last trial.
    Expected: the same loop on the host with the oracle's trial and the reference's strategy classes."""
    from oracle import lie_np, optim_np
    S = pp.optim.strategy
    strat = {"constant": S.Constant(damping=0.5), "adaptive": S.Adaptive(damping=0.5), "trustregion": S.TrustRegion(radius=2.0)}[kind]
    pg = dict(strat.defaults)
    reject = 3
    torch.manual_seed(n)
    P = pp.randn_SE3(n, dtype=dtype, device=DEV).tensor().contiguous()
    X = pp.randn_SE3(n, dtype=dtype, device=DEV).tensor().contiguous()
    P0 = P.clone()
    sfx = "_f32" if dtype == torch.float32 else "_f64"
    fn = _C.library().symbol("pplie_lm_se3inv_step" + sfx, fused._STEP_SIG)
    save, part = torch.empty_like(P), torch.empty((4096, 4), dtype=dtype, device=DEV)
    state = torch.zeros((2, 16), dtype=torch.float64, device=DEV)
    state[0, 5] = 1e-30                                  # "previous loss"
    sync = torch.zeros(8, dtype=torch.int32, device=DEV)
    out = torch.zeros(2, dtype=dtype, device=DEV)
    cfg = _cfg(strategy={"constant": 0, "adaptive": 1, "trustregion": 2}[kind], reject=reject, flags=1, damping=pg['damping'],
               host_down=pg.get('down', 0.5), smin=getattr(strat, 'min', 0.0), smax=getattr(strat, 'max', 0.0),
               sdown=getattr(strat, 'down', 0.5))
    code = fn(P.data_ptr(), X.data_ptr(), save.data_ptr(), part.data_ptr(), state[0].data_ptr(), state[1].data_ptr(), sync.data_ptr(),
              cfg, n, out.data_ptr(), out.data_ptr() + P.element_size(), _C.stream_ptr(torch.device(DEV)))
    assert code == 0
    torch.cuda.synchronize()
    st = state[1].tolist()
    assert (st[6], st[7], st[8], st[9]) == (float(reject), 1.0, 0.0, float(reject + 1)), st
    assert sync.tolist()[0] == 0 and torch.equal(save, P0)
    # the same loop on the host (sums from a bounded sample would not be the device's sums: small n only)
    m = min(n, 4000)
    r = lie_np.se3_log_fwd(lie_np.se3_mul_fwd(f64(P0[:m]), f64(X[:m]))[0])[0]
    scale, rejects, last = 1.0, 0, 1e-30
    while True:
        scale *= 1.0 + pg['damping']
        Pn, _, s = optim_np.lm_se3inv_trial(r, f64(P0[:m]), f64(X[:m]), scale, 1e-6, 1e32)
        if m < n:     # the decisions do not depend on the sums' values here (every trial is far above `last`): use the device's
            s = None
        new, jj, jr = (s[0], s[2], s[3]) if s is not None else (1.0, 1.0, -1.0)
        loss, rejects, done = _reference_decision(strat, pg, last, new, jj, jr, rejects, reject)
        if done:
            break
    assert st[3] == pytest.approx(scale, rel=1e-12) and st[0] == pytest.approx(pg['damping'], rel=1e-12)
    assert np.abs(f64(P[:m]) - Pn).max() <= tol * max(1.0, np.abs(Pn).max())
    if m == n:
        assert st[5] == pytest.approx(s[0], rel=100 * tol) and float(out[0]) == pytest.approx(s[0], rel=100 * tol)


def test_lazy_mirrors_and_user_overrides():
    """No synchronisation inside step(); damping / reject_count / last read back on first use; values written into the
    param group or `opt.loss` by the user become the next step's state."""
    init, inp = _far_problem(5000, torch.float64, seed=3)
    net = InvNet(init.clone())
    opt = pp.optim.LM(net, strategy=pp.optim.strategy.TrustRegion(radius=1e4))
    l1 = opt.step(inp)
    assert opt.linearization == "fused:se3inv" and l1.is_cuda and l1.dim() == 0
    dev = opt._device_lm
    l2 = opt.step(inp)
    assert dev.pending and l2 is opt.loss and l1 is not l2
    pg = opt.param_groups[0]
    d2 = pg['damping']                                   # read-back
    assert not dev.pending and pg['radius'] == pytest.approx(1.0 / d2)
    assert float(opt.last) == pytest.approx(float(l1), rel=1e-12)
    assert "damping" in repr(opt) and opt.state_dict()['param_groups'][0]['damping'] == d2
    # the user resets the strategy state and forgets the loss (what bench.py does between repetitions)
    pg.update(opt.strategy.defaults)
    del opt.loss
    net.pose.data.copy_(init.tensor())
    l3 = opt.step(inp)
    assert float(l3) == pytest.approx(float(l1), rel=1e-12) and pg['damping'] == pytest.approx(5e-5)
    # an input edited in place is read by the very next step (the kernel reads the operand's storage, not a copy)
    inp2 = inp.tensor().clone()
    inp.tensor().copy_(pp.randn_SE3(5000, dtype=torch.float64, device=DEV).tensor())
    net.pose.data.copy_(init.tensor())
    del opt.loss
    l4 = float(opt.step(inp))
    net2 = InvNet(init.clone())
    opt2 = pp.optim.LM(net2, strategy=pp.optim.strategy.TrustRegion(radius=1e4))
    opt2.param_groups[0]['damping'] = 5e-5
    assert l4 == pytest.approx(float(opt2.step(pp.SE3(inp.tensor().clone()))), rel=1e-10) and opt._device_lm is dev
    inp.tensor().copy_(inp2)
    # the parameter's version counter moves with every in-place step (autograd would notice a stale saved tensor)
    v = net.pose._version
    opt.step(inp)
    assert net.pose._version > v


def test_custom_strategy_keeps_the_block_path():
    class Mine(pp.optim.strategy.Adaptive):
        def update(self, pg, last, loss, J, D, R, *a, **k):
            assert (J @ D).shape == R.shape          # the real operands, not a 1x1 stand-in
            return super().update(pg, last, loss, J, D, R)
    init, inp = _far_problem(64, torch.float64)
    opt = pp.optim.LM(InvNet(init), strategy=Mine(damping=1e-6))
    opt.step(inp)
    assert opt.linearization == "block"


@pytest.mark.parametrize("kind", ["constant", "adaptive", "trustregion"])
def test_failed_factorisation_leaves_the_parameter_untouched(capsys, kind):
    """A non-positive pivot (here: a NaN input row) makes the reference's solver raise before any update
    (solver.py:214, optimizer.py:667-671): the step is abandoned, P is unchanged, the loss stays -- and so do damping /
    radius / down: strategy.update is never reached (:672)."""
    init, inp = _far_problem(1000, torch.float64)
    net = InvNet(init)
    S = pp.optim.strategy
    st = {"constant": lambda: S.Constant(damping=1e-4), "adaptive": lambda: S.Adaptive(damping=1e-4), "trustregion": lambda: S.TrustRegion(radius=1e4)}[kind]()
    opt = pp.optim.LM(net, strategy=st)
    l0 = float(opt.step(inp))
    P1 = net.pose.detach().tensor().clone()
    before = {k: opt.param_groups[0].get(k) for k in ("damping", "radius", "down")}
    bad = inp.tensor().clone()
    bad[17] = float('nan')
    # same storage, new values: keeps the program, the version counter re-triggers the trace
    inp.tensor().copy_(bad)
    opt.step(inp)
    assert opt.reject_count == 0
    out = capsys.readouterr().out
    assert "Linear solver failed" in out
    got = net.pose.detach().tensor()
    assert torch.equal(got, P1)
    after = {k: opt.param_groups[0].get(k) for k in ("damping", "radius", "down")}
    for k in before:
        if before[k] is not None and not (kind == "trustregion" and k == "radius"):      # (radius is re-derived as 1 / damping)
            assert after[k] == pytest.approx(before[k], rel=1e-12), (k, before, after)


# ---------------------------------------------------------------------------------------------------------------------
# default semantics: the model's forward runs every step (reference optimizer.py:631, 646); static=True is the opt-out
# ---------------------------------------------------------------------------------------------------------------------
class _Switching(InvNet):
    flip = False
    other = None
    calls = 0

    def forward(self, input):
        self.calls += 1
        if self.flip:
            return (self.pose.Inv() @ input).Log().tensor()
        return (self.pose @ (self.other if self.other is not None else input)).Log().tensor()


def _fresh_step_loss(model_cls, pose, inp, **attrs):
    net = model_cls(pp.SE3(pose.clone()))
    for k, v in attrs.items():
        setattr(net, k, v)
    opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(damping=1e-4))
    return float(opt.step(inp)), opt.linearization


def test_default_step_follows_a_rebound_buffer_and_a_flipped_attribute_at_once():
    torch.manual_seed(5)
    n = 4096
    net = _Switching(pp.randn_SE3(n, sigma=0.3, device=DEV))
    inp, inp2 = pp.randn_SE3(n, sigma=0.3, device=DEV), pp.randn_SE3(n, sigma=0.3, device=DEV)
    opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(damping=1e-4))
    for _ in range(4):
        opt.step(inp)
    assert opt.linearization == "fused:se3inv" and opt.__dict__.get('_device_lm') is not None
    calls = net.calls
    opt.step(inp)
    assert net.calls > calls, "the model's Python must run on every step"
    # ... and yet the step stays the two-launch device step: the general path (linearise, verify, host decisions) is not entered
    general, orig = [], opt._step_general
    opt._step_general = lambda *a, **k: (general.append(1), orig(*a, **k))[1]
    for _ in range(5):
        opt.step(inp)
    opt._step_general = orig
    assert not general and opt._device_lm.pending
    # a buffer is REBOUND (no tensor written, same `input` object): the very next step optimises against it
    pose = net.pose.detach().tensor().clone()
    net.other = inp2
    del opt.loss
    got = float(opt.step(inp))
    want, kind = _fresh_step_loss(_Switching, pose, inp, other=inp2)
    assert kind == "fused:se3inv" and abs(got - want) <= 1e-5 * max(want, 1e-12), (got, want)
    # an attribute flips the residual program itself: the next step runs the other program (Log(P^-1 X): the normal-form kernel)
    pose = net.pose.detach().tensor().clone()
    net.flip = True
    del opt.loss
    got = float(opt.step(inp))
    want, kind = _fresh_step_loss(_Switching, pose, inp, flip=True)
    assert opt.linearization == kind == "fused:lpr" and abs(got - want) <= 1e-5 * max(want, 1e-12), (got, want)


def test_static_true_is_the_opt_out():
    torch.manual_seed(6)
    n = 2048
    net = _Switching(pp.randn_SE3(n, sigma=0.3, device=DEV))
    inp = pp.randn_SE3(n, sigma=0.3, device=DEV)
    opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(damping=1e-4), static=True)
    for _ in range(4):
        opt.step(inp)
    calls = net.calls
    for _ in range(8):
        opt.step(inp)
    assert net.calls == calls and opt.linearization == "fused:se3inv"      # the promise: the model is not run


def test_default_pose_graph_step_follows_a_model_change_behind_the_captured_graph():
    from tests.optim_models import PoseGraph

    class Swappable(PoseGraph):
        swapped = False

        def forward(self, edges, poses):
            a, b = (1, 0) if self.swapped else (0, 1)
            return (poses.Inv() @ self.nodes[edges[..., a]].Inv() @ self.nodes[edges[..., b]]).Log().tensor()

    torch.manual_seed(7)
    N, E = 300, 900
    gt = pp.cumprod(pp.randn_SE3(N, sigma=0.3, device=DEV), dim=0, left=False)
    e = torch.cat([torch.stack([torch.arange(N - 1), torch.arange(1, N)], -1), torch.randint(0, N, (E - N + 1, 2))]).to(DEV)
    e[:, 1] = torch.where(e[:, 0] == e[:, 1], (e[:, 1] + 1) % N, e[:, 1])
    rel = gt[e[:, 0]].Inv() @ gt[e[:, 1]] @ pp.randn_SE3(E, sigma=0.01, device=DEV)
    init = gt @ pp.randn_SE3(N, sigma=0.05, device=DEV)

    def make(nodes, **attrs):
        g = Swappable(pp.SE3(nodes.clone()))
        for k, v in attrs.items():
            setattr(g, k, v)
        return g, pp.optim.LM(g, solver=pp.optim.solver.PCG(tol=1e-6, maxiter=500), strategy=pp.optim.strategy.TrustRegion(radius=1e4))
    graph, opt = make(init.tensor())
    for _ in range(6):
        opt.step((e, rel))
    assert opt.linearization == "fused:pgo" and opt.__dict__.get('_pgo_graph_step') is not None
    nodes = graph.nodes.detach().tensor().clone()
    damping = opt.param_groups[0]['damping']
    graph.swapped = True                         # residual becomes Log(Z^-1 n_j^-1 n_i): a different problem
    del opt.loss
    got = float(opt.step((e, rel)))
    g2, o2 = make(nodes, swapped=True)
    o2.param_groups[0]['damping'] = damping
    want = float(o2.step((e, rel)))
    assert abs(got - want) <= 1e-4 * want, (got, want)


def test_pose_graph_model_that_reads_parameter_values_is_never_run_speculatively():
    """the captured pose-graph step is enqueued before the model's dry run has finished (fused.checked_shortcut); a model whose
    forward looks at the parameter's VALUES must see them as the reference's would (before the step): the tracer notices the
    access, the launch is undone exactly and speculation is switched off -- same iterates as the plain model"""
    from tests.optim_models import PoseGraph

    class Logging(PoseGraph):
        seen = None

        def forward(self, edges, poses):
            with torch.no_grad():
                type(self).seen = float(self.nodes.tensor().abs().sum())         # a host read of the parameter, every forward
            return super().forward(edges, poses)

    torch.manual_seed(11)
    N, E = 300, 900
    gt = pp.cumprod(pp.randn_SE3(N, sigma=0.3, device=DEV), dim=0, left=False)
    e = torch.cat([torch.stack([torch.arange(N - 1), torch.arange(1, N)], -1), torch.randint(0, N, (E - N + 1, 2))]).to(DEV)
    e[:, 1] = torch.where(e[:, 0] == e[:, 1], (e[:, 1] + 1) % N, e[:, 1])
    rel = gt[e[:, 0]].Inv() @ gt[e[:, 1]] @ pp.randn_SE3(E, sigma=0.01, device=DEV)
    init = gt @ pp.randn_SE3(N, sigma=0.05, device=DEV)
    runs = {}
    for cls in (PoseGraph, Logging):
        g = cls(pp.SE3(init.tensor().clone()))
        opt = pp.optim.LM(g, solver=pp.optim.solver.PCG(tol=1e-6, maxiter=500), strategy=pp.optim.strategy.TrustRegion(radius=1e4))
        losses, seen = [], []
        for _ in range(8):
            before = float(g.nodes.detach().tensor().abs().sum())
            losses.append(float(opt.step((e, rel))))
            seen.append((Logging.seen, before))
        runs[cls] = (losses, seen, opt)
    assert runs[Logging][2].__dict__.get('speculate') is False and runs[PoseGraph][2].__dict__.get('speculate', True) is True
    np.testing.assert_allclose(runs[Logging][0], runs[PoseGraph][0], rtol=1e-4)
    # every forward of the step saw the parameters a reference run would have shown it: the LAST forward of a step is the loss
    # evaluation at the accepted point or the dry run at its start -- never a half-updated state (a finite, plausible sum)
    assert all(np.isfinite(s) and abs(s - b) <= 0.2 * b for s, b in runs[Logging][1])
