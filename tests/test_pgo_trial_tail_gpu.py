"""The two entry points of a captured pose-graph LM trial, called on their own through the C ABI (include/pplie.h):

* pplie_pgo_trial_tail -- what optimizer.py:669-673 does between the linear solve and the accept test: retraction
  (lietensor.py:60-65), the loss at the candidate, the gain-ratio terms of strategy.py:144 / :261 -- against the same quantities
  from the package's own ops; the result block lands in HOST-PINNED memory, the sequence number last;
* pplie_pcg_begin -- clears a control block and brings a scalar from pinned host memory into device memory.
"""
import ctypes

import pytest
import torch

import pypose_amd as pp
from pypose_amd import _C

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")
_TAIL_SIG = [ctypes.c_void_p] * 11 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
_BEGIN_SIG = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float64, 1e-12)])
@pytest.mark.parametrize("N,E", [(40, 110), (3000, 70_001)])
def test_trial_tail_equals_the_unfused_ops(dtype, tol, N, E):
    torch.manual_seed(0)
    nodes = pp.randn_SE3(N, dtype=dtype, device=DEV).tensor().contiguous()
    Z = pp.randn_SE3(E, dtype=dtype, device=DEV).tensor().contiguous()
    idx = torch.randint(0, N, (E, 2), device=DEV)
    J = torch.randn(E, 2, 6, 6, dtype=dtype, device=DEV)
    J[:, 0] = -J[:, 1]          # the program's two blocks per edge are opposite (pplie_pgo_linearize); the tail reads J[:, 1] only
    R = torch.randn(E, 6, dtype=dtype, device=DEV)
    x = 0.01 * torch.randn(N, 6, dtype=dtype, device=DEV)
    info = torch.tensor([5.0, 1e-3, 1.0, 1.0], dtype=dtype, device=DEV)
    partial = torch.empty(3 * 1024, dtype=dtype, device=DEV)
    ring = torch.zeros(16, dtype=dtype, device=DEV)
    state = torch.tensor([6, ring.data_ptr(), 16, 6, 0, 0, 0, 0], dtype=torch.int64, device=DEV)    # (six executions so far, six retractions; ticket at rest)
    out = torch.zeros(8, dtype=torch.float64).pin_memory()
    backup = torch.empty_like(nodes)
    want_nodes = (pp.se3(x).Exp() @ pp.SE3(nodes)).tensor()
    rel = pp.SE3(Z).Inv() @ pp.SE3(want_nodes[idx[:, 0]]).Inv() @ pp.SE3(want_nodes[idx[:, 1]])
    want_loss = rel.Log().tensor().double().square().sum().item()
    JD = (J[:, 0] @ x[idx[:, 0]].unsqueeze(-1) + J[:, 1] @ x[idx[:, 1]].unsqueeze(-1)).squeeze(-1).double()
    want_a, want_b = (JD * JD).sum().item(), (JD * R.double()).sum().item()
    before = nodes.clone()
    fn = _C.library().symbol("pplie_pgo_trial_tail" + ("_f32" if dtype == torch.float32 else "_f64"), _TAIL_SIG)
    with _C._on_device(DEV):
        code = fn(nodes.data_ptr(), backup.data_ptr(), idx.data_ptr(), Z.data_ptr(), J.data_ptr(), R.data_ptr(), x.data_ptr(),
                  info.data_ptr(), partial.data_ptr(), state.data_ptr(), out.data_ptr(), N, E, _C.stream_ptr(DEV))
    assert code == 0
    torch.cuda.synchronize()
    got = out.tolist()
    assert got[7] == 7.0 and state.tolist()[0] == 7 and state.tolist()[3] == 7     # executions and retractions are counted
    assert state.tolist()[4:] == [0, 0, 0, 0]                                       # the arrival ticket is back at rest
    assert got[3:7] == [5.0, pytest.approx(1e-3, rel=1e-6), 1.0, 1.0]
    assert got[0] == pytest.approx(want_a, rel=tol) and got[1] == pytest.approx(want_b, rel=tol, abs=tol * want_a)
    assert got[2] == pytest.approx(want_loss, rel=tol)
    assert ring[7].item() == pytest.approx(got[2], rel=1e-7) and ring.count_nonzero().item() == 1
    assert torch.equal(backup, before)
    torch.testing.assert_close(nodes, want_nodes, rtol=0, atol=4 * torch.finfo(dtype).eps * 10)


def test_pcg_begin_clears_and_fetches_from_pinned_memory():
    ctl = torch.full((4096,), 0x5A, dtype=torch.uint8, device=DEV)
    src = torch.tensor([1.0 + 2.5e-5], dtype=torch.float64).pin_memory()
    dst = torch.zeros(1, dtype=torch.float64, device=DEV)
    fn = _C.library().symbol("pplie_pcg_begin", _BEGIN_SIG)
    with _C._on_device(DEV):
        assert fn(ctl.data_ptr(), ctl.numel(), src.data_ptr(), dst.data_ptr(), _C.stream_ptr(DEV)) == 0
        assert fn(ctl.data_ptr(), 12, None, None, _C.stream_ptr(DEV)) != 0            # (not a multiple of 8)
    torch.cuda.synchronize()
    assert ctl.count_nonzero().item() == 0 and dst.item() == 1.0 + 2.5e-5


def test_a_failed_tail_in_the_ordinary_loop_restores_the_parameters_and_resyncs():
    """the un-captured trial's tail overwrites the parameters in its first launch; a failure reported by the same C
    call afterwards must leave the parameters where the trial started and the host's execution count equal to the device's.
    The failure is injected AFTER a real execution's retraction (the call runs, then reports an error), and as a call that
    never ran."""
    from pypose_amd.optim import pgograph as PG
    from tests.optim_models import PoseGraph
    from tests.test_optim_gpu import _synthetic_graph
    edges, rel, init = _synthetic_graph(40_000, 120_000, torch.float32)     # beyond the persistent solve: the ordinary trial loop
    graph = PoseGraph(init.clone())
    opt = pp.optim.LM(graph, solver=pp.optim.solver.PCG(tol=1e-4, maxiter=100), strategy=pp.optim.strategy.TrustRegion(radius=1e4))
    opt.graph_step = False
    l0 = float(opt.step((edges, rel)))
    tt = opt.__dict__.get("_trial_tail")
    assert tt is not None, "the fused tail was not used on this path"
    real_enqueue = PG.TrialTail.enqueue
    before = graph.nodes.detach().clone()
    mode = {"k": "after"}

    def failing(self, pt, backup, prog, lin, Dn, info):
        if mode["k"] == "after":
            real_enqueue(self, pt, backup, prog, lin, Dn, info)
            torch.cuda.synchronize()
            self.state[0] -= 1                     # as if the last kernel (pack) had never run: retracted, not reported
        raise RuntimeError("injected")
    PG.TrialTail.enqueue = failing
    try:
        for k in ("after", "never"):
            mode["k"] = k
            seq_dev = int(tt.state[0])
            with pytest.raises(RuntimeError, match="injected"):
                opt.step((edges, rel))
            torch.cuda.synchronize()
            assert torch.equal(graph.nodes.detach(), before), k          # back at the trial's starting point
            assert tt.seq == int(tt.state[0]) == seq_dev, k              # host mirror = device count
            assert int(tt.state[3]) == int(tt.state[0]), k
    finally:
        PG.TrialTail.enqueue = real_enqueue
    l1 = float(opt.step((edges, rel)))                                    # and the loop carries on
    assert l1 < l0


def test_wait_is_bounded_by_time_and_reports_an_idle_stream():
    from pypose_amd.optim import pgograph as PG
    tt = PG.TrialTail(torch.float32, torch.device(DEV))
    tt.advance()                                                          # a result is expected, nothing was enqueued
    with pytest.raises(RuntimeError, match="without reporting"):
        tt.wait()
    assert tt.seq == 0                                                    # resynchronised with the device's count
