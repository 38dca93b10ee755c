"""The default LM.step runs the model's Python EVERY step (reference optimizer.py:631, 646) as a dry trace that launches
nothing (optim/fused.py DryTracer): these tests drive the matcher on CPU tensors -- nothing is launched, so no GPU and no
stand-in backend is needed -- and pin that a Python-side change of the model is seen by the very next trace."""
import pytest
import torch

import pypose_amd as pp
from pypose_amd import _C
from pypose_amd.optim import fused
from tests.optim_models import InvNet, PoseGraph


class _Opt:                                  # what dry_program needs of an optimizer: the RobustModel-wrapped model
    def __init__(self, model):
        from pypose_amd.optim.optimizer import RobustModel
        self.model = RobustModel(model)


def _se3(n, seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(n, 4, generator=g)
    return pp.SE3(torch.cat([torch.randn(n, 3, generator=g), q / q.norm(dim=-1, keepdim=True)], -1))


def test_invnet_is_matched_without_a_launch():
    net = InvNet(_se3(5, 0))
    inp = _se3(5, 1)
    m = fused.dry_program(_Opt(net), [net.pose], inp, None)
    assert m[0] == "se3inv" and m[1] is net.pose and m[2].data_ptr() == inp.data_ptr()
    assert not _C.dry_tracing()


def test_pose_graph_is_matched_without_a_launch():
    graph = PoseGraph(_se3(6, 0))
    edges = torch.tensor([[0, 1], [1, 2], [2, 3], [5, 0]])
    poses = _se3(4, 2)
    m = fused.dry_program(_Opt(graph), [graph.nodes], (edges, poses), None)
    assert m[0] == "pgo" and m[1] is graph.nodes and m[4].data_ptr() == poses.data_ptr()
    assert torch.equal(m[2], edges[:, 0]) and torch.equal(m[3], edges[:, 1])


def test_python_side_changes_show_in_the_next_trace():
    class Switching(InvNet):
        flip = False
        other = None

        def forward(self, input):
            if self.flip:
                return (self.pose.Inv() @ input).Log().tensor()
            return (self.pose @ (self.other if self.other is not None else input)).Log().tensor()
    net = Switching(_se3(5, 0))
    inp, inp2 = _se3(5, 1), _se3(5, 3)
    opt = _Opt(net)
    assert fused.dry_program(opt, [net.pose], inp, None)[2].data_ptr() == inp.data_ptr()
    net.other = inp2                                                  # a rebound buffer: no tensor was written to
    assert fused.dry_program(opt, [net.pose], inp, None)[2].data_ptr() == inp2.data_ptr()
    net.flip = True                                                   # a different program altogether: Log(P^-1 X)
    m = fused.dry_program(opt, [net.pose], inp, None)
    assert m[0] == "lpr" and m[1].sign == -1 and m[1].kind == 0 and not m[1].left and len(m[1].right) == 1


def test_value_dependent_models_escape_the_dry_run():
    class Peeking(InvNet):
        def forward(self, input):
            r = (self.pose @ input).Log().tensor()
            return r * 0 if float(r.abs().max()) > 100 else r         # needs a VALUE: a meta tensor refuses
    net = Peeking(_se3(5, 0))
    assert fused.dry_program(_Opt(net), [net.pose], _se3(5, 1), None) is False
    assert not _C.dry_tracing()

    class Scaled(InvNet):
        def forward(self, input):
            return 2.0 * (self.pose @ input).Log().tensor()           # not the recognised chain
    net = Scaled(_se3(5, 0))
    assert not fused.dry_program(_Opt(net), [net.pose], _se3(5, 1), None)


def test_product_chains_reduce_to_the_normal_form():
    """r = Log(A^-1 * (P^-1 * B) * C^-1): constants to the left and right of the one occurrence of the parameter"""
    class Chain(InvNet):
        def __init__(self, init, A, B, C):
            super().__init__(init)
            self.A, self.B, self.C = A, B, C

        def forward(self, input):
            return (self.A.Inv() @ (self.pose.Inv() @ self.B) @ self.C.Inv()).Log().tensor()
    A, B, C = _se3(5, 2), _se3(5, 3), _se3(1, 4)
    net = Chain(_se3(5, 0), A, B, C)
    m = fused.dry_program(_Opt(net), [net.pose], None, None)
    assert m[0] == "lpr"
    pr = m[1]
    assert pr.sign == -1 and [(c.data_ptr(), inv) for c, inv in pr.left] == [(A.data_ptr(), True)]
    assert [(c.data_ptr(), inv) for c, inv in pr.right] == [(B.data_ptr(), False), (C.data_ptr(), True)]

    class ActNet(InvNet):
        def forward(self, pts):
            return self.pose.Inv().Act(pts)
    net = ActNet(_se3(5, 0))
    pts, tgt = torch.randn(5, 3), torch.randn(5, 3)
    m = fused.dry_program(_Opt(net), [net.pose], pts, tgt)
    assert m[0] == "lpr" and m[1].kind == 1 and m[1].sign == -1 and m[1].a.data_ptr() == pts.data_ptr() and m[1].b.data_ptr() == tgt.data_ptr()

    class TwoP(InvNet):                                                # the parameter twice: not the normal form
        def forward(self, input):
            return (self.pose @ input @ self.pose).Log().tensor()
    net = TwoP(_se3(5, 0))
    assert fused.dry_program(_Opt(net), [net.pose], _se3(5, 1), None) is None


def test_other_launches_are_refused_during_a_dry_trace():
    with fused.DryTracer():
        with pytest.raises(_C.DryTraceEscape):
            _C.stream_ptr(torch.device("cpu"))
    assert not _C.dry_tracing()


def test_value_reads_of_real_lietensors_are_noticed():
    """what lets a pose-graph step be launched before its dry run finished (fused.checked_shortcut): the tracer knows whether
    the model could have looked at parameter VALUES"""
    graph = PoseGraph(_se3(6, 0))
    edges, poses = torch.tensor([[0, 1], [1, 2], [2, 3], [5, 0]]), _se3(4, 2)
    o = _Opt(graph)
    assert fused.dry_program(o, [graph.nodes], (edges, poses), None)[0] == "pgo"
    assert o._dry_state.touched is False

    class Logging(PoseGraph):
        def forward(self, edges, poses):
            self.norm = self.nodes.tensor().norm()                     # reads the parameter's values
            return super().forward(edges, poses)
    g2 = Logging(_se3(6, 0))
    o2 = _Opt(g2)
    assert fused.dry_program(o2, [g2.nodes], (edges, poses), None)[0] == "pgo"
    assert o2._dry_state.touched is True

    class Scaling(PoseGraph):
        def forward(self, edges, poses):
            self.c = self.nodes.abs().max()                            # a torch function on the parameter itself
            return super().forward(edges, poses)
    g3 = Scaling(_se3(6, 0))
    o3 = _Opt(g3)
    fused.dry_program(o3, [g3.nodes], (edges, poses), None)
    assert o3._dry_state.touched is True


def test_a_view_of_the_result_is_not_the_result():
    """a model that returns a SLICE of the Log result (same storage offset) is not the full program, and a model
    that changes the view it returns between steps must not keep the previous step's match"""
    class Sliced(InvNet):
        cut = False

        def forward(self, input):
            r = (self.pose.Inv() @ input).Log().tensor()
            return r[..., :3] if self.cut else r
    net = Sliced(_se3(5, 0))
    inp = _se3(5, 1)
    opt = _Opt(net)
    m = fused.dry_program(opt, [net.pose], inp, None)
    assert m is not None and m[0] == "lpr"
    net.cut = True                                                    # same kernels, same operands, same offset: another view
    m2 = fused.dry_program(opt, [net.pose], inp, None)
    assert m2 is None or m2 is False
    net.cut = False
    assert fused.dry_program(opt, [net.pose], inp, None)[0] == "lpr"


def test_trace_positions_of_different_dtypes_do_not_collide():
    st = fused._DryState()
    with fused.DryTracer(st) as tr:
        outs = [tr._out((4, 6), torch.float32), tr._out((4, 6), torch.float32), tr._out((4, 6), torch.float64),
                tr._out((4, 6), torch.float64)]
    assert len({fused._key(o) for o in outs}) == 4 and len({fused._tok(o) for o in outs}) == 4
