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
    net.flip = True                                                   # a different program altogether
    assert fused.dry_program(opt, [net.pose], inp, None) is None


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


def test_other_launches_are_refused_during_a_dry_trace():
    with fused.DryTracer():
        with pytest.raises(_C.DryTraceEscape):
            _C.stream_ptr(torch.device("cpu"))
    assert not _C.dry_tracing()
