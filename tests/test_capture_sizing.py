"""Sizing of a captured LM trial on a graph beyond the persistent solve (optim/posegraph.py FusedPCG.unwatched_for; host logic, no GPU):
the capture queues the next multiple of eight above the longest watched solve, never fewer than 16, and is not made at all when that
exceeds the cap or the solver's own iteration limit."""
from pypose_amd.optim.posegraph import FusedPCG


def test_unwatched_iterations_are_the_next_multiple_of_eight():
    f = FusedPCG.unwatched_for
    assert f(0) is None                                     # no watched solve yet: nothing to size a capture by
    assert f(1) == 16 and f(14) == 16 and f(15) == 16       # at least 16
    assert f(16) == 24 and f(18) == 24 and f(23) == 24      # one more than the longest solve, rounded up (BASELINE configs[3]: 18 / 20 / 23)
    assert f(24) == 32 and f(39) == 40 and f(47) == 48
    assert f(48) is None and f(200) is None                 # beyond the cap the no-op launches outweigh the saved host latency
    assert f(23, maxiter=20) is None and f(23, maxiter=24) == 24 and f(23, maxiter=None) == 24
