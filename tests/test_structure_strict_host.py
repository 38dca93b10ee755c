"""LM(structure="strict"): structure verdicts are cached per shape signature and re-probed every 64 linearisations;
a model whose row dependence changes with a flag at FIXED shapes runs on a stale "block" verdict in between -- silently wrong J.
The strict switch probes at every linearisation: its trajectory equals the dense linearisation's (always correct) through the change."""
import numpy as np
import pytest
import torch

import pypose_amd as pp
from tests.oracle_backend import oracle_backend


class Switch(torch.nn.Module):
    """r_i = (p_i [+ 0.5 p_{i-1} once `couple` is set]) * x_i - 1: block-diagonal Jacobian until the flag flips, same shapes after"""

    def __init__(self, n):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        self.p = torch.nn.Parameter(torch.randn(n, 3, generator=g, dtype=torch.float64))
        self.couple = False

    def forward(self, x):
        q = self.p + 0.5 * self.p.roll(1, 0) if self.couple else self.p
        return q * x - 1.0


def _run(steps, flip_at, **kw):
    torch.manual_seed(0)
    net = Switch(12)
    x = torch.linspace(0.5, 2.0, 36, dtype=torch.float64).reshape(12, 3)
    structured = kw.pop("structured", True)
    opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(damping=1e-3), **kw)
    opt.structured = structured
    out, kinds = [], []
    for k in range(steps):
        if k == flip_at:
            net.couple = True
        out.append(float(opt.step(x)))
        kinds.append(opt.linearization)
    return np.array(out), kinds


def test_strict_reprobes_every_linearisation(monkeypatch):
    with oracle_backend():
        dense, kd = _run(6, 3, structured=False)
        strict, ks = _run(6, 3, structure="strict")
        cached, kc = _run(6, 3)
    assert set(kd) == {"dense"}
    assert ks[:3] == ["block"] * 3 and ks[3:] == ["dense"] * 3, ks          # the probe sees the coupling in the very step it appears
    np.testing.assert_allclose(strict, dense, rtol=1e-7, atol=1e-20)      # (equal up to the rounding of losses that reach 1e-16)
    # the default keeps its verdict for up to 64 linearisations: the block path's J misses the coupling (the documented exposure)
    assert kc == ["block"] * 6, kc
    assert not np.allclose(cached[3:], dense[3:], rtol=1e-6)


def test_strict_by_environment(monkeypatch):
    monkeypatch.setenv("PPLIE_STRUCTURE", "strict")
    with oracle_backend():
        net = Switch(4)
        opt = pp.optim.LM(net)
    assert opt.structure == "strict"
    with pytest.raises(AssertionError):
        pp.optim.LM(Switch(4), structure="loose")
