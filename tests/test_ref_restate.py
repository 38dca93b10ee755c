"""oracle/ref_restate.py (the full-size CPU restatement of the reference's LM loop, built from reference functions) pinned
to trajectories recorded from the REAL reference optimizer (tests/golden/lm_golden2.npz, make_lm_golden2.py)."""
import numpy as np
import pytest
import torch

from oracle import ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="oracle/_ref not present (make -C oracle)")


def _golden():
    from tests.lm_golden2_util import G2
    return G2()


@pytest.mark.parametrize("B", [64, 1024])
@pytest.mark.parametrize("name,kw", [("constant", dict(damping=1e-4)), ("trustregion", dict(radius=10.0))])
def test_invnet_restatement_equals_reference_lm(B, name, kw):
    from oracle import ref_restate
    from tests.lm_golden2_util import invnet_problem
    G = _golden()
    inp, init = invnet_problem(G, B)
    rec = ref_restate.invnet_lm(init.tensor(), inp.tensor(), 5, strategy=name, strategy_kw=kw)
    ref = G[f"invnet{B}/{name}/loss"]
    for k, (a, b) in enumerate(zip(rec["loss"], ref)):
        if b > 1e-16:
            assert abs(a - b) <= 1e-8 * b, (k, rec["loss"], ref)
    above = ref > 1e-16
    np.testing.assert_allclose(np.asarray(rec["damping"])[above], G[f"invnet{B}/{name}/damping"][above], rtol=1e-12)
    np.testing.assert_array_equal(np.asarray(rec["reject"])[above], G[f"invnet{B}/{name}/reject"][above])
    fin = rec["final"][::max(1, B // 64)].numpy()
    np.testing.assert_allclose(fin, G[f"invnet{B}/{name}/final"], atol=1e-7)


@pytest.mark.parametrize("tag", ["pgo50", "pgo200"])
def test_pose_graph_restatement_equals_reference_dense_lm(tag):
    """CSR normal equations + the reference's CG (run to 1e-12) against the reference's dense Cholesky LM"""
    from oracle import ref_restate
    G = _golden()
    edges, poses, init = (torch.from_numpy(G[f"{tag}/{k}"]) for k in ("edges", "poses", "init"))
    rec = ref_restate.pgo_lm(init, edges, poses, 4, radius=1e4, tol=1e-12, maxiter=5000)
    np.testing.assert_allclose(rec["loss"], G[f"{tag}/noweight/loss"], rtol=1e-7)
    np.testing.assert_allclose(rec["damping"], G[f"{tag}/noweight/damping"], rtol=1e-12)
    np.testing.assert_array_equal(rec["reject"], G[f"{tag}/noweight/reject"])
    np.testing.assert_allclose(rec["final"].numpy(), G[f"{tag}/noweight/final"], atol=1e-6)


def test_pose_graph_blocks_equal_autograd_of_the_reference_model():
    """the closed-form per-edge blocks are the Jacobian torch.autograd computes through the reference's own Functions"""
    from oracle import ref_restate
    rpp = ref_loader.load()
    torch.manual_seed(3)
    nodes = rpp.randn_SE3(6, dtype=torch.float64)
    poses = rpp.randn_SE3(5, dtype=torch.float64)
    edges = torch.tensor([[0, 1], [1, 2], [2, 3], [5, 0], [4, 2]])
    r, J1, J2 = ref_restate.pgo_blocks(nodes.tensor(), edges, poses.tensor())

    def f(n):
        n = rpp.SE3(n)
        return (poses.Inv() @ n[edges[:, 0]].Inv() @ n[edges[:, 1]]).Log().tensor()
    J = torch.autograd.functional.jacobian(f, nodes.tensor())            # [5,6,6,7]
    for e, (i, j) in enumerate(edges.tolist()):
        np.testing.assert_allclose(J[e, :, i, :6].numpy(), J1[e].numpy(), atol=1e-10)
        np.testing.assert_allclose(J[e, :, j, :6].numpy(), J2[e].numpy(), atol=1e-10)
        assert float(J[e, :, :, 6].abs().max()) == 0.0
