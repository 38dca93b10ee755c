"""Conversions + g2o loader on the CPU: the oracle against goldens recorded from the real reference, and the
host-side logic of pypose_amd.lietensor.convert / pypose_amd.io (through the oracle backend: no GPU here)."""
import os
import warnings

import numpy as np
import pytest
import torch

import pypose_amd as pp
from oracle import convert_np
from tests.oracle_backend import oracle_backend

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def G():
    return dict(np.load(os.path.join(HERE, "golden", "convert_golden.npz")))


def T(a):
    return torch.from_numpy(np.array(a))


def test_oracle_matches_reference_goldens(G):
    R = G["mat2so3/R"].reshape(-1, 9)
    np.testing.assert_allclose(convert_np.mat2so3_fwd(R)[0], G["mat2so3/q"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(convert_np.mat2so3_bwd(R, G["mat2so3/g"])[0], G["mat2so3/gR"].reshape(-1, 9), rtol=0, atol=2e-8)
    np.testing.assert_allclose(convert_np.euler2so3_fwd(G["euler2so3/e"])[0], G["euler2so3/q"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(convert_np.euler2so3_bwd(G["euler2so3/e"], G["euler2so3/g"])[0], G["euler2so3/ge"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(convert_np.so3_euler_fwd(G["euler/Q"])[0], G["euler/e"], rtol=0, atol=1e-14)
    gq = convert_np.so3_euler_bwd(G["euler/Q"], G["euler/g"])[0]
    ok = np.isfinite(G["euler/gQ"]).all(-1)                       # (the reference's asin gradient is inf at |t2| = 1)
    np.testing.assert_allclose(gq[ok], G["euler/gQ"][ok], rtol=1e-9, atol=1e-9)
    ids, nodes, edges, poses, infos = convert_np.read_g2o(os.path.join(HERE, "golden", "sample.g2o"))
    for a, k in ((ids, "ids"), (nodes, "nodes"), (edges, "edges"), (poses, "poses"), (infos, "infos")):
        np.testing.assert_array_equal(a, G["g2o/" + k])


def test_host_logic_matches_reference(G):
    with oracle_backend():
        q = pp.mat2SO3(T(G["mat2so3/R"]), check=False)
        assert q.ltype == pp.SO3_type and q.shape == (G["mat2so3/R"].shape[0], 4)
        np.testing.assert_allclose(q.tensor().numpy(), G["mat2so3/q"], atol=1e-14)
        np.testing.assert_allclose(pp.mat2SE3(T(G["mat2se3/M"])).tensor().numpy(), G["mat2se3/X"], atol=1e-14)
        np.testing.assert_allclose(pp.mat2SE3(T(G["mat2se3/M"])[:, :3, :]).tensor().numpy(), G["mat2se3/X34"], atol=1e-14)
        np.testing.assert_allclose(pp.mat2Sim3(T(G["mat2sim3/M"])).tensor().numpy(), G["mat2sim3/X"], atol=1e-13)
        np.testing.assert_allclose(pp.mat2RxSO3(T(G["mat2rxso3/M"])).tensor().numpy(), G["mat2rxso3/X"], atol=1e-13)
        X = pp.from_matrix(T(G["mat2se3/M"]), pp.SE3_type)
        assert X.ltype == pp.SE3_type
        np.testing.assert_allclose(X.tensor().numpy(), G["from_matrix/SE3"], atol=1e-14)
        np.testing.assert_allclose(pp.euler2SO3(T(G["euler2so3/e"])).tensor().numpy(), G["euler2so3/q"], atol=1e-15)
        np.testing.assert_allclose(pp.SO3(T(G["euler/Q"])).euler().numpy(), G["euler/e"], atol=1e-14)
        np.testing.assert_allclose(pp.euler(pp.se3(T(G["euler/se3"]))).numpy(), G["euler/se3_e"], atol=1e-13)
        np.testing.assert_allclose(pp.quat2unit(pp.SE3(T(G["quat2unit/in"]))).tensor().numpy(), G["quat2unit/out"], atol=1e-15)
        # autograd wiring: d mat2SO3 / d R and d euler2SO3 / d e through the backward entry points
        R = T(G["mat2so3/R"]).requires_grad_(True)
        (gR,) = torch.autograd.grad(pp.mat2SO3(R, check=False).tensor(), R, T(G["mat2so3/g"]))
        np.testing.assert_allclose(gR.numpy(), G["mat2so3/gR"], atol=2e-8)
        # batched leading dims and the 3x4 / 4x4 slicing
        M = T(G["mat2se3/M"]).reshape(4, 4, 4, 4)
        assert pp.mat2SO3(M).shape == (4, 4, 4) and pp.mat2SE3(M).shape == (4, 4, 7)


def test_argument_errors_like_the_reference():
    with oracle_backend():
        with pytest.raises(ValueError, match="at least 2 dimensions"):
            pp.mat2SO3(torch.zeros(3))
        with pytest.raises(ValueError, match="3 x 3 or"):
            pp.mat2SE3(torch.zeros(2, 2, 2))
        with pytest.raises(ValueError, match="not all orthogonal"):
            pp.mat2SO3(2 * torch.eye(3))
        with pytest.raises(ValueError, match="determinant"):
            pp.mat2SO3(torch.diag(torch.tensor([1., 1., -1.])))
        with pytest.raises(ValueError, match="not full rank"):
            pp.mat2Sim3(torch.zeros(3, 3))
        with pytest.raises(ValueError, match="must be one of"):
            pp.from_matrix(torch.eye(3), pp.so3_type)
        with pytest.raises(ValueError, match="zero quaternions"):
            pp.quat2unit(pp.SO3(torch.zeros(2, 4)))
        M = torch.eye(4)
        M[3, 0] = 0.5
        with pytest.warns(UserWarning, match="last rows"):
            pp.mat2SE3(M)
        with pytest.warns(UserWarning, match="not Lie group"):
            assert pp.quat2unit(pp.randn_so3(2)).ltype == pp.so3_type
        assert pp.mat2SO3(2 * torch.eye(3), check=False).shape == (4,)     # check=False skips the validation


def test_conversions_raise_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU compute path"):
        pp.mat2SO3(torch.eye(3))
    with pytest.raises(RuntimeError, match="no CPU compute path"):
        pp.euler2SO3(torch.zeros(2, 3))


def test_g2o_reader_and_writer(G, tmp_path):
    path = os.path.join(HERE, "golden", "sample.g2o")
    d = pp.io.read_g2o(path, dtype=torch.float64)
    for k in ("ids", "edges", "infos"):
        np.testing.assert_array_equal(d[k].numpy(), G["g2o/" + k])
    assert d["nodes"].ltype == pp.SE3_type and d["poses"].ltype == pp.SE3_type
    np.testing.assert_array_equal(d["nodes"].tensor().numpy(), G["g2o/nodes"])
    np.testing.assert_array_equal(d["poses"].tensor().numpy(), G["g2o/poses"])
    assert d["edges"].dtype == torch.int64 and d["ids"].dtype == torch.int64
    assert pp.io.read_g2o(path)["nodes"].dtype == torch.get_default_dtype()
    out = tmp_path / "roundtrip.g2o"
    pp.io.write_g2o(str(out), d["nodes"].tensor(), d["edges"], d["poses"].tensor(), d["infos"], d["ids"])
    d2 = pp.io.read_g2o(str(out), dtype=torch.float64)
    for k in ("ids", "edges", "infos"):
        assert torch.equal(d[k], d2[k])
    assert torch.equal(d["nodes"].tensor(), d2["nodes"].tensor()) and torch.equal(d["poses"].tensor(), d2["poses"].tensor())
    ds = pp.io.G2OPGO(os.path.join(HERE, "golden"), "sample.g2o")
    assert len(ds) == 20 and ds.init_value().shape == (12, 7) and ds[3][2].shape == (6, 6)
    bad = tmp_path / "bad.g2o"
    bad.write_text("VERTEX_SE3:QUAT 0 1 2 3\n")
    with pytest.raises(ValueError, match="8 fields"):
        pp.io.read_g2o(str(bad))
