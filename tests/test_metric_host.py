"""pp.metric.ape / rpe and pp.svdtf / svdstf against goldens recorded from the real reference
(tests/golden/make_metric_golden.py); host logic through the oracle backend."""
import os
import warnings

import numpy as np
import pytest
import torch

import pypose_amd as pp
from tests.oracle_backend import oracle_backend

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "metric_golden.npz"))
ORDER = ['Max', 'Min', 'Mean', 'Median', 'RMSE', 'SSE', 'STD']
ETYPES = ["translation", "rotation", "pose", "radian", "degree"]
APE = {"plain": {}, "align": {"align": True}, "scale": {"align": True, "scale": True}, "origin": {"origin": True},
       "offset": {"offset": 0.1, "diff": 0.02}}
RPE = {"frame1": {}, "frame3_all": {"delta": 3, "all": True}, "dist": {"associate": "distance", "delta": 0.8, "rtol": 0.5},
       "dist_all": {"associate": "distance", "delta": 0.8, "rtol": 0.3, "all": True},
       "rpair_align": {"rpair": True, "align": True, "delta": 2}}


def inputs(device="cpu"):
    t = lambda k: torch.from_numpy(G[k]).to(device)
    return t("rstamp"), pp.SE3(t("gt")), t("estamp"), pp.SE3(t("est"))


def vec(d):
    return torch.stack([d[k] for k in ORDER]).cpu()


def check_all(device, rtol):
    args = inputs(device)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, kw in APE.items():
            for et in ETYPES:
                torch.testing.assert_close(vec(pp.metric.ape(*args, etype=et, **kw)), torch.from_numpy(G[f"ape/{name}/{et}"]),
                                           rtol=rtol, atol=rtol, msg=lambda m: f"ape/{name}/{et}: {m}")
        for name, kw in RPE.items():
            for et in ETYPES:
                torch.testing.assert_close(vec(pp.metric.rpe(*args, etype=et, **kw)), torch.from_numpy(G[f"rpe/{name}/{et}"]),
                                           rtol=rtol, atol=rtol, msg=lambda m: f"rpe/{name}/{et}: {m}")
        single = pp.metric.ape(*args, otype="RMSE")
        torch.testing.assert_close(single.cpu(), torch.from_numpy(G["ape/single"]), rtol=rtol, atol=rtol)
        t = lambda k: torch.from_numpy(G[k]).to(device)
        doc = (t("doc_rstamp"), pp.SE3(t("doc_rpose")), t("doc_estamp"), pp.SE3(t("doc_epose")))
        torch.testing.assert_close(vec(pp.metric.ape(*doc)), torch.from_numpy(G["doc/ape"]), rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(vec(pp.metric.rpe(*doc)), torch.from_numpy(G["doc/rpe"]), rtol=1e-5, atol=1e-7)


def check_registration(device, tol):
    t = lambda k: torch.from_numpy(G[k]).to(device)
    same = lambda X, k: torch.testing.assert_close(X.tensor().cpu(), torch.from_numpy(G[k]), rtol=tol, atol=tol)
    same(pp.svdtf(t("reg_src"), t("reg_tgt")), "svdtf")
    same(pp.svdstf(t("reg_src"), t("reg_tgt")), "svdstf")
    same(pp.svdstf(t("reg_src"), t("reg_tgt"), with_scale=False), "svdstf_noscale")


def test_metrics_match_reference():
    with oracle_backend():
        check_all("cpu", 1e-9)


def test_registration_matches_reference():
    with oracle_backend():
        check_registration("cpu", 1e-10)
        flat = torch.from_numpy(G["reg_flat"])
        got = pp.svdtf(flat, -flat.flip(-1)).tensor()
        ref = torch.from_numpy(G["svdtf_flat"])
        sign = torch.sign((got[..., 3:] * ref[..., 3:]).sum(-1, keepdim=True))
        torch.testing.assert_close(got[..., :3], ref[..., :3], rtol=1e-8, atol=1e-8)
        torch.testing.assert_close(got[..., 3:], sign * ref[..., 3:], rtol=1e-8, atol=1e-8)


def test_argument_errors():
    with oracle_backend():
        args = inputs()
        with pytest.raises(ValueError, match="Unknown output type"):
            pp.metric.ape(*args, etype="metres")
        with pytest.raises(ValueError, match="Unknown output metric type"):
            pp.metric.ape(*args, otype="Mode")
        with pytest.raises(ValueError, match="unsupported delta unit"):
            pp.metric.rpe(*args, associate="seconds")
        with pytest.raises(AssertionError):
            pp.metric.ape(args[0], args[1], args[2] + 100.0, args[3])        # no stamp within `diff`
        with pytest.raises(AssertionError):
            pp.metric.StampedSE3(args[0].flip(0), args[1])                   # stamps must ascend
        with pytest.warns(Warning, match="not enough"):
            drift = torch.tensor([0.0, 0.05, 0.05, 0.05], dtype=torch.float64)      # only the first stamp still matches
            pp.metric.ape(args[0], args[1], args[2][:4] + drift, args[3][:4], thresh=0.5)


@pytest.mark.gpu
def test_metrics_on_device():
    from pypose_amd import _C
    assert _C._test_backend is None
    check_all("cuda:0", 1e-8)
    check_registration("cuda:0", 1e-9)
