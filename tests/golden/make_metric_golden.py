"""Goldens for pp.metric.ape / rpe and pp.svdtf / svdstf from the REAL reference:
    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_metric_golden.py"""
import os
import sys
import warnings

import numpy as np
import torch

sys.dont_write_bytecode = True
import pypose as pp  # noqa: E402

D = torch.float64
torch.manual_seed(11)
S = {}
n = 40
gt = pp.cumprod(pp.randn_SE3(n, sigma=0.2, dtype=D), dim=0, left=False)
est = pp.randn_SE3(1, sigma=0.4, dtype=D) * gt * pp.randn_SE3(n, sigma=0.03, dtype=D)
est = pp.SE3(torch.cat([1.3 * est.translation(), est.rotation().tensor()], -1))          # scale drift
rstamp = torch.arange(n, dtype=D) * 0.1
estamp = rstamp[3:35] + 0.004 * torch.rand(32, dtype=D)
est_sub = est[3:35]
S.update(gt=gt.tensor(), est=est_sub.tensor(), rstamp=rstamp, estamp=estamp)
CASES = {
    "plain": {}, "align": {"align": True}, "scale": {"align": True, "scale": True}, "origin": {"origin": True},
    "offset": {"offset": 0.1, "diff": 0.02},
}
ETYPES = ["translation", "rotation", "pose", "radian", "degree"]
ORDER = ['Max', 'Min', 'Mean', 'Median', 'RMSE', 'SSE', 'STD']
vec = lambda d: torch.stack([d[k] for k in ORDER])
# NB the reference shifts the longer trajectory's stamps IN PLACE by `offset` (ape_rpe.py:133 `stamps_2 += offset_2`,
# and StampedSE3 keeps the caller's float64 tensor): every call below therefore gets fresh copies.
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for name, kw in CASES.items():
        for et in ETYPES:
            S[f"ape/{name}/{et}"] = vec(pp.metric.ape(rstamp.clone(), gt, estamp.clone(), est_sub, etype=et, **kw))
    RPE = {"frame1": {}, "frame3_all": {"delta": 3, "all": True}, "dist": {"associate": "distance", "delta": 0.8, "rtol": 0.5},
           "dist_all": {"associate": "distance", "delta": 0.8, "rtol": 0.3, "all": True}, "rpair_align": {"rpair": True, "align": True, "delta": 2}}
    for name, kw in RPE.items():
        for et in ETYPES:
            S[f"rpe/{name}/{et}"] = vec(pp.metric.rpe(rstamp.clone(), gt, estamp.clone(), est_sub, etype=et, **kw))
    S["ape/single"] = pp.metric.ape(rstamp.clone(), gt, estamp.clone(), est_sub, otype="RMSE")
    # the docstring example (fp32 stamps of ~1.3e9 s collapse to one value: every estimate matches reference pose 0)
    drs = torch.tensor([1311868163.8696999550, 1311868163.8731000423, 1311868163.8763999939])
    drp = pp.SE3([[-0.1357000023, -1.4217000008, 1.4764000177, 0.6452999711, -0.5497999787, 0.3362999856, -0.4101000130],
                  [-0.1357000023, -1.4218000174, 1.4764000177, 0.6453999877, -0.5497000217, 0.3361000121, -0.4101999998],
                  [-0.1358000040, -1.4219000340, 1.4764000177, 0.6455000043, -0.5498999953, 0.3357999921, -0.4101000130]])
    des = torch.tensor([1311868164.3631811142, 1311868164.3990259171, 1311868164.4309399128])
    dep = pp.SE3([[0.0000000000, 0.0000000000, 0.0000000000, 0.0000000000, 0.0000000000, 0.0000000000, 1.0000000000],
                  [-0.0005019300, 0.0010138600, -0.0020097860, -0.0020761820, -0.0010706080, -0.0007627490, 0.9999969602],
                  [0.0004298200, 0.0019603260, -0.0048985220, -0.0043526068, -0.0036625920, -0.0023494449, 0.9999810457]])
    S.update(doc_rstamp=drs, doc_rpose=drp.tensor(), doc_estamp=des, doc_epose=dep.tensor())
    S["doc/ape"] = vec(pp.metric.ape(drs, drp, des, dep))
    S["doc/rpe"] = vec(pp.metric.rpe(drs, drp, des, dep))
# point-cloud registration
src = torch.randn(2, 50, 3, dtype=D)
T = pp.randn_Sim3(2, dtype=D)
tgt = T.unsqueeze(-2).Act(src) + 0.01 * torch.randn(2, 50, 3, dtype=D)
S.update(reg_src=src, reg_tgt=tgt, svdtf=pp.svdtf(src, tgt).tensor(), svdstf=pp.svdstf(src, tgt).tensor(),
         svdstf_noscale=pp.svdstf(src, tgt, with_scale=False).tensor())
mirror = src.clone()
mirror[..., 2] = 0                                   # planar cloud: the reflection branch can trigger
S.update(reg_flat=mirror, svdtf_flat=pp.svdtf(mirror, -mirror.flip(-1)).tensor())
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "metric_golden.npz"),
                    **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in S.items()})
print("ok", len(S))
