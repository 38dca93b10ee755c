"""Run ONE leg of bench.py (for rocprofv3 --kernel-trace --stats / --pmc around it):
    python tools/prof_leg.py imu_train | imu | ops_10m | lm_pgo | lm_pgo_100k | lm_invnet | ba_reproj | c1 | scan_bwd"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
leg = sys.argv[1]
if leg == "scan_bwd":
    import pypose_amd as pp
    res = {}
    for name, rnd in (("so3", pp.randn_SO3), ("se3", pp.randn_SE3)):
        for left in (True, False):
            X = rnd(4096, 1025, device=dev, requires_grad=True)
            W = torch.randn(4096, 1025, X.shape[-1], device=dev)

            def f():
                Y = pp.cumprod(X, dim=1, left=left)
                return torch.autograd.grad([Y.tensor()], [X], [W])
            res[f"{name}_{'left' if left else 'right'}_fwd_bwd_ms"] = bench._event_ms(dev, f, 10)
    print(json.dumps(res))
else:
    fn = {"imu_train": bench.imu_train_rate, "imu": bench.imu_rate, "ops_10m": bench.ops_10m_rates, "lm_pgo": bench.pgo_lm_rate,
          "lm_pgo_100k": lambda d: bench.pgo_lm_rate(d, 100_000, 400_000, with_static=False), "lm_invnet": bench.invnet_lm_rate,
          "ba_reproj": bench.reproj_rate, "c1": bench.c1_latency}[leg]
    print(json.dumps(fn(dev)))
