"""Covariance kernel time at BASELINE configs[4] (4096 x 1024 fp32): module forward with and without covariance, their difference.
    python tools/time_imu_cov.py [B ...]      (tuning switches of csrc/scan.hip are read from the environment by the library)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
out = {}
for B in ([int(a) for a in sys.argv[1:]] or [4096]):
    r = bench.imu_rate(dev, B, 1024, reps=8)
    out[B] = {"with_covariance_us": round(r["with_covariance"]["ms"] * 1e3, 1), "states_only_us": round(r["states_only"]["ms"] * 1e3, 1),
              "difference_us": round((r["with_covariance"]["ms"] - r["states_only"]["ms"]) * 1e3, 1)}
print(json.dumps(out))
