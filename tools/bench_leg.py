"""One leg of bench.py by name, without the rest of the line:  python tools/bench_leg.py lm_invnet | lm_pgo | lm_pgo_100k | c1 | imu ...
Prints the leg's value and a few of its fields (the full block goes to gpurun_out/bench_leg_<name>.json)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pypose_amd as pp

name = sys.argv[1]
dev = torch.device("cuda:0")
pp.optim.freeze_gc()
legs = {"lm_invnet": lambda: bench.invnet_lm_rate(dev), "lm_pgo": lambda: bench.pgo_lm_rate(dev),
        "lm_pgo_100k": lambda: bench.pgo_lm_rate(dev, 100_000, 400_000, reps=3, with_static=False),
        "c1": lambda: bench.c1_latency(dev), "imu": lambda: bench.imu_rate(dev), "imu_train": lambda: bench.imu_train_rate(dev),
        "ba_reproj": lambda: bench.reproj_rate(dev)}
blk = legs[name]()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(blk, open(f"gpurun_out/bench_leg_{name}.json", "w"), default=str)
keep = {k: blk[k] for k in ("value", "unit", "static_model_value", "pcg_iterations", "repetitions_ms_per_step", "fwd_us", "fwd_bwd_us") if k in blk}
print(name, json.dumps(keep, default=str))
