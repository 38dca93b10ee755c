#!/bin/bash
# Round 5: two-launch trial tail -- tests, the two pose-graph legs, 10k timeline
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/r05_${1:-i}; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest -q -m gpu --tb=short -p no:cacheprovider tests/test_pgo_trial_tail_gpu.py tests/test_pgo_capture_large_gpu.py tests/test_optim_gpu.py tests/test_determinism_gpu.py tests/test_fullsize_parity_gpu.py tests/test_robust_gpu.py 2>&1 | tail -12 | cut -c1-300
timeout 600 python - <<'P'
import sys, json, os, torch
sys.path.insert(0, '.')
import bench
dev = torch.device("cuda:0")
inst = bench._host_instances(False, False)
r = bench.pgo_lm_rate(dev, 10_000, 40_000, reps=25, problem=inst.get("lm_pgo"))
print("10k", json.dumps({k: r.get(k) for k in ("value", "pcg_iterations", "losses", "static_model_value")})[:600])
r = bench.pgo_lm_rate(dev, 100_000, 400_000, reps=9, with_static=False, problem=inst.get("lm_pgo_100k"))
print("100k", json.dumps({k: r.get(k) for k in ("value", "pcg_iterations", "losses")})[:600])
P
bash tools/gpu_timeline.sh 2>&1 | grep -A14 "^---- step" | head -34 | cut -c1-120
