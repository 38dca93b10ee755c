#!/bin/bash
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/r05_${1:-v10}; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest -q -m gpu --tb=short -p no:cacheprovider tests/test_pcg_gauge_gpu.py tests/test_optim_gpu.py tests/test_pgo_trial_tail_gpu.py tests/test_determinism_gpu.py tests/test_fullsize_parity_gpu.py 2>&1 | tail -5 | cut -c1-300
PPLIE_PCG_GAUGE=1 timeout 300 python tools/time_pcg_iter.py 2>&1 | tail -1 > $O/pcg_iter_gauge1.json; cut -c1-1500 $O/pcg_iter_gauge1.json; echo
timeout 600 python - <<'P'
import sys, json, time, torch
sys.path.insert(0, '.')
import bench
dev = torch.device("cuda:0")
inst = bench._host_instances(False, False)
r = bench.pgo_lm_rate(dev, 10_000, 40_000, reps=25, problem=inst.get("lm_pgo"))
print(json.dumps({k: r[k] for k in ("value", "pcg_iterations", "losses", "static_model_value")}))
P
