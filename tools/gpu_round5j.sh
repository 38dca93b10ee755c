#!/bin/bash
# Round 5: ghost kernel with Z^T r advanced locally -- gauge tests, iteration timing, 10k leg
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/r05_${1:-j}; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest -q -m gpu --tb=short -p no:cacheprovider tests/test_pcg_gauge_gpu.py tests/test_optim_gpu.py tests/test_fullsize_parity_gpu.py tests/test_determinism_gpu.py 2>&1 | tail -8 | cut -c1-300
timeout 300 python tools/time_pcg_iter.py 2>&1 | tail -1 > $O/pcg_iter.json; python - <<P
import json
d=json.load(open("$O/pcg_iter.json"))
print({k: (v.get("marginal_us_per_iteration") if isinstance(v, dict) else v) for k, v in d.items() if k.startswith("ghost_grid")})
print(d.get("ghost_grid256_phase_us_per_iteration"))
P
timeout 600 python - <<'P'
import sys, json, os, torch
sys.path.insert(0, '.')
import bench
dev = torch.device("cuda:0")
inst = bench._host_instances(False, False)
r = bench.pgo_lm_rate(dev, 10_000, 40_000, reps=25, problem=inst.get("lm_pgo"))
print("10k", json.dumps({k: r.get(k) for k in ("value", "pcg_iterations", "losses", "static_model_value")})[:600])
P
