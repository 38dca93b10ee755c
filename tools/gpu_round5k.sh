#!/bin/bash
# Round 5: packed diagonal blocks in the two-level two-launch iteration -- tests, iteration timing A/B, 100k leg A/B
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/r05_${1:-k}; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest -q -m gpu --tb=short -p no:cacheprovider tests/test_pcg_gauge_gpu.py tests/test_pack_blocks_gpu.py tests/test_pgo_capture_large_gpu.py "tests/test_fullsize_parity_gpu.py" -k "not 10k" 2>&1 | tail -12 | cut -c1-300
for dp in 1 0; do echo "PACK_DIAG=$dp"; PPLIE_PACK_DIAG=$dp timeout 300 python tools/time_pcg2.py 2>&1 | tail -1 | cut -c1-400; done
for dp in 1 0; do
PPLIE_PACK_DIAG=$dp timeout 600 python - <<'P'
import sys, json, os, torch
sys.path.insert(0, '.')
import bench
dev = torch.device("cuda:0")
inst = bench._host_instances(False, False)
r = bench.pgo_lm_rate(dev, 100_000, 400_000, reps=9, with_static=False, problem=inst.get("lm_pgo_100k"))
print("100k pack_diag", os.environ["PPLIE_PACK_DIAG"], json.dumps({k: r.get(k) for k in ("value", "pcg_iterations", "losses")})[:600])
P
done
