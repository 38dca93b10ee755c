#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3g
export TMPDIR=/tmp
echo "== pytest"
timeout 1500 python -m pytest tests/test_lpr_gpu.py tests/test_lm_device_gpu.py tests/test_fullsize_parity_gpu.py tests/test_activate_gpu.py \
    -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tee gpurun_out/r3g/pytest_full.log | tail -30
echo "== time_lpr"; timeout 600 python tools/time_lpr.py 2>&1 | tee gpurun_out/r3g/time_lpr.log | tail -14 | cut -c1-400
echo "== time_pgo static"; timeout 200 python tools/time_pgo.py 10000 40000 static 2>&1 | head -2 | tee gpurun_out/r3g/time_pgo_static.log
