#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3h
export TMPDIR=/tmp
echo "== pytest"
timeout 1500 python -m pytest tests/test_lpr_gpu.py tests/test_lm_device_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tee gpurun_out/r3h/pytest_full.log | tail -12 | cut -c1-300
echo "== prof host"; timeout 300 python tools/prof_c3_host.py 2>&1 | tee gpurun_out/r3h/prof_c3_host.log | grep -E "us|ncalls|fused|optimizer|lietensor|posegraph|_C.py" | head -60 | cut -c1-200
