#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3j
export TMPDIR=/tmp
echo "== pytest"
timeout 1500 python -m pytest tests/test_pcg_p2p_gpu.py tests/test_lm_device_gpu.py tests/test_optim_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tee gpurun_out/r3j/pytest_full.log | tail -12 | cut -c1-300
echo "== prof bench invnet"; timeout 300 python tools/prof_bench_invnet.py 2>&1 | tee gpurun_out/r3j/prof_bench_invnet.log | grep -v "^$" | head -50 | cut -c1-180
