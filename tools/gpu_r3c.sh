#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c
export TMPDIR=/tmp
echo "== pingpong"; timeout 120 tools/micro/build/pingpong 2>&1 | tee gpurun_out/r3c/pingpong.log
echo "== pytest"
timeout 1500 python -m pytest tests/test_matrix_helpers.py tests/test_activate_module_gpu.py tests/test_activate_gpu.py tests/test_fullsize_parity_gpu.py tests/test_lm_device_gpu.py tests/test_imu_gpu.py \
    -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tee gpurun_out/r3c/pytest_full.log | tail -30
