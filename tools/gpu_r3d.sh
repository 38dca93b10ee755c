#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3d
export TMPDIR=/tmp
echo "== pcg iteration"; timeout 300 python tools/time_pcg_iter.py 2>&1 | tail -1 | tee gpurun_out/r3d/pcg_iter.log
echo "== pytest"
timeout 1500 python -m pytest tests/test_matrix_helpers.py tests/test_activate_module_gpu.py tests/test_activate_gpu.py tests/test_fullsize_parity_gpu.py tests/test_lm_device_gpu.py tests/test_optim_gpu.py tests/test_lm_golden2_gpu.py \
    -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tee gpurun_out/r3d/pytest_full.log | tail -15
echo "== time_pgo default"; timeout 200 python tools/time_pgo.py 2>&1 | tail -14 | tee gpurun_out/r3d/time_pgo.log
echo "== time_pgo static"; timeout 200 python tools/time_pgo.py 10000 40000 static 2>&1 | head -2 | tee gpurun_out/r3d/time_pgo_static.log
