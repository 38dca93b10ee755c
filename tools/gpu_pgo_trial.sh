#!/bin/bash
# the captured pose-graph trial with pinned-memory hand-off: tests, timing, timeline, host profile
set -u
cd "$(dirname "$0")/.."; R=$PWD; mkdir -p gpurun_out/r3p; export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_optim_gpu.py tests/test_lm_device_gpu.py tests/test_lm_golden2_gpu.py tests/test_fullsize_parity_gpu.py tests/test_determinism_gpu.py tests/test_distributed_gpu.py tests/test_examples_gpu.py -v -m gpu -x --tb=short -p no:cacheprovider > gpurun_out/r3p/pytest.log 2>&1
grep -n "FAILED\|ERROR" gpurun_out/r3p/pytest.log | tail -4; grep -n "Fatal\|fault\|Aborted" gpurun_out/r3p/pytest.log | head -10; tail -3 gpurun_out/r3p/pytest.log | cut -c1-200
for mode in 0 1; do echo "== loop static=$mode"; timeout 300 python tools/pgo_loop.py 10000 40000 8 $mode 2>&1 | tail -3; done
timeout 300 python tools/time_pgo_host.py 2>&1 | tail -1; timeout 300 python tools/time_pgo_host.py static 2>&1 | tail -1
bash tools/gpu_timeline.sh 2>&1 | grep -v "^[EW]2026" | grep -A12 "steps seen" | head -34
