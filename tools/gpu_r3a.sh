#!/bin/bash
# round 3, visit A: the changed paths (default dry-trace LM step, one-exchange persistent PCG, full-size parity)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
echo "== pytest (changed areas)"
timeout 1500 python -m pytest tests/test_lm_device_gpu.py tests/test_optim_gpu.py tests/test_lm_golden2_gpu.py tests/test_fullsize_parity_gpu.py tests/test_distributed_gpu.py \
    -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tee gpurun_out/r3a/pytest_full.log | tail -40
echo "== pcg iteration"; timeout 300 python tools/time_pcg_iter.py 2>&1 | tail -3 | tee gpurun_out/r3a/pcg_iter.log
echo "== time_pgo default"; timeout 200 python tools/time_pgo.py 2>&1 | tail -16 | tee gpurun_out/r3a/time_pgo.log
echo "== time_pgo static"; timeout 200 python tools/time_pgo.py 10000 40000 static 2>&1 | tail -16 | tee gpurun_out/r3a/time_pgo_static.log
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; tail -c 600 gpurun_out/r3a/bench.err; head -c 3000 gpurun_out/r3a/bench.json
