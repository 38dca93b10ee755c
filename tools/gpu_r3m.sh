#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3m
export TMPDIR=/tmp
echo "== pcg iteration"; timeout 300 python tools/time_pcg_iter.py 2>&1 | tail -1 | tee gpurun_out/r3m/pcg_iter.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if 'ghost' in k or 'marginal' in k or 'diff' in k})"
echo "== pytest"
timeout 1500 python -m pytest tests/test_optim_gpu.py tests/test_lm_golden2_gpu.py tests/test_fullsize_parity_gpu.py tests/test_gram_mfma_gpu.py tests/test_distributed_gpu.py tests/test_examples_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tee gpurun_out/r3m/pytest_full.log | tail -12 | cut -c1-300
echo "== time_pgo default"; timeout 200 python tools/time_pgo.py 2>&1 | head -3 | tee gpurun_out/r3m/time_pgo.log
echo "== time_pgo static"; timeout 200 python tools/time_pgo.py 10000 40000 static 2>&1 | head -2 | tee gpurun_out/r3m/time_pgo_static.log
