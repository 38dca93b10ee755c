#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3n
export TMPDIR=/tmp
echo "== pcg iteration"; timeout 300 python tools/time_pcg_iter.py 2>&1 | tail -1 | tee gpurun_out/r3n/pcg_iter.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if 'ghost' in k or 'diff' in k})"
timeout 600 python -m pytest tests/test_optim_gpu.py -q -m gpu --tb=short -p no:cacheprovider -k "ghost or pcg or posegraph" 2>&1 | tail -4 | cut -c1-300
