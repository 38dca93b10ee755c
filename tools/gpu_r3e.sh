#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3e
export TMPDIR=/tmp
echo "== pcg iteration"; timeout 300 python tools/time_pcg_iter.py 2>&1 | tail -1 | tee gpurun_out/r3e/pcg_iter.log
