#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3k
export TMPDIR=/tmp
timeout 200 python tools/debug_shortcut.py 2>&1 | tail -14 | tee gpurun_out/r3k/debug.log
grep -n "^E \|Error" gpurun_out/r3j/pytest_full.log | head -5
