#!/bin/bash
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/r05_${1:-v8}; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest -q -m gpu --tb=short -p no:cacheprovider tests/test_ba_gpu.py tests/test_bae_compat_gpu.py 2>&1 | tail -4 | cut -c1-300
for c in 128 256 512; do echo "chunk $c"; PPLIE_MG3_CHUNK=$c timeout 300 python tools/prof_ba.py 2>&1 | grep "s/step" | tee $O/prof_ba_chunk$c.log | cut -c1-120; done
cd /tmp && PPLIE_MG3_CHUNK=128 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_ba -o ba -- python $R/tools/prof_ba.py > $R/$O/rocprof_ba.log 2>&1; cd $R
f=$(find $O/prof_ba -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/prof_ba_kernel_stats.csv && head -4 "$f" | grep -o '^"void pplie::[a-z0-9_]*\|",[0-9]*,[0-9]*,[0-9.]*,[0-9.]*,'
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/prof_ba
