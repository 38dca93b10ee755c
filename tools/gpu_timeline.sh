#!/bin/bash
# kernel-by-kernel timeline of the pose-graph LM step (10k / 40k), default and static
set -u
cd "$(dirname "$0")/.."; R=$PWD; mkdir -p gpurun_out/tl; export TMPDIR=/tmp
for mode in 0 1; do
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl/m$mode -o pgo -- python $R/tools/pgo_loop.py 10000 40000 8 $mode > $R/gpurun_out/tl/loop_$mode.log 2>&1
  cd $R; tail -4 gpurun_out/tl/loop_$mode.log
  f=$(find gpurun_out/tl/m$mode -name "*kernel_trace.csv" | head -1)
  python tools/timeline_summary.py "$f" pgo_linearize > gpurun_out/tl/timeline_$mode.txt 2>&1; head -60 gpurun_out/tl/timeline_$mode.txt
  rm -f "$f"; find gpurun_out/tl/m$mode -name "*.db" -delete
done
