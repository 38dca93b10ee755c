#!/bin/bash
set -u
cd "$(dirname "$0")/.."; R=$PWD; mkdir -p gpurun_out/first; export TMPDIR=/tmp
for v in plain idle hot; do
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/first/$v -o t -- python $R/tools/exp_first_step.py $v > $R/gpurun_out/first/$v.log 2>&1
  cd $R; grep "^rep" gpurun_out/first/$v.log | tail -1
  f=$(find gpurun_out/first/$v -name "*kernel_trace.csv" | head -1); python tools/exp_first_step_report.py "$f"; rm -rf gpurun_out/first/$v
done
