#!/bin/bash
# rocprofv3 kernel statistics of one bench leg:  tools/gpu_prof_leg.sh <tag> <leg>   -> gpurun_out/<tag>/<leg>_kernel_stats.csv
set -u
TAG=$1; LEG=$2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof_$LEG -o leg -- python $R/tools/prof_leg.py $LEG > $R/gpurun_out/$TAG/$LEG.log 2>&1
tail -1 $R/gpurun_out/$TAG/$LEG.log | cut -c1-1500
cd $R
f=$(find gpurun_out/$TAG/prof_$LEG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/$TAG/${LEG}_kernel_stats.csv && head -14 "$f" | cut -c1-170
rm -rf gpurun_out/$TAG/prof_$LEG
