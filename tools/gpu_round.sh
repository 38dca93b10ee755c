#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel stats.  Outputs under gpurun_out/.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
echo "== bench"; timeout 600 python bench.py --steps 50 --warmup 10 2>&1 | tail -2 | tee gpurun_out/bench.log
echo "== pgo"; timeout 300 python tools/bench_pgo.py 2>&1 | tail -4 | tee gpurun_out/bench_pgo.log
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; tail -3 $GRAFT_REPO_ROOT/gpurun_out/rocprof.log
cd $GRAFT_REPO_ROOT; find gpurun_out/prof -name "*kernel_trace.csv" -delete; find gpurun_out/prof -name "*.db" -delete; ls -R gpurun_out/prof | head -20
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-220
