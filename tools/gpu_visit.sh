#!/bin/bash
# One GPU-box visit, parametrised: tools/gpu_visit.sh <tag> <part> [<part> ...]
#   parts: tests | tests:<pytest args> | smoke | bench | prof (rocprofv3 kernel stats of the headline) | py:<script and args> | sh:<command>
# Outputs under gpurun_out/<tag>/ (scratch; copy what is to be judged into profiles/).
set -u
cd "$(dirname "$0")/.."
R=$PWD; TAG=${1:-visit}; shift || true
O=$R/gpurun_out/$TAG; mkdir -p "$O"; export TMPDIR=/tmp
for part in "$@"; do
  echo "=== $part"
  case "$part" in
    tests)   timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -15 | cut -c1-300 | tee "$O/pytest_gpu_tail.log" ;;
    tests:*) timeout 1200 python -m pytest -q -m gpu --tb=short -p no:cacheprovider ${part#tests:} 2>&1 | tail -25 | cut -c1-300 | tee -a "$O/pytest_sel_tail.log" ;;
    smoke)   timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee "$O/smoke.log" ;;
    bench)   PPLIE_BENCH_DETAIL=$O/bench_legs.json timeout 900 python bench.py --steps 20 --warmup 5 > "$O/bench_line.json" 2> "$O/bench_stderr.log"
             wc -c "$O/bench_line.json"; cut -c1-6000 "$O/bench_line.json" ;;
    prof)    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o headline -- python "$R/bench.py" --steps 50 --warmup 10 --no-cpu-baseline --no-secondary > "$O/rocprof.log" 2>&1)
             find "$O/prof" -name "*kernel_trace.csv" -delete; find "$O/prof" -name "*.db" -delete
             f=$(find "$O/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" | cut -c1-240 ;;
    py:*)    timeout 900 python ${part#py:} 2>&1 | tail -150 | cut -c1-900 | tee -a "$O/py.log" ;;
    sh:*)    timeout 1200 bash -c "${part#sh:}" 2>&1 | tail -60 | cut -c1-600 | tee -a "$O/sh.log" ;;
    *) echo "unknown part $part" ;;
  esac
done
