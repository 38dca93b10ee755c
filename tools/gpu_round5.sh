#!/bin/bash
# Round 5 GPU-box visits.  tools/gpu_round5.sh <mode> [tag]
#   new     : the tests added / changed this round, smoke, bench, rocprof of the headline
#   full    : whole -m gpu suite, smoke, bench, rocprof headline + all legs
set -u
cd "$(dirname "$0")/.."
R=$PWD
MODE=${1:-new}
O=gpurun_out/r05${2:+_$2}
mkdir -p $O
export TMPDIR=/tmp
if [ "$MODE" = "new" ]; then
  echo "== new tests"
  timeout 1500 python -m pytest -q -m gpu --tb=short -p no:cacheprovider -s \
     tests/test_jinvp_small_angle.py "tests/test_lie_parity_gpu.py::test_random_100k_fp32_vs_oracle_fp64" \
     tests/test_robust_gpu.py::test_robust_kernel_with_foreign_corrector_is_never_captured \
     tests/test_fullsize_parity_gpu.py::test_pgo_100k_400k_tight_solves_equal_reference_restatement 2>&1 | tee $O/pytest_new.log | tail -15 | cut -c1-600
else
  echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tee $O/pytest_gpu_full.log | tail -8 | cut -c1-300
fi
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err; tail -c 300 $O/bench_final.err; tail -c 1800 $O/bench_final.json; echo
echo "== rocprof headline"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_headline -o bench -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary > $R/$O/rocprof_headline.log 2>&1; tail -1 $R/$O/rocprof_headline.log | cut -c1-300
if [ "$MODE" = "full" ]; then
  echo "== rocprof all legs"; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_all -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/rocprof_all.log 2>&1; tail -1 $R/$O/rocprof_all.log | cut -c1-200
fi
cd $R; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
for d in prof_headline prof_all; do f=$(find $O/$d -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/${d}_kernel_stats.csv && head -6 "$f" | cut -c1-160; done
rm -rf $O/prof_headline $O/prof_all
ls $O
