#!/bin/bash
# Round 4: one GPU-box visit with everything the judge reads -- parity tests, smoke, bench, rocprof kernel stats, PMC traffic, SQ counters.
#   tools/gpu_round4.sh [quick]      (quick: skip the full pytest run)
set -u
cd "$(dirname "$0")/.."
R=$PWD
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
if [ "${1:-}" != "quick" ]; then
  echo "== pytest -m gpu"; timeout 1700 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tee $O/pytest_gpu_full.log | tail -6 | cut -c1-200
fi
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
echo "== bench"; timeout 900 python bench.py > $O/bench_final.json 2> $O/bench_final.err; tail -c 200 $O/bench_final.err; head -c 400 $O/bench_final.json; echo
echo "== rocprof headline"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_headline -o bench -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary > $R/$O/rocprof_headline.log 2>&1; tail -1 $R/$O/rocprof_headline.log | cut -c1-300
echo "== rocprof all legs"; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_all -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/rocprof_all.log 2>&1; tail -1 $R/$O/rocprof_all.log | cut -c1-200
cd $R; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
for d in prof_headline prof_all; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${d}_kernel_stats.csv && head -6 "$f" | cut -c1-160; done
rm -rf $O/prof_headline $O/prof_all
echo "== pmc traffic"; bash tools/gpu_pmc.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc/pmc_raw.json $O/pmc_raw.json 2>/dev/null; tail -3 $O/pmc.log | cut -c1-200
echo "== pmc sq"
for leg in imu imu_train lm_invnet; do
  bash tools/gpu_pmc_sq.sh $leg "" -- python $R/tools/prof_leg.py $leg > $O/sq_$leg.log 2>&1
  cp gpurun_out/pmc_sq/$leg.json $O/sq_$leg.json 2>/dev/null
done
bash tools/gpu_pmc_sq.sh scan_bwd "scan" -- python $R/tools/prof_leg.py scan_bwd > $O/sq_scan_bwd.log 2>&1; cp gpurun_out/pmc_sq/scan_bwd.json $O/sq_scan_bwd.json 2>/dev/null
echo "== tools"; timeout 300 python tools/time_pcg2.py 2>&1 | tail -1 > $O/pcg2_100k.json; cat $O/pcg2_100k.json | cut -c1-300
timeout 300 python tools/time_pcg_iter.py 2>&1 | tail -1 > $O/pcg_iter.json
for leg in imu_train scan_bwd; do bash tools/gpu_prof_leg.sh r04 $leg > /dev/null 2>&1; done
ls $O | head -50
