#!/bin/bash
# Round 3: one GPU-box visit with everything the judge reads -- parity tests, smoke, bench, rocprof kernel stats, PMC traffic.
set -u
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tee gpurun_out/r03/pytest_gpu_full.log | tail -6 | cut -c1-200
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r03/smoke.log
echo "== bench"; timeout 900 python bench.py > gpurun_out/r03/bench_final.json 2> gpurun_out/r03/bench_final.err; tail -c 200 gpurun_out/r03/bench_final.err; head -c 600 gpurun_out/r03/bench_final.json; echo
echo "== tools"; timeout 300 python tools/time_pcg_iter.py 2>&1 | tail -1 > gpurun_out/r03/pcg_iter.json
timeout 600 python tools/time_lpr.py 2>&1 | tail -1 > gpurun_out/r03/time_lpr.json
timeout 100 tools/micro/build/pingpong > gpurun_out/r03/pingpong.log 2>&1
echo "== rocprof headline"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/prof_headline -o bench -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary > $R/gpurun_out/r03/rocprof_headline.log 2>&1; tail -1 $R/gpurun_out/r03/rocprof_headline.log | cut -c1-300
echo "== rocprof all legs"; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/prof_all -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r03/rocprof_all.log 2>&1; tail -1 $R/gpurun_out/r03/rocprof_all.log | cut -c1-200
cd $R; find gpurun_out/r03 -name "*kernel_trace.csv" -delete; find gpurun_out/r03 -name "*.db" -delete
for d in prof_headline prof_all; do f=$(find gpurun_out/r03/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r03/${d}_kernel_stats.csv && head -8 "$f" | cut -c1-160; done
echo "== pmc"; bash tools/gpu_pmc.sh > gpurun_out/r03/pmc.log 2>&1; cp gpurun_out/pmc/pmc_raw.json gpurun_out/r03/pmc_raw.json 2>/dev/null; tail -5 gpurun_out/r03/pmc.log | cut -c1-200
echo "== timelines / latency split"; bash tools/gpu_timeline.sh > gpurun_out/r03/timeline.log 2>&1; bash tools/gpu_timeline100k.sh > gpurun_out/r03/timeline100k.log 2>&1
cp gpurun_out/tl/timeline_0.txt gpurun_out/r03/pgo_step_timeline_default.txt; cp gpurun_out/tl/timeline_1.txt gpurun_out/r03/pgo_step_timeline_static.txt; cp gpurun_out/tl/timeline_100k.txt gpurun_out/r03/pgo100k_step_timeline.txt
timeout 200 python tools/time_c1_parts.py > gpurun_out/r03/c1_parts.log 2>&1; tail -7 gpurun_out/r03/c1_parts.log
timeout 300 python tools/time_pgo_host.py 2>&1 | tail -1 > gpurun_out/r03/pgo_host_split.log; timeout 300 python tools/time_pgo_host.py static 2>&1 | tail -1 >> gpurun_out/r03/pgo_host_split.log; cat gpurun_out/r03/pgo_host_split.log
