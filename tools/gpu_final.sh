#!/bin/bash
# The round's record visit: everything profiles/<tag>/SUMMARY.md is generated from.  tools/gpu_final.sh r06
set -u
cd "$(dirname "$0")/.."; R=$PWD; TAG=${1:-r06}; O=$R/gpurun_out/$TAG; mkdir -p "$O"; export TMPDIR=/tmp
bash tools/gpu_visit.sh $TAG tests smoke bench prof
cp "$O/bench_line.json" "$O/bench_final.json"
f=$(find "$O/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/prof_headline_kernel_stats.csv"
# all-legs kernel statistics
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_all" -o all -- python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$O/rocprof_all.log" 2>&1)
f=$(find "$O/prof_all" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/prof_all_kernel_stats.csv"
find "$O/prof_all" "$O/prof" -name "*kernel_trace.csv" -delete 2>/dev/null; find "$O" -name "*.db" -delete 2>/dev/null
for leg in lm_pgo lm_pgo_100k lm_invnet; do bash tools/gpu_prof_leg.sh $TAG $leg > /dev/null 2>&1; ls "$O/${leg}_kernel_stats.csv" 2>/dev/null; done
python tools/time_pcg_iter.py > "$O/pcg_iter.json" 2> "$O/pcg_iter.err"; tail -3 "$O/pcg_iter.json" | cut -c1-200
bash tools/gpu_timeline.sh > "$O/timeline10k.log" 2>&1; cp gpurun_out/tl/timeline_0.txt "$O/pgo10k_step_timeline.txt"
bash tools/gpu_timeline100k.sh > "$O/timeline100k.log" 2>&1; cp gpurun_out/tl/timeline_100k.txt "$O/pgo100k_step_timeline.txt"
bash tools/gpu_pmc.sh > "$O/pmc.log" 2>&1; cp gpurun_out/pmc/pmc_raw.json "$O/pmc_raw.json"; tail -5 "$O/pmc.log" | cut -c1-200
rm -rf "$O/prof_all" "$O/prof"
ls "$O"
