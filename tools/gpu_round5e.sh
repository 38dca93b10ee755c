#!/bin/bash
# Round 5, visit 5: three-launch BA iteration, bae kernel routes (with diagnostics), handles, gauge thresholds
set -u
cd "$(dirname "$0")/.."
R=$PWD
O=gpurun_out/r05_${1:-v5}
mkdir -p $O
export TMPDIR=/tmp
echo "== new tests"
timeout 1500 python -m pytest -q -m gpu --tb=short -p no:cacheprovider -s \
   tests/test_ba_gpu.py tests/test_bae_compat_gpu.py tests/test_pcg_gauge_gpu.py tests/test_lie_parity_gpu.py::test_prepared_handles_carry_the_plain_eager_case \
   tests/test_reproj_gpu.py tests/test_determinism_gpu.py 2>&1 | tee $O/pytest_new.log | grep -v Warning | tail -30 | cut -c1-900
echo "== BA step timing"
for v in 1 0; do PPLIE_MG3=$v timeout 300 python tools/prof_ba.py 2>&1 | grep "s/step" | tee $O/prof_ba_mg3_$v.log | cut -c1-300; done
PPLIE_MG3=1 timeout 300 python tools/prof_ba.py f64 2>&1 | grep "s/step" | tee $O/prof_ba_mg3_1_f64.log | cut -c1-300
echo "== rocprof BA"; cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_ba -o ba -- python $R/tools/prof_ba.py > $R/$O/rocprof_ba.log 2>&1; cd $R
f=$(find $O/prof_ba -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/prof_ba_kernel_stats.csv && head -12 "$f" | cut -c1-170
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/prof_ba
ls $O
