#!/bin/bash
# Round 5, visit 4: two-level exchange, bae kernel routes, handles / gb tests, pose-graph timeline
set -u
cd "$(dirname "$0")/.."
R=$PWD
O=gpurun_out/r05_${1:-v4}
mkdir -p $O
export TMPDIR=/tmp
echo "== new tests"
timeout 1500 python -m pytest -q -m gpu --tb=short -p no:cacheprovider -s \
   tests/test_pcg_gauge_gpu.py tests/test_bae_compat_gpu.py tests/test_lie_parity_gpu.py::test_prepared_handles_carry_the_plain_eager_case \
   "tests/test_lie_parity_gpu.py::test_broadcast_cotangent_variant_equals_the_materialised_launch" \
   "tests/test_lie_parity_gpu.py::test_sum_backward_takes_the_broadcast_route_and_equals_the_materialised_one" \
   2>&1 | tee $O/pytest_new.log | grep -v Warning | tail -30 | cut -c1-900
echo "== pose-graph suites"
timeout 2000 python -m pytest -q -m gpu --tb=short -p no:cacheprovider tests/test_optim_gpu.py tests/test_pgo_trial_tail_gpu.py tests/test_lm_device_gpu.py \
   tests/test_determinism_gpu.py tests/test_pack_blocks_gpu.py tests/test_robust_gpu.py tests/test_fullsize_parity_gpu.py tests/test_reference_suite_gpu.py 2>&1 | tee $O/pytest_pgo.log | grep -v "^\[W\|Gloo\|amdgpu.ids" | tail -12 | cut -c1-400
echo "== pcg timings"
PPLIE_PCG_GAUGE=1 timeout 300 python tools/time_pcg_iter.py 2>&1 | tail -1 > $O/pcg_iter_gauge1.json; cut -c1-1700 $O/pcg_iter_gauge1.json; echo
echo "== timeline"; bash tools/gpu_timeline.sh > $O/timeline.log 2>&1; cp gpurun_out/tl/timeline_0.txt $O/timeline_default.txt 2>/dev/null; cp gpurun_out/tl/timeline_1.txt $O/timeline_static.txt 2>/dev/null; head -45 $O/timeline_default.txt | cut -c1-150
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err; tail -c 1700 $O/bench.json; echo
ls $O
