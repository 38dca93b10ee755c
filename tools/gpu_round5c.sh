#!/bin/bash
# Round 5, visit 3: re-validation of the gauge preconditioner after the fixes, handles, tuned tiles
set -u
cd "$(dirname "$0")/.."
R=$PWD
O=gpurun_out/r05_${1:-v3}
mkdir -p $O
export TMPDIR=/tmp
echo "== new tests"
timeout 1500 python -m pytest -q -m gpu --tb=short -p no:cacheprovider -s \
   tests/test_pcg_gauge_gpu.py tests/test_lie_parity_gpu.py::test_prepared_handles_carry_the_plain_eager_case \
   "tests/test_lie_parity_gpu.py::test_broadcast_cotangent_variant_equals_the_materialised_launch" \
   "tests/test_lie_parity_gpu.py::test_sum_backward_takes_the_broadcast_route_and_equals_the_materialised_one" \
   tests/test_fullsize_parity_gpu.py 2>&1 | tee $O/pytest_new.log | grep -v Warning | tail -30 | cut -c1-900
echo "== pose-graph + lie suites"
timeout 2000 python -m pytest -q -m gpu --tb=short -p no:cacheprovider tests/test_optim_gpu.py tests/test_pgo_trial_tail_gpu.py tests/test_lm_device_gpu.py \
   tests/test_determinism_gpu.py tests/test_pack_blocks_gpu.py tests/test_robust_gpu.py tests/test_pcg_p2p_gpu.py tests/test_distributed_gpu.py \
   tests/test_activate_gpu.py tests/test_lie_parity_gpu.py tests/test_reference_suite_gpu.py tests/test_ba_gpu.py 2>&1 | tee $O/pytest_pgo.log | grep -v "^\[W\|Gloo\|amdgpu.ids" | tail -12 | cut -c1-400
echo "== pcg timings"
for g in 1 0; do
  PPLIE_PCG_GAUGE=$g timeout 300 python tools/time_pcg_iter.py 2>&1 | tail -1 > $O/pcg_iter_gauge$g.json; cut -c1-1600 $O/pcg_iter_gauge$g.json; echo
  PPLIE_PCG_GAUGE=$g timeout 300 python tools/time_pcg2.py 2>&1 | tail -1 > $O/pcg2_gauge$g.json; cut -c1-400 $O/pcg2_gauge$g.json; echo
done
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err; tail -c 1700 $O/bench.json; echo
ls $O
