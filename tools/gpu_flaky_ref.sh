#!/bin/bash
# how often does the reference's own tests/optim/test_optimizer.py fail on this box, and which tests?  (N runs per mode, the whole
# suite of files as tests/test_reference_suite_gpu.py runs them, so that the random streams are where they are there)
cd "$(dirname "$0")/.."; R=$PWD
FILES="lietensor/test_lietensor.py optim/test_optimizer.py optim/test_jacobian.py optim/test_solver.py optim/test_scheduler.py optim/test_sparse_lm.py basics/test_ops.py basics/test_func.py function/test_checking.py function/test_spline.py module/test_loss.py"
ARGS=""; for f in $FILES; do ARGS="$ARGS $R/oracle/_ref/tests/$f"; done
for mode in "" "--default-cuda"; do
  for i in $(seq 1 ${1:-6}); do
    (cd /tmp && PPLIE_QUIET_STAGING=1 timeout 600 python $R/tests/run_reference_tests.py $mode -rf -k "not test_sparse_lm_chain_pgo_runs_and_converges and not test_parameter_dispatch" $ARGS > /tmp/ref_run.log 2>&1; echo "rc=$? $(grep -E "^FAILED|too many|passed|failed" /tmp/ref_run.log | tr '\n' '|' | cut -c1-600) [mode=$mode run=$i]")
  done
done
