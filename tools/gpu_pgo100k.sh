python -m pytest tests/test_optim_gpu.py tests/test_lm_golden2_gpu.py -q 2>&1 | grep -E "passed|failed|Error|^E " | head
python tools/time_pgo.py 100000 400000 | head -1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/pgo100k -o p -- python $GRAFT_REPO_ROOT/tools/time_pgo.py 100000 400000 > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/gpurun_out/pgo100k/p_results.db 6
