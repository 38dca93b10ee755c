#!/bin/bash
# Round 5, visit 2: the gauge preconditioner, broadcast cotangents, fp64 tiles -- tests, per-iteration timings, tuning sweeps, bench
set -u
cd "$(dirname "$0")/.."
R=$PWD
O=gpurun_out/r05_${1:-v2}
mkdir -p $O
export TMPDIR=/tmp
echo "== new tests"
timeout 1500 python -m pytest -q -m gpu --tb=short -p no:cacheprovider -s -x \
   tests/test_pcg_gauge_gpu.py "tests/test_lie_parity_gpu.py::test_broadcast_cotangent_variant_equals_the_materialised_launch" \
   "tests/test_lie_parity_gpu.py::test_sum_backward_takes_the_broadcast_route_and_equals_the_materialised_one" \
   tests/test_fullsize_parity_gpu.py 2>&1 | tee $O/pytest_new.log | grep -v Warning | tail -25 | cut -c1-700
echo "== pose-graph suites"
timeout 1500 python -m pytest -q -m gpu --tb=short -p no:cacheprovider tests/test_optim_gpu.py tests/test_pgo_trial_tail_gpu.py tests/test_lm_device_gpu.py \
   tests/test_determinism_gpu.py tests/test_pack_blocks_gpu.py tests/test_robust_gpu.py tests/test_pcg_p2p_gpu.py tests/test_distributed_gpu.py tests/test_activate_gpu.py 2>&1 | tee $O/pytest_pgo.log | tail -8 | cut -c1-400
echo "== pcg timings"
for g in 1 0; do
  PPLIE_PCG_GAUGE=$g timeout 300 python tools/time_pcg_iter.py 2>&1 | tail -1 > $O/pcg_iter_gauge$g.json; cut -c1-900 $O/pcg_iter_gauge$g.json
  PPLIE_PCG_GAUGE=$g timeout 300 python tools/time_pcg2.py 2>&1 | tail -1 > $O/pcg2_gauge$g.json; cut -c1-400 $O/pcg2_gauge$g.json
done
echo "== tune f64"; timeout 600 python tools/tune_general.py --f64 > $O/tune_f64.log 2>&1; cp gpurun_out/tune_general_f64.json $O/ 2>/dev/null; python - <<'P'
import json
try:
    r=json.load(open('gpurun_out/tune_general_f64.json'))
    best={}
    for x in r:
        k=x['op']
        if k not in best or x['ms']<best[k]['ms']: best[k]=x
    base={x['op']:x for x in r if x['block']==256 and x['rpt']==1}
    for k,v in best.items(): print(k,'best',v['block'],v['rpt'],round(v['ms'],4),'GB/s',round(v['GBps']),'| 256x1:',round(base[k]['ms'],4))
except Exception as e: print('tune f64:',e)
P
echo "== tune f32 (so3_exp, rxso3_mul)"; timeout 300 python tools/tune_general.py so3_exp_fwd rxso3_mul_fwd > $O/tune_f32.log 2>&1; cp gpurun_out/tune_general.json $O/tune_general_f32.json 2>/dev/null; grep -c op $O/tune_f32.log; python - <<'P'
import json
try:
    r=json.load(open('gpurun_out/tune_general.json'))
    for op in sorted({x['op'] for x in r}):
        rows=sorted([x for x in r if x['op']==op], key=lambda x:x['ms'])
        print(op, [(x['block'],x['rpt'],round(x['ms'],4)) for x in rows[:4]])
except Exception as e: print('tune f32:',e)
P
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err; tail -c 1500 $O/bench.json; echo
ls $O
