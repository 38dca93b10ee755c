#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3i
export TMPDIR=/tmp
echo "== pytest"
timeout 1500 python -m pytest tests/test_lm_device_gpu.py tests/test_optim_gpu.py tests/test_fullsize_parity_gpu.py tests/test_examples_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tee gpurun_out/r3i/pytest_full.log | tail -12 | cut -c1-300
echo "== time_pgo default"; timeout 200 python tools/time_pgo.py 2>&1 | head -3 | tee gpurun_out/r3i/time_pgo.log
echo "== time_pgo static"; timeout 200 python tools/time_pgo.py 10000 40000 static 2>&1 | head -2 | tee gpurun_out/r3i/time_pgo_static.log
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3i/bench.json 2> gpurun_out/r3i/bench.err; tail -c 400 gpurun_out/r3i/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3i/bench.json'))
for k in ('value','ms_per_step'): print(k, d[k])
print('roofline', d['roofline']['frac'])
for leg in ('c1','lm_invnet','lm_pgo','lm_pgo_100k','imu','ba_reproj'):
    v=d.get(leg,{})
    print(leg, {k:v.get(k) for k in ('value','static_model_value','error') if k in v}, (v.get('roofline') or {}).get('frac'), 'cpu:', (v.get('cpu_baseline') or {}).get('value'))
print('cpu_baseline', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('cores'))
PY
