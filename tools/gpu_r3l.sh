#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3l
export TMPDIR=/tmp
echo "== pytest (all gpu)"
timeout 2400 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tee gpurun_out/r3l/pytest_full.log | tail -12 | cut -c1-300
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3l/bench.json 2> gpurun_out/r3l/bench.err; tail -c 300 gpurun_out/r3l/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3l/bench.json'))
for k in ('value','ms_per_step'): print(k, d[k])
print('roofline', d['roofline']['frac'])
for leg in ('c1','lm_invnet','lm_pgo','lm_pgo_100k','imu','ba_reproj'):
    v=d.get(leg,{})
    print(leg, {k:v.get(k) for k in ('value','static_model_value','error') if k in v}, (v.get('roofline') or {}).get('frac'), 'cpu:', (v.get('cpu_baseline') or {}).get('value'))
PY
