#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_scan_grad_gpu.py tests/test_imu_gpu.py tests/test_activate_module_gpu.py tests/test_determinism_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -2
for leg in imu imu_train scan_bwd; do bash tools/gpu_prof_leg.sh v7 $leg 2>&1 | grep "pplie::\(imu\|scan\)" | awk -F'",' '{split($1,a,"("); print substr(a[1],1,80), $2}' | cut -c1-150; done
timeout 300 python tools/time_imu_cov.py 4096 | tail -1
