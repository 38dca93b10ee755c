#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest ${@:-tests} -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
