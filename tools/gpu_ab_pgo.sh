#!/bin/bash
# A/B of the pose-graph LM step (tools/pgo_loop.py, no profiler): tools/gpu_ab_pgo.sh "<env assignments of B>" [nodes edges]
cd "$(dirname "$0")/.."
N=${2:-10000}; E=${3:-40000}
for rep in 1 2; do
  echo "--- A (default)"; python tools/pgo_loop.py $N $E 8 0 2>&1 | grep "^rep" | tail -4
  echo "--- B ($1)"; env $1 python tools/pgo_loop.py $N $E 8 0 2>&1 | grep "^rep" | tail -4
done
