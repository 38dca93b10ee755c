#!/bin/bash
# round-robin timing of the pose-graph LM step with an environment switch on and off: tools/gpu_ab_env.sh VAR [nodes edges] (3 rounds)
cd "$(dirname "$0")/.."
V=$1; N=${2:-10000}; E=${3:-40000}
for round in 1 2 3; do
  for val in 1 0; do
    printf "%-34s" "$V=$val"; env $V=$val python tools/pgo_loop.py $N $E 8 0 2>&1 | grep "^rep" | tail -4 | awk '{printf "%s ", $4} END {print ""}'
  done
done
