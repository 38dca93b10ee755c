#!/bin/bash
# round-robin timing of the pose-graph LM step over several builds of the library: tools/gpu_ab_lib.sh lib1.so lib2.so ... (3 rounds)
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for lib in "$@"; do
    printf "%-28s" "$lib"; PPLIE_LIBRARY_FILE=$lib python tools/pgo_loop.py 10000 40000 8 0 2>&1 | grep "^rep" | tail -4 | awk '{printf "%s ", $4} END {print ""}'
  done
done
