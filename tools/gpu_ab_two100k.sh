#!/bin/bash
# round-robin of the pose-graph LM step (100k / 400k) over libpplie_head.so (the previous build) and libpplie.so: 3 rounds
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for lib in libpplie_head.so libpplie.so; do
    printf "%-20s" "$lib"; PPLIE_LIBRARY_FILE=$lib python tools/pgo_loop.py 100000 400000 6 0 2>&1 | grep "^rep" | tail -3 | awk '{printf "%s ", $4} END {print $6,$7,$8}'
  done
done
