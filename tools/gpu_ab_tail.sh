#!/bin/bash
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for cfg in "head 0 libpplie_head.so" "new_off 0 libpplie.so" "new_on 1 libpplie.so"; do
    set -- $cfg
    printf "%-10s" "$1"; PPLIE_PCG_TAIL_IN_SOLVE=$2 PPLIE_LIBRARY_FILE=$3 python tools/pgo_loop.py 10000 40000 8 0 2>&1 | grep "^rep" | tail -4 | awk '{printf "%s ", $4} END {print ""}'
  done
done
