#!/bin/bash
# A/B of the dynamic tick inside one gpurun call:  bash tools/ab_dynamic.sh [reps] variant ...   (variants as in tools/ab.sh)
reps=${1:-3}; shift 1
for i in $(seq $reps); do
  for v in "$@"; do
    lib=${v%%:*}; kv=""; [[ $v == *:* ]] && kv=${v#*:}
    (
      if [ "$lib" != product ]; then export MGX_LIB=$lib; fi
      if [ -n "$kv" ]; then export "$kv"; fi
      timeout -k 10 90 python tools/dynamic_tick_breakdown.py 2>&1 | tail -1 | sed "s|^|[$v] |"
    ) || exit 1
  done
done
