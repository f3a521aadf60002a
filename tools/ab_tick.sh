#!/bin/bash
# A/B inside one gpurun call:  bash tools/ab_tick.sh [reps] [robots] variant ...   (variants as in tools/ab.sh; runs tools/quick_tick_bench.py)
reps=${1:-3}; n=${2:-1000}; shift 2
for i in $(seq $reps); do
  for v in "$@"; do
    lib=${v%%:*}; kv=""; [[ $v == *:* ]] && kv=${v#*:}
    (
      if [ "$lib" != product ]; then export MGX_LIB=$lib; fi
      if [ -n "$kv" ]; then export "$kv"; fi
      timeout -k 10 120 python tools/quick_tick_bench.py $n 2>&1 | tail -1 | sed "s|^|[$v] |"
    ) || exit 1
  done
done
