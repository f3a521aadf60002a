#!/bin/bash
# A/B inside one gpurun call:  bash tools/ab.sh [reps] [robots] variant ...
#   variant = "product" | path of a library | either one followed by ":VAR=VALUE" (an environment variable for that run)
# runs tools/quick_ir_bench.py alternately with each variant (box-to-box differences are as large as most kernel changes)
reps=${1:-3}; n=${2:-1000}; shift 2
for i in $(seq $reps); do
  for v in "$@"; do
    lib=${v%%:*}; kv=""; [[ $v == *:* ]] && kv=${v#*:}
    (
      if [ "$lib" != product ]; then export MGX_LIB=$lib; fi
      if [ -n "$kv" ]; then export "$kv"; fi
      timeout -k 10 90 python tools/quick_ir_bench.py $n 2>&1 | tail -1 | sed "s|^|[$v] |"
    ) || exit 1
  done
done
