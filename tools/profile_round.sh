#!/bin/bash
# Round profile on the GPU box (gpurun):  bash tools/profile_round.sh <tag>
# kernel trace + stats of the default bench command, separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ counters) of the
# same command with and without resident launches, kernel stats of the other BASELINE configs (tools/bench_configs.py).
# Everything lands under gpurun_out/<tag>_*; tools/profile_round.py turns it into profiles/<tag>_*.
set -o pipefail
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-extras"
out=gpurun_out/${tag}
rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_kt -- $B > ${out}_kt.json 2> ${out}_kt.err || exit 1
echo "kernel trace done"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d ${out}_pmc_${name} -- $B > ${out}_pmc_${name}.json 2> ${out}_pmc_${name}.err || exit 1
  MGX_PERSISTENT=0 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d ${out}_pmc_${name}_np -- $B --no-configs1 > ${out}_pmc_${name}_np.json 2> ${out}_pmc_${name}_np.err || exit 1
  echo "pmc $name done"
done
MGX_PERSISTENT=0 rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_kt_np -- $B --no-configs1 > ${out}_kt_np.json 2> ${out}_kt_np.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_cfg -- python3 tools/bench_configs.py > ${out}_cfg.log 2> ${out}_cfg.err || exit 1
echo "configs done"
