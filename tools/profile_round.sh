#!/bin/bash
# Round profile on the GPU box (gpurun):  bash tools/profile_round.sh <tag> [part]
#   part a (default: all): kernel trace + stats of the bench command, the same under the DRIVER's command line, separate --pmc
#           passes (FETCH_SIZE / WRITE_SIZE / SQ counters) of the bench command with and without resident launches
#   part b: kernel stats and the same three --pmc passes of the other BASELINE configs (tools/bench_configs.py: K = 32 kernels)
# Everything lands under gpurun_out/<tag>_*; tools/profile_round.py turns it into profiles/<tag>_*.
# (the profiled program comes right after `--`: no env / shell hop between rocprofv3 and python3)
set -o pipefail
tag=${1:-r04}
part=${2:-ab}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
# (the driver's own block shape — 20 steps = 2 ticks, one mgx_iterate call each: the first launches, the second is posted into the
# lingering launch, the closing synchronisation ends it — repeated often: every dispatch of the timed blocks holds 20 iterations)
B="python3 bench.py --steps 20 --warmup 20 --repeats 60 --preheat-ms 20 --no-cpu-baseline --no-extras --sustained-seconds 0"
out=gpurun_out/${tag}
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
# what the VALU instructions ARE (f64 arithmetic by kind) and how many of a wave's 64 lanes they keep busy
F64="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"
if [[ $part == *a* ]]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_kt -- $B > ${out}_kt.json 2> ${out}_kt.err || exit 1
  echo "kernel trace done"
  # the driver's own command line (bench.py's defaults otherwise): the per-dispatch durations behind the round's bench line
  rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_kt_driver -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > ${out}_kt_driver.json 2> ${out}_kt_driver.err || exit 1
  echo "driver command done"
  # (GRBM_GUI_ACTIVE: shader-clock cycles per dispatch — with the dispatch's duration the clock the box ran at)
  for pass in "FETCH_SIZE" "WRITE_SIZE" "$SQ" "$F64" "GRBM_GUI_ACTIVE"; do
    name=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --pmc $pass --kernel-trace --output-format csv -d ${out}_pmc_${name} -- $B > ${out}_pmc_${name}.json 2> ${out}_pmc_${name}.err || exit 1
    export MGX_PERSISTENT=0
    rocprofv3 --pmc $pass --kernel-trace --output-format csv -d ${out}_pmc_${name}_np -- $B --no-configs1 > ${out}_pmc_${name}_np.json 2> ${out}_pmc_${name}_np.err || exit 1
    unset MGX_PERSISTENT
    echo "pmc $name done"
  done
  export MGX_PERSISTENT=0
  rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_kt_np -- $B --no-configs1 > ${out}_kt_np.json 2> ${out}_kt_np.err || exit 1
  unset MGX_PERSISTENT
fi
if [[ $part == *b* ]]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_cfg -- python3 tools/bench_configs.py > ${out}_cfg.log 2> ${out}_cfg.err || exit 1
  for pass in "FETCH_SIZE" "WRITE_SIZE" "$SQ"; do
    name=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --pmc $pass --kernel-trace --output-format csv -d ${out}_cfg_pmc_${name} -- python3 tools/bench_configs.py > ${out}_cfg_pmc_${name}.log 2> ${out}_cfg_pmc_${name}.err || exit 1
    echo "configs pmc $name done"
  done
  echo "configs done"
fi
# the raw traces are tens of megabytes per pass: what travels back is the summary made HERE
python3 tools/profile_round.py ${tag} gpurun_out/${tag}_profiles > gpurun_out/${tag}_profile_summary.log 2>&1 || exit 1
rm -rf ${out}_kt ${out}_kt_np ${out}_kt_driver ${out}_pmc_* ${out}_cfg ${out}_cfg_pmc_*
echo "summary written to gpurun_out/${tag}_profiles"

