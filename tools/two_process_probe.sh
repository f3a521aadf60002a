#!/bin/bash
# Two PROCESSES, ROBOTS (default 450) x 16 each, on the one GPU of a box (2 x 500 + the two decider workgroups do not get onto one device
# together — the ranks' agreement declines those launches; on a node every rank has a GPU of its own): bench.py's direct-child role (peer-mapped stores + resident schedule
# launches, hipIpc), next to the single-world figure of the same 1000 robots.   bash tools/two_process_probe.sh
port=$((20000 + RANDOM % 20000))
[ -n "$MGX_TIMING_ON" ] && export MGX_TIMING=1
export HSA_ENABLE_IPC_MODE_LEGACY=0 MGX_HALO_TIMEOUT_MS=20000 MGX_BENCH_DEVICE=0 MGX_RESIDENT_CENSUS_SHARDED_US=500000
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$port timeout -k 10 200 python bench.py --role direct-child --gpus 2 \
    --steps 200 --warmup 40 --robots-per-gpu ${ROBOTS:-450} --horizon 16 > gpurun_out/two_proc_$r.json 2> gpurun_out/two_proc_$r.err &
  pids[$r]=$!
done
wait ${pids[0]}; wait ${pids[1]}
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/two_proc_0.json") if l.startswith("{")][-1])
print("two processes, robots each:", d.get("config", {}).get("robots_per_gpu", "?"), "-", d.get("transport"), "launches/tick", d.get("launches_per_tick"), "us per iteration", round(d["ms_per_step"] * 1e3, 2),
      "verified", d.get("verified_against_host_driven_exchange"), "resident launches / declined", d.get("resident_launches"), d.get("resident_declined"))
PY
timeout -k 10 100 python tools/quick_ir_bench.py $((2 * ${ROBOTS:-450})) 2>&1 | tail -1
