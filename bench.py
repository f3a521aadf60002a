"""bench.py — GBP iterations/s of the MI355X engine on BASELINE.json's synthetic graphs.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE GBP iteration (one schedule step: internal factor + variable sweep, external factor sweep ->
routing -> external variable sweep -> routing) over 1000 robots x 16 horizon per GPU.  Steps are
issued the way the reference's driver issues them: one `mgx_iterate` call per tick's schedule (10 steps,
Junction-Twoway 10/10), inputs already resident in HBM, priors frozen.

Headline (`value`) — the workload BASELINE.json's north star scales: dynamics + obstacle + INTER-ROBOT factors
(comm radius 8).  N = 1: configs[2], 1000 robots x 16 on one GPU.  N > 1: configs[3]'s layout, 1000 * N robots
sharded N ways in (y, x) strips with ONE exchange of boundary snapshot records per external iteration (RCCL
all-to-all-v over xGMI, or the in-engine transports), weak scaling: `value` = N * steps / time, i.e. iterations
of 1000 robots' worth of graph per second, the same unit at every N.
`configs1`: BASELINE configs[1] (the same robots without inter-robot factors; at N > 1 independent shards, no
exchange), with its own roofline and CPU baseline.

Timing protocol (declared in the line): W warm-up steps, a clock pre-heat of `preheat_ms`, then `repeats`
repetitions of EXACTLY K steps, each bracketed by barrier + synchronize, MAX over ranks per repetition; `value`
comes from the MEDIAN repetition (`ms_per_step`), the spread is reported beside it.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s peak, ~6.3 achievable)
# FP64 vector peak: 256 CUs x 4 SIMDs x 32 lanes x 2 (FMA) x 2.4 GHz / 2 (f64 issues at half the f32 rate) = 78.6 TFLOP/s
# = half the guide's 157.3 TFLOP/s FP32 vector figure; as an issue rate: one f64 VALU wave-instruction per SIMD per 4 clocks
F64_VALU_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 4.0
SCHEDULE_LEN = 10
PROFILE_TAG = "r05"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--repeats", type=int, default=15, help="repetitions of the timed K-step block (median reported)")
    ap.add_argument("--preheat-ms", type=float, default=60.0, help="untimed work in front of the repetitions (clock ramp)")
    ap.add_argument("--robots-per-gpu", type=int, default=1000)
    ap.add_argument("--horizon", type=int, default=16)
    ap.add_argument("--ticks-per-submission", type=int, default=0,
                    help="1 = the reference's call pattern: one mgx_iterate per tick, nothing bracketed — schedules issued back to back ride "
                         "in one lingering launch by themselves (the default at N = 1); > 1: that many ticks handed over inside "
                         "mgx_batch_begin / _end (the default at N > 1 is 2: the launches of a sharded world do not linger)")
    ap.add_argument("--no-configs1", action="store_true", help="skip the BASELINE configs[1] block")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="N = 1: skip the tick / dynamic_tick / scenario measurements")
    ap.add_argument("--no-children", action="store_true", help="N > 1: skip the in-engine transports (measured in child processes)")
    ap.add_argument("--deadline", type=float, default=150.0,
                    help="N > 1 only: seconds a phase with collectives may take before the run is abandoned (non-zero exit)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--sustained-seconds", type=float, default=10.0,
                    help="length of the sustained run of the headline workload behind the repetitions (0: skip)")
    ap.add_argument("--fma", action="store_true", help="use the FMA-contracting build (not the product)")
    ap.add_argument("--no-resident", action="store_true", help="N > 1: keep the direct transport on its push / wait kernels (no resident launches)")
    ap.add_argument("--role", default="main", choices=["main", "direct-child", "rccl-child"],
                    help="*-child: one rank of an isolated measurement of the sharded workload with the exchange inside the "
                         "engine (direct: peer-mapped stores; rccl: grouped ncclSend / ncclRecv), spawned by the main role")
    return ap.parse_args()


def run_steps(iterate, n, steps_one_tick, batch=None, group=1):
    """issue exactly n iterations as ticks of len(steps_one_tick) plus a remainder.  batch (a callable that gives a context
    manager: World.batch / ShardedWorld.batch) with group > 1: the ticks are handed over `group` at a time inside
    mgx_batch_begin / mgx_batch_end — the engine submits each group together, merged into one resident launch where its
    segments fit — every tick still its own mgx_iterate call, nothing skipped, results bit-identical (tests/test_gpu_batch.py).
    Returns the sweep-kernel launches the groups were submitted as (None: not batched)."""
    full, rem = divmod(n, len(steps_one_tick))
    if batch is None or group <= 1:
        for _ in range(full):
            iterate(steps_one_tick)
        if rem:
            iterate(steps_one_tick[:rem])
        return None
    launches, t = 0, 0
    while t < full or (rem and t == 0 and full == 0):
        g = min(group, full - t)
        with batch() as b:
            for _ in range(g):
                iterate(steps_one_tick)
            if rem and t + g >= full:  # the remainder rides with the last group
                iterate(steps_one_tick[:rem])
                rem = 0
        launches += getattr(b, "launches", 0) or 0
        t += max(g, 1)
    return launches


def timed(torch, dist, iterate, steps_one_tick, a, multi, red_dev="cuda", sync=None, batch=None, group=1, flush=None):
    """The contract's measurement, repeated: W warm-up steps, a pre-heat, then `repeats` x [barrier + synchronize,
    EXACTLY K steps, barrier + synchronize].  Returns per-repetition (wall seconds MAX over ranks, device seconds
    between HIP events on the launch stream MAX over ranks).
    sync: the world's own synchronize (mgx_synchronize: a launch that lingers for the next schedule is told to end first — waiting
    on the stream by other means would wait out its bound); flush: the same without the wait, in front of the closing event."""
    sync = sync or torch.cuda.synchronize
    flush = flush or (lambda: None)
    run_steps(iterate, a.warmup, steps_one_tick, batch, group)
    sync()
    t_end = time.perf_counter() + a.preheat_ms * 1e-3
    while time.perf_counter() < t_end:  # the same work, untimed: clocks ramp up in the first tens of milliseconds
        run_steps(iterate, max(a.steps, SCHEDULE_LEN), steps_one_tick, batch, group)
        sync()
    walls, devs = [], []
    for _ in range(a.repeats):  # wall clock: nothing but the K steps between the two barrier + synchronize pairs
        if multi:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        run_steps(iterate, a.steps, steps_one_tick, batch, group)
        if multi:
            dist.barrier()
        sync()
        walls.append(time.perf_counter() - t0)
    for _ in range(a.repeats):  # device time of the same block: HIP events on the launch stream (own repetitions: recording them costs host time)
        sync()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        run_steps(iterate, a.steps, steps_one_tick, batch, group)
        flush()
        ev1.record()
        sync()
        devs.append(ev0.elapsed_time(ev1) * 1e-3)
    if multi:
        t = torch.tensor([walls, devs], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        walls, devs = [float(x) for x in t[0]], [float(x) for x in t[1]]
    return walls, devs


def sustained(torch, dist, iterate, steps_one_tick, seconds, ms_per_step, multi, red_dev="cuda", sync=None, units_per_step=1.0, batch=None, group=1):
    """The headline workload for `seconds` of wall clock in ONE timed block (the repetitions above are a fraction of a
    millisecond each: clocks, caches and the host's launch queue in a steady state are a different regime).  The number of
    steps is fixed in advance from the measured step time — the same on every rank — and the block is bracketed like every
    other: barrier + synchronize on both sides, MAX over ranks."""
    sync = sync or torch.cuda.synchronize
    n = max(len(steps_one_tick), int(seconds / max(ms_per_step * 1e-3, 1e-9)))
    n -= n % len(steps_one_tick)
    if multi:
        t = torch.tensor([n], dtype=torch.int64, device=red_dev)
        dist.broadcast(t, src=0)
        n = int(t[0])
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    run_steps(iterate, n, steps_one_tick, batch, group)
    if multi:
        dist.barrier()
    sync()
    wall = time.perf_counter() - t0
    if multi:
        t = torch.tensor([wall], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t[0])
    return {"seconds": round(wall, 3), "steps": n, "value": round(units_per_step * n / wall, 2), "ms_per_step": wall / n * 1e3,
            "what": "one timed block of the headline workload, barrier + synchronize on both sides, MAX over ranks"}


def settle_resident(sw, ref, steps_one_tick, agree, max_ticks=40):
    """Ticks in front of the verification until the ranks' resident launches run: the very first launches of a process load code
    objects and the ranks' hosts reach them tens of milliseconds apart — the ranks' agreement then declines (by design) and the
    engines back off for a number of external iterations, in which nothing resident is even tried.  Judging the transport by
    those ticks would switch it off on every real node.  `ref` (the host-driven twin, if any) runs the same ticks.
    Returns (ticks run, declined launches during them)."""
    n = 0
    while n < max_ticks:
        sw.iterate(steps_one_tick)
        if ref is not None:
            ref.iterate(steps_one_tick)
        sw.synchronize()
        n += 1
        st = sw.world.resident_stats()
        if n >= 2 and agree(int(st[2]) == 0 and sw.world.last_launch_count() == 1):
            break
    return n, int(sw.world.resident_stats()[1])


def summary(walls, devs, steps, units_per_step=1.0):
    """median repetition -> throughput; spread of the repetitions beside it"""
    w, d = statistics.median(walls), statistics.median(devs)
    return {"value": units_per_step * steps / w, "ms_per_step": w / steps * 1e3, "device_ms_per_step": d / steps * 1e3,
            "wall_s_median": w, "device_s_median": d,
            "spread": {"repeats": len(walls), "ms_per_step_min": min(walls) / steps * 1e3, "ms_per_step_max": max(walls) / steps * 1e3}}


def cpu_baseline(sc, seconds, check=None):
    """The oracle (a from-scratch port of the reference's CPU path, oracle/gbp_oracle.c) timed on
    this box's host cores on the same workload, bounded to ~`seconds` of CPU work.  Robots run in
    parallel across threads for the internal sweeps, external phase serial (robot.rs:1789-1859).
    check = (ticks, beliefs): the beliefs (eta, lam, mean) a FRESH engine world held after `ticks` ticks of the schedule issued
    exactly as the timed block issues them; a fresh oracle world runs the same ticks (outside every timed region) and the
    result — identical bit for bit or not — is returned under "parity" (BASELINE.md §3)."""
    import oracle
    from magics_amd import scenarios as S
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    lib_path = None
    try:
        import tempfile
        lib_path = oracle.build(native=True, out_dir=tempfile.mkdtemp(prefix="orc_native_"))
    except Exception:
        lib_path = None  # fall back to the prebuilt x86-64-v3 library
    # thread counts tried (the reference runs robots on Bevy's compute pool, a subset of the cores):
    # the fastest one is reported, with the count actually used
    cands = sorted({1, min(avail, 8), min(avail, 16), min(avail, 32), min(avail, 64)})
    per = max(1.0, seconds / len(cands))
    res = {}
    for threads in cands:
        w = oracle.OracleWorld(sc["params"], threads=threads, lib_path=lib_path)
        S.populate(w, sc)
        w.iterate(sc["steps"])  # warm-up tick
        n, t0 = 0, time.perf_counter()
        while True:
            w.iterate(sc["steps"])
            n += len(sc["steps"])
            el = time.perf_counter() - t0
            if el >= per or n >= 20000:
                break
        res[threads] = (n / el, n)
        w.close()
    best = max(res, key=lambda t: res[t][0])
    parity = None
    if check is not None:
        import numpy as np
        ticks, got = check
        w = oracle.OracleWorld(sc["params"], threads=best, lib_path=lib_path)
        S.populate(w, sc)
        for _ in range(ticks):
            w.iterate(sc["steps"])
        want = w.read_beliefs()
        w.close()
        same = [bool(np.array_equal(x, y)) for x, y in zip(got, want)]
        parity = {"checked": True, "identical": all(same), "finite": bool(all(np.isfinite(x).all() for x in got)),
                  "steps": ticks * len(sc["steps"]), "entries": int(sum(x.size for x in got)),
                  "max_abs_diff": float(max(np.nanmax(np.abs(x - y)) for x, y in zip(got, want))),
                  "what": f"a fresh world ran the timed block's exact submission ({ticks} tick(s) of the {len(sc['steps'])}-step schedule, "
                          "batched as in the timed block) and a fresh CPU oracle the same steps, outside the timed region: belief eta, lam "
                          "and mean of every variable compared bit for bit"}
    return {
        "parity": parity,
        "value": round(res[best][0], 2), "unit": "GBP iterations/s", "cores": best, "kind": "port",
        "sample": f"{sc['name']}: {res[best][1]} iterations on {best} of {avail} available host threads (robots parallel in "
                  f"internal sweeps, external phase serial), -O3 -march={'native' if lib_path else 'x86-64-v3'}; "
                  f"fastest of thread counts {cands}",
        "single_thread_value": round(res[1][0], 2),
        "by_threads": {str(t): round(v[0], 2) for t, v in res.items()},
    }


def kernel_digest():
    """content hash of what the sweep kernel is compiled from: a counter profile belongs to ONE such state"""
    import hashlib
    h = hashlib.sha256()
    for f in ("mgx_sweep.h", "gbp_math.h", "mgx_dev.h", "mgx_sweep_inst.hip"):
        with open(os.path.join(ROOT, "magics_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:20]


def profiled(key):
    """What rocprofv3 measured for THIS build of the sweep kernel (profiles/traffic_<tag>.json, written by tools/profile_round.py
    from separate --pmc passes of this command): HBM bytes per dispatch (FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950
    correction + WRITE_SIZE) and the SQ instruction counters.  (None, reason) when there is no such file, no such key, or the
    file was made from other kernel sources than the ones this library was built from — stale counters are not quoted."""
    path = os.path.join(ROOT, "profiles", f"traffic_{PROFILE_TAG}.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except Exception:  # noqa: BLE001
        return None, f"profiles/traffic_{PROFILE_TAG}.json is absent"
    if d.get("_kernel_digest") != kernel_digest():
        return None, (f"profiles/traffic_{PROFILE_TAG}.json was measured on other kernel sources (digest {d.get('_kernel_digest')}) than this "
                      f"build's ({kernel_digest()}): re-run tools/profile_round.sh")
    if key not in d:
        return None, f"no counters for '{key}' in profiles/traffic_{PROFILE_TAG}.json"
    return d[key], ""


def roofline(kernel, alg_bytes_per_launch, launches, dev_seconds, prof_key, iterations_per_launch, prof_note=""):
    """SURVEY §8(d)'s figure at the top level — ALGORITHMIC bytes per launch (every live message / belief read and written once
    per sweep: the survey's per-robot figure x robots x iterations of the launch) / the kernel's live average launch duration (HIP
    events on the launch stream), against 8 TB/s: `achieved`, `peak`, `frac`.  It exceeds 1 where the graphs stay in LDS across the
    iterations of a launch: a WORK RATE in the survey's byte model then, not a bandwidth — which is what `physical` is for: what
    rocprofv3's counters measured for this build of the kernel (profiles/, separate --pmc passes), scaled per iteration to this
    run's launches:
      physical.hbm   FETCH_SIZE (doubled per MI355X_MICROARCH.md's gfx950 correction) + WRITE_SIZE against 8 TB/s
      physical.valu  SQ_INSTS_VALU counts EVERY VALU wave-instruction; SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 the f64 ones;
                     SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64) the share of a wave's 64 lanes that work; from those the
                     useful f64 FLOP/s against the 78.6 TFLOP/s vector peak, and the issue time of the instruction mix (an f64
                     wave-instruction holds its SIMD 4 clocks, any other VALU one 2)."""
    avg = dev_seconds / launches
    ach = alg_bytes_per_launch / avg / 1e9
    out = {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
           "traffic": None, "avg_launch_us": round(avg * 1e6, 3), "iterations_per_launch": iterations_per_launch,
           "algorithmic_bytes_per_launch": alg_bytes_per_launch,
           "note": "SURVEY §8d: algorithmic bytes per launch / live launch duration against the HBM peak.  Above 1 this is a work rate in the "
                   "survey's byte model, not a bandwidth (the graphs stay in LDS across the iterations of a launch: `traffic` is what HBM "
                   "really moved); `physical` holds the measured fractions of physical peaks"}
    p, why = profiled(prof_key) if prof_key else (None, "no counter profile for this shape")
    if not p:
        out["physical"] = {"note": "no counters: " + why}
        return out
    ipd = float(p.get("iterations_per_dispatch") or (1 if prof_key == "config2" else SCHEDULE_LEN))
    scale = iterations_per_launch / ipd  # counters are per dispatch of `ipd` iterations: per iteration, times this run's
    traffic = int((2 * p["fetch_kib"] + p["write_kib"]) * 1024 * scale)
    t_hbm = traffic / (HBM_PEAK_GBS * 1e9)
    out["traffic"] = traffic
    phys = {"hbm": {"bytes_per_launch": traffic, "gbs": round(traffic / avg / 1e9, 1), "peak_gbs": HBM_PEAK_GBS, "frac": round(t_hbm / avg, 4),
                    "traffic_over_algorithmic": round(traffic / alg_bytes_per_launch, 4)}}
    if p.get("valu_wave_instr"):
        n_all = p["valu_wave_instr"] * scale
        v = {"valu_wave_instr_per_launch": round(n_all), "g_valu_wave_instr_per_s": round(n_all / avg / 1e9, 2),
             "valu_active_pct_of_wave_cycles": p.get("valu_busy_pct"), "wait_pct_of_wave_cycles": p.get("wait_pct"),
             "note": "SQ_INSTS_VALU: EVERY VALU wave-instruction (f64 arithmetic, moves, selects, integer address work, lane reads / writes "
                     "of spilled scalars)"}
        if p.get("f64_wave_instr"):
            n64 = p["f64_wave_instr"] * scale
            occ = p.get("lane_occupancy") or 1.0
            flops = p.get("f64_flops_per_wave_instr", 1.0) * n64 * 64.0 * occ  # (FMA counts 2)
            t_issue = (4.0 * n64 + 2.0 * (n_all - n64)) / (1024 * 2.4e9)
            v.update({"f64_wave_instr_per_launch": round(n64), "f64_frac": round(n64 / n_all, 4), "lane_occupancy": round(occ, 4),
                      "f64_tflops": round(flops / avg / 1e12, 3), "f64_peak_tflops": 78.6, "f64_frac_of_peak": round(flops / avg / 78.6e12, 4),
                      "issue_frac": round(t_issue / avg, 4),
                      "issue_model": "an f64 VALU wave-instruction holds its SIMD 4 clocks, any other VALU wave-instruction 2; 1024 SIMDs x 2.4 GHz"})
        else:
            v["issue_frac_if_all_were_f64"] = round(n_all / F64_VALU_WAVE_INSTR_PER_S / avg, 4)
        phys["valu"] = v
    phys["source"] = p.get("source", f"profiles/{PROFILE_TAG}_*: rocprofv3 --pmc, separate passes") + (f" — {prof_note}" if prof_note else "")
    phys["counters_per_dispatch_of_iterations"] = ipd
    # counters per iteration are duration-independent; the profile box's own duration and clock are quoted beside THIS run's
    phys["profile_box"] = {"kernel_trace_avg_us": p.get("kernel_trace_avg_us"), "shader_clock_ghz": p.get("shader_clock_ghz"),
                           "counter_pass_avg_us": p.get("counter_pass_avg_us"),
                           "note": "the box the counters were collected on (profiles/); `avg_launch_us` above is this run's own, live"}
    phys["note"] = ("neither HBM nor VALU issue bounds the launch: two waves per SIMD, each waiting on its own dependent chain (LDS round "
                    "trips, f64 latency, the hand-off between neighbouring workgroups) — see wait_pct_of_wave_cycles")
    out["physical"] = phys
    return out


# ---- N > 1: the sharded workload with the exchange INSIDE the engine, one rank per child process ------------------
def transport_child(a):
    """One rank of the sharded inter-robot workload with the exchange inside the engine: DIRECT (peer-mapped
    stores over xGMI + device-side arrival counters, include/mgx.h) or RCCL grouped send / recv enqueued by
    mgx_iterate.  Runs as a child process of each bench rank so that nothing it does can take the collective
    measurement down.  Control plane: gloo."""
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world_size = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ.get("MGX_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    dist.init_process_group("gloo")
    from magics_amd import World, scenarios as S, sharded
    stream = torch.cuda.current_stream().cuda_stream
    n_loc, K = a.robots_per_gpu, a.horizon
    n_tot = n_loc * world_size

    def agree(ok):  # every rank succeeded so far?
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t[0]) == 1

    out, sw, err, got = {}, None, "", None
    try:
        sc2 = S.grid_scenario(n_tot, K, interrobot=True, seed=805)
        comm = sharded.TorchDistComm()
        sw = sharded.ShardedWorld(sc2, rank, world_size, lambda p: World(p, stream=stream, fma=a.fma), comm=comm)
        sw.world.sweep(0, 0, 0)
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    if not agree(not err):
        out = {"error": err or "another rank failed to set up"}
    elif a.role == "direct-child":
        # peer-mapped stores first, and on top of them resident schedule launches (ghost records travel inside ONE launch per
        # schedule and rank); sharded.connect makes the ranks agree after every step
        try:
            got = sharded.connect(sw, comm, "direct", resident=not a.no_resident)
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        if not agree(not err and got is not None and got.startswith("direct")):
            out = {"error": err or f"the direct transport is not available (got {got})"}
    else:
        try:
            sharded.connect_rccl(sw, comm)  # collective (ncclCommInitRank); a failure on one rank ends in the deadline
            got = "rccl"
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        if not agree(not err):
            out = {"error": err or "another rank failed to join the RCCL communicator"}
    if not out:
        def sync():
            sw.synchronize()  # mgx_synchronize: raises if a wait on the device gave up
        verified = None
        declined_note = ""
        try:
            # the same two ticks on a world whose exchange is driven from the host (pack / all-to-all-v over the control plane /
            # unpack): the in-engine transport has to leave bit-identical beliefs
            ref = sharded.ShardedWorld(sc2, rank, world_size, lambda p: World(p, stream=stream, fma=a.fma),
                                       comm=sharded.TorchDistComm(stage_through_host=True))
            settled, declined0 = (settle_resident(sw, ref, sc2["steps"], agree) if got == "direct+resident" else (0, 0))
            for _ in range(2):
                sw.iterate(sc2["steps"])
                ref.iterate(sc2["steps"])
            sync()
            ref.synchronize()
            same = all(np.array_equal(x, y) for x, y in zip(sw.read_beliefs()[1:], ref.read_beliefs()[1:]))
            verified = agree(same)
            del ref
            # resident launches the ranks could not agree on (a rank late, crowded out, the agreement word out of reach) cost
            # a wait each before they fall back: if the two ticks above (behind the settling ticks) saw any, the measurement is the
            # plain direct transport's
            if got == "direct+resident" and not agree(sw.world.resident_stats()[1] == declined0):
                sw.world.set_resident_launches(False)
                got = "direct"
                declined_note = "resident launches were declined by the ranks' agreement during verification: switched off"
            walls, devs = timed(torch, dist, sw.iterate, sc2["steps"], a, True, red_dev="cpu", sync=sync)
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
            walls = devs = None
        if not agree(not err):
            out = {"error": err or "another rank timed out"}
        else:
            res = summary(walls, devs, a.steps, units_per_step=world_size)
            sw.iterate(sc2["steps"])
            n_launch = sw.world.last_launch_count()
            n_ex = sw.world.halo_direct_status() if a.role == "direct-child" else None  # exchanges run; raises if one timed out
            out = {"value": round(res["value"], 2), "transport": got, "verified_against_host_driven_exchange": verified,
                   "launches_per_tick": n_launch, "exchanges": n_ex, "ms_per_step": res["ms_per_step"],
                   "device_ms_per_step": res["device_ms_per_step"], "spread": res["spread"],
                   "exchange": ("direct+resident: ONE launch per schedule and rank; boundary robots store their snapshot records and progress "
                                "words into the other ranks' peer-mapped ghost areas from inside it (hipIpc, system-scope stores), no exchange "
                                "kernel, no collective" if got == "direct+resident" else
                                "direct: peer-mapped stores into the consumers' receive areas (hipIpc) + device-side arrival "
                                "counters, one C call per tick, no collective") if a.role == "direct-child" else
                               "RCCL inside the engine: grouped ncclSend / ncclRecv per external iteration on the launch stream, "
                               "one C call per tick",
                   "ghost_robots_this_rank": len(sw.plan.ghosts)}
            if a.role == "direct-child":
                out["settling_ticks"], out["declined_while_settling"] = settled, declined0
            if got.startswith("direct"):
                st = sw.world.resident_stats()
                out["resident_launches"], out["resident_declined"] = int(st[0]), int(st[1])
            if declined_note:
                out["note"] = declined_note
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.barrier()
    if sw is not None:
        sw.direct_close()
    dist.destroy_process_group()


def run_children(a, rank, local_rank, world_size, role, port_shift):
    """Spawn this rank's child for an in-engine transport and wait for it (bounded).  Returns the child's JSON
    (rank 0) or a description of what went wrong; never raises."""
    # the children rendezvous among themselves: nothing of the launcher's elastic agent may leak in
    # (with TORCHELASTIC_USE_AGENT_STORE set, rank 0 would not host the store on the new port)
    base = {k: v for k, v in os.environ.items() if not k.startswith(("TORCHELASTIC_", "GROUP_", "ROLE_"))}
    env = dict(base, RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world_size),
               MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"),
               MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + port_shift),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.setdefault("MGX_HALO_TIMEOUT_MS", "2000")
    env.setdefault("MGX_RESIDENT_TIMEOUT_MS", "2000")
    cmd = [sys.executable, os.path.abspath(__file__), "--role", role, "--gpus", str(world_size), "--steps", str(a.steps),
           "--warmup", str(a.warmup), "--repeats", str(a.repeats), "--preheat-ms", str(a.preheat_ms),
           "--robots-per-gpu", str(a.robots_per_gpu), "--horizon", str(a.horizon)] + (["--fma"] if a.fma else []) + (
               ["--no-resident"] if a.no_resident else [])
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=a.deadline)
    except subprocess.TimeoutExpired:
        return {"error": f"{role} did not finish within {a.deadline} s"}
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}
    if rank != 0:
        return None
    for ln in reversed(r.stdout.decode(errors="replace").splitlines()):
        ln = ln.strip()
        if ln.startswith("{"):
            try:
                return json.loads(ln)
            except ValueError:
                pass
    return {"error": f"child exit code {r.returncode}: {r.stderr.decode(errors='replace')[-400:]}"}


class _Deadline:
    """A phase with collectives that overruns ends the process from a timer thread: rank 0 prints what it has (the
    line without a headline value, marked as failed) and EVERY rank exits non-zero — a stalled collective on a
    process that has used the GPU must never be reported as success."""

    def __init__(self, seconds, rank, line):
        import threading
        self._t = None
        if seconds > 0:
            self._t = threading.Timer(seconds, self._expire, args=(seconds, rank, line))
            self._t.daemon = True
            self._t.start()

    @staticmethod
    def _expire(seconds, rank, line):
        if rank == 0:
            out = dict(line)
            out["error"] = f"the sharded phase did not finish within {seconds} s: no headline value"
            print(json.dumps(out), flush=True)
        os._exit(3)

    def cancel(self):
        if self._t is not None:
            self._t.cancel()


def scenario_run(World, name="Junction Twoway", sim_seconds=30.0):
    """The reference's scenario files (parsed form: tests/golden/scenarios.json) run headless on the
    engine: environment rasterised on the device, formations spawned over time, whole ticks of the
    driver chain (topology pass, prior updates, GBP schedule, waypoints / despawn) — simulated seconds
    per wall-clock second, host logic included."""
    from magics_amd import config, sim
    with open(os.path.join(ROOT, "tests", "golden", "scenarios.json"), encoding="utf-8") as f:
        sc = json.load(f)[name]
    s = sim.Simulation(sc, World(config.world_params(sc["config"])))
    t0 = time.perf_counter()
    s.run(max_time=sim_seconds)
    s.w.synchronize()
    wall = time.perf_counter() - t0
    return {"name": name, "simulated_s": round(s.elapsed(), 2), "wall_s": round(wall, 3), "ticks": s.tick_no,
            "ticks_per_s": round(s.tick_no / wall, 1), "robots_spawned": len(s.robots), "horizon": s.K,
            "what": "config.toml + environment.yaml + formation.yaml of the reference, unmodified: device rasteriser, "
                    "spawner, topology pass, prior updates, GBP schedule, waypoint logic (Python host loop included)"}


def launch_ranks(a):
    """`python bench.py --gpus N` with no launcher around it: this process becomes the launcher.  It starts the N ranks as fresh
    CHILD processes — before anything here has touched the GPU (torch is not even imported): a process that has initialised the
    GPU must never be replaced or forked — gives each its RANK / LOCAL_RANK / WORLD_SIZE and a rendezvous on 127.0.0.1, relays
    rank 0's JSON line and exits with the worst exit code.  The ranks themselves take the same path as under torch.distributed.run."""
    import socket
    with socket.socket() as sk:  # a free port for the rendezvous (+ room above it for the probes' own rendezvous, see run_children)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    if port > 65000:
        port -= 2000
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(a.gpus):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MGX_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode(errors="replace"))
    sys.stdout.flush()
    sys.exit(max(abs(rc) for rc in rcs))


def main():
    a = parse()
    if a.role in ("direct-child", "rccl-child"):
        return transport_child(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    multi = world_size > 1
    if a.gpus != world_size:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world_size} processes", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the engine has no CPU path", file=sys.stderr)
        sys.exit(2)
    # MGX_BENCH_BACKEND=gloo + MGX_BENCH_DEVICE=0 is the dry-run mode of the N > 1 control flow on a
    # one-GPU box (collectives staged through the host); the measured configuration is nccl = RCCL.
    backend = os.environ.get("MGX_BENCH_BACKEND", "nccl")
    device_index = int(os.environ.get("MGX_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(device_index)
    red_dev = "cuda" if backend == "nccl" else "cpu"
    dist = None
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    from magics_amd import World, scenarios as S, sharded
    stream = torch.cuda.current_stream().cuda_stream
    n_loc, K = a.robots_per_gpu, a.horizon
    n_tot = n_loc * world_size
    full_size = (n_loc, K) == (1000, 16)

    line = {
        "metric": "GBP iterations/sec (whole node), N robots x K horizon",
        "value": None,
        "unit": f"GBP iterations/s (one iteration = every factor and variable of {n_loc} robots x {K} horizon updated once; "
                f"{n_tot} robots on {world_size} GPU(s) advance {world_size} such units per step)",
        "n_gpus": world_size, "steps": a.steps, "warmup": a.warmup, "repeats": a.repeats, "preheat_ms": a.preheat_ms,
        "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "timing": "median of `repeats` repetitions of exactly `steps` steps, each between barrier + synchronize, MAX over ranks "
                  "per repetition, after `warmup` steps and `preheat_ms` of untimed work",
    }

    # who ran: what torch.distributed saw, each rank's device, the collective library — so that "RCCL saw N ranks on N
    # different GPUs" can be checked from the line
    try:
        prop = torch.cuda.get_device_properties(device_index)
        me = {"rank": rank, "device_index": device_index, "name": prop.name, "uuid": str(getattr(prop, "uuid", "")),
              "pci_bus_id": getattr(prop, "pci_bus_id", None)}
    except Exception as e:  # noqa: BLE001
        me = {"rank": rank, "device_index": device_index, "error": f"{type(e).__name__}: {e}"}
    ranks_seen = [me]
    if multi:
        ranks_seen = [None] * world_size
        dist.all_gather_object(ranks_seen, me)
    try:
        rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001
        rccl = None
    line["ranks"] = {"world_size_seen_by_torch_distributed": dist.get_world_size() if multi else 1, "backend": backend if multi else None,
                     "rccl_version": rccl, "distinct_devices": len({(r or {}).get("uuid") or (r or {}).get("device_index") for r in ranks_seen}),
                     "devices": ranks_seen, "started_by": "bench.py itself (child processes)" if os.environ.get("MGX_BENCH_SELF_LAUNCHED") else
                     ("a launcher (WORLD_SIZE in the environment)" if multi else "one process")}

    # Every phase below contains collectives at N > 1 (barriers, reductions, the halo exchange): the run carries a
    # deadline (see _Deadline) that ends it with a non-zero exit code instead of hanging.
    guard = _Deadline(a.deadline if multi else 0, rank, line)

    # ---- configs[1]: dynamics + obstacle factors, no inter-robot factors; ranks are independent shards ------------
    sc1 = None
    if not a.no_configs1:
        sc1 = S.grid_scenario(n_loc, K, interrobot=False, seed=805 + rank)
        w1 = World(sc1["params"], stream=stream, fma=a.fma)
        S.populate(w1, sc1)
        assert len(sc1["steps"]) == SCHEDULE_LEN
        G1 = max(a.ticks_per_submission, 2)  # (no inter-robot factors: nothing to keep a launch resident for — its ticks are bracketed two at a time)
        walls, devs = timed(torch, dist, w1.iterate, sc1["steps"], a, multi, red_dev, sync=w1.synchronize, batch=w1.batch, group=G1, flush=w1.flush)
        r1 = summary(walls, devs, a.steps, units_per_step=world_size)
        bytes1 = S.algorithmic_bytes_per_robot_iter(K, 0.0) * n_loc
        # launches of the timed K-step block: as the engine counted them for the same block, issued once more
        n_launch = run_steps(w1.iterate, a.steps, sc1["steps"], w1.batch, G1) or -(-a.steps // SCHEDULE_LEN)
        it1 = a.steps / n_launch
        line["configs1"] = {
            "value": round(r1["value"], 2), "unit": "GBP iterations/s (same unit as the headline)",
            "ms_per_step": r1["ms_per_step"], "device_ms_per_step": r1["device_ms_per_step"], "spread": r1["spread"],
            "config": {"workload": f"BASELINE configs[1]: synthetic {n_loc} robots x {K} horizon per GPU, dynamics + obstacle factors, "
                                   "seed 805, 10-step schedule" + f", {G1} ticks per submission (mgx_batch_*) = one launch",
                       "parallelism": f"{world_size} independent shard(s), no exchange (robots do not interact)"},
            "roofline": roofline("k_robot_sweep<16,0,false>", bytes1 * it1, n_launch, r1["device_s_median"], "config1" if full_size else "",
                                 it1),
        }
        w1.synchronize()

    # ---- headline: + inter-robot factors, robots sharded over the ranks ------------------------------------------------
    # Every rank first builds its shard; the ranks then agree that all of them succeeded BEFORE the first collective, so
    # a failure on one rank can never leave the others waiting in RCCL.
    sw, err, sc2 = None, "", None
    try:
        sc2 = S.grid_scenario(n_tot, K, interrobot=True, seed=805)
        comm = sharded.TorchDistComm(stage_through_host=backend != "nccl") if multi else None
        sw = sharded.ShardedWorld(sc2, rank, world_size, lambda p: World(p, stream=stream, fma=a.fma), comm=comm)
        sw.world.sweep(0, 0, 0)  # commit: device arrays built, no phase run
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=red_dev)
    if multi:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok[0]) != 1:
        if rank == 0:
            line["error"] = err or "another rank failed to build its shard"
            print(json.dumps(line), flush=True)
        os._exit(3)
    D = len(sc2["ir"]) / n_tot
    bytes2 = S.algorithmic_bytes_per_robot_iter(K, D) * n_loc  # per GPU per iteration

    def agree(flag):  # every rank says yes?
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=red_dev)
        if multi:
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t[0]) == 1

    # ---- N > 1: the in-engine transports are PROBED in child processes first (a transport that cannot map its peers on this
    # node, or faults, takes a child down, not the bench); the one that ran there and left the beliefs of the host-driven
    # exchange is then wired and measured HERE, in the bench process, and gives the headline
    children, probe_ok = {}, False
    if multi and not a.no_children:
        guard.cancel()  # the children carry their own bound (run_children); the phases with collectives get a fresh one below
        children["direct"] = run_children(a, rank, local_rank, world_size, "direct-child", 23)
        if backend == "nccl":  # the in-library RCCL transport needs one GPU per rank
            children["rccl_in_engine"] = run_children(a, rank, local_rank, world_size, "rccl-child", 41)
        r = children["direct"] if rank == 0 else None
        flag = [bool(isinstance(r, dict) and isinstance(r.get("value"), (int, float)) and r.get("verified_against_host_driven_exchange"))]
        guard = _Deadline(a.deadline, rank, line)
        dist.broadcast_object_list(flag, src=0)
        probe_ok = flag[0]

    G = a.ticks_per_submission if a.ticks_per_submission > 0 else (2 if multi else 1)
    walls, devs = timed(torch, dist, sw.iterate, sc2["steps"], a, multi, red_dev, sync=sw.synchronize, batch=sw.batch, group=G, flush=sw.flush)
    sw.iterate(sc2["steps"])
    resident = sw.world.last_launch_count() == 1  # the engine ran the 10-step schedule as ONE resident launch (or posted it into one)
    sw.synchronize()
    # the timed block once more, counted: sweep-kernel launches (a schedule POSTED into a lingering launch is no launch)
    res0 = int(sw.world.resident_stats()[0])
    head_launches = run_steps(sw.iterate, a.steps, sc2["steps"], sw.batch, G)
    sw.flush()
    if resident:
        head_launches = int(sw.world.resident_stats()[0]) - res0
    lst = [int(x) for x in sw.world.linger_stats()] if hasattr(sw.world, "linger_stats") else None
    r2 = summary(walls, devs, a.steps, units_per_step=world_size)
    per_tick, variants = None, {}
    a_warm = argparse.Namespace(**{**vars(a), "preheat_ms": 0.0})  # (clocks are up: a second timed pass needs no pre-heat of its own)
    if G > 1:  # the same workload, one submission per tick — always beside a batched headline
        walls_t, devs_t = timed(torch, dist, sw.iterate, sc2["steps"], a_warm, multi, red_dev, sync=sw.synchronize, flush=sw.flush)
        rt = summary(walls_t, devs_t, a.steps, units_per_step=world_size)
        per_tick = {"value": round(rt["value"], 2), "ms_per_step": rt["ms_per_step"], "device_ms_per_step": rt["device_ms_per_step"],
                    "what": "the same workload without mgx_batch_*: every 10-step tick its own mgx_iterate call, nothing bracketed"}
    elif not multi and not a.no_extras:
        # what the two ways of keeping the graphs in LDS across ticks are worth, each against the same world: ticks bracketed two at
        # a time (mgx_batch_*: round 4's headline), and lingering switched off (every tick a launch of its own)
        walls_t, devs_t = timed(torch, dist, sw.iterate, sc2["steps"], a_warm, multi, red_dev, sync=sw.synchronize, batch=sw.batch, group=2, flush=sw.flush)
        rt = summary(walls_t, devs_t, a.steps, units_per_step=world_size)
        variants["two_ticks_per_batch"] = {"value": round(rt["value"], 2), "ms_per_step": rt["ms_per_step"], "device_ms_per_step": rt["device_ms_per_step"],
                                           "what": "ticks handed over two at a time inside mgx_batch_begin / _end (merged into one launch)"}
        sw.world.set_linger(0)
        walls_t, devs_t = timed(torch, dist, sw.iterate, sc2["steps"], a_warm, multi, red_dev, sync=sw.synchronize, flush=sw.flush)
        rt = summary(walls_t, devs_t, a.steps, units_per_step=world_size)
        variants["launch_per_tick"] = {"value": round(rt["value"], 2), "ms_per_step": rt["ms_per_step"], "device_ms_per_step": rt["device_ms_per_step"],
                                       "what": "mgx_set_linger(world, 0): every 10-step tick a resident launch of its own (graphs HBM -> LDS -> HBM per tick)"}
        sw.world.set_linger(None)
    sustained_head = None
    if a.sustained_seconds > 0:
        sustained_head = sustained(torch, dist, sw.iterate, sc2["steps"], a.sustained_seconds, r2["ms_per_step"], multi, red_dev,
                                   sync=sw.synchronize, units_per_step=world_size, batch=sw.batch, group=G)
    by_transport = {("one GPU, no exchange" if not multi else "collective"): round(r2["value"], 2)}
    transport = "none (one GPU)" if not multi else "collective (torch.distributed all_to_all_single over RCCL, host-driven)"
    sw.synchronize()
    guard.cancel()

    in_engine = None
    if multi and probe_ok:
        guard = _Deadline(a.deadline, rank, line)
        sw_in, got, err = None, None, ""
        try:
            sw_in = sharded.ShardedWorld(sc2, rank, world_size, lambda p: World(p, stream=stream, fma=a.fma), comm=comm)
            sw_in.world.sweep(0, 0, 0)
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        if agree(not err):
            try:
                got = sharded.connect(sw_in, comm, "direct", resident=not a.no_resident)  # the ranks agree after every step
            except Exception as e:  # noqa: BLE001
                err = f"{type(e).__name__}: {e}"
            if agree(not err and got is not None and got.startswith("direct")):
                import numpy as np
                verified = False
                settled, declined0 = 0, 0
                try:
                    ref = sharded.ShardedWorld(sc2, rank, world_size, lambda p: World(p, stream=stream, fma=a.fma), comm=comm)
                    settled, declined0 = (settle_resident(sw_in, ref, sc2["steps"], agree) if got == "direct+resident" else (0, 0))
                    if rank == 0:
                        line["in_engine_settling"] = {"ticks": settled, "declined_while_settling": declined0}
                    for _ in range(2):  # the same two ticks with the exchange driven from the host: bit-identical beliefs
                        sw_in.iterate(sc2["steps"])
                        ref.iterate(sc2["steps"])
                    sw_in.synchronize()
                    ref.synchronize()
                    same = all(np.array_equal(x, y) for x, y in zip(sw_in.read_beliefs()[1:], ref.read_beliefs()[1:]))
                    del ref
                except Exception as e:  # noqa: BLE001
                    err, same = f"{type(e).__name__}: {e}", False
                verified = agree(same)
                walls_i = devs_i = None
                if verified and got == "direct+resident" and not agree(sw_in.world.resident_stats()[1] == declined0):
                    # (see the probe: declined launches cost a wait each — measure the plain direct transport instead)
                    sw_in.world.set_resident_launches(False)
                    got = "direct"
                    if rank == 0:
                        line["in_engine_note"] = "resident launches were declined by the ranks' agreement during verification: switched off"
                if verified:
                    try:
                        walls_i, devs_i = timed(torch, dist, sw_in.iterate, sc2["steps"], a, True, red_dev, sync=sw_in.synchronize,
                                                batch=sw_in.batch, group=G)
                        sw_in.iterate(sc2["steps"])
                        sw_in.synchronize()
                        lpt_in = sw_in.world.last_launch_count()
                        launches_in = run_steps(sw_in.iterate, a.steps, sc2["steps"], sw_in.batch, G)
                        sw_in.synchronize()
                        if G > 1:  # one submission per tick, for the same wiring
                            walls_t, devs_t = timed(torch, dist, sw_in.iterate, sc2["steps"], a_warm, True, red_dev, sync=sw_in.synchronize)
                            rt = summary(walls_t, devs_t, a.steps, units_per_step=world_size)
                            per_tick = {"value": round(rt["value"], 2), "ms_per_step": rt["ms_per_step"], "device_ms_per_step": rt["device_ms_per_step"],
                                        "what": "the same workload and wiring without mgx_batch_*: every 10-step tick submitted by itself"}
                    except Exception as e:  # noqa: BLE001
                        err = f"{type(e).__name__}: {e}"
                if verified and agree(not err):
                    ri = summary(walls_i, devs_i, a.steps, units_per_step=world_size)
                    in_engine = {"transport": got, "result": ri, "launches_per_tick": lpt_in, "launches": launches_in,
                                 "resident_stats": [int(x) for x in sw_in.world.resident_stats()]}
                    try:
                        if a.sustained_seconds > 0:
                            in_engine["sustained"] = sustained(torch, dist, sw_in.iterate, sc2["steps"], a.sustained_seconds, ri["ms_per_step"],
                                                               True, red_dev, sync=sw_in.synchronize, units_per_step=world_size,
                                                               batch=sw_in.batch, group=G)
                            in_engine["resident_stats"] = [int(x) for x in sw_in.world.resident_stats()]
                        if got == "direct+resident":
                            # the same wiring with resident launches switched off on every rank: what the exchange costs WITHOUT the
                            # in-launch hand-off (the scaling curve's other half)
                            sw_in.world.set_resident_launches(False)
                            walls_s, devs_s = timed(torch, dist, sw_in.iterate, sc2["steps"], a, True, red_dev, sync=sw_in.synchronize)
                            sw_in.iterate(sc2["steps"])
                            sw_in.synchronize()
                            rs = summary(walls_s, devs_s, a.steps, units_per_step=world_size)
                            in_engine["launch_per_segment"] = {
                                "value": round(rs["value"], 2), "ms_per_step": rs["ms_per_step"], "device_ms_per_step": rs["device_ms_per_step"],
                                "launches_per_tick": sw_in.world.last_launch_count(),
                                "what": "the same sharded workload and direct wiring with mgx_set_resident_launches(world, 0) on every rank: push / "
                                        "wait kernels around one launch per segment"}
                            sw_in.world.set_resident_launches(True)
                    except Exception as e:  # noqa: BLE001
                        in_engine["sustained_error"] = f"{type(e).__name__}: {e}"
                    by_transport[got + " (in the bench process)"] = round(ri["value"], 2)
                    if rank == 0:  # (resident launches so far, declined by the ranks' agreement, back-off left)
                        line["in_engine_resident_stats"] = in_engine["resident_stats"]
        if in_engine is None and rank == 0:
            line["in_engine_in_process"] = {"error": err or "another rank failed, or the beliefs differed from the host-driven exchange's"}
        if sw_in is not None:
            dist.barrier()
            sw_in.direct_close()
        guard.cancel()

    # launches per 10-step tick: ONE when the whole schedule runs as a resident launch, else 11 ([I], 9 x [E I], [E])
    head = r2
    if in_engine is not None:
        head = in_engine["result"]
        resident = in_engine["launches_per_tick"] == 1
        sustained_head = in_engine.get("sustained", None)
        if "launch_per_segment" in in_engine:
            line["launch_per_segment"] = in_engine["launch_per_segment"]
        line["resident_stats"] = {"launches": in_engine["resident_stats"][0], "declined": in_engine["resident_stats"][1],
                                  "backoff_left": in_engine["resident_stats"][2] if len(in_engine["resident_stats"]) > 2 else None}
    elif not multi:
        try:
            st = [int(x) for x in sw.world.resident_stats()]
            line["resident_stats"] = {"launches": st[0], "declined": st[1], "backoff_left": st[2] if len(st) > 2 else None}
        except Exception:  # noqa: BLE001
            pass
    if sustained_head is not None:
        line["sustained"] = sustained_head
    if in_engine is not None:
        transport = {"direct+resident": "direct+resident: ONE resident launch per schedule and rank; boundary robots store their snapshot records "
                                        "and progress words into the other ranks' peer-mapped ghost areas from inside it (xGMI, system-scope "
                                        "stores), no exchange kernel, no collective, no host work between the iterations",
                     "direct": "direct (peer-mapped stores over xGMI + device-side arrival counters, in-engine, one C call per tick)"}[in_engine["transport"]]
    ticks = -(-a.steps // SCHEDULE_LEN)
    launches = ticks if resident else a.steps + ticks
    counted = in_engine.get("launches") if in_engine is not None else head_launches
    if counted:  # batched submissions: as the engine counted them
        launches = counted
    it_per_launch = (a.steps / launches) if resident else 1
    line["submission"] = {"ticks_per_submission": G, "launches_per_timed_block": launches, "iterations_per_launch": it_per_launch,
                          "what": ("every tick is one mgx_iterate call; the ticks are handed to the engine " + str(G) + " at a time inside mgx_batch_begin / "
                                   "mgx_batch_end, which submits them together — merged into one resident launch where their segments fit (the graphs "
                                   "go HBM -> LDS and back once per launch instead of once per tick); nothing is skipped or reordered, results are "
                                   "bit-identical (tests/test_gpu_batch.py)") if G > 1 else
                                  "the reference's call pattern (robot.rs:85-108): one mgx_iterate call per 10-step tick, nothing bracketed.  Schedules "
                                  "issued back to back are POSTED into the resident launch that is there (it lingers for them: the graphs stay in LDS "
                                  "across the calls, include/mgx.h); the timed block's closing synchronisation ends the launch.  Bit-identical "
                                  "(tests/test_gpu_linger.py)"}
    if lst is not None and not multi:
        line["submission"]["linger_stats"] = {"launches_that_lingered": lst[0], "schedules_posted": lst[1], "posts_rerun_as_launches": lst[2],
                                              "launches_ended_by_device": lst[3]}
    if per_tick is not None:
        line["one_submission_per_tick"] = per_tick
    elif G == 1:
        line["one_submission_per_tick"] = {"value": round(head["value"], 2), "what": "the headline itself (ticks_per_submission = 1)"}
    if variants:
        line["variants"] = variants
    kernel = (("k_robot_sweep<16,2,true,shard>" if multi else "k_robot_sweep<16,2,true>") + " (whole schedule resident)") if resident else \
        "k_robot_sweep<16,2,false> (one iteration per launch)"
    line.update({
        "value": round(head["value"], 2), "ms_per_step": head["ms_per_step"], "device_ms_per_step": head["device_ms_per_step"],
        "spread": head["spread"],
        "config": {"workload": f"BASELINE configs[{'2' if not multi else '3 layout'}]: synthetic {n_tot} robots x {K} horizon, dynamics + "
                               f"obstacle + inter-robot factors (comm radius 8, {D:.2f} neighbours/robot), seed 805, 10/10 schedule"
                               + (f", {G} ticks per submission" if G > 1 else ", one mgx_iterate call per tick"),
                   "robots_per_gpu": n_loc, "horizon": K, "robots_total": n_tot,
                   "parallelism": (f"robots sharded over {world_size} GPUs in (y, x) strips, one exchange of boundary snapshot records "
                                   "per external iteration" + (", inside ONE resident launch per schedule and rank" if resident else "")) if multi else
                                  (f"1 GPU: {G} ticks of the 10-step schedule are ONE resident launch" if G > 1 else "1 GPU: the ticks of a timed block "
                                   "run in ONE resident launch (the first launches it, the others are posted into it)") +
                                  ", neighbouring workgroups hand their snapshot records over inside it" if resident else "1 GPU",
                   "ghost_robots_this_rank": len(sw.plan.ghosts)},
        "roofline": roofline(kernel, bytes2 * it_per_launch, launches, head["device_s_median"],
                             ("config2_resident" if resident else "config2") if full_size else "", it_per_launch,
                             prof_note="counters of the N = 1 instantiation of the same per-GPU workload" if multi else ""),
    })
    if multi and rank == 0 and children:
        line["in_engine_transports"] = children  # the probes (isolated child processes)
        for key, r in children.items():
            if isinstance(r, dict) and isinstance(r.get("value"), (int, float)):
                by_transport[key + " (child process)"] = r["value"]
    line["transport"], line["by_transport"] = transport, by_transport

    # ---- N = 1: the same workload with resident launches switched off for THIS world (one launch per iteration): what a
    # rank of a sharded world would pay WITHOUT the in-launch hand-off, so that the scaling curve separates the cost of
    # the exchange from the loss of residency
    if not multi and resident:
        try:
            w_seg = World(sc2["params"], stream=stream, fma=a.fma)
            S.populate(w_seg, sc2)
            w_seg.set_resident_launches(False)
            walls_s, devs_s = timed(torch, dist, w_seg.iterate, sc2["steps"], a, False, red_dev, sync=w_seg.synchronize, flush=w_seg.flush)
            w_seg.iterate(sc2["steps"])
            rs = summary(walls_s, devs_s, a.steps)
            line["launch_per_segment"] = {"value": round(rs["value"], 2), "ms_per_step": rs["ms_per_step"], "device_ms_per_step": rs["device_ms_per_step"],
                                          "launches_per_tick": w_seg.last_launch_count(),
                                          "what": "the same 1000 x 16 inter-robot workload with mgx_set_resident_launches(world, 0): one launch per "
                                                  "[external iteration] internal* segment (= MGX_PERSISTENT=0)"}
            w_seg.synchronize()
            del w_seg
        except Exception as e:  # noqa: BLE001
            line["launch_per_segment"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- N = 1 extras: whole driver ticks, topology churn, a reference scenario end to end ----------------------------
    if not multi and not a.no_extras:
        w = sw.world
        tk = S.tick_inputs(sc2)
        n_ticks = max(200, a.steps // 10)

        def ticks_per_s(tick_once, n, reps=5):
            """the headline's protocol for whole driver ticks: a pre-heat of the same work (the blocks before this one leave the GPU
            idle for seconds of host work: clocks are down), then `reps` blocks of n ticks between synchronisations, the median"""
            t_end = time.perf_counter() + a.preheat_ms * 1e-3
            while time.perf_counter() < t_end:
                for _ in range(10):
                    tick_once()
                w.synchronize()
            rates = []
            for _ in range(reps):
                w.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    tick_once()
                w.synchronize()
                rates.append(n / (time.perf_counter() - t0))
            return statistics.median(rates), min(rates), max(rates)
        med, lo, hi = ticks_per_s(lambda: w.tick(steps=sc2["steps"], **tk), n_ticks)  # mgx_tick: the prior updates ride in the launch / the post
        line["tick"] = {"value": round(med, 1), "unit": "driver ticks/s per GPU", "min": round(lo, 1), "max": round(hi, 1), "ticks_per_block": n_ticks,
                        "what": "update_prior_of_horizon_state + update_prior_of_current_state_v3 for all robots, then the 10/10 "
                                "schedule (inter-robot workload), one mgx_tick call per tick, ticks back to back (robot.rs:85-108); median of 5 "
                                "blocks after a pre-heat"}
        w.set_linger(0)
        med0, _, _ = ticks_per_s(lambda: w.tick(steps=sc2["steps"], **tk), n_ticks, reps=3)
        w.set_linger(None)
        line["tick"]["launch_per_tick"] = round(med0, 1)  # mgx_set_linger(world, 0): every tick a launch of its own
        try:
            import numpy as np
            sc3 = S.grid_scenario(n_loc, K, interrobot=True, seed=805)
            sc3["ir"] = []
            wd = World(sc3["params"], stream=stream, fma=a.fma)
            S.populate(wd, sc3)
            rng = np.random.default_rng(805)
            base = np.array([[rb["pos"][0], 0.5, rb["pos"][1]] for rb in sc3["robots"]], dtype=np.float32)
            tk3 = S.tick_inputs(sc3)
            nxt, _, _ = wd.update_topology(base, 8.0, 1)
            wd.iterate(sc3["steps"])
            wd.synchronize()
            n_dyn, n_blocks, n_warm, made, gone = 60, 3, 30, 0, 0
            # (the robots' positions of every tick are this measurement's INPUT: drawn before the clock starts, like every other
            # synthetic input of the bench; untimed ticks first — the first one sizes tables and asks for the launch's capacity, and
            # the GPU has idled through seconds of host work: the `tick` figure's protocol, a pre-heat and the median of the blocks)
            poss = [base + rng.normal(0, 0.15, size=base.shape).astype(np.float32) for _ in range(n_warm + n_blocks * n_dyn)]
            for pos in poss[:n_warm]:
                nxt, _, _ = wd.update_topology(pos, 8.0, nxt)
                wd.tick(steps=sc3["steps"], **tk3)
            rates = []
            for b in range(n_blocks):
                wd.synchronize()
                t0 = time.perf_counter()
                for pos in poss[n_warm + b * n_dyn:n_warm + (b + 1) * n_dyn]:
                    nxt, c, d = wd.update_topology(pos, 8.0, nxt)
                    wd.tick(steps=sc3["steps"], **tk3)
                    made, gone = made + c, gone + d
                wd.synchronize()
                rates.append(n_dyn / (time.perf_counter() - t0))
            n_all = n_blocks * n_dyn
            line["dynamic_tick"] = {"value": round(statistics.median(rates), 1), "unit": "driver ticks/s per GPU", "min": round(min(rates), 1),
                                    "max": round(max(rates), 1), "ticks_per_block": n_dyn,
                                    "what": f"{n_loc} robots x {K}: comms-range search + factor create/delete (on average "
                                            f"{made / n_all:.0f} connections created, {gone / n_all:.0f} pairs deleted per tick) + prior "
                                            f"updates + 10 GBP iterations with inter-robot factors; median of {n_blocks} blocks after {n_warm} untimed ticks"}
        except Exception as e:  # noqa: BLE001
            line["dynamic_tick"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            line["scenario"] = scenario_run(World)
        except Exception as e:  # noqa: BLE001
            line["scenario"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- CPU baseline: rank 0, N = 1 only ---------------------------------------------------------------------------
    if rank == 0 and not multi and not a.no_cpu_baseline:
        # parity of the exact path the headline ran: a fresh world, the timed block's submission, read back
        check = None
        try:
            w_par = World(sc2["params"], stream=stream, fma=a.fma)
            S.populate(w_par, sc2)
            n_par = min(a.steps, 20 * SCHEDULE_LEN)  # (the whole timed block, up to 200 steps: the oracle runs ~25 steps a second)
            par_launches = run_steps(w_par.iterate, n_par, sc2["steps"], w_par.batch, G)
            par_linger = [int(x) for x in w_par.linger_stats()]
            par_launches = par_launches or int(w_par.resident_stats()[0])
            if n_par % SCHEDULE_LEN == 0:
                check = (n_par // SCHEDULE_LEN, w_par.read_beliefs())
            del w_par
        except Exception as e:  # noqa: BLE001
            line["parity"] = {"checked": False, "error": f"{type(e).__name__}: {e}"}
        cb = cpu_baseline(sc2, a.cpu_seconds, check=check)
        par = cb.pop("parity", None)
        if par is not None:
            par["engine_launches"] = par_launches
            par["engine_linger_stats"] = {"launches_that_lingered": par_linger[0], "schedules_posted": par_linger[1], "posts_rerun_as_launches": par_linger[2]}
            line["parity"] = par
        elif "parity" not in line:
            line["parity"] = {"checked": False, "why": "--steps is not a whole number of ticks"}
        line["cpu_baseline"] = cb
        line["speedup_vs_cpu_baseline"] = round(line["value"] / cb["value"], 1)
        if sc1 is not None:
            cb1 = cpu_baseline(sc1, a.cpu_seconds)
            cb1.pop("parity", None)
            line["configs1"]["cpu_baseline"] = cb1
            line["configs1"]["speedup_vs_cpu_baseline"] = round(line["configs1"]["value"] / cb1["value"], 1)

    if rank == 0:
        print(json.dumps(line))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
