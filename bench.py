"""bench.py — GBP iterations/s of the MI355X engine on BASELINE.json's synthetic graphs.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE GBP iteration (one schedule step: internal factor + variable sweep and, when the
workload has inter-robot factors, external factor sweep -> routing -> external variable sweep ->
routing) over every robot.  Steps are issued the way the reference's driver issues
them: one `mgx_iterate` call per tick's schedule (10 steps, Junction-Twoway 10/10), inputs already
resident in HBM, priors frozen.

Primary line (`value`): BASELINE.json configs[1] — synthetic 1000 robots x 16 horizon per GPU,
dynamics + obstacle factors; robots are independent, so at N > 1 ranks are independent shards
(no data-path collective) and scaling is weak.  `secondary`: configs[2]/[3] — the same robots with
inter-robot factors (comm radius 8), 1000*N robots sharded N ways with one RCCL all-to-all-v of
boundary snapshots per external iteration.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s peak, ~6.3 achievable)
SCHEDULE_LEN = 10


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--robots-per-gpu", type=int, default=1000)
    ap.add_argument("--horizon", type=int, default=16)
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dynamic", action="store_true", help="N = 1: skip the dynamic_tick measurement")
    ap.add_argument("--no-direct", action="store_true", help="N > 1: skip the direct-exchange child measurement")
    ap.add_argument("--secondary-deadline", type=float, default=150.0,
                    help="N > 1 only: seconds the sharded (collective) phase may take before it is abandoned")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--fma", action="store_true", help="use the FMA-contracting build (not the product)")
    ap.add_argument("--role", default="main", choices=["main", "direct-child", "rccl-child"],
                    help="*-child: one rank of an isolated measurement of the sharded workload with the exchange inside the "
                         "engine (direct: peer-mapped stores; rccl: grouped ncclSend / ncclRecv), spawned by the main role")
    return ap.parse_args()


def run_steps(iterate, n, steps_one_tick):
    """issue exactly n iterations as ticks of len(steps_one_tick) plus a remainder"""
    full, rem = divmod(n, len(steps_one_tick))
    for _ in range(full):
        iterate(steps_one_tick)
    if rem:
        iterate(steps_one_tick[:rem])


def timed(torch, dist, iterate, steps_one_tick, n_steps, n_warm, multi, red_dev="cuda"):
    """W warm-up steps, then exactly K steps between barrier + synchronize; returns
    (wall seconds MAX over ranks, device seconds between HIP events on the launch stream)."""
    run_steps(iterate, n_warm, steps_one_tick)
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run_steps(iterate, n_steps, steps_one_tick)
    ev1.record()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev = ev0.elapsed_time(ev1) * 1e-3
    if multi:
        t = torch.tensor([wall, dev], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev = float(t[0]), float(t[1])
    return wall, dev


def cpu_baseline(sc, seconds):
    """The oracle (a from-scratch port of the reference's CPU path, oracle/gbp_oracle.c) timed on
    this box's host cores on the same workload, bounded to ~`seconds` of CPU work.  Robots run in
    parallel across threads for the internal sweeps, external phase serial (robot.rs:1789-1859)."""
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
    cands = sorted({1, min(avail, 8), min(avail, 16), min(avail, 32), min(avail, 64), avail})
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
    return {
        "value": round(res[best][0], 2), "unit": "GBP iterations/s", "cores": best, "kind": "port",
        "sample": f"{sc['name']}: {res[best][1]} iterations on {best} of {avail} available host threads (robots parallel in "
                  f"internal sweeps, external phase serial), -O3 -march={'native' if lib_path else 'x86-64-v3'}; "
                  f"fastest of thread counts {cands}",
        "single_thread_value": round(res[1][0], 2),
        "by_threads": {str(t): round(v[0], 2) for t, v in res.items()},
    }


def measured_traffic(key):
    """HBM bytes per dispatch from the committed PMC passes (profiles/traffic_r01.json): FETCH_SIZE
    doubled (gfx950 correction for wide streaming reads) + WRITE_SIZE; None when absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_r01.json")) as f:
            t = json.load(f)[key]
        return int((2 * t["fetch_kib"] + t["write_kib"]) * 1024)
    except Exception:  # noqa: BLE001
        return None


def direct_child(a):
    """One rank of the sharded inter-robot workload with the DIRECT halo exchange (peer-mapped
    stores over xGMI, include/mgx.h) — run as a child process of each bench rank so that nothing
    it does can take the main measurement down.  Control plane: gloo; no RCCL in this process."""
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

    out, sw, err = {}, None, ""
    try:
        sc2 = S.grid_scenario(n_tot, K, interrobot=True, seed=805)
        comm = sharded.TorchDistComm()
        sw = sharded.ShardedWorld(sc2, rank, world_size, lambda p: World(p, stream=stream, fma=a.fma), comm=comm)
        sw.world.sweep(0, 0, 0)
        info = sw.direct_setup(export_ipc=True) if a.role == "direct-child" else None
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    if not agree(not err):
        out = {"error": err or "another rank failed to set up"}
    elif a.role == "direct-child":
        infos = comm.all_gather_object(info)
        try:
            sw.direct_connect({i["rank"]: i for i in infos})
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        if not agree(not err):
            out = {"error": err or "another rank failed to map its peers"}
    else:
        try:
            sharded.connect_rccl(sw, comm)  # collective (ncclCommInitRank); a failure on one rank ends in the deadline
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        if not agree(not err):
            out = {"error": err or "another rank failed to join the RCCL communicator"}
    if not out:
        steps2 = sc2["steps"]
        n2, w2 = max(SCHEDULE_LEN, a.steps // 4), max(SCHEDULE_LEN, a.warmup // 4)
        run_steps(sw.iterate, w2, steps2)
        try:
            if a.role == "direct-child":
                sw.world.halo_direct_status()  # synchronises; raises if a wait timed out
            else:
                sw.synchronize()
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        if not agree(not err):
            out = {"error": err or "another rank timed out in the warm-up"}
    if not out:
        dist.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        run_steps(sw.iterate, n2, steps2)
        ev1.record()
        torch.cuda.synchronize()
        dist.barrier()
        wall = time.perf_counter() - t0
        t = torch.tensor([wall, ev0.elapsed_time(ev1) * 1e-3], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev = float(t[0]), float(t[1])
        n_ex = None
        try:
            if a.role == "direct-child":
                n_ex = sw.world.halo_direct_status()
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        if not agree(not err):
            out = {"error": err or "another rank timed out"}
        else:
            D = len(sc2["ir"]) / n_tot
            bytes2 = S.algorithmic_bytes_per_robot_iter(K, D) * n_loc
            out = {"value": round(n2 / wall, 2), "unit": f"GBP iterations/s (one iteration over all {n_tot} robots)",
                   "steps": n2, "ms_per_step": wall / n2 * 1e3, "exchanges": n_ex,
                   "exchange": ("direct: peer-mapped stores into the consumers' receive areas (hipIpc) + device-side arrival "
                                "counters, one C call per tick, no collective") if a.role == "direct-child" else
                               "RCCL inside the engine: grouped ncclSend / ncclRecv per external iteration on the launch stream, "
                               "one C call per tick",
                   "ghost_robots_this_rank": len(sw.plan.ghosts),
                   "roofline": {"bound": "hbm", "achieved": round(bytes2 * n2 / dev / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(bytes2 * n2 / dev / 1e9 / HBM_PEAK_GBS, 4)}}
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.barrier()
    if sw is not None:
        sw.direct_close()
    dist.destroy_process_group()


def run_direct_children(a, rank, local_rank, world_size, role="direct-child", port_shift=23):
    """Spawn this rank's direct-exchange child and wait for it (bounded).  Returns the child's JSON
    (rank 0) or a description of what went wrong; never raises."""
    # the children rendezvous among themselves: nothing of the launcher's elastic agent may leak in
    # (with TORCHELASTIC_USE_AGENT_STORE set, rank 0 would not host the store on the new port)
    base = {k: v for k, v in os.environ.items() if not k.startswith(("TORCHELASTIC_", "GROUP_", "ROLE_"))}
    env = dict(base, RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world_size),
               MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"),
               MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + port_shift),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.setdefault("MGX_HALO_TIMEOUT_MS", "2000")
    cmd = [sys.executable, os.path.abspath(__file__), "--role", role, "--gpus", str(world_size), "--steps", str(a.steps),
           "--warmup", str(a.warmup), "--robots-per-gpu", str(a.robots_per_gpu), "--horizon", str(a.horizon)] + (["--fma"] if a.fma else [])
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=a.secondary_deadline)
    except subprocess.TimeoutExpired:
        return {"error": f"{role} did not finish within {a.secondary_deadline} s"}
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
    """Ends the process from a timer thread if a phase with collectives overruns (see main)."""

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
            out["secondary"] = {"error": f"sharded phase did not finish within {seconds} s; skipped"}
            print(json.dumps(out), flush=True)
        os._exit(0)

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


def main():
    a = parse()
    if a.role in ("direct-child", "rccl-child"):
        return direct_child(a)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    multi = world_size > 1
    if a.gpus != world_size:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} needs {a.gpus} processes (torch.distributed.run --nproc-per-node {a.gpus}); "
                  f"found WORLD_SIZE={world_size}", file=sys.stderr)
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

    # ---- primary: configs[1], dynamics + obstacle factors, ranks are independent shards --------
    sc = S.grid_scenario(n_loc, K, interrobot=False, seed=805 + rank)
    w = World(sc["params"], stream=stream, fma=a.fma)
    S.populate(w, sc)
    steps = sc["steps"]
    assert len(steps) == SCHEDULE_LEN
    wall, dev = timed(torch, dist, w.iterate, steps, a.steps, a.warmup, multi, red_dev)
    bytes_iter = S.algorithmic_bytes_per_robot_iter(K, 0.0) * n_loc  # per GPU per iteration
    n_launch = -(-a.steps // SCHEDULE_LEN)
    line = {
        "metric": "GBP iterations/sec (whole node), N robots x K horizon",
        "value": round(world_size * a.steps / wall, 2),
        "unit": "GBP iterations/s (one iteration over 1000 robots x 16 horizon per GPU)",
        "n_gpus": world_size, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": wall / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: synthetic {n_loc} robots x {K} horizon per GPU, dynamics + "
                               "obstacle factors, seed 805, 10-step schedule per launch",
                   "robots_per_gpu": n_loc, "horizon": K, "robots_total": n_loc * world_size,
                   "parallelism": f"{world_size} independent shard(s), no collective"},
        "roofline": {
            "bound": "hbm", "kernel": "k_robot_sweep",
            "achieved": round(bytes_iter * a.steps / dev / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(bytes_iter * a.steps / dev / 1e9 / HBM_PEAK_GBS, 4),
            "traffic": measured_traffic("config2") if (n_loc, K) == (1000, 16) else None,
            "algorithmic_bytes_per_launch": bytes_iter * SCHEDULE_LEN,
            "avg_launch_us": round(dev / n_launch * 1e6, 3),
            "note": "algorithmic bytes = SURVEY §8d model (38 656 B per robot-iteration); the launch keeps each "
                    "robot's graph in LDS for its 10 iterations, so real HBM traffic is far below it",
        },
    }
    # whole driver ticks: prior updates of the current and horizon state of every robot
    # (one launch, robot.rs:2182-2338) + the 10-step schedule; includes the host-side argument upload
    tk = S.tick_inputs(sc)
    n_ticks = max(10, a.steps // 20)
    for _ in range(5):
        w.update_priors(**tk)
        w.iterate(steps)
    w.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_ticks):
        w.update_priors(**tk)
        w.iterate(steps)
    w.synchronize()
    two_calls = n_ticks / (time.perf_counter() - t0)
    for _ in range(5):
        w.tick(steps=steps, **tk)
    w.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_ticks):
        w.tick(steps=steps, **tk)   # mgx_tick: the prior updates ride in the launch that opens the tick
    w.synchronize()
    line["tick"] = {"value": round(n_ticks / (time.perf_counter() - t0), 1), "unit": "driver ticks/s per GPU",
                    "what": "update_prior_of_horizon_state + update_prior_of_current_state_v3 for all robots, then 10 GBP "
                            "iterations, one mgx_tick call per tick",
                    "as_two_calls": round(two_calls, 1)}
    w.synchronize()

    # ---- secondary: configs[2]/[3], + inter-robot factors, robots sharded with halo exchange ------
    sc2 = None
    # The sharded workload is the only phase with a collective in it.  Should that collective ever
    # stall on some node, the primary measurement must still be reported: every rank carries a
    # deadline for this phase, and on expiry rank 0 prints the line without the secondary figures
    # and all ranks leave.
    guard = _Deadline(a.secondary_deadline if multi else 0, rank, line)
    if not a.no_secondary:
        # Every rank first builds its shard; the ranks then agree that all of them succeeded BEFORE the
        # first collective, so a failure on one rank can never leave the others waiting in RCCL.
        n_tot = n_loc * world_size
        sw, err = None, ""
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
        if int(ok[0]) == 1:
            steps2 = sc2["steps"]
            n2, w2 = max(SCHEDULE_LEN, a.steps // 4), max(SCHEDULE_LEN, a.warmup // 4)
            wall2, dev2 = timed(torch, dist, sw.iterate, steps2, n2, w2, multi, red_dev)
            D = len(sc2["ir"]) / n_tot
            bytes2 = S.algorithmic_bytes_per_robot_iter(K, D) * n_loc
            line["secondary"] = {
                "value": round(n2 / wall2, 2), "unit": f"GBP iterations/s (one iteration over all {n_tot} robots)",
                "steps": n2, "ms_per_step": wall2 / n2 * 1e3,
                "config": {"workload": f"BASELINE configs[2]/[3]: synthetic {n_tot} robots x {K} horizon + inter-robot "
                                       f"factors (comm radius 8, {D:.2f} neighbours/robot), sharded {world_size} way(s)",
                           "parallelism": (f"robots sharded over {world_size} GPUs, one RCCL all-to-all-v of boundary "
                                           "snapshots per external iteration") if multi else "1 GPU, no collective",
                           "ghost_robots_this_rank": len(sw.plan.ghosts)},
                "roofline": {"bound": "hbm", "kernel": "k_robot_sweep",
                             "achieved": round(bytes2 * n2 / dev2 / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(bytes2 * n2 / dev2 / 1e9 / HBM_PEAK_GBS, 4),
                             "traffic": measured_traffic("config3") if (n_loc, K, world_size) == (1000, 16, 1) else None,
                             "algorithmic_bytes_per_launch": bytes2,
                             "avg_launch_us": round(dev2 / (n2 + -(-n2 // SCHEDULE_LEN)) * 1e6, 3)},
            }
            sw.synchronize()
        else:
            line["secondary"] = {"error": err or "another rank failed to build its shard"}
            sc2 = None

    # ---- whole ticks with topology churn (N = 1 only): positions jitter every tick, so the comms-range
    # search finds pairs that cross the radius and the engine creates / deletes their factors
    if not multi and not a.no_secondary and not a.no_dynamic:
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
            n_dyn, made, gone = 30, 0, 0
            t0 = time.perf_counter()
            for _ in range(n_dyn):
                pos = base + rng.normal(0, 0.15, size=base.shape).astype(np.float32)
                nxt, c, d = wd.update_topology(pos, 8.0, nxt)
                wd.tick(steps=sc3["steps"], **tk3)
                made, gone = made + c, gone + d
            wd.synchronize()
            line["dynamic_tick"] = {"value": round(n_dyn / (time.perf_counter() - t0), 1), "unit": "driver ticks/s per GPU",
                                    "what": f"{n_loc} robots x {K}: comms-range search + factor create/delete (on average "
                                            f"{made / n_dyn:.0f} connections created, {gone / n_dyn:.0f} pairs deleted per tick) + prior "
                                            "updates + 10 GBP iterations with inter-robot factors"}
        except Exception as e:  # noqa: BLE001
            line["dynamic_tick"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- a reference scenario end to end (N = 1): the front-end of SURVEY §8 f3 ----------------------------
    if not multi and not a.no_secondary and not a.no_dynamic:
        try:
            line["scenario"] = scenario_run(World)
        except Exception as e:  # noqa: BLE001
            line["scenario"] = {"error": f"{type(e).__name__}: {e}"}

    guard.cancel()

    # ---- the same sharded workload with the direct exchange, isolated in child processes ---------
    if multi and not a.no_secondary and not a.no_direct:
        res = run_direct_children(a, rank, local_rank, world_size)
        if rank == 0 and isinstance(line.get("secondary"), dict):
            line["secondary"]["direct_exchange"] = res
        if backend == "nccl":  # the in-library RCCL transport needs one GPU per rank
            res = run_direct_children(a, rank, local_rank, world_size, role="rccl-child", port_shift=41)
            if rank == 0 and isinstance(line.get("secondary"), dict):
                line["secondary"]["rccl_in_engine"] = res
        # the three transports carry the same exchange; the headline of the secondary block is the fastest one that
        # ran (the host-driven collective pays ~70 us of Python per external iteration, the in-engine ones do not)
        if rank == 0 and isinstance(line.get("secondary"), dict) and "value" in line["secondary"]:
            sec = line["secondary"]
            sec["transport"] = "collective (torch.distributed all_to_all_single, host-driven)"
            sec["by_transport"] = {"collective": sec["value"]}
            for key, label in (("direct_exchange", "direct (peer-mapped stores, in-engine)"), ("rccl_in_engine", "RCCL grouped send/recv, in-engine")):
                r = sec.get(key)
                if isinstance(r, dict) and isinstance(r.get("value"), (int, float)):
                    sec["by_transport"][key] = r["value"]
                    if r["value"] > sec["value"]:
                        sec["value"], sec["ms_per_step"], sec["transport"] = r["value"], r.get("ms_per_step"), label
                        if isinstance(r.get("roofline"), dict):
                            sec["roofline"] = dict(sec.get("roofline", {}), **r["roofline"])
                            sec["roofline"]["avg_launch_us"] = None  # measured for the collective run only

    # ---- CPU baseline: rank 0, N = 1 only ---------------------------------------------------------
    if rank == 0 and not multi and not a.no_cpu_baseline:
        cb = cpu_baseline(sc, a.cpu_seconds)
        line["cpu_baseline"] = cb
        line["speedup_vs_cpu_baseline"] = round(line["value"] / cb["value"], 1)
        if sc2 is not None and "value" in line.get("secondary", {}):
            cb2 = cpu_baseline(sc2, a.cpu_seconds)
            line["secondary"]["cpu_baseline"] = cb2
            line["secondary"]["speedup_vs_cpu_baseline"] = round(line["secondary"]["value"] / cb2["value"], 1)

    if rank == 0:
        print(json.dumps(line))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
