"""Direct halo exchange across PROCESSES: two / three ranks, each its own process on the one GPU
of the test box, map each other's receive areas with hipIpc and exchange through peer stores and
device-side arrival counters.  The assembled beliefs must equal the single-world oracle's."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle
from magics_amd import scenarios as S, sharded

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


@pytest.mark.parametrize("world_size,mode", [(2, "direct"), (3, "direct"), (2, "direct+resident"), (3, "direct+resident"),
                                             (3, "direct+resident+decline"), (3, "direct+resident+late"), (2, "direct+resident+batch")])
def test_ranks_in_separate_processes(world_size, mode, tmp_path):
    """mode "direct+resident": every rank's schedule is ONE resident launch and the ghost records cross the process
    boundary inside it (hipIpc-mapped ghost areas, system-scope stores and polls).
    "+decline": the last rank says no to the launch of the second tick — on the word in rank 0's area where the ranks agree on
    every schedule, so ALL ranks' launches return with their worlds untouched, every engine runs that schedule launch by launch
    with the direct exchange, and the third tick (inside the back-off) runs launch by launch on every rank too.
    "+late": the last rank issues the second tick 0.3 s after the others — whose launches have given up waiting for it on that word
    (MGX_RESIDENT_CENSUS_SHARDED_US, here 20 ms) and returned; its own launch finds their "no" and returns as well.
    "+batch": two more ticks inside mgx_batch_begin / mgx_batch_end on every rank: submitted together as ONE resident launch per rank."""
    port = _free_port()
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(world_size)]
    # (how long the ranks' launches wait for each other before they all fall back: generous where every schedule is expected to
    # run as one launch — three processes share the one GPU of a test box —, short where a rank is 0.3 s late on purpose)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MGX_HALO_TIMEOUT_MS="20000", MGX_RESIDENT_TIMEOUT_MS="20000",
               MGX_RESIDENT_CENSUS_SHARDED_US="20000" if mode.endswith("+late") else "500000")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "direct_halo_worker.py"), str(r), str(world_size), port, outs[r], mode],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world_size)]
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=240)[0].decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"

    sc = S.grid_scenario(64, 10, interrobot=True, pitch=2.5, comm_radius=5.0)
    K = sc["K"]
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    steps = sc["steps"] + [1, 1, 2, 3, 2]
    boundary = sorted({g for r in range(world_size) for g in sharded.ShardPlan(sc, r, world_size).ghosts})
    for tick in range(3):
        if tick == 1:
            ref.set_antenna(boundary[0], False)
            ref.change_prior(boundary[2], 9, np.array([0.5, 0.25, 1.0, -1.0]))
        if tick == 2:
            ref.set_antenna(boundary[0], True)
        ref.iterate(steps)
    batch, mode = mode.endswith("+batch"), mode.replace("+batch", "")
    if batch:
        ref.iterate(sc["steps"])
        ref.iterate(sc["steps"])
    eta_r, lam_r, mu_r = ref.read_beliefs()
    # exchanges by the push / wait kernels: one per external iteration — or none at all when the schedule (which opens with an
    # internal iteration) runs as one resident launch
    n_ext = {"direct": 3, "direct+resident": 0, "direct+resident+decline": 2, "direct+resident+late": 2}[mode] * sum(1 for s in steps if s & 2)
    seen = 0
    for o in outs:
        d = np.load(o)
        assert int(d["n"]) == n_ext
        assert int(d["launches"]) == (1 if mode == "direct+resident" else len(sharded.segments(steps)))
        if batch:
            assert tuple(int(x) for x in d["batched"]) == (2, 1), d["batched"]  # two ticks of the 10 / 10 schedule: one launch
        if mode.endswith("+decline") or mode.endswith("+late"):  # (resident launches, declined ones, back-off left) — the same on every rank
            assert d["stats"][0] == 2 and d["stats"][1] == 1 and d["stats"][2] > 0, d["stats"]
        for j, g in enumerate(d["ids"]):
            sl, dl = slice(g * K, (g + 1) * K), slice(j * K, (j + 1) * K)
            assert np.array_equal(d["eta"][dl], eta_r[sl]) and np.array_equal(d["lam"][dl], lam_r[sl]) and np.array_equal(d["mu"][dl], mu_r[sl]), (o, g)
            seen += 1
    assert seen == len(sc["robots"])


def test_bench_direct_child_role(tmp_path):
    """bench.py's isolated direct-exchange measurement (what every bench rank spawns at N > 1),
    here two ranks on the one GPU of the box."""
    import json
    port = _free_port()
    bench = os.path.join(os.path.dirname(HERE), "bench.py")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   MGX_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", MGX_HALO_TIMEOUT_MS="20000", MGX_RESIDENT_CENSUS_SHARDED_US="500000")
        procs.append(subprocess.Popen([sys.executable, bench, "--role", "direct-child", "--gpus", "2", "--steps", "80", "--warmup", "20",
                                       "--robots-per-gpu", "100", "--horizon", "10"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=240))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, outs[r][1].decode(errors="replace")[-3000:]
    res = json.loads([ln for ln in outs[0][0].decode().splitlines() if ln.startswith("{")][-1])
    assert "error" not in res, res
    assert res["value"] > 0 and res["ghost_robots_this_rank"] > 0
    assert res["transport"] == "direct+resident" and res["launches_per_tick"] == 1 and res["verified_against_host_driven_exchange"], res


def test_bench_two_rank_control_flow_dry_run():
    """bench.py as the driver launches it at N = 2 (torch.distributed.run, one process per rank),
    in its dry-run mode for one-GPU boxes: both ranks on cuda:0, gloo instead of RCCL.  Checks the
    control flow end to end: one JSON line, whole-job aggregate, sharded secondary, and the
    direct-exchange measurement from the child processes."""
    import json
    bench = os.path.join(os.path.dirname(HERE), "bench.py")
    env = dict(os.environ, MGX_BENCH_BACKEND="gloo", MGX_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
               MGX_HALO_TIMEOUT_MS="20000", MGX_RESIDENT_CENSUS_SHARDED_US="500000")  # (both ranks share one GPU here)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), bench, "--gpus", "2", "--steps", "60", "--warmup", "20", "--robots-per-gpu", "144",
           "--horizon", "10", "--deadline", "150", "--repeats", "3"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak" and "cpu_baseline" not in d and "error" not in d
    # the headline is the SHARDED inter-robot workload (288 robots over two ranks, ghosts on each), configs[1] beside it
    assert d["config"]["robots_total"] == 288 and d["config"]["ghost_robots_this_rank"] > 0
    assert d["configs1"]["value"] > 0 and d["by_transport"]["collective"] > 0
    # the probe (child processes): peer-mapped stores + resident schedule launches, beliefs equal to the host-driven exchange's
    probe = d["in_engine_transports"]["direct"]
    assert probe.get("value", 0) > 0 and probe["transport"] == "direct+resident" and probe["verified_against_host_driven_exchange"], probe
    assert probe["launches_per_tick"] == 1
    # ... and the headline: the same transport wired and measured in the bench process itself
    assert d["transport"].startswith("direct+resident"), d.get("in_engine_in_process", d["transport"])
    assert d["value"] == d["by_transport"]["direct+resident (in the bench process)"]
    assert "resident" in d["roofline"]["kernel"]


def test_bench_bare_launch_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (no WORLD_SIZE in the environment): bench.py starts its two ranks
    itself, as fresh child processes of a parent that never touches the GPU, and relays rank 0's line — the driver's N = 1 command
    is a bare `python3 bench.py --gpus 1 ...`, and the first multi-GPU run must not end in a usage error.  Dry-run mode for a
    one-GPU box (both ranks on cuda:0, gloo); the line says who ran, carries a sustained figure, the resident launches'
    statistics and the launch-per-segment figure of the same wiring."""
    import json
    bench = os.path.join(os.path.dirname(HERE), "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MGX_BENCH_BACKEND="gloo", MGX_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", MGX_HALO_TIMEOUT_MS="20000",
               MGX_RESIDENT_CENSUS_SHARDED_US="500000")
    cmd = [sys.executable, bench, "--gpus", "2", "--steps", "40", "--warmup", "10", "--robots-per-gpu", "144", "--horizon", "10",
           "--deadline", "150", "--repeats", "3", "--sustained-seconds", "0.5", "--no-configs1", "--ticks-per-submission", "2"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert "error" not in d and d["n_gpus"] == 2 and d["value"] > 0
    assert d["ranks"]["world_size_seen_by_torch_distributed"] == 2 and len(d["ranks"]["devices"]) == 2
    assert d["ranks"]["started_by"].startswith("bench.py itself")
    assert {x["rank"] for x in d["ranks"]["devices"]} == {0, 1}
    assert d["transport"].startswith("direct+resident"), d.get("in_engine_in_process", d["transport"])
    assert d["sustained"]["value"] > 0 and d["sustained"]["seconds"] >= 0.4
    assert d["resident_stats"]["launches"] > 0
    assert d["launch_per_segment"]["value"] > 0 and d["launch_per_segment"]["launches_per_tick"] > 1
    # (--ticks-per-submission 2; the default is the reference's call pattern, one call per tick — sharded worlds' launches do not
    # linger) the ticks are handed to the engines two at a time (mgx_batch_*): ONE resident launch of 20 iterations per rank and submission
    assert d["submission"]["ticks_per_submission"] == 2 and d["submission"]["iterations_per_launch"] == 20, d["submission"]
    assert d["one_submission_per_tick"]["value"] > 0
    print("2 ranks on one GPU:", d["value"], "it/s batched,", d["one_submission_per_tick"]["value"], "one submission per tick")


@pytest.mark.parametrize("n", [2])
def test_bench_ranks_fall_back_together_when_one_probe_fails(n):
    """One rank's in-engine probe fails (its child cannot set up its receive area: injected): the children agree, every rank
    falls back at the same step, the bench line comes out with the collective transport's figure and the probe's error in it —
    no rank is left waiting in a transport the others never wired.  (n ranks are 2 n + 1 processes on the card — the ranks, their
    probe children, this test process: three ranks are seven, one more than a one-GPU box allows; the eight-rank form of this
    rehearsal needs a node.)"""
    import json
    bench = os.path.join(os.path.dirname(HERE), "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MGX_BENCH_BACKEND="gloo", MGX_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", MGX_HALO_TIMEOUT_MS="20000",
               MGX_TEST_FAIL_DIRECT_SETUP_RANK=str(n - 1))
    cmd = [sys.executable, bench, "--gpus", str(n), "--steps", "20", "--warmup", "10", "--robots-per-gpu", "64", "--horizon", "10",
           "--deadline", "120", "--repeats", "3", "--sustained-seconds", "0.2", "--no-configs1"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert "error" not in d and d["n_gpus"] == n and d["value"] > 0
    assert d["transport"].startswith("collective"), d["transport"]
    assert "error" in d["in_engine_transports"]["direct"], d["in_engine_transports"]
    assert d["one_submission_per_tick"]["value"] > 0 and d["submission"]["ticks_per_submission"] == 2  # (N > 1: two ticks per batch by default)
