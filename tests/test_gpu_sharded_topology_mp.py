"""A sharded world that follows its topology, ranks in SEPARATE processes on the one GPU of the test
box (gloo control plane and exchange, as in a multi-GPU run with RCCL): each rank drives the same
mission; assembled beliefs, events, robot numbers and trajectories equal the single-world oracle's."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle
from magics_amd import scenarios as S
from magics_amd.driver import Driver

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


@pytest.mark.parametrize("mode", ["collective", "direct", "direct+resident", "collective+migrate", "direct+resident+migrate", "direct+spawn",
                                  "direct+resident+spawn"])
def test_two_processes_follow_the_topology(tmp_path, mode):
    """mode "direct": the exchange inside the engines (hipIpc-mapped record slots, one per ghost robot, re-aimed after every topology
    pass that changed the lists) — no host-driven all-to-all in any tick; "direct+resident": and the ghosts' exchange records inside
    ONE resident launch per schedule and rank (`len(ghosts) < n_robots - n_local` in what is exchanged: only robots connected across
    the ranks travel; every rank still holds a record of every robot — the replicated bookkeeping);
    "+migrate": every ten ticks the robots are dealt out again by where they are and change ranks (ShardedWorld.migrate: the records
    over the control plane, the hipIpc areas closed and wired again) — results unchanged;
    "+spawn": two of the robots join AFTER the in-engine transports were wired (ShardedWorld.add_robot: a slot more in every receive
    area, push tables for the new lists — wired again collectively; the engine refuses an exchange over stale tables)"""
    ws, n, K, ticks = 2, 8, 10, 60
    port = _free_port()
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(ws)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MGX_HALO_TIMEOUT_MS="20000")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dynamic_topology_worker.py"), str(r), str(ws), port, outs[r], mode],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(ws)]
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=300)[0].decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"

    sc = S.circle_scenario(n, K, circle_radius=12.0, n_internal=10, n_external=10)
    sc["ir"] = []
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    drv = Driver(ref, n, K, waypoints=[[tuple(rb["goal"])] for rb in sc["robots"]], radii=[rb["radius"] for rb in sc["robots"]],
                 t0=[rb["t0"] for rb in sc["robots"]], steps=sc["steps"], comms_radius=12.0, target_speed=sc["target_speed"])
    events = np.array([drv.tick() for _ in range(ticks)])
    eta_r, lam_r, mu_r = ref.read_beliefs()
    assert events[:, 0].sum() > 0
    seen = 0
    migrate, mode = mode.endswith("+migrate"), mode.replace("+migrate", "").replace("+spawn", "")
    for o in outs:
        z = np.load(o)
        if migrate:
            assert int(z["moved"]) >= n // 2, int(z["moved"])  # (the exchange and launch counts start over with every re-wiring)
        else:
            assert (int(z["exchanges"]) > (10 if mode == "direct+resident" else ticks)) == mode.startswith("direct")
        if mode == "direct+resident" and not migrate:
            assert int(z["resident"][0]) > ticks // 2, z["resident"]  # most schedules ran as ONE launch on this rank
        assert np.array_equal(z["events"], events) and np.array_equal(z["translation"], drv.translation)
        assert np.array_equal(z["finished_at"], drv.finished_at) and int(z["next_number"]) == drv.next_number
        for j, g in enumerate(z["ids"]):
            sl, sg = slice(j * K, (j + 1) * K), slice(g * K, (g + 1) * K)
            assert np.array_equal(z["eta"][sl], eta_r[sg]) and np.array_equal(z["lam"][sl], lam_r[sg]) and np.array_equal(z["mu"][sl], mu_r[sg])
            seen += 1
    assert seen == n
