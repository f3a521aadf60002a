"""Soak run of sharded worlds with ghost records inside resident launches, every rank a PROCESS of its own on the one GPU (hipIpc,
gloo as the control plane): random grids on three ranks, random schedules, flags, prior changes, switches of the inter-robot
factors — and ranks that say no to a launch or issue it tens of milliseconds late, so that the ranks' agreement sends everybody to
the launch-by-launch path, which here is each ENGINE's own (tools/soak_sharded.py drives several ranks from one thread and re-runs
such schedules itself).  Beliefs of the single-world oracle, bit for bit, after every script.
usage: python tools/soak_sharded_mp.py [seconds] [ranks]"""
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank, ws, port, budget):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(ws))
    import numpy as np
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    torch.cuda.set_device(0)
    from magics_amd import World, scenarios as S, sharded
    if rank == 0:
        import oracle
    comm = sharded.TorchDistComm()
    t0, seed, done, declined, resident, dropped, failed = time.time(), 7000, 0, 0, 0, 0, 0
    while True:
        go = [time.time() - t0 < budget and failed == 0]
        dist.broadcast_object_list(go, src=0)
        if not go[0]:
            break
        rng = np.random.default_rng(seed)
        K = int(rng.choice([10, 12, 16]))
        n = int(rng.integers(8, 40)) * ws
        sc = S.grid_scenario(n, K, interrobot=True, seed=seed, pitch=2.5, comm_radius=float(rng.choice([4.0, 5.0, 6.5])))
        sw = sharded.ShardedWorld(sc, rank, ws, World, comm=comm)
        got = sharded.connect(sw, comm, "direct", resident=True)
        ref = None
        if rank == 0:
            ref = oracle.OracleWorld(sc["params"])
            S.populate(ref, sc)
        on, ir_on = sc["params"]["enable_mask"], True
        for step in range(8):
            op, r = int(rng.integers(0, 8)), int(rng.integers(0, n))
            who = int(rng.integers(0, ws))
            worlds = [sw] + ([ref] if ref is not None else [])
            if op == 0:
                v = bool(rng.integers(0, 2))
                for w in worlds:
                    w.set_antenna(r, v)
            elif op == 1:
                v = bool(rng.integers(0, 2))
                for w in worlds:
                    w.set_idle(r, v)
            elif op == 2:
                var, m = int(rng.choice([0, K - 1])), rng.normal(size=4) * 3
                for w in worlds:
                    w.change_prior(r, var, m)
            elif op == 3 and rng.random() < 0.5:
                ir_on = not ir_on
                for w in worlds:
                    w.set_enabled(on if ir_on else on & ~S.EN_IR)
            steps = [int(x) for x in rng.integers(1, 4, size=int(rng.integers(2, 14)))]
            if op == 4 and rank == who:
                sw.world.set_resident_launches("decline")  # this rank says no: every rank's launch returns untouched
            if op == 5 and rank == who:
                time.sleep(0.06)  # ... or comes too late for the others
            before = sw.world.resident_stats()
            sw.iterate(steps)
            if ref is not None:
                ref.iterate(steps)
            after = sw.world.resident_stats()  # (waits for the launch's decision)
            if op == 4 and rank == who:
                sw.world.set_resident_launches(True)
            declined += after[1] - before[1]
            resident += (after[0] - before[0]) - (after[1] - before[1])
        ids, eta, lam, mu = sw.read_beliefs()
        parts = comm.all_gather_object((list(ids), eta, lam, mu))
        if rank == 0:
            e_r, l_r, m_r = ref.read_beliefs()
            if not all(np.isfinite(x).all() for x in (e_r, l_r, m_r)):
                dropped += 1
            else:
                ok = True
                for pid, pe, pl, pm in parts:
                    for j, g in enumerate(pid):
                        sl, dl = slice(g * K, (g + 1) * K), slice(j * K, (j + 1) * K)
                        ok = ok and np.array_equal(pe[dl], e_r[sl]) and np.array_equal(pl[dl], l_r[sl]) and np.array_equal(pm[dl], m_r[sl])
                if ok:
                    done += 1
                else:
                    failed += 1
                    print(f"[soak sharded mp] seed {seed}: MISMATCH ({ws} ranks, {n} robots x {K}, transport {got})", flush=True)
        dist.barrier()
        sw.direct_close()
        del sw
        seed += 1
        if rank == 0 and done % 20 == 0:
            print(f"[soak sharded mp] {done} scripts identical after {time.time() - t0:.0f} s", flush=True)
    tot = comm.all_gather_object((resident, declined))
    if rank == 0:
        print(f"soak sharded mp: {done} random scripts on {ws} processes sharing one GPU (transport {got}), seeds 7000..{seed - 1}: rank 0 saw "
              f"{tot[0][0]} schedules run as ONE launch per rank and {tot[0][1]} declined by the ranks' agreement (a rank said no, or came "
              f"60 ms late) and re-run launch by launch by every engine; {'all bit-identical to the single-world oracle' if not failed else str(failed) + ' MISMATCHED'}; "
              f"{dropped} more drove the oracle itself to NaN / inf ({time.time() - t0:.0f} s)", flush=True)
    dist.destroy_process_group()
    sys.exit(1 if failed else 0)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        return worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], float(sys.argv[5]))
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    ws = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MGX_HALO_TIMEOUT_MS="20000", MGX_RESIDENT_TIMEOUT_MS="20000")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(ws), port, str(budget)], env=env) for r in range(ws)]
    rc = 0
    for p in procs:
        rc = rc or p.wait()
    sys.exit(rc)


if __name__ == "__main__":
    main()
