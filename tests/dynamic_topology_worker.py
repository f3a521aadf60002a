"""One rank of the multi-process dynamic-topology test (tests/test_gpu_sharded_topology_mp.py): every
rank is its own process on cuda:0 with a sharded world that follows its topology; the control plane
and the halo all-to-all-v go through gloo (device buffers staged through the host — a dry run of the
RCCL path), or — mode "direct" — the exchange lives in the engines: peer-mapped stores (hipIpc) into one record slot per ghost
robot, wired once, re-aimed when the lists change.  Every rank runs the same driver.
A mode with "+migrate" re-balances every ten ticks: the robots are dealt out again in strips of where they are, and the ones whose
strip changed move to their new rank (ShardedWorld.migrate: records over the control plane, transports wired again).
A mode with "+spawn" builds and WIRES the world with the first n - 2 robots and lets the last two join afterwards (ShardedWorld.add_robot,
collective): every rank's receive areas need a slot more and the push tables name the old lists — the transports are wired again.
usage: dynamic_topology_worker.py RANK WORLD_SIZE PORT OUT.npz [collective|direct|direct+resident][+migrate|+spawn]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, ws, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    mode = sys.argv[5] if len(sys.argv) > 5 else "collective"
    migrate, spawn = "+migrate" in mode, "+spawn" in mode
    mode = mode.replace("+migrate", "").replace("+spawn", "")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(ws))
    import numpy as np
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    torch.cuda.set_device(0)
    from magics_amd import World, scenarios as S, sharded
    from magics_amd.driver import Driver
    n, K, ticks = 8, 10, 60
    sc = S.circle_scenario(n, K, circle_radius=12.0, n_internal=10, n_external=10)
    sc["ir"] = []
    comm = sharded.TorchDistComm(stage_through_host=True)
    stream = torch.cuda.Stream()
    n0 = n - 2 if spawn else n
    sc0 = dict(sc, robots=sc["robots"][:n0])
    sw = sharded.ShardedWorld(sc0, rank, ws, lambda p: World(p, stream=stream.cuda_stream), comm=comm, owner=np.arange(n0) % ws, dynamic=True)
    if mode.startswith("direct"):
        os.environ.setdefault("MGX_RESIDENT_CENSUS_SHARDED_US", "500000")  # (two processes share the one GPU of a test box)
        got = sharded.connect(sw, comm, "direct", resident=mode == "direct+resident")
        assert got == mode and sw.direct, got
    for g in range(n0, n):  # robots that join a world whose exchange is wired already
        rb = sc["robots"][g]
        assert sw.add_robot(rb["mean0"], rb["prior_diag"], rb["dt"], rb["radius"], path=rb["path"], owner=g % ws, order_key=rb["order_key"]) == g
        assert sw.transport == mode and (sw.direct or mode == "collective"), sw.transport  # (wired again as it was)
    drv = Driver(sw, n, K, waypoints=[[tuple(rb["goal"])] for rb in sc["robots"]], radii=[rb["radius"] for rb in sc["robots"]],
                 t0=[rb["t0"] for rb in sc["robots"]], steps=sc["steps"], comms_radius=12.0, target_speed=sc["target_speed"])
    events, moved = [], 0
    for tick in range(ticks):
        events.append(drv.tick())
        if migrate and tick % 10 == 9:
            live = np.nonzero(drv.alive)[0]
            new = sw.plan.owner.copy()
            if len(live) >= ws:
                new[live] = sharded.partition_strips(sw.read_variable_means(0)[live, :2], ws)
            moved += sw.migrate(new)
            assert sw.transport == mode, sw.transport  # (wired again as it was)
    ids, eta, lam, mu = sw.read_beliefs()
    exchanges = sw.world.halo_direct_status() if mode.startswith("direct") else 0  # (raises if one of them timed out)
    resident = [int(x) for x in sw.world.resident_stats()] if mode == "direct+resident" else [0, 0, 0]
    np.savez(out, moved=moved, exchanges=exchanges, resident=np.array(resident), ids=np.array(ids), eta=eta, lam=lam, mu=mu, events=np.array(events), translation=drv.translation,
             finished_at=drv.finished_at, next_number=drv.next_number)
    dist.barrier()
    if mode.startswith("direct"):
        sw.direct_close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
