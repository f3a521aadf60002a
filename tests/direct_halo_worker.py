"""One rank of the multi-process direct-halo test (tests/test_gpu_direct_halo_mp.py): every rank
is its own process on cuda:0, the control plane is gloo, the data plane is hipIpc-mapped
peer stores.  usage: direct_halo_worker.py RANK WORLD_SIZE PORT OUT.npz [direct | direct+resident | direct+resident+decline | direct+resident+late | direct+resident+batch]
(direct+resident: the ghost records travel inside ONE resident launch per schedule and rank; +batch: two more ticks inside
mgx_batch_begin / mgx_batch_end on every rank — submitted together, merged into as few launches as their segments fit)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, ws, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    mode = sys.argv[5] if len(sys.argv) > 5 else "direct"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(ws))
    import numpy as np
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    torch.cuda.set_device(0)
    from magics_amd import World, scenarios as S, sharded
    sc = S.grid_scenario(64, 10, interrobot=True, pitch=2.5, comm_radius=5.0)
    comm = sharded.TorchDistComm()
    sw = sharded.ShardedWorld(sc, rank, ws, World, comm=comm)
    decline = mode.endswith("+decline")  # the last rank says no to the resident launch of tick 1: every rank falls back
    late = mode.endswith("+late")        # the last rank starts tick 1 long after the others have given up waiting for it
    batch = mode.endswith("+batch")
    mode = mode.replace("+decline", "").replace("+late", "").replace("+batch", "")
    got = sharded.connect(sw, comm, resident=mode == "direct+resident")  # the default wiring: in-engine transports first
    assert got == mode, got
    steps = sc["steps"] + [1, 1, 2, 3, 2]
    boundary = sorted({g for r in range(ws) for g in sharded.ShardPlan(sc, r, ws).ghosts})
    for tick in range(3):
        if tick == 1:
            sw.set_antenna(boundary[0], False)
            sw.change_prior(boundary[2], 9, np.array([0.5, 0.25, 1.0, -1.0]))
        if tick == 2:
            sw.set_antenna(boundary[0], True)
        if decline and rank == ws - 1:
            sw.world.set_resident_launches("decline" if tick == 1 else True)
        if late and tick == 1:
            sw.synchronize()
            dist.barrier()
            if rank == ws - 1:
                import time
                time.sleep(0.3)
        sw.iterate(steps)
        if (decline or late) and tick == 0:
            assert sw.world.resident_stats()[:2] == (1, 0), sw.world.resident_stats()
    batched = (0, 0)
    if batch:
        with sw.batch() as b:
            sw.iterate(sc["steps"])
            sw.iterate(sc["steps"])
        batched = (b.schedules, b.launches)
    launches = sw.world.last_launch_count()
    stats = sw.world.resident_stats()
    ids, eta, lam, mu = sw.read_beliefs()
    n = sw.world.halo_direct_status()
    np.savez(out, batched=np.array(batched), ids=np.array(ids), eta=eta, lam=lam, mu=mu, n=n, launches=launches, stats=np.array(stats, dtype=np.int64))
    dist.barrier()
    sw.direct_close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
