"""An in-process cluster of MANY ranks whose schedules run as resident launches (tests/test_gpu_sharded.py): a process of its own
because the hardware queues a process gets are fixed when HIP starts (GPU_MAX_HW_QUEUES, default 4) and ranks whose streams share
a queue cannot have their launches on the device together.  usage: resident_cluster_worker.py RANKS ROBOTS K OUT.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ws, n, K, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch
    import oracle
    from magics_amd import World, scenarios as S, sharded
    from parity import assert_identical
    sc = S.grid_scenario(n, K, interrobot=True, pitch=2.5, comm_radius=5.0)
    streams = [torch.cuda.Stream() for _ in range(ws)]
    it = iter(streams)
    cluster = sharded.LocalCluster(sc, ws, lambda p: World(p, stream=next(it).cuda_stream), direct=True, resident=True)
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    one_launch = 0
    scripts = [sc["steps"], [2, 3, 3, 1, 3], sc["steps"], sc["steps"] + [1, 1, 2, 3, 2]]
    for tick, steps in enumerate(scripts):
        declined = getattr(cluster, "declined", 0)
        cluster.iterate(steps)
        ref.iterate(steps)
        for sw in cluster.ranks:
            sw.synchronize()
        if getattr(cluster, "declined", 0) == declined and all(sw.world.last_launch_count() == 1 for sw in cluster.ranks):
            one_launch += 1
        assert_identical(cluster, ref, what=f"{ws} ranks in one process, tick {tick}")
    json.dump(dict(resident=bool(cluster.resident), one_launch=one_launch, declined=getattr(cluster, "declined", 0), schedules=len(scripts),
                   ghosts=[len(sw.plan.ghosts) for sw in cluster.ranks]), open(out, "w"))


if __name__ == "__main__":
    main()
