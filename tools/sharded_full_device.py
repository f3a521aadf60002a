"""Diagnostics: does the SHARDED resident kernel (k_robot_sweep<K, 2, true, shard>: more registers, scratch) hold a GPU's full share
of robots at once?  Two ranks of N / 2 robots inside ONE process (a stream each), wired direct + resident: N + 2 workgroups of the
sharded instantiation on the one device of a development box — what every rank of a real node launches alone on its GPU.
usage: python tools/sharded_full_device.py [n_robots_total] [K] [ticks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from magics_amd import World, scenarios as S, sharded

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 100
first = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # robots of rank 0 (0: equal strips)
sc = S.grid_scenario(n, K, interrobot=True)
owner = None
if first:
    import numpy as np
    order = np.lexsort((np.asarray(sc["positions"])[:, 0], np.asarray(sc["positions"])[:, 1]))  # (y, x) order, like the strips
    owner = np.ones(n, dtype=np.int64)
    owner[order[:first]] = 0
streams = [torch.cuda.Stream() for _ in range(2)]
it = iter(streams)
cluster = sharded.LocalCluster(sc, 2, lambda p: World(p, stream=next(it).cuda_stream), direct=True, resident=True, owner=owner)
print("resident wiring:", cluster.resident, [len(sw.plan.local) for sw in cluster.ranks], "ghosts", [len(sw.plan.ghosts) for sw in cluster.ranks])
for _ in range(10):
    cluster.iterate(sc["steps"])
for sw in cluster.ranks:
    sw.synchronize()
t0 = time.perf_counter()
for _ in range(ticks):
    cluster.iterate(sc["steps"])
for sw in cluster.ranks:
    sw.synchronize()
dt = time.perf_counter() - t0
print(f"{n} robots on 2 in-process ranks: {dt / (ticks * len(sc['steps'])) * 1e6:.2f} us per iteration; launches per tick "
      f"{[sw.world.last_launch_count() for sw in cluster.ranks]}; resident launches / declined / back-off left "
      f"{[tuple(int(x) for x in sw.world.resident_stats()) for sw in cluster.ranks]}; schedules declined (cluster) {getattr(cluster, 'declined', 0)}")
