"""What re-balancing buys on a mission where robots leave their strips: records exchanged per tick (sum over ranks of the send lists)
with ownership fixed at the initial strips against ownership that follows the positions (LocalCluster.migrate every M ticks).
usage: python tools/migration_traffic.py [n_robots] [world_size] [M]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magics_amd import World, scenarios as S, sharded  # noqa: E402
from magics_amd.driver import Driver  # noqa: E402


def run(n, ws, every):
    K = 10
    sc = S.circle_scenario(n, K, circle_radius=max(12.0, n * 0.6), n_internal=10, n_external=10)
    sc["ir"] = []
    cluster = sharded.LocalCluster(sc, ws, World, dynamic=True)
    drv = Driver(cluster, n, K, waypoints=[[tuple(rb["goal"])] for rb in sc["robots"]], radii=[rb["radius"] for rb in sc["robots"]],
                 t0=[rb["t0"] for rb in sc["robots"]], steps=sc["steps"], comms_radius=12.0, target_speed=sc["target_speed"])
    sent, cross, moved, ticks = 0, 0, 0, 0
    for tick in range(600):
        if not (drv.finished_at < 0).any():
            break
        drv.tick()
        ticks += 1
        sent += sum(sum(sw._send_counts_robots) for sw in cluster.ranks)
        owner = cluster.ranks[0].plan.owner
        cross += sum(1 for r in np.nonzero(drv.alive)[0] for q in cluster.connections(int(r)) if owner[int(q)] != owner[int(r)])
        if every and tick % every == every - 1:
            live = np.nonzero(drv.alive)[0]
            new = owner.copy()
            if len(live) >= ws:
                new[live] = sharded.partition_strips(cluster.read_variable_means(0)[live, :2], ws)
            moved += cluster.migrate(new)
    return dict(ticks=ticks, records_sent=sent, cross_rank_connections=cross, migrations=moved)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    ws = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    fixed, moving = run(n, ws, 0), run(n, ws, every)
    print(f"{n} robots crossing a circle on {ws} ranks")
    print("ownership fixed at the initial strips:   ", fixed)
    print(f"ownership follows positions (every {every}):", moving)
