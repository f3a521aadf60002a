import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from magics_amd import World, scenarios as S, sharded
import oracle

ws = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sc = S.grid_scenario(64, 10, interrobot=True, pitch=2.5, comm_radius=5.0)
streams = []
def make(params):
    st = torch.cuda.Stream(); streams.append(st)
    return World(params, stream=st.cuda_stream)
cluster = sharded.LocalCluster(sc, ws, make, direct=True)
for sw in cluster.ranks:
    p = sw.plan
    print("rank", p.rank, "send", [len(l) for l in p.send_lists], "recv", [len(l) for l in p.recv_lists])
ref = oracle.OracleWorld(sc["params"]); S.populate(ref, sc)
steps = [3]
for tick in range(3):
    t0 = time.time()
    cluster.iterate(steps); ref.iterate(steps)
    a = cluster.read_beliefs(); b = ref.read_beliefs()
    print("tick", tick, "identical", all(np.array_equal(x, y) for x, y in zip(a, b)), "%.2fs" % (time.time() - t0))
    for sw in cluster.ranks:
        try:
            print("  rank", sw.plan.rank, "exchanges", sw.world.halo_direct_status())
        except Exception as e:
            print("  rank", sw.plan.rank, "ERR", e)
