"""Worker of tests/test_gpu_topology.py::test_topology_differences_are_cross_checked: a world that follows its topology — churn on
a jittering grid, robots removed on the way, ticks and plain schedules mixed, a read-back (hence a pull) in between — on the
engine and on the oracle, in a process of its own so that MGX_CHECK_INDEX (read once per process) is on from the first topology
change: every block of differences the engine sends is compared with tables built from its whole connection list.
usage: python tests/topology_check_worker.py [robots] [ticks]   -> prints one summary line, exit code 0 = identical"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
os.environ["MGX_CHECK_INDEX"] = "1"
import numpy as np
import oracle
from magics_amd import scenarios as S
from parity import assert_identical, make_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 180
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 24
sc = S.grid_scenario(n, 12, interrobot=True, comm_radius=8.0)
sc["ir"] = []
eng, ref = make_pair(sc)
rng = np.random.default_rng(77)
base = np.array([[rb["pos"][0], 0.5, rb["pos"][1]] for rb in sc["robots"]], dtype=np.float32)
tk = S.tick_inputs(sc)
nxt_e = nxt_r = 1
made = gone = 0
for tick in range(ticks):
    pos = base + rng.normal(0, 0.25, size=base.shape).astype(np.float32)
    oe, orf = eng.update_topology(pos, 8.0, nxt_e), ref.update_topology(pos, 8.0, nxt_r)
    assert oe == orf, (tick, oe, orf)
    nxt_e, nxt_r = oe[0], orf[0]
    made, gone = made + oe[1], gone + oe[2]
    for w in (eng, ref):
        if tick % 3 == 2:
            w.iterate(sc["steps"])
        else:
            w.tick(steps=sc["steps"], **tk)
    if tick == ticks // 3:  # a robot leaves: its neighbours drop their factors in the passes that follow
        for w in (eng, ref):
            w.remove_robot(n // 2)
        tk = dict(tk)
        keep = np.asarray(tk["robots"]) != n // 2
        tk = {k: (np.asarray(v)[keep] if k in ("robots", "waypoints_xy", "time_scale", "what") else v) for k, v in tk.items()}
    if tick == ticks // 2:  # a read-back in the middle: the host mirror is refreshed from the layout the differences produced
        assert_identical(eng, ref, what=f"tick {tick}")
assert_identical(eng, ref, what="the end")
assert all(eng.connections(r) == ref.connections(r) for r in range(0, n, 7))
print(f"OK {n} robots, {ticks} ticks: {made} connections created, {gone} pairs deleted, every topology block cross-checked")
