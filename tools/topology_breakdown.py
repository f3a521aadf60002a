"""Where a topology-changing tick spends its time at 1000 robots x 16 (diagnostic): neighbour search alone, the whole
pass (search + connection bookkeeping), the table rebuild the next device call triggers, the tick itself."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402,F401
from magics_amd import World, scenarios as S  # noqa: E402

sc = S.grid_scenario(1000, 16, interrobot=True, comm_radius=8.0)
sc["ir"] = []
w = World(sc["params"])
S.populate(w, sc)
rng = np.random.default_rng(0)
base = np.array([[rb["pos"][0], 0.5, rb["pos"][1]] for rb in sc["robots"]], dtype=np.float32)
tk = S.tick_inputs(sc)
nxt, c, d = w.update_topology(base, 8.0, 1)
w.tick(steps=sc["steps"], **tk)
w.synchronize()
T = {"search": 0.0, "pass": 0.0, "rebuild": 0.0, "tick": 0.0}
n = 40
for _ in range(n):
    pos = base + rng.normal(0, 0.15, size=base.shape).astype(np.float32)
    t0 = time.perf_counter()
    w.neighbours(pos, 8.0)
    t1 = time.perf_counter()
    nxt, c, d = w.update_topology(pos, 8.0, nxt)
    t2 = time.perf_counter()
    w.read_variable_means(0)  # first device call after the pass: rebuilds the edge tables
    t3 = time.perf_counter()
    w.tick(steps=sc["steps"], **tk)
    w.synchronize()
    t4 = time.perf_counter()
    T["search"] += t1 - t0; T["pass"] += t2 - t1; T["rebuild"] += t3 - t2; T["tick"] += t4 - t3
w.read_variable_means(0)
t0 = time.perf_counter()
for _ in range(100):
    w.read_variable_means(0)
base_read = (time.perf_counter() - t0) / 100
print({k: round(v / n * 1e6, 1) for k, v in T.items()}, "us; read_variable_means alone", round(base_read * 1e6, 1), "us")
