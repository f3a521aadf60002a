"""Cost of a topology pass that CHANGES something (factors created / deleted) at 1000 robots x 16:
positions jitter every tick so that a few pairs cross the comms radius."""
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
nxt = 1
nxt, c, d = w.update_topology(base, 8.0, nxt)
w.iterate(sc["steps"])
w.synchronize()
print("initial pass: created", c)
tot_c = tot_d = 0
t_top = t_it = 0.0
for tick in range(30):
    pos = base + rng.normal(0, 0.15, size=base.shape).astype(np.float32)
    t0 = time.perf_counter()
    nxt, c, d = w.update_topology(pos, 8.0, nxt)
    w.synchronize()
    t1 = time.perf_counter()
    w.iterate(sc["steps"])
    w.synchronize()
    t2 = time.perf_counter()
    t_top += t1 - t0
    t_it += t2 - t1
    tot_c += c
    tot_d += d
print(f"per tick: topology pass {t_top / 30 * 1e3:.2f} ms (created {tot_c / 30:.1f}, deleted {tot_d / 30:.1f} per tick), "
      f"10 iterations {t_it / 30 * 1e3:.2f} ms")
