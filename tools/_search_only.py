import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa
from magics_amd import World, scenarios as S
sc = S.grid_scenario(1000, 16, interrobot=True, seed=805)
sc["ir"] = []
w = World(sc["params"]); S.populate(w, sc)
rng = np.random.default_rng(805)
base = np.array([[rb["pos"][0], 0.5, rb["pos"][1]] for rb in sc["robots"]], dtype=np.float32)
w.synchronize()
ts = []
for i in range(40):
    pos = base + rng.normal(0, 0.15, size=base.shape).astype(np.float32)
    t0 = time.perf_counter()
    try:
        w.neighbours(pos, 8.0)
    except Exception as e:
        pass
    ts.append((time.perf_counter() - t0) * 1e6)
print("neighbours() median %.1f us" % np.median(ts))
