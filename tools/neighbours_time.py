"""Diagnostics: what one mgx_neighbours call costs on an idle device (1000 robots, radius 8) — launch + kernel + synchronisation."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from magics_amd import World, scenarios as S
sc = S.grid_scenario(1000, 16, interrobot=True, seed=805); sc["ir"] = []
w = World(sc["params"]); S.populate(w, sc)
base = np.array([[rb["pos"][0], 0.5, rb["pos"][1]] for rb in sc["robots"]], dtype=np.float32)
for _ in range(50): w.neighbours(base, 8.0)
w.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(300): w.neighbours(base, 8.0)
    print("mgx_neighbours on an idle device: %.1f us per call" % ((time.perf_counter() - t0) / 300 * 1e6))
