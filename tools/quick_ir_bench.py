"""Experiments only: us / iteration of configs[2] for the engine build named by MGX_LIB (default: the product)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from magics_amd import World, scenarios as S
sc = S.grid_scenario(1000, 16, interrobot=True)
w = World(sc["params"]); S.populate(w, sc)
for _ in range(20): w.iterate(sc["steps"])
w.synchronize(); t0 = time.perf_counter()
n = 200
for _ in range(n): w.iterate(sc["steps"])
w.synchronize(); dt = time.perf_counter() - t0
print(os.environ.get("MGX_LIB", "product"), os.environ.get("MGX_PERSISTENT", ""), "launches/tick", w.last_launch_count(), "us/iter %.2f" % (dt / (n * 10) * 1e6))
