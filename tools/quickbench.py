"""Quick A/B of engine builds on the GPU box: prints us/iteration for config 2 and config 3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from magics_amd import World, scenarios as S
def run(sc, n_ticks=100):
    w = World(sc["params"]); S.populate(w, sc)
    for _ in range(10): w.iterate(sc["steps"])
    w.synchronize(); t0 = time.perf_counter()
    for _ in range(n_ticks): w.iterate(sc["steps"])
    w.synchronize(); dt = time.perf_counter() - t0
    return dt / (n_ticks * len(sc["steps"])) * 1e6
sc2 = S.grid_scenario(1000, 16, interrobot=False)
sc3 = S.grid_scenario(1000, 16, interrobot=True)
print(os.environ.get("MGX_LIB", "default"), "config2 us/iter %.2f" % run(sc2), "config3 us/iter %.2f" % run(sc3, 30))
