"""Diagnostics: us / iteration of the inter-robot workload (configs[2] shape) for the engine build named by MGX_LIB
(default: the product).  usage: python tools/quick_ir_bench.py [n_robots]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from magics_amd import World, scenarios as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sc = S.grid_scenario(n, 16, interrobot=True)
w = World(sc["params"]); S.populate(w, sc)
for _ in range(20): w.iterate(sc["steps"])
w.synchronize(); t0 = time.perf_counter()
reps = 200
for _ in range(reps): w.iterate(sc["steps"])
w.synchronize(); dt = time.perf_counter() - t0
print(os.environ.get("MGX_LIB", "product"), "persistent", os.environ.get("MGX_PERSISTENT", "1"), "robots", n, "launches/tick", w.last_launch_count(), "us/iter %.2f" % (dt / (reps * 10) * 1e6))
