import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from magics_amd import World, scenarios as S
for name, sc in (("grid 500x32 +ir +trk", S.grid_scenario(500, 32, interrobot=True, tracking=True)),
                 ("junction 500x32 7x7", S.junction_scenario(500, 32, tiles=7)),
                 ("grid 500x21 +ir", S.grid_scenario(500, 21, interrobot=True))):
    w = World(sc["params"]); S.populate(w, sc)
    for _ in range(10): w.iterate(sc["steps"])
    w.synchronize(); t0 = time.perf_counter()
    for _ in range(100): w.iterate(sc["steps"])
    w.synchronize(); dt = time.perf_counter() - t0
    print(name, "persistent", os.environ.get("MGX_PERSISTENT", "1"), "launches", w.last_launch_count(), "us/iter %.2f" % (dt / (100 * len([s for s in sc["steps"] if s & 2 or True]) ) * 1e6 * len(sc["steps"]) / max(1, sum(1 for s in sc["steps"] if s in (1, 3)))))
