import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa
from magics_amd import World, scenarios as S
n = 1000
sc = S.grid_scenario(n, 16, interrobot=True)
for m in (1, 2, 3, 1, 3):
    w = World(sc["params"]); S.populate(w, sc)
    steps = sc["steps"] * m
    for _ in range(20): w.iterate(steps)
    w.synchronize(); t0 = time.perf_counter()
    reps = 240 // m
    for _ in range(reps): w.iterate(steps)
    w.synchronize(); dt = time.perf_counter() - t0
    print("ticks per call", m, "launches/call", w.last_launch_count(), "us/iter %.3f" % (dt / (reps * 10 * m) * 1e6), flush=True)
    del w
