"""How the sweep kernel's throughput grows with the number of robots on one GPU (configs[1]
workload, 16 horizon): robot-iterations per second and time per iteration."""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402,F401
from magics_amd import World, scenarios as S  # noqa: E402

for n in (250, 500, 1000, 2000, 4000, 8000, 16000):
    sc = S.grid_scenario(n, 16, interrobot=False)
    w = World(sc["params"])
    S.populate(w, sc)
    for _ in range(5):
        w.iterate(sc["steps"])
    w.synchronize()
    ticks = max(10, 40000 // n)
    t0 = time.perf_counter()
    for _ in range(ticks):
        w.iterate(sc["steps"])
    w.synchronize()
    dt = (time.perf_counter() - t0) / (ticks * len(sc["steps"]))
    print(f"{n:6d} robots: {dt * 1e6:8.2f} us / iteration, {n / dt / 1e6:8.1f} M robot-iterations/s")
