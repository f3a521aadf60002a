"""Per-GPU shapes of all BASELINE.json configs on one MI355X (informational, next to bench.py):
microseconds per GBP iteration."""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402,F401
from magics_amd import World, scenarios as S  # noqa: E402


def run(sc, n_ticks):
    w = World(sc["params"])
    S.populate(w, sc)
    for _ in range(5):
        w.iterate(sc["steps"])
    w.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_ticks):
        w.iterate(sc["steps"])
    w.synchronize()
    return (time.perf_counter() - t0) / (n_ticks * len(sc["steps"])) * 1e6


for name, sc, ticks in (
    ("configs[0] circle, 10 robots x 10", S.circle_scenario(10, 10), 20),
    ("configs[1] 1000 x 16 dyn+obs", S.grid_scenario(1000, 16, interrobot=False), 100),
    ("configs[2] 1000 x 16 +inter-robot", S.grid_scenario(1000, 16, interrobot=True), 30),
    ("configs[3] per-GPU share: 1000 x 16 +inter-robot (8000 on 8 GPUs)", S.grid_scenario(1000, 16, interrobot=True), 30),
    ("configs[4] per-GPU share: 500 x 32 +inter-robot +tracking (4000 on 8 GPUs)", S.grid_scenario(500, 32, interrobot=True, tracking=True), 30),
    ("configs[4] whole on one GPU: 4000 x 32 +inter-robot +tracking", S.grid_scenario(4000, 32, interrobot=True, tracking=True), 10),
    ("configs[4] as SURVEY §8d words it: 20 x 20 crossroads rasterised on the device, 4000 x 32 on the lanes +inter-robot +tracking",
     S.junction_scenario(4000, 32), 10),
    ("configs[4] per-GPU share of that: 500 x 32 on 7 x 7 crossroads", S.junction_scenario(500, 32, tiles=7), 30),
):
    print(f"{name}: {run(sc, ticks):.2f} us / iteration ({len(sc['steps'])}-step schedule)")
