"""Times the comms-range neighbour search (mgx_neighbours) at several world
sizes: all-pairs vs hash-grid kernel, host wall time per call including the CSR download."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from magics_amd import World, hostlib, scenarios as S  # noqa: E402


def main():
    for n in (1000, 10000, 100000):
        sc = S.grid_scenario(4, 10, interrobot=False, obstacles=False)
        w = World(sc["params"])
        rb = sc["robots"][0]
        for r in range(n):
            w.add_robot(rb["mean0"], rb["prior_diag"], rb["dt"], rb["radius"], order_key=r)
        side = np.sqrt(n) * 5.0
        rng = np.random.default_rng(n)
        pos = np.stack([rng.uniform(0, side, n), np.full(n, 0.5), rng.uniform(0, side, n)], axis=1).astype(np.float32)
        radius = 8.0
        for name, method in (("pairs", hostlib.NEIGHBOURS_PAIRS), ("grid", hostlib.NEIGHBOURS_GRID)):
            if method == hostlib.NEIGHBOURS_PAIRS and n > 20000:
                reps = 2
            else:
                reps = 20
            ptr, idx = w.neighbours(pos, radius, method)
            t0 = time.perf_counter()
            for _ in range(reps):
                w.neighbours(pos, radius, method)
            dt = (time.perf_counter() - t0) / reps
            print(f"n={n:6d} {name:5s}: {dt * 1e3:8.3f} ms/call (2 passes: size + fill), mean degree {len(idx) / n:.1f}")


if __name__ == "__main__":
    main()
