"""Soak run of the comms-range neighbour search (all-pairs and hash-grid kernels) against the oracle's scan: random world
sizes, radii over six decades, uniform / clustered / lattice positions, a sprinkle of NaN / inf / huge / denormal / coincident
coordinates; CSR output identical.  usage: python tools/soak_neighbours.py [seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from magics_amd import hostlib  # noqa: E402
from test_gpu_topology import bare_pair, same_csr  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(424242)
t0, last, n_q = time.time(), time.time(), 0
worlds = {n: bare_pair(n) for n in (1, 2, 3, 17, 64, 65, 200, 1000, 2100)}
while time.time() - t0 < budget:
    n = int(rng.choice(list(worlds)))
    eng, ref, _ = worlds[n]
    kind = int(rng.integers(0, 4))
    span = float(10.0 ** rng.uniform(-1, 3))
    if kind == 0:
        pos = rng.uniform(-span, span, size=(n, 3))
    elif kind == 1:  # clusters: heavy buckets
        centres = rng.uniform(-span, span, size=(max(1, n // 40), 3))
        pos = centres[rng.integers(0, len(centres), size=n)] + rng.normal(0, span * 0.01, size=(n, 3))
    elif kind == 2:  # lattice: many distances exactly on the boundary
        step = float(rng.choice([0.5, 1.0, 2.0]))
        pos = np.stack([rng.integers(-8, 8, size=n) * step, np.zeros(n), rng.integers(-8, 8, size=n) * step], axis=1)
    else:
        pos = rng.normal(0, span, size=(n, 3))
    pos = pos.astype(np.float32)
    if rng.random() < 0.3 and n > 3:
        for v in (np.nan, np.inf, -np.inf, 3e38, 1e-30):
            if rng.random() < 0.5:
                pos[int(rng.integers(0, n)), int(rng.choice([0, 2]))] = v
        pos[int(rng.integers(0, n))] = pos[int(rng.integers(0, n))]
    radius = float(rng.choice([span * 10.0 ** rng.uniform(-3, 1), 1.0, 2.0, 0.0, -1.0, np.inf, np.nan, 1e30, 1e-30],
                              p=[0.6, 0.1, 0.1, 0.04, 0.04, 0.04, 0.04, 0.02, 0.02]))
    want = ref.neighbours(pos, radius)
    if len(want[1]) > 3_000_000:
        continue
    for method in (hostlib.NEIGHBOURS_PAIRS, hostlib.NEIGHBOURS_GRID, hostlib.NEIGHBOURS_AUTO):
        assert same_csr(eng.neighbours(pos, radius, method), want), (n, kind, span, radius, method)
    n_q += 1
    if time.time() - last > 45:
        last = time.time()
        print(f"[soak] {n_q} queries x 3 methods after {last - t0:.0f} s", flush=True)
print(f"soak: {n_q} random neighbour queries x 3 methods (all pairs, hash grid, auto), CSR identical to the oracle's scan ({time.time() - t0:.0f} s)")
