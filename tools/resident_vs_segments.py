"""Diagnostics: the resident schedule launch against the launch-per-segment path of the SAME library, schedule prefix by
schedule prefix — where (after how many steps, on how many robots) the two part.  Both are held to the oracle by the
test-suite; this narrows a disagreement down without the oracle's minutes.
usage: python tools/resident_vs_segments.py [n_robots] [K] [ticks]      (MGX_LIB names another build)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
from magics_amd import World, scenarios as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sc = S.grid_scenario(n, K, interrobot=True)
bad_total = 0
for n_steps in list(range(2, len(sc["steps"]) + 1)):
    ws = []
    for resident in (1, 0):
        w = World(sc["params"])
        S.populate(w, sc)
        w.set_resident_launches(resident)
        for _ in range(ticks):
            w.iterate(sc["steps"][:n_steps])
        ws.append((w.read_beliefs(), w.last_launch_count()))
    (a, la), (b, lb) = ws
    bad = np.zeros(n, bool)
    for x, y in zip(a, b):
        d = ~((x == y) | (np.isnan(x) & np.isnan(y)))
        bad |= d.reshape(n, -1).any(axis=1)
    bad_total += int(bad.sum())
    print(f"steps {n_steps:2d} x {ticks} ticks: launches {la} vs {lb}; robots that differ: {int(bad.sum())}", np.flatnonzero(bad)[:12])
print("TOTAL", bad_total)
