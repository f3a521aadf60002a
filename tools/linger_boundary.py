"""Diagnostics: what a plan boundary of a lingering launch costs.  Back-to-back mgx_iterate calls of 10-, 20- and 30-step schedules
(11, 21, 31 segments) at 1000 x 16 + inter-robot factors: T(n) = n c + b  ->  c (us per iteration), b (us per boundary)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from magics_amd import World, scenarios as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sc = S.grid_scenario(n, 16, interrobot=True)
w = World(sc["params"]); S.populate(w, sc)
tk = S.tick_inputs(sc)
for mode in ("iterate", "tick"):  # (tick: the same schedules with the prior updates of all robots riding in every post)
    res = {}
    for mult in (1, 2, 3, 1, 2, 3):
        steps = bytes(sc["steps"] * mult)
        call = (lambda: w.iterate(steps)) if mode == "iterate" else (lambda: w.tick(steps=steps, **tk))
        reps = 600 // mult
        for _ in range(30): call()
        w.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): call()
        w.synchronize(); dt = (time.perf_counter() - t0) / reps * 1e6
        res[mult] = min(res.get(mult, 1e9), dt)
    c = (res[3] - res[1]) / 20.0
    b = res[1] - 10 * c
    print(os.environ.get("MGX_LIB", "product"), "linger", os.environ.get("MGX_LINGER", "default"), mode, {k: round(v, 2) for k, v in res.items()},
          "us/iteration %.3f  us/boundary %.2f" % (c, b), w.linger_stats() if hasattr(w._L, "mgx_linger_stats") else None)
