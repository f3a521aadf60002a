"""Diagnostics: bench.py's dynamic_tick leg on its own (comms-range search + factor create / delete + mgx_tick, 1000 x 16), with the
two halves of a tick timed apart (host clocks around update_topology and around tick; the tick's launch runs on into the next
search) — for the build named by MGX_LIB (default: the product).  usage: python tools/dynamic_tick_bench.py [n_ticks] [blocks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
from magics_amd import World, scenarios as S
n_dyn = int(sys.argv[1]) if len(sys.argv) > 1 else 60
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sc = S.grid_scenario(1000, 16, interrobot=True, seed=805)
sc["ir"] = []
w = World(sc["params"]); S.populate(w, sc)
rng = np.random.default_rng(805)
base = np.array([[rb["pos"][0], 0.5, rb["pos"][1]] for rb in sc["robots"]], dtype=np.float32)
tk = S.tick_inputs(sc)
nxt, _, _ = w.update_topology(base, 8.0, 1)
w.iterate(sc["steps"]); w.synchronize()
out = []
per_tick = []
for b in range(blocks):
    poss = [base + rng.normal(0, 0.15, size=base.shape).astype(np.float32) for _ in range(n_dyn + 5)]
    for pos in poss[:5]:
        nxt, _, _ = w.update_topology(pos, 8.0, nxt); w.tick(steps=sc["steps"], **tk)
    w.synchronize()
    t_top = t_tick = 0.0
    t0 = time.perf_counter()
    for pos in poss[5:]:
        a = time.perf_counter(); nxt, c, d = w.update_topology(pos, 8.0, nxt); m = time.perf_counter()
        w.tick(steps=sc["steps"], **tk); e = time.perf_counter(); t_tick += e - m; t_top += m - a
        per_tick.append((e - a) * 1e6)
    w.synchronize()
    dt = time.perf_counter() - t0
    out.append(f"{n_dyn / dt:.0f}/s (topology {t_top / n_dyn * 1e6:.0f} us, tick {t_tick / n_dyn * 1e6:.0f} us)")
pt = np.array(per_tick)
print(f"per tick: median {np.median(pt):.0f} us, 90 % {np.percentile(pt, 90):.0f}, max {pt.max():.0f}; ticks over twice the median: {int((pt > 2 * np.median(pt)).sum())} of {len(pt)}")
st = w.linger_stats() if hasattr(w._L, "mgx_linger_stats") else None
print("resident launches / declined / back-off left", w.resident_stats(), end=" | ")
print(os.environ.get("MGX_LIB", "product"), "linger", os.environ.get("MGX_LINGER", "default"), "|", " | ".join(out), "| linger stats", st)
