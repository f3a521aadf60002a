"""Diagnostics: the inter-robot workload (configs[2] shape) issued tick after tick — us / iteration of back-to-back mgx_iterate
calls and driver ticks / s of back-to-back mgx_tick calls — for the engine build named by MGX_LIB (default: the product);
MGX_LINGER=0 keeps every schedule a launch of its own.  usage: python tools/quick_tick_bench.py [n_robots] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from magics_amd import World, scenarios as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
sc = S.grid_scenario(n, 16, interrobot=True)
w = World(sc["params"]); S.populate(w, sc)
tick = S.tick_inputs(sc)
out = []
for name, call in (("iterate", lambda: w.iterate(sc["steps"])), ("tick", lambda: w.tick(steps=sc["steps"], **tick))):
    best = 1e9
    for _ in range(3):
        for _ in range(20): call()
        w.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): call()
        w.synchronize(); best = min(best, time.perf_counter() - t0)
    out.append(f"{name} {best / reps * 1e6:.1f} us/call ({reps / best:.0f}/s, {best / (reps * 10) * 1e6:.2f} us/iter)")
st = w.linger_stats() if hasattr(w._L, "mgx_linger_stats") else None
print(os.environ.get("MGX_LIB", "product"), "linger", os.environ.get("MGX_LINGER", "default"), "robots", n, "|", " | ".join(out), "| linger stats", st)
