"""Diagnostics: are there stalls in a long run of back-to-back mgx_iterate calls?  Times every call on the host (the host runs a post
ahead of the device, so a call's duration follows the device's progress), prints the calls that took more than ten times the
median, how far apart they are, and the rate without them.  usage: python tools/iterate_stalls.py [calls] [iterate|tick]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if not os.environ.get("MGX_NO_TORCH"): import torch  # noqa
from magics_amd import World, scenarios as S
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
mode = sys.argv[2] if len(sys.argv) > 2 else "iterate"
sc = S.grid_scenario(1000, 16, interrobot=True)
w = World(sc["params"]); S.populate(w, sc)
tk = S.tick_inputs(sc)
steps = bytes(sc["steps"])
call = (lambda: w.iterate(steps)) if mode == "iterate" else (lambda: w.tick(steps=steps, **tk))
for _ in range(500): call()
w.synchronize()
t = np.empty(calls + 1)
t[0] = time.perf_counter()
for i in range(calls):
    call()
    t[i + 1] = time.perf_counter()
w.synchronize()
total = time.perf_counter() - t[0]
d = np.diff(t) * 1e6
med = float(np.median(d))
big = np.nonzero(d > float(os.environ.get("MGX_STALL_FACTOR", "10")) * med)[0]
print(f"{mode}: {calls} calls in {total * 1e3:.1f} ms = {total / calls / 10 * 1e6:.2f} us per iteration; median call {med:.1f} us; "
      f"{len(big)} calls over {10 * med:.0f} us, {d[big].sum() / 1e3:.1f} ms in all; without them {(total * 1e6 - d[big].sum() + len(big) * med) / calls / 10:.2f} us per iteration")
for i in big[:12]:
    print(f"   call {i}: {d[i] / 1e3:.2f} ms, at {(t[i] - t[0]) * 1e3:.1f} ms")
print("linger stats", w.linger_stats(), "resident stats", w.resident_stats())
