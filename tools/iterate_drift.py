"""Diagnostics: does the time of an iteration depend on how long the world has been iterated?  Blocks of 2000 mgx_iterate calls
(20000 iterations each) at 1000 x 16 + inter-robot factors, us / iteration of every block, with and without prior updates
(mgx_tick) — and how many beliefs are still finite at the end.  usage: python tools/iterate_drift.py [blocks] [calls per block] [modes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
from magics_amd import World, scenarios as S
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 12
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 2000  # calls per block
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["iterate", "tick"]
for mode in modes:
    sc = S.grid_scenario(1000, 16, interrobot=True)
    w = World(sc["params"]); S.populate(w, sc)
    tk = S.tick_inputs(sc)
    call = (lambda: w.iterate(sc["steps"])) if mode == "iterate" else (lambda: w.tick(steps=sc["steps"], **tk))
    out = []
    for b in range(blocks):
        w.synchronize(); t0 = time.perf_counter()
        for _ in range(calls): call()
        w.synchronize(); out.append((time.perf_counter() - t0) / (10 * calls) * 1e6)
    eta, lam, mu = w.read_beliefs()
    fin = float(np.isfinite(mu).mean())
    tiny = float(((np.abs(eta) < 1e-300) & (eta != 0)).mean())
    print(mode, " ".join(f"{x:.2f}" for x in out), "| finite means", fin, "| |eta| < 1e-300", tiny, "| linger", w.linger_stats())
