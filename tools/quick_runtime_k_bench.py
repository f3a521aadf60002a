"""Diagnostics: what a horizon WITHOUT a constant-K instantiation costs (the run-time-K kernels k_robot_sweep<0, ...>) next to
its constant-K neighbours.  usage: python tools/quick_runtime_k_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from magics_amd import World, scenarios as S
for K in (32, 33, 34, 35):
    for ir in (False, True):
        try:
            sc = S.grid_scenario(500, K, interrobot=ir)
        except Exception as e:  # noqa: BLE001
            print("K", K, "scenario not available:", type(e).__name__, e)
            break
        w = World(sc["params"]); S.populate(w, sc)
        steps = sc["steps"] if ir else [1] * 10
        for _ in range(10): w.iterate(steps)
        w.synchronize(); t0 = time.perf_counter()
        for _ in range(100): w.iterate(steps)
        w.synchronize(); dt = time.perf_counter() - t0
        print("K", K, "inter-robot" if ir else "dyn+obs    ", "launches/call", w.last_launch_count(), "us/iter %.2f" % (dt / 1000 * 1e6))
