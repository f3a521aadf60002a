"""Soak run of the environment rasteriser + blur: random environments (tests/test_gpu_env.py::_random_env) at random
resolutions / expansions / blur widths, device image vs the numpy restatement of env_to_png, byte for byte.
usage: python tools/soak_env.py [seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_env as T  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(99805)
t0, last, n = time.time(), time.time(), 0
while time.time() - t0 < budget:
    env = T._random_env(rng)
    T._same(env, int(rng.choice([7, 17, 33, 50, 64, 101, 128])), float(rng.uniform(0.0, 0.15)), float(rng.choice([0.0, 0.01, 0.03, 0.05, 0.1, 0.3, 1.0])))
    n += 1
    if time.time() - last > 45:
        last = time.time()
        print(f"[soak] {n} environments after {last - t0:.0f} s", flush=True)
print(f"soak: {n} random environments rasterised and blurred on the device, every image byte-identical to the oracle's ({time.time() - t0:.0f} s)")
