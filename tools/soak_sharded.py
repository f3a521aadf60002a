"""Soak run of sharded worlds whose ghost records travel INSIDE resident schedule launches (mgx_halo_resident_*): random
grids on 2 .. 4 ranks inside this process (a stream per rank), random schedules (opening with an internal or an external
iteration, long and short), antenna / idle flags and prior changes on random robots, inter-robot factors switched off and on —
beliefs of the single-world oracle bit for bit after every script.
The ranks' launches of one schedule have to be on the device together; one process whose streams share a few hardware queues
(GPU_MAX_HW_QUEUES, default 4) cannot always give them that — they then find out by themselves, agree on "no" and the schedule
runs launch by launch (counted below as declined): both paths are soaked.
usage: python tools/soak_sharded.py [seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import oracle  # noqa: E402
from magics_amd import World, scenarios as S, sharded  # noqa: E402
from parity import assert_identical  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
t0, seed, done, resident_ticks, dropped, declined, failed = time.time(), 5000, 0, 0, 0, 0, 0
last = t0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    ws = int(rng.integers(2, 5))
    K = int(rng.choice([10, 12, 16, 21]))
    n = int(rng.integers(6, 40)) * ws
    sc = S.grid_scenario(n, K, interrobot=True, seed=seed, pitch=2.5, comm_radius=float(rng.choice([4.0, 5.0, 6.5])))
    if "POOL" not in globals():  # the same four streams for every script
        POOL = [torch.cuda.Stream() for _ in range(4)]
    streams = POOL[:ws]
    it = iter(streams)
    cluster = sharded.LocalCluster(sc, ws, lambda p: World(p, stream=next(it).cuda_stream), direct=True, resident=True)
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    on = sc["params"]["enable_mask"]
    ir_on = True
    history = []
    try:
      for step in range(10):
          op = int(rng.integers(0, 7))
          r = int(rng.integers(0, n))
          if op == 0:
              v = bool(rng.integers(0, 2))
              for w in (cluster, ref):
                  w.set_antenna(r, v)
          elif op == 1:
              v = bool(rng.integers(0, 2))
              for w in (cluster, ref):
                  w.set_idle(r, v)
          elif op == 2:
              var, m = int(rng.choice([0, K - 1])), rng.normal(size=4) * 3
              for w in (cluster, ref):
                  w.change_prior(r, var, m)
          elif op == 3 and rng.random() < 0.5:
              ir_on = not ir_on
              for w in (cluster, ref):
                  w.set_enabled(on if ir_on else on & ~S.EN_IR)
          steps = [int(x) for x in rng.integers(1, 4, size=int(rng.integers(1, 14)))]
          history.append((op, r, ir_on, steps))
          before = getattr(cluster, "declined", 0)
          for w in (cluster, ref):
              w.iterate(steps)
          if getattr(cluster, "declined", 0) != before:
              declined += 1
          elif cluster.resident and all(sw.world.last_launch_count() == 1 for sw in cluster.ranks):
              resident_ticks += 1
      for sw in cluster.ranks:
          sw.synchronize()
    except Exception as e:  # noqa: BLE001
        print(f"[soak sharded] seed {seed}: {ws} ranks, {n} robots x {K}, streams {[hex(st.cuda_stream) for st in streams]}: {type(e).__name__}: {str(e)[:120]}")
        print("   history:", history, flush=True)
        failed += 1
        del cluster
        seed += 1
        if failed >= 3:
            break
        continue
    if not all(np.isfinite(x).all() for x in ref.read_beliefs()):
        dropped += 1  # the oracle itself left the finite range: not reproduced from there on
    else:
        assert_identical(cluster, ref, what=f"seed {seed}: {ws} ranks, {n} robots x {K}")
        done += 1
    del cluster
    seed += 1
    if time.time() - last > 45:
        last = time.time()
        print(f"[soak sharded] {done} scripts identical, {resident_ticks} schedules as one launch per rank, {declined} declined, {dropped} dropped after {last - t0:.0f} s", flush=True)
print(f"soak sharded: {done} random scripts on 2 .. 4 ranks with ghost records inside resident launches ({resident_ticks} schedules ran as ONE "
      f"launch per rank, {declined} were declined by the ranks' agreement and ran launch by launch), seeds 5000..{seed - 1}, "
      f"{'all bit-identical to the single-world oracle' if not failed else str(failed) + ' FAILED'}; {dropped} more drove the oracle itself "
      f"to NaN / inf ({time.time() - t0:.0f} s)")
sys.exit(1 if failed else 0)
