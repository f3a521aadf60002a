"""Soak run of the reference's local-planning scenarios (parsed form: tests/golden/scenarios.json), engine and oracle side
by side for as long as a wall-clock budget per scenario allows (the oracle is the slow side): same spawns, topology events,
trajectories and beliefs bit for bit, same export.
usage: python tools/soak_scenarios.py [seconds per scenario] [ranks]     (ranks > 1: the engine side is a world sharded over
that many ranks in this process, following its topology — magics_amd.sharded.LocalCluster)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from magics_amd import World, config, sim  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 1
with open(os.path.join(ROOT, "tests", "golden", "scenarios.json"), encoding="utf-8") as f:
    known = json.load(f)
names = [n for n, sc in sorted(known.items())
         if all(f["planning-strategy"] == "only-local" for f in sc["formation"]["formations"]) and sum(f["robots"] for f in sc["formation"]["formations"])]
print("| scenario | K | simulated s | ticks | robots spawned | finished | topology events | identical |")
print("|---|---|---|---|---|---|---|---|", flush=True)
for name in names:
    sc = known[name]
    p = config.world_params(sc["config"])
    if ranks > 1:
        from magics_amd import sharded
        engine = sharded.LocalCluster(dict(params=p, robots=[], ir=[], K=None), ranks, World, dynamic=True)
    else:
        engine = World(p)
    a, b = sim.Simulation(sc, engine), sim.Simulation(sc, oracle.OracleWorld(p))
    limit = sc["config"]["simulation"]["max-time"]
    t0, ok = time.time(), True
    while time.time() - t0 < budget and not a.finished() and a.elapsed() < limit:
        a.tick()
        b.tick()
        if a.tick_no % 10 == 0:
            ok = ok and len(a.robots) == len(b.robots) and np.array_equal(a.translation, b.translation)
            if ok and a.robots:
                ok = all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a.w.read_beliefs(), b.w.read_beliefs()))
            if not ok:
                break
    ok = ok and a.events == b.events
    if ranks == 1:  # message counts (part of the export) are kept for unsharded worlds only
        ok = ok and json.dumps(a.export(), sort_keys=True) == json.dumps(b.export(), sort_keys=True)
    print(f"| {name} | {a.K} | {a.elapsed():.1f} | {a.tick_no} | {len(a.robots)} | {sum(1 for r in a.robots if r['completed'])} | "
          f"{len(a.events)} | {'yes' if ok else 'NO (tick %d)' % a.tick_no} |", flush=True)
