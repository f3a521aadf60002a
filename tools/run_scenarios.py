"""Runs the reference's scenarios (parsed form: tests/golden/scenarios.json, or directories given on the command
line) headless on the engine and prints what the reference's export would say about them: makespan, robots
finished, distance travelled.  usage: python tools/run_scenarios.py [--max-time S] [scenario-dir | name ...]"""
import json
import os
import sys
import time

sys.path.insert(0, ".")
from magics_amd import World, config, sim  # noqa: E402

args = sys.argv[1:]
max_time = 60.0
if "--max-time" in args:
    i = args.index("--max-time")
    max_time = float(args[i + 1])
    del args[i:i + 2]
with open(os.path.join("tests", "golden", "scenarios.json"), encoding="utf-8") as f:
    known = json.load(f)
names = args or [n for n, sc in sorted(known.items())
                 if all(f["planning-strategy"] == "only-local" for f in sc["formation"]["formations"]) and sum(f["robots"] for f in sc["formation"]["formations"])]
for name in names:
    sc = config.load_scenario(name) if os.path.isdir(name) else known[name]
    t0 = time.perf_counter()
    s = sim.Simulation(sc, World(config.world_params(sc["config"])))
    s.run(max_time=max_time)
    wall = time.perf_counter() - t0
    done = [r for r in s.robots if r["completed"]]
    trav = [r["travelled"] for r in s.robots]
    print(json.dumps({"scenario": sc.get("name", name), "simulated_s": round(s.elapsed(), 1), "wall_s": round(wall, 2), "robots": len(s.robots),
                      "finished": len(done), "all_finished": s.finished(), "K": s.K,
                      "mean_distance": round(sum(trav) / max(1, len(trav)), 1),
                      "last_finish_s": round(max((r["finished_at"] for r in done), default=0.0), 1),
                      "topology_events": len(s.events)}), flush=True)
