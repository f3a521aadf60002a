"""Diagnostics: where the wall time of bench.py's scenario leg goes (cProfile over Simulation.run of a reference scenario).
usage: python tools/scenario_profile.py [scenario name] [simulated seconds]"""
import cProfile, io, json, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from magics_amd import World, config, sim
name = sys.argv[1] if len(sys.argv) > 1 else "Junction Twoway"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
with open(os.path.join(ROOT, "tests", "golden", "scenarios.json"), encoding="utf-8") as f:
    sc = json.load(f)[name]
for rep in range(2):
    s = sim.Simulation(sc, World(config.world_params(sc["config"])))
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    if rep: pr.enable()
    s.run(max_time=secs); s.w.synchronize()
    if rep: pr.disable()
    print(f"run {rep}: {s.tick_no} ticks in {time.perf_counter() - t0:.4f} s, {len(s.robots)} robots")
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(22)
print(out.getvalue()[:6000])
