"""Where a device-side mission tick spends its host time (MGX_TIMING=1 prints the stages of mgx_mission_tick on stderr)."""
import os, sys, time
os.environ["MGX_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magics_amd import World, scenarios as S
from magics_amd.driver import DeviceDriver
n, K = 60, 12
sc = S.circle_scenario(n, K, circle_radius=60.0, n_internal=10, n_external=10)
sc["ir"] = []
w = World(sc["params"]); S.populate(w, sc)
d = DeviceDriver(w, n, K, waypoints=[[tuple(rb["goal"])] for rb in sc["robots"]], radii=[rb["radius"] for rb in sc["robots"]],
                 t0=[rb["t0"] for rb in sc["robots"]], steps=sc["steps"], comms_radius=20.0, target_speed=sc["target_speed"])
for _ in range(30): d.tick()
w.synchronize()
print("---- steady ticks ----", file=sys.stderr)
t0 = time.perf_counter()
for _ in range(3): d.tick()
w.synchronize()
print("3 ticks", (time.perf_counter() - t0) * 1e6 / 3, "us each; launches/tick", w.last_launch_count(), file=sys.stderr)
