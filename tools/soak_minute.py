"""One minute of driver ticks posted into lingering launches at the headline size (1000 x 16 + inter-robot factors): mgx_tick back to
back, a synchronisation every `sync_every` ticks (which ends the launch; the next tick launches a new one).  Reports ticks, us per
iteration, the lingering statistics and that no wait on the device gave up.   usage: python tools/soak_minute.py [seconds] [sync_every]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from magics_amd import World, scenarios as S
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
every = int(sys.argv[2]) if len(sys.argv) > 2 else 500
sc = S.grid_scenario(1000, 16, interrobot=True)
w = World(sc["params"]); S.populate(w, sc)
tick = S.tick_inputs(sc)
t0, n = time.time(), 0
while time.time() - t0 < seconds:
    for _ in range(every):
        w.tick(steps=sc["steps"], **tick)
    w.synchronize()  # raises if a wait inside a launch gave up
    n += every
dt = time.time() - t0
print(f"OK {n} ticks in {dt:.1f} s: {dt / (n * 10) * 1e6:.2f} us per iteration; lingering launches / posts / re-run / ended by the device: "
      f"{w.linger_stats()}; resident launches / declined / back-off left: {w.resident_stats()}")
