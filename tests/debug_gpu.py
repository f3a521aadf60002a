import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle
from magics_amd import World, scenarios as S
from parity import make_pair, errors

def run(name, sc, script, strict=True):
    eng, ref = make_pair(sc, strict=strict)
    script(eng); script(ref)
    ee, le, me = eng.read_beliefs(); er, lr, mr = ref.read_beliefs()
    d = np.abs(me - mr).max(axis=1)
    from parity import max_abs_diff
    print(name, 'strict' if strict else 'fast', 'maxabs', max_abs_diff(eng, ref), 'err', errors(eng, ref), 'worst var', int(d.argmax()), d.max())
    return me, mr

for strict in (True, False):
  for K in (10, 16):
    for obs in (False, True):
        sc = S.grid_scenario(4, K, interrobot=False, obstacles=obs)
        for n in (1, 2, 3, 10, 40):
            run(f'K{K} obs{obs} fused{n}', sc, lambda w: w.iterate([1] * n), strict)
  sc = S.grid_scenario(36, 10, interrobot=True, pitch=1.5, comm_radius=4.0)
  for n in (1, 2, 3, 10, 30):
      run(f'dense ir x{n}', sc, lambda w: w.iterate([3] * n), strict)
  sc = S.grid_scenario(12, 32, interrobot=True, tracking=True, pitch=3.0, comm_radius=6.0)
  for n in (5, 12, 20, 60):
      run(f'K32 trk x{n}', sc, lambda w: w.iterate([3] * n), strict)
