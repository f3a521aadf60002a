"""Soak run of LINGERING resident launches (include/mgx.h): seeded random worlds and random scripts of back-to-back ticks and
schedules of random shapes, pauses of random length (the launch waits out its bound, or almost), calls that end the launch
(prior changes, gating, read-backs), random bounds from 50 us to 20 ms.  The engine runs a whole script first — so that its
schedules do come back to back — then the oracle; beliefs and message counts bit-identical after every script.
usage: python tools/soak_linger.py [seconds]   -> progress lines and a one-line summary"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle  # noqa: E402
from magics_amd import World, scenarios as S  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
KS = [10, 12, 16, 16, 16, 21, 32]
tot = dict(scripts=0, calls=0, launches=0, posts=0, reruns=0, ended=0, diverged=0, ticks=0)
t0 = last = time.time()
seed = 5000
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    K = KS[seed % len(KS)]
    n = int(rng.choice([16, 36, 64, 100, 144, 256, 400]))
    sc = S.grid_scenario(n, K, interrobot=True, seed=seed, tracking=False)
    if seed % 3 == 0:  # ragged: a third of the directed connections gone (one-sided pairs among them)
        sc = dict(sc, ir=[c for c in sc["ir"] if rng.random() > 0.33])
    eng, ref = World(sc["params"]), oracle.OracleWorld(sc["params"], threads=8)
    assert S.populate(eng, sc) == S.populate(ref, sc)
    eng.set_linger(int(rng.choice([50, 300, 300, 2000, 20000])))
    tick = S.tick_inputs(sc)
    script = []
    for _ in range(int(rng.integers(10, 40))):
        op = rng.random()
        if op < 0.55:
            script.append(("tick", None))
        elif op < 0.75:
            m = int(rng.integers(1, 13))
            steps = [int(x) for x in rng.choice([1, 2, 3, 3, 3], size=m)]
            script.append(("iterate", steps))
        elif op < 0.83:
            script.append(("sleep", float(rng.choice([0.0001, 0.0004, 0.001, 0.004]))))
        elif op < 0.89:
            script.append(("prior", (int(rng.integers(0, n)), int(rng.choice([0, K - 1])), rng.normal(size=4) * 3)))
        elif op < 0.94:
            script.append(("antenna", (int(rng.integers(0, n)), bool(rng.integers(0, 2)))))
        elif op < 0.97:
            script.append(("idle", (int(rng.integers(0, n)), bool(rng.integers(0, 2)))))
        else:
            script.append(("read", int(rng.integers(0, n))))
    reads = {}
    for w in (eng, ref):
        for i, (op, arg) in enumerate(script):
            if op == "tick":
                w.tick(steps=sc["steps"], **tick)
            elif op == "iterate":
                w.iterate(arg)
            elif op == "sleep" and w is eng:
                time.sleep(arg)
            elif op == "prior":
                w.change_prior(*arg)
            elif op == "antenna":
                w.set_antenna(*arg)
            elif op == "idle":
                w.set_idle(*arg)
            elif op == "read":
                got = w.get_belief(arg, K // 2)
                if w is eng:
                    reads[i] = got
                else:
                    assert all(np.array_equal(reads[i][k], got[k], equal_nan=True) for k in ("mean", "cov", "eta", "lam")), (seed, i)
    la, po, rr, en = eng.linger_stats()
    eb, rb = eng.read_beliefs(), ref.read_beliefs()
    if not all(np.isfinite(x).all() for x in rb):
        tot["diverged"] += 1  # (the oracle's own beliefs left the finite range: documented not to be reproduced there)
    else:
        for name, a, b in zip(("eta", "lam", "mean"), eb, rb):
            assert np.array_equal(a, b), f"seed {seed}: {name} differs ({n} robots x {K}, linger stats {(la, po, rr, en)})"
        for r in range(0, n, max(1, n // 7)):
            assert eng.message_counts(r) == ref.message_counts(r), (seed, r)
    eng.synchronize()
    tot["scripts"] += 1; tot["calls"] += len(script); tot["launches"] += la; tot["posts"] += po; tot["reruns"] += rr; tot["ended"] += en
    tot["ticks"] += sum(1 for op, _ in script if op in ("tick", "iterate"))
    ref.close()
    del eng, ref
    seed += 1
    if time.time() - last > 45:
        last = time.time()
        print(f"[soak linger] {tot} after {last - t0:.0f} s (seed {seed})", flush=True)
print(f"soak linger: {tot['scripts']} random scripts ({tot['calls']} calls, {tot['ticks']} schedules), seeds 5000..{seed - 1}: {tot['launches']} launches "
      f"lingered, {tot['posts']} schedules were posted into them, {tot['reruns']} posts taken back and re-run as launches, {tot['ended']} launches had "
      f"ended by themselves when the next schedule came; beliefs, read-backs and message counts bit-identical to the oracle after every script; "
      f"{tot['diverged']} scripts drove the oracle itself to NaN / inf and were dropped there ({time.time() - t0:.0f} s)")
