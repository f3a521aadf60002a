"""Seeded random scripts against the oracle: horizon lengths that hit every kernel instantiation
(constant-K templates 10 / 12 / 16 / 21 / 32 and the runtime-K fallback), ragged inter-robot
topologies (robots with no, few and many neighbours), random gating, prior changes, prior-update
ticks, late connections and disconnections.  Beliefs must stay bit-identical throughout."""
import numpy as np
import pytest

from magics_amd import scenarios as S
from magics_amd import hostlib

from parity import assert_identical, make_pair

pytestmark = pytest.mark.gpu


def _scenario(K, n, seed, tracking=False):
    if K in S.HORIZON_FOR_K:
        return S.grid_scenario(n, K, interrobot=True, tracking=tracking, seed=seed, pitch=2.5, comm_radius=4.5)
    # horizon lengths outside the table: build timesteps by hand (runtime-K kernel)
    S.HORIZON_FOR_K[K] = {5: 4, 7: 9, 8: 10, 13: 30, 19: 60}[K]
    try:
        return S.grid_scenario(n, K, interrobot=True, tracking=tracking, seed=seed, pitch=2.5, comm_radius=4.5)
    finally:
        del S.HORIZON_FOR_K[K]


@pytest.mark.parametrize("K,n,seed", [(5, 7, 1), (8, 9, 2), (10, 12, 3), (12, 10, 4), (13, 6, 5), (16, 14, 6), (19, 5, 7),
                                      (21, 8, 8), (32, 6, 9)])
def test_random_script(K, n, seed):
    sc = _scenario(K, n, seed, tracking=(seed % 3 == 0))
    # ragged topology: drop a random third of the directed connections, isolate robot 0 entirely
    rng = np.random.default_rng(seed)
    ir = [c for c in sc["ir"] if rng.random() > 0.33 and 0 not in c[:2]]
    dropped = [c for c in sc["ir"] if c not in ir and 0 not in c[:2]]
    sc = dict(sc, ir=ir)
    eng, ref = make_pair(sc)
    tick = S.tick_inputs(sc)
    script = []
    for step in range(14):
        op = rng.integers(0, 8)
        if op == 0:
            r = int(rng.integers(0, n)); v = bool(rng.integers(0, 2))
            script.append(lambda w, r=r, v=v: w.set_antenna(r, v))
        elif op == 1:
            r = int(rng.integers(0, n)); v = bool(rng.integers(0, 2))
            script.append(lambda w, r=r, v=v: w.set_idle(r, v))
        elif op == 2:
            r = int(rng.integers(0, n)); var = int(rng.choice([0, K - 1])); m = rng.normal(size=4) * 3
            script.append(lambda w, r=r, var=var, m=m: w.change_prior(r, var, m))
        elif op == 3 and dropped:
            a, b, n0 = dropped.pop()
            script.append(lambda w, a=a, b=b, n0=n0: w.ir_connect(a, b, n0 + 100000))
        elif op == 4 and ir:
            a, b, _ = ir[int(rng.integers(0, len(ir)))]
            def disc(w, a=a, b=b):
                w.ir_disconnect(a, b)
            script.append(disc)
            ir = [c for c in ir if set(c[:2]) != {a, b}]
        elif op == 5:
            script.append(lambda w: w.update_priors(**tick))
        steps = [int(x) for x in rng.integers(1, 4, size=int(rng.integers(1, 6)))]
        script.append(lambda w, steps=steps: w.iterate(steps))
    for k, f in enumerate(script):
        f(eng)
        f(ref)
        if k % 5 == 4:
            assert_identical(eng, ref, what=f"K={K} n={n} seed={seed} after op {k}")
    assert_identical(eng, ref, what=f"K={K} n={n} seed={seed} final")


def test_many_neighbours_falls_back_to_unstaged_messages():
    # a dense cluster gives a robot more inter-robot edges than LDS staging allows (> 64 KB per
    # workgroup): the kernel variant that reads the messages from L2 must give the same beliefs
    sc = S.grid_scenario(49, 16, interrobot=True, pitch=1.2, comm_radius=20.0, obstacles=False)
    per_robot = max(sum(1 for c in sc["ir"] if c[1] == r) for r in range(49)) * 15
    assert per_robot * 21 * 8 > 64 * 1024
    eng, ref = make_pair(sc)
    for w in (eng, ref):
        w.iterate([3, 3, 1, 3])
    assert_identical(eng, ref, what="unstaged inter-robot messages")


def test_all_schedule_kinds_full_tick_lengths():
    sc = S.grid_scenario(16, 10, interrobot=True, pitch=2.5, comm_radius=4.5)
    for kind in range(5):
        for ni, ne in ((10, 10), (50, 10), (3, 9)):
            eng, ref = make_pair(sc)
            steps = hostlib.schedule(kind, ni, ne)
            for w in (eng, ref):
                w.iterate(steps)
            assert_identical(eng, ref, what=f"schedule {kind} ({ni},{ne})")
