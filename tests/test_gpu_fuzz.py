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
@pytest.mark.parametrize("batched", [False, True])
def test_random_script(K, n, seed, batched):
    """batched: the engine's whole script runs inside ONE open batch (mgx_batch_begin): consecutive schedules merge into one
    submission, every other call first submits what was recorded — the results must not know the difference"""
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
    if batched:
        eng.batch_begin()
    for k, f in enumerate(script):
        f(eng)
        f(ref)
        if k % 5 == 4:
            assert_identical(eng, ref, what=f"K={K} n={n} seed={seed} after op {k}")
    if batched:
        schedules, _ = eng.batch_end()
        assert schedules == 14
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


@pytest.mark.parametrize("K,n,seed", [(10, 14, 11), (12, 9, 12), (16, 20, 13), (21, 7, 14), (8, 10, 15)])
@pytest.mark.parametrize("batched", [False, True])
def test_random_script_with_dynamic_topology(K, n, seed, batched):
    """The same idea with the topology systems in the mix: robots wander (positions are the caller's
    Transform inputs), whole update_robot_neighbours / delete / create passes run between sweeps
    (in-place edge-table rebuilds on the device), robots despawn, antennas are rewritten in bulk;
    connection sets, robot numbers, message counts and beliefs must all match."""
    sc = _scenario(K, n, seed)
    sc = dict(sc, ir=[])
    eng, ref = make_pair(sc)
    rng = np.random.default_rng(seed)
    pos = np.array([[rb["pos"][0], 0.5, rb["pos"][1]] for rb in sc["robots"]], dtype=np.float32)
    alive = np.ones(n, dtype=bool)
    nxt = {id(eng): 1, id(ref): 1}
    tick = S.tick_inputs(sc)
    if batched:
        eng.batch_begin()
    for step in range(24):
        op = int(rng.integers(0, 7))
        if op <= 2:
            pos = pos + rng.normal(0, 1.2, size=pos.shape).astype(np.float32) * np.array([1, 0, 1], dtype=np.float32)
            radius = float(rng.choice([3.0, 4.5, 6.0]))
            res = []
            for w in (eng, ref):
                out = w.update_topology(pos, radius, nxt[id(w)])
                nxt[id(w)] = out[0]
                res.append(out)
            assert res[0] == res[1], (step, res)
        elif op == 3 and alive.sum() > 3:
            r = int(rng.choice(np.nonzero(alive)[0]))
            alive[r] = False
            for w in (eng, ref):
                w.remove_robot(r)
        elif op == 4:
            live = np.nonzero(alive)[0].astype(np.int32)
            on = rng.random(len(live)) > 0.3
            for w in (eng, ref):
                w.set_antennas(live, on)
        elif op == 5:
            live = np.nonzero(alive)[0]
            args = dict(tick, robots=tick["robots"][live], waypoints_xy=tick["waypoints_xy"][live], time_scale=tick["time_scale"][live],
                        what=tick["what"][live])
            for w in (eng, ref):
                w.update_priors(**args)
        steps = [int(x) for x in rng.integers(1, 4, size=int(rng.integers(1, 5)))]
        for w in (eng, ref):
            w.iterate(steps)
        if step % 6 == 5:
            assert [eng.connections(r) for r in range(n)] == [ref.connections(r) for r in range(n)]
            assert [eng.message_counts(r) for r in range(n)] == [ref.message_counts(r) for r in range(n)], step
            assert_identical(eng, ref, what=f"dynamic K={K} n={n} seed={seed} step {step}")
    if batched:
        assert eng.batch_end()[0] == 24
    assert_identical(eng, ref, what=f"dynamic K={K} n={n} seed={seed} final")
