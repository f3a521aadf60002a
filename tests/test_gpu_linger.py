"""Lingering resident launches (include/mgx.h, mgx_dev.h): schedules issued back to back are POSTED into the launch that is there —
the graphs stay in LDS across mgx_iterate / mgx_tick calls — and every other call ends the launch first.  The bar is the one of
every other path: beliefs and message counts of the CPU oracle bit for bit, whatever mix of launches, posts, closes and posts
taken back and re-run the calling pattern produces; and every wait bounded (a host that goes away leaves no spinning launch)."""
import time

import numpy as np
import pytest

import oracle
from magics_amd import World, scenarios as S

from parity import assert_identical, make_pair

pytestmark = pytest.mark.gpu


def test_ticks_back_to_back_ride_in_one_launch():
    """mgx_tick after mgx_tick: ONE launch, the rest posts (prior updates included), closed by the read-back"""
    sc = S.grid_scenario(144, 16, interrobot=True)
    eng, ref = make_pair(sc)
    tick = S.tick_inputs(sc)
    n = 12
    for _ in range(n):
        eng.tick(steps=sc["steps"], **tick)
        assert eng.last_launch_count() == 1  # (a pure query: the launch stays)
    launches, posts, reruns, ended = eng.linger_stats()
    assert launches == 1 and posts + reruns == n - 1 and posts >= 1, (launches, posts, reruns, ended)
    assert eng.resident_stats()[0] == 1 + reruns
    for _ in range(n):
        ref.tick(steps=sc["steps"], **tick)
    assert_identical(eng, ref, what=f"{n} ticks back to back: {posts} posted into one lingering launch")
    for r in (0, 17, 143):
        assert eng.message_counts(r) == ref.message_counts(r)


def test_schedules_of_every_shape_posted_or_launched():
    """iterate calls of mixed shapes: schedules that open with an internal iteration are posted (also a single segment, also one
    with internal iterations only), one that opens with an external iteration ends the launch and runs as a launch of its own;
    calls in between (prior change, antenna, read) end it as well — the oracle's beliefs after every step"""
    sc = S.grid_scenario(64, 16, interrobot=True)
    eng, ref = make_pair(sc)
    mean = np.array([0.5, 0.25, 1.0, -1.0])
    eng.set_linger(20000)  # (a bound no host jitter reaches: what is posted and what is launched follows from the calls alone)
    groups = [[sc["steps"], sc["steps"], [1]],                   # launch, post, post (a single segment)
              [[3, 3, 1], [1, 1, 1], sc["steps"], [2, 3, 3]],    # launch, post (internal iterations only), post; opens with an external iteration: a launch of its own
              [sc["steps"], sc["steps"][:5], [3]],                # (behind a prior change) launch, post, post
              [sc["steps"]]]                                      # (behind an antenna going off)
    for gi, group in enumerate(groups):
        for w in (eng, ref):  # the engine's calls back to back, then the oracle's
            for st in group:
                w.iterate(st)
            if gi == 1:
                w.change_prior(5, 15, mean)
            if gi == 2:
                w.set_antenna(9, False)
        assert_identical(eng, ref, what=f"mixed schedules, group {gi}")
    launches, posts, reruns, ended = eng.linger_stats()
    assert (posts, reruns, ended) == (6, 0, 0) and launches >= 4, (launches, posts, reruns, ended)
    for r in range(0, 64, 7):
        assert eng.message_counts(r) == ref.message_counts(r)


def test_a_host_that_goes_away_leaves_no_spinning_launch():
    """nothing follows the tick: the workgroups wait out their bound and end the launch themselves; the next tick — issued while
    they wait, or after they have gone — is posted, or taken back and launched: either way it runs exactly once"""
    sc = S.grid_scenario(100, 16, interrobot=True)
    eng, ref = make_pair(sc)
    eng.set_linger(2000)
    tick = S.tick_inputs(sc)
    import torch
    for pause in (0.0, 0.0, 0.0005, 0.003, 0.02, 0.0, 0.0015, 0.0025, 0.0):
        eng.tick(steps=sc["steps"], **tick)
        ref.tick(steps=sc["steps"], **tick)
        if pause:
            time.sleep(pause)
    t0 = time.perf_counter()
    torch.cuda.synchronize()  # NOT mgx_synchronize: nobody tells the launch to end — it ends by itself within its bound
    assert time.perf_counter() - t0 < 0.5
    launches, posts, reruns, ended = eng.linger_stats()
    assert launches >= 2 and launches + posts == 9, (launches, posts, reruns, ended)  # (a post taken back became a launch)
    assert_identical(eng, ref, what=f"ticks with pauses: {launches} launches, {posts} posts, {reruns} re-run")
    eng.synchronize()


def test_lingering_at_the_headline_size_and_switched_off():
    """1000 x 16 + inter-robot factors, every workgroup resident: 6 ticks back to back, posted; the same with
    mgx_set_linger(w, 0): every tick its own launch — identical beliefs, the oracle's"""
    sc = S.grid_scenario(1000, 16, interrobot=True)
    eng, ref = World(sc["params"]), oracle.OracleWorld(sc["params"], threads=16)
    off = World(sc["params"])
    for w in (eng, ref, off):
        S.populate(w, sc)
    off.set_linger(0)
    tick = S.tick_inputs(sc)
    for _ in range(6):
        eng.tick(steps=sc["steps"], **tick)
    eng.flush()  # (the launch ends, nothing is waited for)
    for _ in range(6):
        off.tick(steps=sc["steps"], **tick)
        ref.tick(steps=sc["steps"], **tick)
    launches, posts, reruns, ended = eng.linger_stats()
    assert launches == 1 and posts + reruns == 5 and posts >= 3, (launches, posts, reruns, ended)
    assert off.linger_stats() == (0, 0, 0, 0) and off.resident_stats()[0] == 6
    assert_identical(eng, ref, what="1000 x 16, 6 ticks in one lingering launch")
    assert_identical(off, ref, what="1000 x 16, lingering switched off")
    assert all(np.isfinite(x).all() for x in eng.read_beliefs())


def test_lingering_gives_up_where_nothing_follows():
    """iterate, read, iterate, read ...: two lingering launches that ended without a post are evidence enough — the launches that
    follow end with their schedule, until schedules come back to back again"""
    sc = S.grid_scenario(64, 16, interrobot=True)
    eng, ref = make_pair(sc)
    for i in range(5):
        eng.iterate(sc["steps"]); ref.iterate(sc["steps"])
        assert_identical(eng, ref, what=f"iterate / read {i}")
    assert eng.linger_stats()[0] == 2
    for _ in range(3):
        eng.iterate(sc["steps"]); ref.iterate(sc["steps"])
    launches, posts, reruns, _ = eng.linger_stats()
    assert launches == 3 and posts + reruns == 1, (launches, posts, reruns)  # the second of the three lingered, the third was posted
    assert_identical(eng, ref, what="back to back again")


@pytest.mark.parametrize("K,n,tracking", [(10, 64, False), (12, 60, True), (21, 100, False), (32, 64, True)])
def test_lingering_on_other_horizons(K, n, tracking):
    """run-time paths that differ by horizon: K <= 16 (side-by-side variable sweeps), longer ones (barriers in between, response
    means kept in LDS from the other branch), tracking factors' gate counting on over the plans"""
    sc = S.grid_scenario(n, K, interrobot=True, tracking=tracking)
    eng, ref = World(sc["params"]), oracle.OracleWorld(sc["params"], threads=8)
    assert S.populate(eng, sc) == S.populate(ref, sc)
    tick = S.tick_inputs(sc)
    from parity import assert_identical_where_finite
    eng.set_linger(20000)
    for rep in range(2):
        for w in (eng, ref):  # (the engine's ticks back to back: the oracle's would have the launch wait out its bound in between)
            for _ in range(4):
                w.tick(steps=sc["steps"], **tick)
        if tracking:
            assert_identical_where_finite(eng, ref, what=f"K = {K}, tracking, 4 ticks back to back ({rep})", max_nan_only_mismatch=5e-3)
        else:
            assert_identical(eng, ref, what=f"K = {K}, 4 ticks back to back ({rep})")
    launches, posts, reruns, _ = eng.linger_stats()
    assert (launches, posts, reruns) == (2, 6, 0), (launches, posts, reruns)
