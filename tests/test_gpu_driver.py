"""Whole missions through the per-tick system chain (magics_amd/driver.py) on the engine and on
the oracle: robots cross a circle to the antipode (the Circle Experiment's formation), connect and
disconnect on the way, reach their waypoint and despawn; everything the driver sees — topology
events, finishing ticks, distances, message counts, beliefs — must be the same."""
import numpy as np
import pytest

from magics_amd import scenarios as S
from magics_amd.driver import Driver

from parity import assert_identical, make_pair

pytestmark = pytest.mark.gpu


def circle_driver(w, sc, n, K, **kw):
    return Driver(w, n, K, waypoints=[[tuple(rb["goal"])] for rb in sc["robots"]], radii=[rb["radius"] for rb in sc["robots"]],
                  t0=[rb["t0"] for rb in sc["robots"]], steps=sc["steps"], comms_radius=12.0, target_speed=sc["target_speed"], **kw)


def test_circle_mission_until_everybody_has_arrived():
    n, K = 8, 10
    sc = S.circle_scenario(n, K, circle_radius=12.0, n_internal=10, n_external=10)
    sc["ir"] = []
    eng, ref = make_pair(sc)
    de, dr = circle_driver(eng, sc, n, K), circle_driver(ref, sc, n, K)
    events = []
    for tick in range(400):
        if not (de.finished_at < 0).any():
            break
        ee, er = de.tick(), dr.tick()
        assert ee == er, (tick, ee, er)
        events.append(ee)
        assert np.array_equal(de.translation, dr.translation) and np.array_equal(de.finished_at, dr.finished_at), tick
        if tick % 20 == 19:
            assert_identical(eng, ref, what=f"circle mission, tick {tick + 1}")
    se, sr = de.summary(), dr.summary()
    assert se == sr
    assert se["finished"] == n and se["makespan_s"] is not None, se
    assert sum(c for c, _ in events) > n and sum(d for _, d in events) > 0      # connections came and went
    assert min(se["distance_travelled"]) > 20.0                                   # everybody crossed the circle
    print("makespan", se["makespan_s"], "s; ticks", se["ticks"])


def test_mission_with_an_intermediate_waypoint_and_comms_failures():
    n, K = 6, 10
    sc = S.circle_scenario(n, K, circle_radius=10.0, n_internal=10, n_external=10)
    sc["ir"] = []
    eng, ref = make_pair(sc)
    rng = np.random.default_rng(7)
    draws = rng.random((200, n)) > 0.3                                            # 30 % failure rate, same draws for both

    def failures(tick, k):
        return draws[tick, :k]
    ways = [[(0.6 * rb["goal"][0] + 3.0, 0.6 * rb["goal"][1] - 2.0), tuple(rb["goal"])] for rb in sc["robots"]]
    drv = [Driver(w, n, K, waypoints=ways, radii=[rb["radius"] for rb in sc["robots"]], t0=[rb["t0"] for rb in sc["robots"]],
                  steps=sc["steps"], comms_radius=10.0, target_speed=sc["target_speed"], failure_draws=failures,
                  despawn_when_finished=False) for w in (eng, ref)]
    for tick in range(120):
        a, b = drv[0].tick(), drv[1].tick()
        assert a == b, tick
    assert drv[0].summary() == drv[1].summary()
    assert all(len(wl) < 2 for wl in drv[0].way)                                  # every robot got past its first waypoint
    assert_identical(eng, ref, what="two-waypoint mission with comms failures")
