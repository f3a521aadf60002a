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


# ---- the same missions with the mission state on the device (mgx_mission_*): one call per tick, one synchronisation ----
def test_device_driver_equals_the_host_driver_on_the_oracle():
    from magics_amd.driver import DeviceDriver
    n, K = 8, 10
    sc = S.circle_scenario(n, K, circle_radius=12.0, n_internal=10, n_external=10)
    sc["ir"] = []
    eng, ref = make_pair(sc)
    kw = dict(waypoints=[[tuple(rb["goal"])] for rb in sc["robots"]], radii=[rb["radius"] for rb in sc["robots"]],
              t0=[rb["t0"] for rb in sc["robots"]], steps=sc["steps"], comms_radius=12.0, target_speed=sc["target_speed"])
    de, dr = DeviceDriver(eng, n, K, **kw), Driver(ref, n, K, **kw)
    events = []
    for tick in range(400):
        if not (dr.finished_at < 0).any():
            break
        ee, er = de.tick(), dr.tick()
        assert ee == er, (tick, ee, er)
        events.append(ee)
        if tick % 10 == 9 or not (dr.finished_at < 0).any():
            tr, left, fin = de.state()
            assert np.array_equal(tr, dr.translation), tick
            assert np.array_equal(fin, dr.finished_at), tick
            assert left.tolist() == [len(wl) for wl in dr.way]
            assert_identical(eng, ref, what=f"device-side circle mission, tick {tick + 1}")
    se, sr = de.summary(), dr.summary()
    assert se["finished"] == n and se["finished_at_tick"] == sr["finished_at_tick"] and se["makespan_s"] == sr["makespan_s"]
    assert se["messages"] == sr["messages"] and se["ticks"] == sr["ticks"]
    assert sum(c for c, _ in events) > n and sum(d for _, d in events) > 0


def test_device_driver_two_waypoints_and_comms_failures():
    from magics_amd.driver import DeviceDriver
    n, K = 6, 10
    sc = S.circle_scenario(n, K, circle_radius=10.0, n_internal=10, n_external=10)
    sc["ir"] = []
    eng, ref = make_pair(sc)
    rng = np.random.default_rng(7)
    draws = rng.random((200, n)) > 0.3

    def failures(tick, k):
        return draws[tick, :k]
    ways = [[(0.6 * rb["goal"][0] + 3.0, 0.6 * rb["goal"][1] - 2.0), tuple(rb["goal"])] for rb in sc["robots"]]
    kw = dict(waypoints=ways, radii=[rb["radius"] for rb in sc["robots"]], t0=[rb["t0"] for rb in sc["robots"]], steps=sc["steps"],
              comms_radius=10.0, target_speed=sc["target_speed"], failure_draws=failures, despawn_when_finished=False)
    de, dr = DeviceDriver(eng, n, K, **kw), Driver(ref, n, K, **kw)
    for tick in range(120):
        assert de.tick() == dr.tick(), tick
    tr, left, fin = de.state()
    assert np.array_equal(tr, dr.translation) and np.array_equal(fin, dr.finished_at)
    assert left.tolist() == [len(wl) for wl in dr.way] and all(x < 2 for x in left)
    assert de.summary()["messages"] == dr.summary()["messages"]
    assert_identical(eng, ref, what="device-side two-waypoint mission with comms failures")


def test_device_driver_ticks_without_belief_readbacks():
    """the point of the exercise: ticks per second of a small world, host-driven vs device-side missions"""
    import time
    from magics_amd.driver import DeviceDriver
    from magics_amd import World
    n, K = 60, 12
    sc = S.circle_scenario(n, K, circle_radius=60.0, n_internal=10, n_external=10)
    sc["ir"] = []
    kw = dict(waypoints=[[tuple(rb["goal"])] for rb in sc["robots"]], radii=[rb["radius"] for rb in sc["robots"]],
              t0=[rb["t0"] for rb in sc["robots"]], steps=sc["steps"], comms_radius=20.0, target_speed=sc["target_speed"])
    rates = {}
    for name, cls in (("host", Driver), ("device", DeviceDriver)):
        w = World(sc["params"])
        S.populate(w, sc)
        d = cls(w, n, K, **kw)
        for _ in range(20):
            d.tick()
        w.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            d.tick()
        w.synchronize()
        rates[name] = 200 / (time.perf_counter() - t0)
    print(f"[driver] 60 robots x 12, ticks / s: host-driven {rates['host']:.0f}, device-side missions {rates['device']:.0f}")
    assert rates["device"] > rates["host"]
