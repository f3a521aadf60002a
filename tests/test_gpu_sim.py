"""GPU parity of whole scenario runs: the reference's scenario files (as parsed into
tests/golden/scenarios.json) driven headless on the HIP engine and on the CPU oracle — same spawns,
same topology events, same trajectories, beliefs bit for bit, same export."""
import json
import os

import numpy as np
import pytest

import oracle
from magics_amd import World, config, sim

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scenario(name):
    with open(os.path.join(ROOT, "tests", "golden", "scenarios.json"), encoding="utf-8") as f:
        return json.load(f)[name]


def _pair(sc):
    p = config.world_params(sc["config"])
    return sim.Simulation(sc, World(p)), sim.Simulation(sc, oracle.OracleWorld(p))


def _run_both(name, ticks, check_every=10, tweak=None):
    sc = _scenario(name)
    if tweak:
        tweak(sc)
    a, b = _pair(sc)
    for t in range(ticks):
        a.tick()
        b.tick()
        if (t + 1) % check_every == 0 or t == ticks - 1:
            assert len(a.robots) == len(b.robots)
            assert np.array_equal(a.translation, b.translation), f"tick {t}"
            if a.robots:
                for x, y in zip(a.w.read_beliefs(), b.w.read_beliefs()):
                    assert np.array_equal(x, y), f"tick {t}"
    assert a.events == b.events
    assert json.dumps(a.export(), sort_keys=True) == json.dumps(b.export(), sort_keys=True)
    return a


def test_junction_twoway():
    s = _run_both("Junction Twoway", 90)
    assert len(s.robots) >= 12 and s.K == 12 and s.events
    moved = np.hypot(s.translation[:4, 0] - [r["waypoints"][0][0] for r in s.robots[:4]],
                     s.translation[:4, 2] - [r["waypoints"][0][1] for r in s.robots[:4]])
    assert (moved > 20).all()


def test_circle_experiment():
    def fewer(sc):  # 12 of the 30 robots: the oracle side stays within seconds
        sc["formation"]["formations"][0]["robots"] = 12
    s = _run_both("Circle Experiment", 45, tweak=fewer)
    assert len(s.robots) == 12 and s.K == 21 and s.events


def test_environment_obstacles_experiment_with_comms_failures():
    def failing(sc):
        sc["config"]["robot"]["communication"]["failure-rate"] = 0.3
    s = _run_both("Environment Obstacles Experiment", 70, tweak=failing)
    assert len(s.robots) == 5


def test_robots_finish_and_despawn():
    def short(sc):  # bring the far side of the circle within reach of a short run
        f = sc["formation"]["formations"][0]
        f["robots"] = 6
        f["initial-position"]["shape"]["radius"] = 12.0
        f["waypoints"][0]["shape"]["radius"] = 12.0
    s = _run_both("Circle Experiment", 60, tweak=short)
    assert any(r["completed"] for r in s.robots)
    assert all((not r["alive"]) == r["completed"] for r in s.robots)


def _only_local_scenarios():
    with open(os.path.join(ROOT, "tests", "golden", "scenarios.json"), encoding="utf-8") as f:
        scenarios = json.load(f)
    return sorted(n for n, sc in scenarios.items()
                  if all(f["planning-strategy"] == "only-local" for f in sc["formation"]["formations"]) and
                  sum(f["robots"] for f in sc["formation"]["formations"]) > 0)


@pytest.mark.parametrize("name", _only_local_scenarios())
def test_every_local_planning_scenario_of_the_reference(name):
    """Every scenario directory of the reference that plans locally (the RRT* ones need the global
    planner, out of scope), unmodified, for its first simulated seconds: environment, spawns, topology,
    prior updates, schedule — engine and oracle stay bit-identical."""
    sc = _scenario(name)
    n_first = max(f["robots"] for f in sc["formation"]["formations"])
    first_spawn = min(f["delay"] for f in sc["formation"]["formations"]) / 1e9
    ticks = int((first_spawn + (1.5 if n_first > 12 else 3.0)) * sc["config"]["simulation"]["hz"]) + 1
    a, b = _pair(sc)
    for t in range(ticks):
        a.tick()
        b.tick()
    assert len(a.robots) == len(b.robots) and len(a.robots) > 0
    assert np.array_equal(a.translation, b.translation) and a.events == b.events
    for x, y in zip(a.w.read_beliefs(), b.w.read_beliefs()):
        assert np.array_equal(x, y)


def test_scenario_on_a_sharded_world():
    """The scenario runner on a world sharded over three ranks that follows its topology: formations spawn robots
    over time (each joins as a real robot on its owner's rank and as a ghost on the others), connections come and
    go across rank boundaries, robots finish and despawn — same trajectories and beliefs as the single-world oracle."""
    from magics_amd import sharded
    sc = _scenario("Junction Twoway")
    params = config.world_params(sc["config"])
    cluster = sharded.LocalCluster(dict(params=params, robots=[], ir=[], K=None), 3, World, dynamic=True)
    a, b = sim.Simulation(sc, cluster), sim.Simulation(sc, oracle.OracleWorld(params))
    for t in range(130):
        a.tick()
        b.tick()
        if t % 10 == 9:
            assert len(a.robots) == len(b.robots)
            assert np.array_equal(a.translation, b.translation), t
            for x, y in zip(a.w.read_beliefs(), b.w.read_beliefs()):
                assert np.array_equal(x, y), t
    assert a.events == b.events and len(a.robots) >= 20
    assert len({sw.plan.owner[r["id"]] for sw in cluster.ranks for r in a.robots}) == 3   # robots live on all three ranks


def test_robot_robot_collisions_are_counted_like_the_reference():
    """update_robot_robot_collisions (planner/collisions.rs:72-140): with the inter-robot factors switched off the robots
    of the Circle Experiment drive through each other at the centre — the Free -> Colliding edges are counted once per
    contact, engine (Transforms on the device, samples one tick behind) and oracle (host loop) export the same."""
    def blind(sc):
        f = sc["formation"]["formations"][0]
        f["robots"] = 6
        f["initial-position"]["shape"]["radius"] = 14.0
        f["waypoints"][0]["shape"]["radius"] = 14.0
        sc["config"]["gbp"]["factors-enabled"]["interrobot"] = False
    s = _run_both("Circle Experiment", 40, tweak=blind)
    ex = s.export()
    total = sum(r["collisions"]["robots"] for r in ex["robots"].values())
    pairs = ex["collisions"]["robots"]
    assert total > 0 and total == 2 * sum(len(p["aabbs"]) for p in pairs)       # every collision counts for both robots
    assert all(p["robot_a"] < p["robot_b"] and all(a["mins"][0] <= a["maxs"][0] for a in p["aabbs"]) for p in pairs)
    assert ex["goal_areas"] == {} and ex["collisions"]["environment"] == []


def _run_chunked(name, ticks, tweak=None, chunk=256):
    """the engine through Simulation.run — whole stretches between two spawns as ONE mgx_mission_run call — against the oracle
    tick by tick: same spawns, events, Transforms, beliefs, random stream and export"""
    sc = _scenario(name)
    if tweak:
        tweak(sc)
    a, b = _pair(sc)
    calls = []
    real = a.w.mission_run

    def counted(n, *args, **kw):
        out = real(n, *args, **kw)
        calls.append((n, out["ticks"]))
        return out
    a.w.mission_run = counted
    a.run(max_ticks=ticks, chunk=chunk)
    for _ in range(a.tick_no):
        b.tick()
    assert a.tick_no == b.tick_no and len(a.robots) == len(b.robots) and a.rng.state == b.rng.state
    assert np.array_equal(a.translation, b.translation) and a.events == b.events
    if a.robots:
        for x, y in zip(a.w.read_beliefs(), b.w.read_beliefs()):
            assert np.array_equal(x, y)
    assert json.dumps(a.export(), sort_keys=True) == json.dumps(b.export(), sort_keys=True)
    return a, calls


def test_many_ticks_per_call_junction_twoway():
    s, calls = _run_chunked("Junction Twoway", 130)
    assert len(s.robots) >= 24 and s.events
    assert max(n for n, _ in calls) >= 15 and len(calls) <= 12, calls  # (twelve staggered spawners: some spawner acts every ~20 ticks)


def test_many_ticks_per_call_with_comms_failures_and_despawns():
    def failing(sc):
        sc["config"]["robot"]["communication"]["failure-rate"] = 0.3
    s, calls = _run_chunked("Environment Obstacles Experiment", 70, tweak=failing)
    assert len(s.robots) == 5

    def short(sc):  # (as test_robots_finish_and_despawn) the far side of the circle within reach: missions complete, robots despawn,
        f = sc["formation"]["formations"][0]  # and the run ends by itself behind the tick that completed the last one
        f["robots"] = 6
        f["initial-position"]["shape"]["radius"] = 12.0
        f["waypoints"][0]["shape"]["radius"] = 12.0
        sc["config"]["robot"]["communication"]["failure-rate"] = 0.1
    s, calls = _run_chunked("Circle Experiment", 400, tweak=short, chunk=64)
    assert s.finished() and all(r["completed"] and not r["alive"] for r in s.robots) and s.tick_no < 400
    assert calls[-1][1] <= calls[-1][0]
