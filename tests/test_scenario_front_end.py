"""CPU checks of the scenario front-end's host logic: the readers of the reference's scenario files
(magics_amd/config.py, environment.py), the spawner and its random stream (spawner.py, prng.py) and
the headless runner (sim.py) on the CPU oracle backend.  No GPU compute here."""
import copy
import json
import math
import os

import numpy as np
import pytest

import oracle
from magics_amd import config, environment, hostlib, sim, spawner
from magics_amd.prng import WyRand

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/config/scenarios"
F = np.float32


@pytest.fixture(scope="module")
def scenarios():
    with open(os.path.join(ROOT, "tests", "golden", "scenarios.json"), encoding="utf-8") as f:
        return json.load(f)


def _plain(x):
    return json.loads(json.dumps(x))


# ---- readers ----------------------------------------------------------------------------------------
@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's scenario files are not on this machine")
def test_readers_reproduce_the_fixtures(scenarios):
    """tests/golden/scenarios.json IS what the readers make of the reference's files."""
    names = sorted(os.listdir(REF))
    assert names == sorted(scenarios)
    for name in names:
        sc = config.load_scenario(os.path.join(REF, name))
        assert _plain(sc["environment"]) == scenarios[name]["environment"], name
        assert _plain(sc["formation"]) == scenarios[name]["formation"], name
        for key in ("gbp", "robot", "simulation"):
            assert _plain(sc["config"][key]) == scenarios[name]["config"][key], (name, key)


def test_config_values_are_f32(scenarios):
    gbp = scenarios["Circle Experiment"]["config"]["gbp"]
    assert gbp["sigma-factor-interrobot"] == float(F(0.005)) != 0.005
    assert scenarios["Junction Twoway"]["config"]["gbp"]["sigma-factor-dynamics"] == float(F(0.1))
    p = config.world_params(scenarios["Junction Twoway"]["config"])
    assert p["enable_mask"] == 1 | 2 | 4 and p["safety_multiplier"] == 2.5


MINIMAL = '''
environment_image = "x"
environment = "e"
formation_group = "f"
[gbp]
sigma-pose-fixed = 1e-15
sigma-factor-dynamics = 0.1
sigma-factor-interrobot = 0.01
sigma-factor-obstacle = 0.01
sigma-factor-tracking = 0.1
lookahead-multiple = 3
[gbp.iteration-schedule]
internal = 10
external = 10
schedule = "centered"
[robot]
planning-horizon = 5.0
target-speed = 4.0
inter-robot-safety-distance-multiplier = 2.2
[robot.radius]
min = 1.0
max = 1.0
[robot.communication]
radius = 20.0
failure-rate = 0.2
'''


def test_config_defaults_and_errors():
    cfg = config.parse_config(MINIMAL)
    assert cfg["gbp"]["factors-enabled"] == {"dynamic": True, "interrobot": True, "obstacle": True, "tracking": False}  # lib.rs:454-494
    assert cfg["gbp"]["tracking"] == {"switch-padding": 1.0, "attraction-distance": 2.0} and cfg["gbp"]["variables"] == 10
    assert cfg["simulation"]["hz"] == 60.0 and cfg["simulation"]["prng-seed"] == 0                                      # lib.rs:333-351
    with pytest.raises(config.ConfigError):
        config.parse_config(MINIMAL.replace("lookahead-multiple = 3\n", ""))
    with pytest.raises(config.ConfigError):
        config.parse_config(MINIMAL.replace('"centered"', '"sideways"'))
    with pytest.raises(config.ConfigError):
        config.parse_config(MINIMAL.replace("target-speed = 4.0", "target-speed = -4.0"))
    with pytest.raises(config.ConfigError):
        config.parse_config(MINIMAL + "\n[gbp")


def test_environment_reader_validation():
    ok = "tiles:\n  grid:\n  - ┼─\n  - │┘\n  settings:\n    tile-size: 50.0\n    path-width: 0.2\n    obstacle-height: 1.0\nobstacles: []\n"
    env = environment.parse(ok)
    assert environment.shape(env) == (2, 2) and environment.world_size(env) == (100.0, 100.0)
    assert env["tiles"]["settings"]["sdf"] == {"resolution": 200, "expansion": 0.1, "blur": 0.05}   # SdfSettings::default
    with pytest.raises(environment.EnvironmentError):
        environment.parse(ok.replace("  - │┘\n", "  - │\n"))            # DifferentLengthRows
    with pytest.raises(environment.EnvironmentError):
        environment.parse(ok.replace("  - ┼─\n  - │┘\n", "  []\n"))     # EmptyGrid
    shaped = ok.replace("obstacles: []", "obstacles:\n- shape: !circle\n    radius: 0.1\n  rotation: 7.0\n  translation:\n    x: 0.5\n    y: 0.5\n"
                        "  tile-coordinates:\n    row: 0\n    col: 1\n")
    with pytest.raises(environment.EnvironmentError):
        environment.parse(shaped)                                         # Angle outside [0, 2 pi]
    env = environment.parse(shaped.replace("7.0", "1.0"))
    assert env["obstacles"][0]["shape"] == {"kind": "circle", "radius": 0.1}


def test_formation_reader(scenarios):
    f = scenarios["Junction Twoway"]["formation"]["formations"]
    assert len(f) == 12 and f[0]["repeat"] == {"every": 6_000_000_000, "times": None} and f[1]["delay"] == 2_000_000_000
    assert f[0]["finished-when-intersects"] == {"distance": ["meter", 10.0], "intersects-with": ["variable", 5]}
    c = scenarios["Circle Experiment"]["formation"]["formations"][0]
    assert c["repeat"]["times"] == 1 and c["initial-position"]["placement-strategy"] == ["equal", None]
    assert c["waypoints"][0]["projection-strategy"] == "cross" and c["finished-when-intersects"]["intersects-with"] == ["current", None]


# ---- random stream ----------------------------------------------------------------------------------------
def test_wyrand_definition():
    r = WyRand(0)
    s = 0xA0761D6478BD642F
    t = s * (s ^ 0xE7037ED1A0B428DB)
    assert r.next_u64() == ((t >> 64) ^ t) & (2 ** 64 - 1)
    a, b = WyRand(805), WyRand(805)
    assert [a.next_u64() for _ in range(5)] == [b.next_u64() for _ in range(5)]
    assert WyRand(1).next_u64() != WyRand(2).next_u64()


def test_rand_samplers():
    r = WyRand(42)
    xs = np.array([r.gen_range_f32(0.0, 1.0) for _ in range(4000)])
    assert xs.dtype == np.float32 and xs.min() >= 0.0 and xs.max() < 1.0 and abs(xs.mean() - 0.5) < 0.03
    ys = np.array([r.gen_range_f32_inclusive(2.0, 3.0) for _ in range(4000)])
    assert ys.min() >= 2.0 and ys.max() <= 3.0 and abs(ys.mean() - 2.5) < 0.03
    assert r.gen_range_f32_inclusive(1.0, 1.0) == 1.0                        # Junction Twoway: min == max, one draw all the same
    before = r.state
    r.gen_range_f32_inclusive(1.0, 1.0)
    assert r.state != before
    idx = np.array([r.gen_index(14) for _ in range(5000)])
    assert idx.min() == 0 and idx.max() == 13 and len(np.unique(idx)) == 14
    assert not any(r.gen_bool(0.0) for _ in range(100)) and all(r.gen_bool(1.0) for _ in range(100))
    assert abs(np.mean([r.gen_bool(0.3) for _ in range(5000)]) - 0.3) < 0.03
    child = r.fork()
    assert child.state != r.state
    with pytest.raises(ValueError):
        r.gen_range_f32(1.0, 1.0)


# ---- spawner --------------------------------------------------------------------------------------------------
def _spawn_ticks(formation, n_ticks, dt_ns=100_000_000):
    sp, out = spawner.FormationSpawner(0, formation), []
    for t in range(n_ticks):
        sp.tick(dt_ns)
        if sp.ready_to_spawn():
            sp.spawn()
            out.append(t)
    return out, sp


def test_formation_spawner_timing(scenarios):
    circle = scenarios["Circle Experiment"]["formation"]["formations"][0]      # delay 1 s, repeat every 10 s, finite 1
    ticks, sp = _spawn_ticks(circle, 400)
    assert ticks == [9] and sp.exhausted()
    junction = scenarios["Junction Twoway"]["formation"]["formations"]         # every 6 s forever, delays 0 / 2 / 4 s
    assert _spawn_ticks(junction[0], 200)[0] == [0, 60, 120, 180]
    assert _spawn_ticks(junction[1], 200)[0] == [19, 79, 139, 199]
    once = dict(circle, repeat=None)
    ticks, sp = _spawn_ticks(once, 50)
    assert ticks == [9] and sp.exhausted()
    thrice = dict(circle, repeat={"every": 1_000_000_000, "times": 3})
    assert _spawn_ticks(thrice, 100)[0] == [9, 19, 29]


def test_circle_formation_positions(scenarios):
    sc = scenarios["Circle Experiment"]
    robots = spawner.spawn_formation(sc["formation"]["formations"][0], sc["config"], (100.0, 100.0), WyRand(805))
    assert len(robots) == 30 and len(robots[0]["timesteps"]) == 21                 # 15 m/s * 5 s = 75 -> K = 21
    for i, rb in enumerate(robots):
        start, goal = rb["waypoints"][0], rb["waypoints"][-1]
        assert 2.0 <= rb["radius"] <= 3.0 and len(rb["waypoints"]) == 2
        assert math.hypot(start[0], start[1]) == pytest.approx(50.0, abs=1e-3)
        assert np.allclose(goal[:2], -start[:2], atol=1e-3)                       # projection-strategy: cross
        assert math.hypot(start[2], start[3]) == pytest.approx(15.0, abs=1e-3)     # target speed towards the waypoint
        assert np.array_equal(goal[2:], start[2:])                               # last.update_velocity(second_last)
        assert math.atan2(start[1], start[0]) % (2 * math.pi) == pytest.approx(2 * math.pi * i / 30, abs=1e-4)


def test_line_segment_formations(scenarios):
    sc = scenarios["Junction Twoway"]
    rng = WyRand(2)
    rb = spawner.spawn_formation(sc["formation"]["formations"][0], sc["config"], (100.0, 100.0), rng)[0]
    assert rb["radius"] == 1.0 and len(rb["waypoints"]) == 3 and len(rb["timesteps"]) == 12
    start = rb["waypoints"][0]
    assert start[0] == -50.0 and 1.5 <= start[1] <= 6.5 and tuple(start[2:]) == (5.0, 0.0)   # left edge, driving right
    # several robots on one segment: placements never overlap, `cross` reverses their order
    f = copy.deepcopy(sc["formation"]["formations"][0])
    f["robots"] = 3
    f["initial-position"]["shape"]["points"] = [[0.0, 0.1], [0.0, 0.9]]
    f["waypoints"] = [{"shape": {"kind": "line-segment", "points": [[1.0, 0.1], [1.0, 0.9]]}, "projection-strategy": "cross"}]
    cfg = copy.deepcopy(sc["config"])
    cfg["robot"]["radius"] = {"min": 1.0, "max": 3.0}
    robots = spawner.spawn_formation(f, cfg, (100.0, 100.0), WyRand(7))
    ys = [r["waypoints"][0][1] for r in robots]
    for i in range(3):
        for j in range(i):
            assert abs(ys[i] - ys[j]) >= robots[i]["radius"] + robots[j]["radius"]
    assert [r["waypoints"][1][1] for r in robots] == pytest.approx(ys[::-1], abs=1e-4)
    f["initial-position"]["placement-strategy"] = ["equal", None]
    even = spawner.spawn_formation(f, cfg, (100.0, 100.0), WyRand(7))
    assert len(even) == 3 and all(-40.0 <= r["waypoints"][0][1] <= 40.0 for r in even)
    f["initial-position"]["shape"]["points"] = [[0.0, 0.5], [0.0, 0.51]]       # 1 m for three robots of radius >= 1
    f["initial-position"]["placement-strategy"] = ["random", 50]
    assert spawner.spawn_formation(f, cfg, (100.0, 100.0), WyRand(7)) is None  # "failed to spawn formation"


# ---- headless runs on the CPU backend ------------------------------------------------------------------
def test_junction_twoway_runs_headless(scenarios):
    sc = scenarios["Junction Twoway"]
    runs = []
    for _ in range(2):
        s = sim.Simulation(sc, oracle.OracleWorld(config.world_params(sc["config"])))
        s.run(max_ticks=70)
        runs.append(s)
    a, b = runs
    assert a.tick_no == 70 and len(a.robots) == 16 and a.K == 12 and not a.finished()
    assert np.array_equal(a.translation, b.translation) and a.events == b.events          # deterministic
    first = a.robots[0]
    assert a.translation[0, 0] > first["waypoints"][0][0] + 25.0                           # 7 s at 5 m/s along the lane
    ex = a.export()
    assert set(ex) == {"scenario", "makespan", "delta_t", "gbp", "robots", "prng_seed", "config", "obstacles", "collisions", "goal_areas"}
    r0 = ex["robots"]["0"]
    assert set(r0) == {"radius", "positions", "velocities", "collisions", "messages", "mission", "planning_strategy", "color"}
    assert r0["messages"]["sent"]["internal"] > 0 and len(r0["positions"]) == 70 and ex["makespan"] == pytest.approx(7.0)
    assert len(r0["velocities"]) == 69 and set(r0["velocities"][0]) == {"velocity", "timestamp", "measured_over"}
    speed = [math.hypot(v["velocity"][0], v["velocity"][2]) for v in r0["velocities"][5:]]
    assert 1.0 < min(speed) and max(speed) < 15.0 and r0["velocities"][0]["measured_over"] == {"secs": 0, "nanos": 100000000}
    json.dumps(ex)


def test_short_circle_finishes(scenarios):
    sc = copy.deepcopy(scenarios["Circle Experiment"])
    f = sc["formation"]["formations"][0]
    f["robots"] = 4
    f["initial-position"]["shape"]["radius"] = 8.0
    f["waypoints"][0]["shape"]["radius"] = 8.0
    s = sim.Simulation(sc, oracle.OracleWorld(config.world_params(sc["config"])))
    s.run(max_ticks=300)
    assert s.finished() and all(r["completed"] and not r["alive"] for r in s.robots)
    assert s.tick_no < 300 and all(r["finished_at"] > r["started_at"] for r in s.robots)
    assert all(r["travelled"] > 10.0 for r in s.robots)


def test_rrt_star_is_rejected(scenarios):
    sc = scenarios["Solo GP"]
    s = sim.Simulation(sc, oracle.OracleWorld(config.world_params(sc["config"])))
    with pytest.raises(NotImplementedError):
        s.run(max_ticks=100)


def test_schedules_follow_the_config(scenarios):
    sc = scenarios["Circle Experiment"]
    s = sim.Simulation(sc, oracle.OracleWorld(config.world_params(sc["config"])))
    assert s.steps == hostlib.schedule(hostlib.SCHEDULE_INTERLEAVE_EVENLY, 50, 10) and len(s.steps) == 50


def test_collision_state_machine_counts_contacts_once():
    """CollisionHistory (planner/collisions.rs:455-495) through sim.Simulation._collide: Free -> Colliding is one collision,
    staying in contact is not another, parting and touching again is; the bounding-sphere predicate is <= in f32"""
    s = sim.Simulation.__new__(sim.Simulation)
    s.collisions = {}
    F = np.float32
    a, b = {"id": 0, "radius": F(1.0)}, {"id": 1, "radius": F(1.5)}

    def at(x):
        return np.array([[0.0, -1.5, 0.0], [x, -1.5, 0.0]], dtype=F)
    for x, times in ((3.0, 0), (2.5, 1), (2.0, 1), (2.6, 1), (2.4, 2), (0.0, 2)):   # 2.5 = r_a + r_b: touching counts
        s._collide([a, b], at(x))
        assert s.collisions.get((0, 1), {"times": 0})["times"] == times, x
    h = s.collisions[(0, 1)]
    assert len(h["aabbs"]) == 2 and h["aabbs"][0] == {"mins": [1.0, -1.0], "maxs": [1.0, 1.0]}


def test_entity_allocator_orders_like_bevy():
    """Bevy 0.13 `Entities`: freed indices come back last-freed-first with the generation raised, and an Entity orders by
    generation first (to_bits) — the order of the factor graphs (id.rs:19-54)."""
    from magics_amd.spawner import EntityAllocator
    a = EntityAllocator()
    e = [a.alloc() for _ in range(4)]
    assert e == [(1 << 32) | i for i in range(4)] and e == sorted(e)
    a.free(e[1])
    a.free(e[3])
    x, y, z = a.alloc(), a.alloc(), a.alloc()          # index 3 (last freed) first, then 1, then a fresh one
    assert (x & 0xffffffff, x >> 32) == (3, 2) and (y & 0xffffffff, y >> 32) == (1, 2) and (z & 0xffffffff, z >> 32) == (4, 1)
    assert max(e) < y < x                              # reused indices sort behind every first-generation entity ...
    assert z < y                                       # ... a fresh first-generation one before them, whenever it is spawned ...
    assert y < x                                       # ... and within a generation the index decides, against the spawn order here
    with pytest.raises(AssertionError):
        a.free(e[1])                                   # stale handle
