"""Dynamic inter-robot topology on the device: comms-range neighbour search
(all-pairs and hash-grid kernels) against the oracle's scan, and whole
update_robot_neighbours + delete_/create_interrobot_factors passes (robot.rs:1362-1586) between
GBP ticks — connection sets, robot numbers and beliefs must equal the oracle's exactly."""
import numpy as np
import pytest

import oracle
from magics_amd import World, hostlib, scenarios as S

from parity import assert_identical, make_pair

pytestmark = pytest.mark.gpu

PAIRS, GRID = hostlib.NEIGHBOURS_PAIRS, hostlib.NEIGHBOURS_GRID


def bare_pair(n, K=10):
    sc = S.grid_scenario(n, K, interrobot=False, obstacles=False)
    eng, ref = World(sc["params"]), oracle.OracleWorld(sc["params"])
    S.populate(eng, sc)
    S.populate(ref, sc)
    return eng, ref, sc


def same_csr(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("n", [1, 2, 63, 257, 1500])
def test_neighbours_random_positions(n):
    eng, ref, _ = bare_pair(n)
    rng = np.random.default_rng(n)
    for radius in (0.7, 3.0, 11.0):
        pos = rng.uniform(-40, 40, size=(n, 3)).astype(np.float32)
        pos[:, 1] = 0.5
        want = ref.neighbours(pos, radius)
        for method in (PAIRS, GRID):
            assert same_csr(eng.neighbours(pos, radius, method), want), (n, radius, method)


def test_neighbours_degenerate_inputs():
    n = 300
    eng, ref, _ = bare_pair(n)
    rng = np.random.default_rng(9)
    pos = rng.uniform(-20, 20, size=(n, 3)).astype(np.float32)
    pos[10] = pos[20]                              # coincident
    pos[30, 0] = np.nan                            # NaN distance counts as "in range"
    pos[31, 2] = np.inf
    pos[32] = [-np.inf, 0, 0]
    pos[33] = [3e38, 0, -3e38]                     # squares overflow to inf
    pos[34] = [1e-30, 0, 1e-30]                    # denormal products
    pos[35] = [0, 0, 0]
    for radius in (2.5, 0.0, -1.0, np.nan, np.inf, 1e-30, 1e30):
        want = ref.neighbours(pos, radius)
        for method in (PAIRS, GRID, hostlib.NEIGHBOURS_AUTO):
            assert same_csr(eng.neighbours(pos, radius, method), want), (radius, method)


def test_neighbours_exact_boundary_and_cell_edges():
    # integer lattice, radius 5: the 3-4-5 pairs sit exactly on the boundary (in range), and
    # with cell size ~radius many robots sit exactly on cell edges
    side = 40
    n = side * side
    eng, ref, _ = bare_pair(n)
    xs, zs = np.meshgrid(np.arange(side), np.arange(side))
    pos = np.stack([xs.ravel() - 20.0, np.zeros(n), zs.ravel() - 20.0], axis=1).astype(np.float32)
    want = ref.neighbours(pos, 5.0)
    deg = np.diff(want[0])
    assert deg.max() == 80  # interior lattice point: 80 other points within distance 5 (incl. 3-4-5)
    for method in (PAIRS, GRID):
        assert same_csr(eng.neighbours(pos, 5.0, method), want)


def test_neighbours_clustered_heavy_buckets():
    n = 2500  # AUTO picks the grid here
    eng, ref, _ = bare_pair(n)
    rng = np.random.default_rng(2)
    centres = rng.uniform(-500, 500, size=(12, 3))
    pos = (centres[rng.integers(0, 12, n)] + rng.normal(0, 2.0, size=(n, 3))).astype(np.float32)
    want = ref.neighbours(pos, 1.5)
    assert same_csr(eng.neighbours(pos, 1.5), want)
    assert same_csr(eng.neighbours(pos, 1.5, PAIRS), want)


def test_order_keys_not_monotone():
    sc = S.grid_scenario(40, 10, interrobot=True, comm_radius=0.01, obstacles=False)
    assert not sc["ir"]
    keys = np.random.default_rng(4).permutation(40)
    for rb, k in zip(sc["robots"], keys):
        rb["order_key"] = int(k) + 100
    eng, ref = make_pair(sc)
    pos = np.random.default_rng(5).uniform(-8, 8, size=(40, 3)).astype(np.float32)
    pos[:, 1] = 0.5
    want = ref.neighbours(pos, 4.0)
    for method in (PAIRS, GRID):
        got = eng.neighbours(pos, 4.0, method)
        assert same_csr(got, want)
    big = int(np.argmax(np.diff(want[0])))
    row = want[1][want[0][big]:want[0][big + 1]]
    assert len(row) > 2 and (np.diff(keys[row]) > 0).all()   # rows ascend in key, not id
    nxt_e = eng.update_topology(pos, 4.0, 1)
    nxt_r = ref.update_topology(pos, 4.0, 1)
    assert nxt_e == nxt_r
    for w in (eng, ref):
        w.iterate([3, 3, 3])
    assert_identical(eng, ref, what="permuted order keys")


def positions_from_beliefs(w, n, K):
    """What the reference's Transform holds after a tick: the current variable's position
    (f32), y = height (robot.rs:2321-2330)."""
    _, _, mu = w.read_beliefs()
    cur = mu.reshape(n, K, 4)[:, 0, :2]
    return np.stack([cur[:, 0], np.full(n, 0.5), cur[:, 1]], axis=1).astype(np.float32)


def test_circle_crossing_with_dynamic_topology():
    """Robots cross the circle: connections appear as they converge and go as they part."""
    n, K = 12, 10
    sc = S.circle_scenario(n, K, circle_radius=30.0, n_internal=10, n_external=10)
    sc["ir"] = []                                  # nothing connected at start
    eng, ref = make_pair(sc)
    args = S.tick_inputs(sc)
    nxt = {id(eng): 1, id(ref): 1}
    seen_created = seen_deleted = 0
    for tick in range(90):
        res = []
        for w in (eng, ref):
            pos = positions_from_beliefs(w, n, K)
            out = w.update_topology(pos, 18.0, nxt[id(w)])
            nxt[id(w)] = out[0]
            res.append(out)
            w.update_priors(**args)
            w.iterate(sc["steps"])
        assert res[0] == res[1], (tick, res)
        seen_created += res[0][1]
        seen_deleted += res[0][2]
        if tick % 10 == 9:
            assert [eng.connections(r) for r in range(n)] == [ref.connections(r) for r in range(n)]
            assert_identical(eng, ref, what=f"dynamic topology, tick {tick + 1}")
    assert seen_created > n and seen_deleted > 0, (seen_created, seen_deleted)


def test_hashmap_quirk_second_set_of_factors():
    """robot.rs:1391-1404: per robot only the largest out-of-range id is deleted in a pass; the
    (0,1) factors survive, and a second set is created when the pair is in range again."""
    sc = S.grid_scenario(4, 10, interrobot=True, comm_radius=0.01, pitch=1.5)
    assert not sc["ir"]
    eng, ref = make_pair(sc)
    near = np.zeros((4, 3), dtype=np.float32)
    near[:, 0] = [0, 1, -1, 2]
    apart = np.zeros((4, 3), dtype=np.float32)
    apart[:, 0] = [0, 100, 200, 300]
    nxt_e = nxt_r = 1
    for step, (pos, expect) in enumerate([(near, (6, 0)), (apart, (0, 4)), (near, (6, 0)), (near, (0, 0)), (apart, (0, 4))]):
        nxt_e, ce, de = eng.update_topology(pos, 1.2, nxt_e)
        nxt_r, cr, dr = ref.update_topology(pos, 1.2, nxt_r)
        assert (ce, de) == (cr, dr) == expect and nxt_e == nxt_r, (step, ce, de, cr, dr)
        assert [eng.connections(r) for r in range(4)] == [ref.connections(r) for r in range(4)]
        for w in (eng, ref):
            w.iterate([3, 3, 3, 3])
        assert_identical(eng, ref, what=f"quirk step {step}")


def test_bulk_antennas_match_single_calls():
    sc = S.grid_scenario(16, 10, interrobot=True, pitch=2.0, comm_radius=5.0)
    eng, ref = make_pair(sc)
    rng = np.random.default_rng(3)
    for _ in range(4):
        active = rng.random(16) > 0.4
        eng.set_antennas(np.arange(16), active)
        ref.set_antennas(np.arange(16), active)
        for w in (eng, ref):
            w.iterate([3, 3, 3])
    assert_identical(eng, ref, what="bulk antenna writes")


def test_ghost_robots_take_part_in_the_search():
    """A sharded world that follows its topology holds ghost copies of the other ranks' robots and
    is handed all positions: the search covers them (magics_amd/sharded.py, dynamic mode)."""
    sc = S.grid_scenario(4, 10, interrobot=False)
    w = World(sc["params"])
    S.populate(w, sc)
    rb = sc["robots"][0]
    g = w.add_robot(rb["mean0"], rb["prior_diag"], rb["dt"], rb["radius"], order_key=99, ghost=True)
    pos = np.zeros((5, 3), dtype=np.float32)
    pos[:, 0] = [0.0, 10.0, 20.0, 30.0, 10.5]
    ptr, idx = w.neighbours(pos, 1.0)
    rows = [list(idx[ptr[r]:ptr[r + 1]]) for r in range(5)]
    assert rows == [[], [g], [], [], [1]]
    nxt, created, deleted = w.update_topology(pos, 1.0, 1)
    assert (created, deleted) == (2, 0) and nxt == 1 + 2 * 9
    assert list(w.connections(1)) == [g] and list(w.connections(g)) == [1]
    w.iterate(sc["steps"])  # only the connection whose target is local has device edges
    assert np.isfinite(w.read_beliefs()[2]).all()


def test_robot_removal_between_ticks():
    """Despawn (robot.rs:2172): removed robots stop iterating, their messages freeze until the
    neighbours' topology passes drop them; ids and the remaining trajectories stay exact."""
    n, K = 12, 10
    sc = S.circle_scenario(n, K, circle_radius=20.0, n_internal=10, n_external=10)
    sc["ir"] = []
    eng, ref = make_pair(sc)
    args = S.tick_inputs(sc)
    nxt = {id(eng): 1, id(ref): 1}
    alive = np.ones(n, dtype=bool)
    for tick in range(40):
        if tick in (8, 15, 16):
            gone = {8: [3], 15: [0, 7], 16: [11]}[tick]
            for w in (eng, ref):
                for r in gone:
                    w.remove_robot(r)
            alive[gone] = False
            keep = np.nonzero(alive)[0]
            args = S.tick_inputs(sc)
            args = dict(args, robots=args["robots"][keep], waypoints_xy=args["waypoints_xy"][keep],
                        time_scale=args["time_scale"][keep], what=args["what"][keep])
        res = []
        for w in (eng, ref):
            pos = positions_from_beliefs(w, n, K)
            out = w.update_topology(pos, 25.0, nxt[id(w)])
            nxt[id(w)] = out[0]
            res.append(out)
            w.update_priors(**args)
            w.iterate(sc["steps"])
        assert res[0] == res[1], (tick, res)
        assert [eng.connections(r) for r in range(n)] == [ref.connections(r) for r in range(n)]
        if tick % 4 == 3:
            assert_identical(eng, ref, what=f"removal, tick {tick + 1}")
    assert np.array_equal(eng.read_means(), ref.read_means())
    assert all(eng.connections(r) == [] for r in (0, 3, 7, 11))
    with pytest.raises(RuntimeError, match="removed"):
        eng.set_idle(3, False)


def test_full_size_world_with_jittering_positions():
    """1000 robots x 16 (BASELINE configs[2] size): topology passes that create and delete dozens of
    connections per tick through the in-place edge-table rebuild, hash-grid search included."""
    sc = S.grid_scenario(1000, 16, interrobot=True, comm_radius=8.0)
    sc["ir"] = []
    eng, ref = make_pair(sc)
    ref._L.orc_set_threads(ref._w, 8)
    rng = np.random.default_rng(21)
    base = np.array([[rb["pos"][0], 0.5, rb["pos"][1]] for rb in sc["robots"]], dtype=np.float32)
    nxt_e = nxt_r = 1
    changed = 0
    for tick in range(4):
        pos = base + rng.normal(0, 0.15, size=base.shape).astype(np.float32)
        oe = eng.update_topology(pos, 8.0, nxt_e)
        orf = ref.update_topology(pos, 8.0, nxt_r)
        assert oe == orf, (tick, oe, orf)
        nxt_e, nxt_r = oe[0], orf[0]
        changed += oe[1] + oe[2]
        for w in (eng, ref):
            w.iterate(sc["steps"])
    assert changed > 7000 + 100          # the initial pass plus real churn afterwards
    assert all(eng.connections(r) == ref.connections(r) for r in range(0, 1000, 37))
    assert_identical(eng, ref, what="1000 robots, jittering positions, 4 ticks")


def test_resident_capacity_follows_the_topology():
    """A world that densifies: with 8 neighbours per robot a K = 16 workgroup of the resident kernel takes under 40 KB of LDS (four
    per CU, 1024 slots) and 900 robots run their schedule as ONE launch; once robots have 20 neighbours the workgroup needs more
    (three per CU, 768 slots) and the same 900 robots no longer fit — the capacity asked for the sparse topology must not be
    reused, or the launch waits for workgroups that never become resident (round 2's advisor finding).  The engine has to take
    the launch-per-segment path by itself: same beliefs as the oracle, no error."""
    n, K = 900, 16
    sc = S.grid_scenario(n, K, interrobot=True, comm_radius=0.01)
    assert not sc["ir"]
    eng, ref = make_pair(sc)
    pos = np.array([[rb["pos"][0], 0.5, rb["pos"][1]] for rb in sc["robots"]], dtype=np.float32)
    nxt = 1
    launches = []
    for radius in (8.0, 11.5, 11.5):
        out_e = eng.update_topology(pos, radius, nxt)
        out_r = ref.update_topology(pos, radius, nxt)
        assert out_e == out_r
        nxt = out_e[0]
        for w in (eng, ref):
            w.iterate(sc["steps"])
        launches.append(eng.last_launch_count())
        eng.synchronize()  # raises if a wait inside a resident launch gave up
        assert_identical(eng, ref, what=f"comms radius {radius}")
    assert launches[0] == 1 and launches[1] == len(sc["steps"]) + 1 and launches[2] == launches[1], launches


def test_one_pass_search_grows_its_rows():
    """AUTO on a small world is ONE kernel that writes rows of a fixed capacity in place; a row that outgrows the capacity makes
    the search run again with more room (and the pinned block it reads the callers' positions from may move)."""
    n = 400
    eng, ref, _ = bare_pair(n)
    rng = np.random.default_rng(3)
    for radius, spread in ((1.0, 30.0), (6.0, 10.0), (40.0, 10.0), (2.0, 30.0)):  # up to every robot in every row
        pos = rng.uniform(-spread, spread, size=(n, 3)).astype(np.float32)
        want = ref.neighbours(pos, radius)
        for method in (hostlib.NEIGHBOURS_AUTO, PAIRS, GRID):
            assert same_csr(eng.neighbours(pos, radius, method), want), (radius, method)
    assert int(np.diff(want[0]).max()) < n


def test_topology_differences_are_cross_checked():
    """Topology changes reach the device as differences (retopo, k_retopo_robots: slot records and peer rows only of the robots
    whose lists changed).  In a process of its own with MGX_CHECK_INDEX on from the start, every block that goes out is compared
    with tables built from the whole connection list, and the world — churn, a robot removed on the way, ticks and plain schedules
    mixed, a read-back in between — stays the oracle's bit for bit (tests/topology_check_worker.py)."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "topology_check_worker.py"), "180", "24"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "OK 180 robots" in out, out[-3000:]
