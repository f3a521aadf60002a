"""Oracle restatement of the dynamic inter-robot topology systems (robot.rs:1362-1586):
update_robot_neighbours against an independent numpy f32 scan, and the bookkeeping of
delete_/create_interrobot_factors including the HashMap quirk (robot.rs:1391-1404)."""
import numpy as np

import oracle
from magics_amd import scenarios as S


def numpy_neighbours(pos, radius):
    """All-pairs scan with numpy float32 arithmetic (every operation rounded to f32)."""
    pos = np.asarray(pos, dtype=np.float32)
    r = np.float32(radius)
    rows = []
    with np.errstate(all="ignore"):
        for i in range(len(pos)):
            d = pos[i] - pos
            s = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            dist = np.sqrt(s)
            inr = ~(r < dist)
            inr[i] = False
            rows.append(np.nonzero(inr)[0])
    return rows


def rows_of(ptr, idx):
    return [idx[ptr[i]:ptr[i + 1]] for i in range(len(ptr) - 1)]


def small_world(n, K=10):
    sc = S.grid_scenario(n, K, interrobot=False)
    w = oracle.OracleWorld(sc["params"])
    S.populate(w, sc)
    return w, sc


def test_neighbours_match_numpy_scan():
    rng = np.random.default_rng(5)
    w, _ = small_world(64)
    for radius in (0.0, 0.5, 3.0, 20.0, np.inf, -1.0, np.nan):
        pos = rng.uniform(-10, 10, size=(64, 3)).astype(np.float32)
        pos[:, 1] = 0.5
        pos[7] = pos[3]                      # coincident pair: distance 0
        pos[11, 0] = np.nan                  # NaN distance counts as in range
        pos[12, 2] = np.inf
        ptr, idx = w.neighbours(pos, radius)
        for i, (a, b) in enumerate(zip(rows_of(ptr, idx), numpy_neighbours(pos, radius))):
            assert np.array_equal(a, b), (radius, i, a, b)


def test_boundary_distance_is_in_range():
    w, _ = small_world(4)
    pos = np.array([[0, 0, 0], [3, 0, 4], [6, 0, 8], [100, 0, 0]], dtype=np.float32)
    ptr, idx = w.neighbours(pos, 5.0)   # |p0-p1| == 5 exactly: `radius < d` is false
    assert [r.tolist() for r in rows_of(ptr, idx)] == [[1], [0, 2], [1], []]


def test_update_topology_numbers_and_sets():
    w, sc = small_world(4, K=10)
    line = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [50, 0, 0]], dtype=np.float32)
    nxt, created, deleted = w.update_topology(line, 1.5, 1)
    # query order 0,1,2: (0,1) (1,0) (1,2) (2,1), K-1 = 9 numbers each
    assert (created, deleted, nxt) == (4, 0, 37)
    assert [w.connections(r) for r in range(4)] == [[1], [0, 2], [1], []]
    nxt, created, deleted = w.update_topology(line, 1.5, nxt)
    assert (created, deleted, nxt) == (0, 0, 37)
    far = line.copy()
    far[2, 0] = 30
    nxt, created, deleted = w.update_topology(far, 1.5, nxt)
    assert (created, deleted) == (0, 2) and w.connections(1) == [0] and w.connections(2) == []


def test_hashmap_quirk_leaves_factors_behind():
    """0-{1,2}, 1-{0,3}: when every pair leaves range in one pass the map keeps (0->2), (1->3),
    (2->0), (3->1): pair (0,1) keeps its factors although both connection sets forget it, and
    gets a second set when it comes back in range."""
    w, sc = small_world(4, K=10)
    near = np.zeros((4, 3), dtype=np.float32)
    near[:, 0] = [0, 1, -1, 2]     # 0-1, 0-2, 1-3 within 1.2; 2-3, 0-3, 1-2 not
    nxt, created, _ = w.update_topology(near, 1.2, 1)
    assert created == 6 and [w.connections(r) for r in range(4)] == [[1, 2], [0, 3], [0], [1]]
    apart = np.zeros((4, 3), dtype=np.float32)
    apart[:, 0] = [0, 100, 200, 300]
    nxt, created, deleted = w.update_topology(apart, 1.2, nxt)
    assert (created, deleted) == (0, 4) and all(w.connections(r) == [] for r in range(4))
    # the (0,1) factors are still in both graphs: variable 1 of robot 1 still has an inbox slot from robot 0
    assert w.variable_inbox_graphs(1, 1).count(0) == 1
    assert w.variable_inbox_graphs(2, 1).count(0) == 0
    nxt, created, _ = w.update_topology(near, 1.2, nxt)
    assert created == 6
    assert w.variable_inbox_graphs(1, 1).count(0) == 2   # old and new factor 0 -> 1
    assert w.variable_inbox_graphs(2, 1).count(0) == 1


def test_removed_robot_leaves_the_queries():
    w, sc = small_world(4, K=10)
    line = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0]], dtype=np.float32)
    nxt, created, _ = w.update_topology(line, 1.5, 1)
    assert created == 6
    w.remove_robot(1)
    ptr, idx = w.neighbours(line, 1.5)
    assert [r.tolist() for r in rows_of(ptr, idx)] == [[], [], [3], [2]]
    nxt, created, deleted = w.update_topology(line, 1.5, nxt)
    # 0 and 2 each drop their factors towards 1 (and the messages robot 1's factors left behind)
    assert (created, deleted) == (0, 2)
    assert [w.connections(r) for r in range(4)] == [[], [], [3], [2]]
    assert w.variable_inbox_graphs(0, 1).count(1) == 0 and w.variable_inbox_graphs(2, 1).count(1) == 0
    import pytest
    with pytest.raises(RuntimeError):
        w.set_antenna(1, True)
    with pytest.raises(RuntimeError):
        w.remove_robot(1)
