"""Whole driver ticks on the device: update_prior_of_horizon_state +
update_prior_of_current_state_v3 (robot.rs:2182-2338) as one launch, then the GBP schedule —
robots actually move; trajectories must equal the oracle's bit for bit."""
import numpy as np
import pytest

from magics_amd import scenarios as S

from parity import assert_identical, make_pair

pytestmark = pytest.mark.gpu


def _run_ticks(w, sc, n_ticks, hz=10.0):
    args = S.tick_inputs(sc, hz)
    for _ in range(n_ticks):
        w.update_priors(**args)
        w.iterate(sc["steps"])


def test_circle_scenario_ticks_robots_move():
    sc = S.circle_scenario(10, 10, n_internal=10, n_external=10)
    eng, ref = make_pair(sc)
    start = np.array([rb["mean0"][0, :2] for rb in sc["robots"]])
    for block in range(4):
        _run_ticks(eng, sc, 15)
        _run_ticks(ref, sc, 15)
        assert_identical(eng, ref, what=f"circle, {15 * (block + 1)} ticks")
    _, _, mu = eng.read_beliefs()
    now = mu.reshape(10, 10, 4)[:, 0, :2]
    moved = np.linalg.norm(now - start, axis=1)
    assert moved.min() > 5.0, moved  # every robot has left its start by metres


def test_grid_scenario_ticks_with_obstacles_and_interrobot():
    sc = S.grid_scenario(36, 16, interrobot=True, pitch=4.0, comm_radius=8.0)
    eng, ref = make_pair(sc)
    for block in range(3):
        _run_ticks(eng, sc, 10)
        _run_ticks(ref, sc, 10)
        assert_identical(eng, ref, what=f"grid 36x16, {10 * (block + 1)} ticks")


def test_update_priors_subsets():
    # only some robots get a horizon update (no next waypoint for the others) / are not idle
    sc = S.grid_scenario(9, 10, interrobot=True, pitch=2.0, comm_radius=5.0)
    eng, ref = make_pair(sc)
    a = S.tick_inputs(sc)
    a["what"][[1, 4]] = 2      # current-state update only
    a["what"][[7]] = 1         # horizon only
    keep = np.array([0, 1, 2, 4, 5, 7, 8])
    a = dict(a, robots=a["robots"][keep], waypoints_xy=a["waypoints_xy"][keep], time_scale=a["time_scale"][keep], what=a["what"][keep])
    for w in (eng, ref):
        w.iterate([3, 3])
        for _ in range(5):
            w.update_priors(**a)
            w.iterate([3, 3, 3])
    assert_identical(eng, ref, what="update_priors on subsets")


def test_one_call_ticks():
    """mgx_tick: the prior updates ride in the launch that opens the tick (applied by each robot's workgroup
    to the image it has just staged).  Whole trajectories, with inter-robot factors, tracking, robots left
    out of the update, idle and silent robots and a message-count check, equal the oracle's bit for bit —
    and a schedule that opens with an external iteration takes the unfused route."""
    sc = S.grid_scenario(36, 16, interrobot=True, pitch=4.0, comm_radius=8.0, tracking=True)
    eng, ref = make_pair(sc)
    a = S.tick_inputs(sc)
    a["what"][[1, 4]] = 2
    a["what"][[7]] = 1
    a["what"][[9]] = 0
    keep = np.array([r for r in range(36) if r not in (3, 20)])
    sub = dict(a, robots=a["robots"][keep], waypoints_xy=a["waypoints_xy"][keep], time_scale=a["time_scale"][keep], what=a["what"][keep])
    for w in (eng, ref):
        w.set_idle(5, True)
        w.set_antenna(6, False)
    for block in range(3):
        for w in (eng, ref):
            for _ in range(8):
                w.tick(steps=sc["steps"], **sub)
        assert_identical(eng, ref, what=f"one-call ticks, block {block}")
        assert eng.message_counts(7) == ref.message_counts(7) and eng.message_counts(5) == ref.message_counts(5)
    for w in (eng, ref):
        w.set_idle(5, False)
        for _ in range(3):
            w.tick(steps=[2, 3, 1, 3], **sub)   # opens with an external iteration: update_priors + iterate
        for _ in range(3):
            w.tick(steps=[1, 1, 3], **sub)
    assert_identical(eng, ref, what="one-call ticks, mixed schedules")
    # K = 10 without inter-robot factors, circle: robots move
    sc2 = S.circle_scenario(10, 10, n_internal=10, n_external=10)
    sc2["ir"] = []
    eng2, ref2 = make_pair(sc2)
    b = S.tick_inputs(sc2)
    for w in (eng2, ref2):
        for _ in range(40):
            w.tick(steps=sc2["steps"], **b)
    assert_identical(eng2, ref2, what="one-call ticks, circle")
