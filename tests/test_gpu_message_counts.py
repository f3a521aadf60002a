"""MessageCount parity (factorgraph/mod.rs:29-137, export.rs:434-439): the engine keeps the
counters on the host from its launch log; they must equal the oracle's, which counts at every
send / receive like the reference."""
import numpy as np
import pytest

from magics_amd import scenarios as S

from parity import assert_identical, make_pair

pytestmark = pytest.mark.gpu


def same_counts(eng, ref, n, what):
    for r in range(n):
        assert eng.message_counts(r) == ref.message_counts(r), (what, r, eng.message_counts(r), ref.message_counts(r))


def test_counts_with_tracking_gate_and_ticks():
    n = 16
    sc = S.grid_scenario(n, 10, interrobot=True, tracking=True, pitch=2.5, comm_radius=5.0)
    eng, ref = make_pair(sc)
    same_counts(eng, ref, n, "after construction")
    args = S.tick_inputs(sc)
    for w in (eng, ref):
        w.iterate([1, 2, 3])            # tracking factors stay closed below 10 factor iterations
    same_counts(eng, ref, n, "3 steps")
    for tick in range(4):
        for w in (eng, ref):
            w.update_priors(**args)
            w.iterate(sc["steps"])
        same_counts(eng, ref, n, f"tick {tick}")
    assert_identical(eng, ref, what="counts scenario")


def test_counts_under_gating_and_fine_grained_calls():
    n = 9
    sc = S.grid_scenario(n, 10, interrobot=True, pitch=2.0, comm_radius=5.0)
    eng, ref = make_pair(sc)
    pattern = np.random.default_rng(1).random(n) > 0.5

    def script(w):
        w.iterate([3, 3])
        w.set_antenna(2, False)
        w.set_idle(5, True)
        w.iterate([3, 1, 2, 3])
        w.change_prior(4, 9, np.array([0.5, 0.25, 1.0, -1.0]))
        w.change_prior(4, 3, np.array([0.1, 0.2, 0.0, 0.0]))
        w.internal_factor_iteration()
        w.internal_variable_iteration()
        w.external_factor_iteration()
        w.external_variable_iteration()
        w.set_antennas(np.arange(n), pattern)
        w.iterate([3, 3, 3])
        w.set_idle(5, False)
        w.iterate([2, 1])
    script(eng)
    script(ref)
    same_counts(eng, ref, n, "gating script")
    assert_identical(eng, ref, what="gating script")


def test_counts_follow_topology_changes_and_removal():
    n, K = 12, 10
    sc = S.circle_scenario(n, K, circle_radius=20.0, n_internal=10, n_external=10)
    sc["ir"] = []
    eng, ref = make_pair(sc)
    args = S.tick_inputs(sc)
    nxt = {id(eng): 1, id(ref): 1}
    alive = np.ones(n, dtype=bool)
    for tick in range(30):
        if tick == 12:
            for w in (eng, ref):
                w.remove_robot(5)
            alive[5] = False
            keep = np.nonzero(alive)[0]
            args = S.tick_inputs(sc)
            args = dict(args, robots=args["robots"][keep], waypoints_xy=args["waypoints_xy"][keep],
                        time_scale=args["time_scale"][keep], what=args["what"][keep])
        for w in (eng, ref):
            _, _, mu = w.read_beliefs()
            cur = mu.reshape(n, K, 4)[:, 0, :2]
            pos = np.stack([cur[:, 0], np.full(n, 0.5), cur[:, 1]], axis=1).astype(np.float32)
            nxt[id(w)] = w.update_topology(pos, 22.0, nxt[id(w)])[0]
            w.update_priors(**args)
            w.iterate(sc["steps"])
        if tick % 5 == 4:
            same_counts(eng, ref, n, f"tick {tick}")
    assert_identical(eng, ref, what="topology + removal with counts")


def test_counts_over_a_log_that_fills():
    """A driver that iterates without reading: the engine's launch log takes three entries per schedule and is flushed when it
    holds four thousand — summed once per class of robots (on air / silent / idle), a robot walking it only while its tracking
    gate is closed (mgx_world_counters.inc).  1500 schedules of a small world with tracking factors, one robot silent and one
    idle from the start, a robot idle until half way: counters and beliefs equal the oracle's at the end."""
    n = 6
    sc = S.grid_scenario(n, 10, interrobot=True, tracking=True, pitch=2.5, comm_radius=5.0)
    eng, ref = make_pair(sc)
    for w in (eng, ref):
        w.set_antenna(1, False)  # silent: internal sweeps only
        w.set_idle(4, True)      # idle: nothing
        w.set_idle(2, True)      # ... until half way (its tracking gate opens in the SECOND log)
        w.iterate([1, 2, 3])     # (the gate of the others: still closed when the long run begins)
    steps = sc["steps"]
    for w in (eng, ref):
        for i in range(1500):
            if i == 1400:
                w.set_idle(2, False)
            w.iterate(steps)
    same_counts(eng, ref, n, "1500 schedules")
    assert_identical(eng, ref, what="1500 schedules, nothing read in between")
