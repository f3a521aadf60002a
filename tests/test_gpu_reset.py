"""FactorGraph::reset_variables / reset_tracking_factors (factorgraph.rs:1541-1590; VariableNode::reset, variable.rs:350-360;
FactorNode::empty_inbox, factor/mod.rs:480-483; TrackingFactor timeout, tracking.rs:153-155,362-371) — what the reference
does to a robot's graph when a global path arrives (robot.rs:766-769): engine == oracle bit for bit, through the reset
and through the ten skipped tracking updates that follow."""
import numpy as np
import pytest

from magics_amd import World, hostlib, scenarios as S
from parity import assert_identical, both, make_pair

pytestmark = pytest.mark.gpu


def _script(sc, robots, resets_tracking=True):
    K = sc["K"]
    rng = np.random.default_rng(11)
    new_means = {r: sc["robots"][r]["mean0"] + rng.normal(0, 0.3, size=(K, 4)) for r in robots}

    def run(w, checkpoints):
        tick = S.tick_inputs(sc)
        for _ in range(2):
            w.tick(steps=sc["steps"], **tick)
        for r in robots:
            w.reset_variables(r, new_means[r])          # the reference's call: (means, 1e30, +inf)
            if resets_tracking:
                w.reset_tracking_factors(r)
        checkpoints.append([x.copy() for x in w.read_beliefs()])
        for _ in range(3):                               # > 10 factor sweeps: the timeout runs out inside
            w.tick(steps=sc["steps"], **tick)
            checkpoints.append([x.copy() for x in w.read_beliefs()])
    return run, new_means


@pytest.mark.parametrize("interrobot", [False, True])
def test_reset_variables_and_tracking_factors(interrobot):
    sc = S.grid_scenario(16, 10, interrobot=interrobot, tracking=True, pitch=2.0, comm_radius=5.0)
    eng, ref = make_pair(sc)
    run, new_means = _script(sc, robots=[3, 8])
    ce, cr = [], []
    run(eng, ce)
    run(ref, cr)
    for k, (a, b) in enumerate(zip(ce, cr)):
        for name, x, y in zip(("eta", "lam", "mean"), a, b):
            assert np.array_equal(x, y, equal_nan=True), f"checkpoint {k}: {name} differs"
    # right after the reset: the belief holds the given means and diag(1e30 | inf) as its precision, eta untouched
    K = sc["K"]
    _, lam, mu = ce[0]
    assert np.array_equal(mu[3 * K:4 * K], new_means[3])
    assert lam[3 * K, 0, 0] == 1e30 and np.isinf(lam[3 * K + 1, 2, 2]) and lam[3 * K + 1, 0, 1] == 0.0
    assert_identical(eng, ref, what="after reset_variables + reset_tracking_factors")


def test_tracking_timeout_changes_the_result():
    """guard: the ten skipped updates are really skipped (with and without reset_tracking_factors differ)"""
    sc = S.grid_scenario(9, 10, interrobot=False, tracking=True, pitch=2.0)
    outs = []
    for flag in (True, False):
        eng, ref = make_pair(sc)
        run, _ = _script(sc, robots=[4], resets_tracking=flag)
        ce, cr = [], []
        run(eng, ce)
        run(ref, cr)
        assert all(np.array_equal(x, y, equal_nan=True) for a, b in zip(ce, cr) for x, y in zip(a, b))
        outs.append(ce[1][2])
    assert not np.array_equal(outs[0], outs[1])


def test_reset_variables_checks_the_number_of_means():
    """the reference asserts variable_indices.len() == means.len() (factorgraph.rs:1548): a short list is an error, not a read
    past its end"""
    sc = S.grid_scenario(4, 10, interrobot=False)
    eng = World(sc["params"])
    S.populate(eng, sc)
    with pytest.raises(hostlib.MgxError):
        eng.reset_variables(0, np.zeros((9, 4)))
    with pytest.raises(hostlib.MgxError):
        eng.reset_variables(0, np.zeros((11, 4)))
    eng.reset_variables(0, np.zeros((10, 4)))
