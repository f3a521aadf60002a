"""Inter-robot factors created WHILE their kind is switched off (factor/mod.rs:307-310: a disabled factor drops what is sent to
it, so the two messages that fill a new factor's inbox are lost): once the kind is enabled again such a factor has no inbox
KEYS until its variables deliver, and FactorNode::update answers the keys it has (factor/mod.rs:336-349,412-449) — nothing is
sent to the other robot while its key is missing, and while only ITS key is there the factor linearises with that variable in
slot 0 whatever the graphs' order.  Round 1 documented this corner as not reproduced; the engine now follows it (the keys fill
structurally, the host knows them, k_keyless_ir evaluates such factors in front of the sweep launch)."""
import numpy as np
import pytest

from magics_amd import scenarios as S
from parity import assert_identical, make_pair

pytestmark = pytest.mark.gpu


def _same_counts(eng, ref, n, what):
    for r in range(n):
        assert eng.message_counts(r) == ref.message_counts(r), (what, r)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_connections_created_while_interrobot_factors_are_off(seed):
    n, K = 9, 10
    full = S.grid_scenario(n, K, interrobot=True, pitch=2.0, comm_radius=3.5)
    sc = dict(full, ir=[])                                     # robots only; the connections come later
    eng, ref = make_pair(sc)
    rng = np.random.default_rng(seed)
    idle, silent = (int(x) for x in rng.choice(n, 2, replace=False))
    mask_on = sc["params"]["enable_mask"]

    steps = []

    def both(fn, what):
        fn(eng)
        fn(ref)
        steps.append(what)
        assert_identical(eng, ref, what=f"seed {seed}: {what}")
        _same_counts(eng, ref, n, what)
    both(lambda w: w.iterate([3, 3]), "two iterations without connections")
    both(lambda w: w.set_enabled(mask_on & ~S.EN_IR), "inter-robot factors switched off")
    both(lambda w: [w.ir_connect(a, b, n0) for a, b, n0 in full["ir"]], "connections created while the kind is off")
    both(lambda w: w.iterate([3, 1, 3]), "iterations while off: deliveries are dropped")
    both(lambda w: (w.set_idle(idle, True), w.set_antenna(silent, False)), "one robot idle, one off the air")
    both(lambda w: w.set_enabled(mask_on), "switched on again: the factors have no keys")
    for k, st in enumerate([2, 2, 1, 2, 3, 2, 1, 3]):          # external sweeps first: only the targets' keys arrive
        both(lambda w: w.iterate([st]), f"step {k} ({st}) after switching on")
    both(lambda w: w.change_prior(idle, K - 1, np.array([0.3, -0.2, 1.0, 0.5])), "a prior change delivers a key too")
    both(lambda w: w.iterate([2, 3]), "after the prior change")
    both(lambda w: (w.set_idle(idle, False), w.set_antenna(silent, True)), "everybody back")
    both(lambda w: w.iterate(sc["steps"]), "a whole schedule: every key is there now")
    both(lambda w: w.iterate(sc["steps"]), "and the resident path again")


def test_half_filled_keys_survive_a_second_switch_off():
    """The soak's seed 10094: created while off, on for one internal sweep (only the owners' keys arrive), off again (the
    records freeze), on again with an external sweep first — the thaw must leave the factors that still lack a key alone."""
    n, K = 9, 10
    full = S.grid_scenario(n, K, interrobot=True, pitch=2.0, comm_radius=3.5)
    sc = dict(full, ir=[])
    eng, ref = make_pair(sc)
    mask_on = sc["params"]["enable_mask"]

    def both(fn, what):
        fn(eng)
        fn(ref)
        assert_identical(eng, ref, what=what)
        _same_counts(eng, ref, n, what)
    both(lambda w: w.iterate([3]), "one iteration without connections")
    both(lambda w: w.set_enabled(mask_on & ~S.EN_IR), "off")
    both(lambda w: [w.ir_connect(a, b, n0) for a, b, n0 in full["ir"]], "connections created while the kind is off")
    both(lambda w: w.set_enabled(mask_on), "on")
    both(lambda w: w.iterate([1]), "one internal sweep: the owners' keys only")
    both(lambda w: w.set_enabled(mask_on & ~S.EN_IR & ~S.EN_DYN), "off again, dynamic factors too")
    both(lambda w: w.iterate([2, 1, 2, 1]), "iterations while off")
    both(lambda w: w.set_enabled(mask_on), "on again: thaw and half-filled keys together")
    for k, st in enumerate([2, 2, 1, 2, 3, 3]):
        both(lambda w: w.iterate([st]), f"step {k} ({st}) after the second switch-on")
