"""MessageCount bookkeeping of the oracle (FG/mod.rs:29-137; variable.rs:179-191,299-332;
factor/mod.rs:307-317,353-367,410-452; factorgraph.rs:876-890) against hand counts."""
import numpy as np

import oracle
from magics_amd import scenarios as S


def test_single_robot_counts_by_hand():
    K = 10
    sc = S.grid_scenario(1, K, interrobot=False)          # dynamics + obstacle enabled, tracking disabled
    w = oracle.OracleWorld(sc["params"])
    S.populate(w, sc)
    n_dyn_edges, n_obs, n_trk = 2 * (K - 1), K - 2, K - 2
    keys = n_dyn_edges + n_obs + n_trk                     # inbox keys of the variables (disabled kinds keep theirs)
    # creation: every edge gives the variable one receive, and the factor one if it is enabled
    assert w.message_counts(0) == (0, 0, keys + n_dyn_edges + n_obs, 0)
    w.internal_factor_iteration()
    sent = n_dyn_edges + n_obs                             # one message per inbox key of every updated factor
    assert w.message_counts(0) == (sent, 0, keys + n_dyn_edges + n_obs + sent, 0)
    w.internal_variable_iteration()
    # one response per inbox key (tracking keys included), received by the enabled factors only
    assert w.message_counts(0) == (sent + keys, 0, keys + 2 * (n_dyn_edges + n_obs) + sent, 0)
    w.change_prior(0, K - 1, np.array([1.0, 2.0, 0.0, 0.0]))
    # change_prior's sends are not added to the counter (variable.rs:208-221 keeps them local);
    # the one dynamic factor of the last variable receives
    assert w.message_counts(0) == (sent + keys, 0, keys + 2 * (n_dyn_edges + n_obs) + sent + 1, 0)


def test_pair_counts_follow_the_radio_gates():
    K = 10
    sc = S.grid_scenario(2, K, interrobot=True, pitch=2.0, comm_radius=5.0)
    assert len(sc["ir"]) == 2
    w = oracle.OracleWorld(sc["params"])
    S.populate(w, sc)
    base = [w.message_counts(r) for r in range(2)]
    # connect: own variables +(K-1) internal, own factors +(K-1) internal and +(K-1) external (the
    # other variable's belief), other's variables +(K-1) external
    keys = 2 * (K - 1) + 2 * (K - 2)
    assert base[0] == (0, 0, keys + 2 * (K - 1) + (K - 2) + 2 * (K - 1), 2 * (K - 1))
    w.iterate([2])                                          # one external iteration, both on air
    after = [w.message_counts(r) for r in range(2)]
    for r in range(2):
        d = np.subtract(after[r], base[r])
        # factors: (K-1) x (1 internal + 1 external) sent; variables: all keys answered
        # (internal keys incl. own inter-robot factors, external keys = the other's factors)
        assert tuple(d) == ((K - 1) + keys + (K - 1), (K - 1) + (K - 1), 0, (K - 1) + (K - 1)), (r, d)
    w.set_antenna(1, False)
    w.iterate([2])
    d0 = np.subtract(w.message_counts(0), after[0])
    d1 = np.subtract(w.message_counts(1), after[1])
    assert tuple(d1) == (0, 0, 0, 0)                        # robot 1 neither iterates externally nor receives
    assert tuple(d0) == ((K - 1) + keys + (K - 1), (K - 1) + (K - 1), 0, 0)  # robot 0 sends, nothing comes back
