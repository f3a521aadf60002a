"""mgx_batch_begin / mgx_batch_end: schedules recorded and submitted together, merged into as few resident launches as their
segments fit.  The bar: bit-identical to the oracle running the same schedules one by one, and every other call on the world
finds it as if each schedule had run when it was issued.  (Sharded worlds with the exchange in the engine: tests/test_gpu_direct_halo_mp.py,
mode "+batch".)"""
import numpy as np
import pytest

from magics_amd import scenarios as S

from parity import assert_identical, make_pair

pytestmark = pytest.mark.gpu


def test_batched_ticks_equal_the_oracle_and_merge_into_fewer_launches():
    sc = S.grid_scenario(64, 16, interrobot=True)
    eng, ref = make_pair(sc)
    eng.iterate(sc["steps"]); ref.iterate(sc["steps"])  # (the first launch of a world sizes its tables)
    assert eng.last_launch_count() == 1
    for n_ticks, want in ((3, 1), (7, 3), (1, 1)):
        with eng.batch() as b:
            for _ in range(n_ticks):
                eng.iterate(sc["steps"])
        assert (b.schedules, b.launches) == (n_ticks, want), (b.schedules, b.launches)  # three ticks of the 10 / 10 schedule per launch
        for _ in range(n_ticks):
            ref.iterate(sc["steps"])
        assert_identical(eng, ref, what=f"{n_ticks} batched ticks")


def test_other_calls_inside_a_batch_see_every_schedule_issued_so_far():
    sc = S.grid_scenario(36, 10, interrobot=True, pitch=2.0, comm_radius=5.0)
    eng, ref = make_pair(sc)
    mean = np.array([0.5, 0.25, 1.0, -1.0])
    eng.batch_begin()
    for w in (eng, ref):
        w.iterate(sc["steps"])
        w.iterate([3, 1, 2])
        w.change_prior(5, 9, mean)            # submits the two schedules first
        w.iterate(sc["steps"])
        w.set_antenna(7, False)               # ... and this one the third
        w.iterate([3, 3])
    got = eng.get_belief(5, 9)                # a read inside the batch: everything issued so far has run
    want = ref.get_belief(5, 9)
    assert all(np.array_equal(got[k], want[k]) for k in ("mean", "cov")) and got["valid"] == want["valid"]
    for w in (eng, ref):
        w.iterate([2, 3, 3])
    schedules, launches = eng.batch_end()
    assert schedules == 5 and launches >= 4
    assert_identical(eng, ref, what="calls between the schedules of a batch")
    for r in range(len(sc["robots"])):
        assert eng.message_counts(r) == ref.message_counts(r)
    with pytest.raises(RuntimeError, match="no batch is open"):
        eng.batch_end()
    eng.batch_begin()
    with pytest.raises(RuntimeError, match="open already"):
        eng.batch_begin()
    eng.batch_end()


def test_batches_on_worlds_that_run_launch_by_launch():
    """no inter-robot factors (nothing to hand over: every segment its own launch), and resident launches switched off"""
    sc = S.grid_scenario(32, 16, interrobot=False)
    eng, ref = make_pair(sc)
    with eng.batch():
        for _ in range(4):
            eng.iterate(sc["steps"])
    for _ in range(4):
        ref.iterate(sc["steps"])
    assert_identical(eng, ref, what="batched ticks without inter-robot factors")
    sc = S.grid_scenario(32, 16, interrobot=True)
    eng, ref = make_pair(sc)
    eng.set_resident_launches(False)
    with eng.batch() as b:
        for _ in range(4):
            eng.iterate(sc["steps"])
    assert b.launches > 4
    for _ in range(4):
        ref.iterate(sc["steps"])
    assert_identical(eng, ref, what="batched ticks, launch per segment")


@pytest.mark.parametrize("ticks_per_batch", [2, 3])
def test_batched_ticks_at_the_headline_size(ticks_per_batch):
    """BASELINE configs[2] at full size — 1000 x 16 + inter-robot factors, every one of the 1000 workgroups resident — with the
    submission bench.py times: `ticks_per_batch` ticks of the 10 / 10 schedule inside one mgx_batch_begin / _end, merged into ONE
    resident launch (the last [external] and the next tick's first internal iteration fused into one segment).  Twice in a row,
    so that the second batch starts from what the first one wrote back; bit-identical to the oracle running tick after tick."""
    import oracle
    sc = S.grid_scenario(1000, 16, interrobot=True)
    from magics_amd import World
    eng, ref = World(sc["params"]), oracle.OracleWorld(sc["params"], threads=16)
    assert S.populate(eng, sc) == S.populate(ref, sc)
    for rep in range(2):
        with eng.batch() as b:
            for _ in range(ticks_per_batch):
                eng.iterate(sc["steps"])
        assert (b.schedules, b.launches) == (ticks_per_batch, 1), (b.schedules, b.launches)
        assert eng.last_launch_count() == 1
        for _ in range(ticks_per_batch):
            ref.iterate(sc["steps"])
        assert_identical(eng, ref, what=f"1000 x 16 + ir, {ticks_per_batch} ticks in one resident launch, batch {rep}")
    assert all(np.isfinite(x).all() for x in eng.read_beliefs())
