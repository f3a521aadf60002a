"""BASELINE.json configs[3] and configs[4] AT THEIR FULL SIZE against the CPU oracle, bit for bit.

configs[3]: 8000 robots x 16 horizon + inter-robot factors, robots sharded 8 ways — the eight
ranks run inside one process on the one GPU of the box (sharded.LocalCluster: same ghosts, same
halo plan, same kernels as eight processes; only the transport of the all-to-all-v differs), over
the collective transport and over the direct (peer-mapped stores) one, one driver tick each, and
the single-world oracle is the checker (robot.rs:1769-1861, 2182-2338).

configs[4]: 4000 robots x 32 horizon with tracking + inter-robot + obstacle + dynamic factors — on
the 20 x 20 crossroads of the Junction Twoway environment rasterised on the device (SURVEY §8d
"Config 5") and on the synthetic grid (7.6 neighbours per robot), two ticks each so that the
tracking factors are past their ten-sweep gate (factorgraph.rs:701, tracking.rs:197-346).  The
reference's arithmetic itself leaves the finite range on this combination of factors within a tick
(parity.assert_identical_where_finite says why): engine and oracle must agree bit for bit wherever the
oracle holds a number, and the same workload WITHOUT inter-robot factors — which stays finite — must
be bit-identical outright.
"""
import os

import numpy as np
import pytest

import oracle
from magics_amd import World, scenarios as S, sharded
from parity import assert_identical, assert_identical_where_finite

RESIDENT = os.environ.get("MGX_PERSISTENT", "1") != "0"  # MGX_PERSISTENT=0 runs the same tests on the launch-per-segment path

pytestmark = pytest.mark.gpu

ORACLE_THREADS = 16


# ---- configs[3] -------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def config3():
    """The scenario and the oracle's beliefs after one tick (computed once for both transports)."""
    sc = S.grid_scenario(8000, 16, interrobot=True)
    assert len(sc["robots"]) == 8000 and sc["K"] == 16
    ref = oracle.OracleWorld(sc["params"], threads=ORACLE_THREADS)
    S.populate(ref, sc)
    tick = S.tick_inputs(sc)
    ref.tick(steps=sc["steps"], **tick)
    beliefs = ref.read_beliefs()
    ref.close()
    return sc, tick, beliefs


class _Frozen:
    """read_beliefs() of an oracle that has already been run and released"""

    def __init__(self, beliefs):
        self._b = beliefs

    def read_beliefs(self):
        return self._b


def _check_cluster(cluster, sc):
    owners = cluster.ranks[0].plan.owner
    assert len(cluster.ranks) == 8 and sorted(np.bincount(owners)) == [1000] * 8
    assert all(sw.plan.ghosts for sw in cluster.ranks), "every strip has neighbours across its border"
    n_ext = sum(1 for s in sc["steps"] if s & 2)
    assert n_ext == 10


def test_config3_full_size_sharded_8_ways_collective(config3):
    sc, tick, beliefs = config3
    cluster = sharded.LocalCluster(sc, 8, World)
    _check_cluster(cluster, sc)
    cluster.tick(steps=sc["steps"], **tick)
    assert_identical(cluster, _Frozen(beliefs), what="configs[3] 8000 x 16 + ir, 8 ranks, collective transport, 1 tick")


def test_config3_full_size_sharded_8_ways_direct(config3):
    import torch
    sc, tick, beliefs = config3
    streams = []

    def make(params):
        streams.append(torch.cuda.Stream())
        return World(params, stream=streams[-1].cuda_stream)
    cluster = sharded.LocalCluster(sc, 8, make, direct=True)
    _check_cluster(cluster, sc)
    assert all(sw.direct for sw in cluster.ranks)
    cluster.tick(steps=sc["steps"], **tick)
    assert_identical(cluster, _Frozen(beliefs), what="configs[3] 8000 x 16 + ir, 8 ranks, direct transport, 1 tick")
    for sw in cluster.ranks:
        assert sw.world.halo_direct_status() == 10  # one exchange per external iteration, none timed out


# ---- configs[4] -------------------------------------------------------------------------------------
def _two_ticks(sc, finite=False):
    eng, ref = World(sc["params"]), oracle.OracleWorld(sc["params"], threads=ORACLE_THREADS)
    assert S.populate(eng, sc) == S.populate(ref, sc)
    tick = S.tick_inputs(sc)
    for t in range(2):
        for w in (eng, ref):
            w.tick(steps=sc["steps"], **tick)
        if finite:
            assert_identical(eng, ref, what=f"{sc['name']} tick {t}")
            assert all(np.isfinite(x).all() for x in eng.read_beliefs())
        else:
            assert_identical_where_finite(eng, ref, what=f"{sc['name']} tick {t}")
    return eng, ref


def test_config4_full_size_junction_tiles():
    """BASELINE configs[4] as SURVEY §8d words it, at full size, in the regime where the reference's arithmetic holds numbers: the
    robots iterate one tick on their own before create_interrobot_factors hooks them up (the reference's driver order; see
    scenarios.junction_scenario) — dynamic + obstacle + inter-robot + tracking factors, 4000 x 32: bit-identical, every belief finite."""
    sc = S.junction_scenario(4000, 32, tiles=20)
    assert len(sc["robots"]) == 4000 and sc["K"] == 32 and sc["params"]["enable_mask"] == 15
    assert all(rb["path"] is not None and len(rb["path"]) >= 2 for rb in sc["robots"])
    assert len(sc["ir_late"]) > 4000 and not sc["ir"] and sc["connect_after_ticks"] == 1
    eng, ref = _two_ticks(sc, finite=True)
    # the tracking factors did fire: switching them off changes the result
    sc2 = dict(sc, params=dict(sc["params"], enable_mask=7))
    off = World(sc2["params"])
    S.populate(off, sc2)
    tick = S.tick_inputs(sc2)
    for _ in range(2):
        off.tick(steps=sc2["steps"], **tick)
    assert not np.array_equal(off.read_beliefs()[2], eng.read_beliefs()[2], equal_nan=True)


def test_config4_full_size_sharded_8_ways():
    """BASELINE configs[4] in its own layout — "4000 robots x 32 horizon ... 8 x MI355X" — in the regime it is quoted on: 4000 x 32
    lanes on the 20 x 20 crossroads, dynamic + obstacle + inter-robot + tracking factors, the robots hooked up after their first
    tick (sc["ir_late"]), sharded 8 ways.  The eight ranks run inside one process on the one GPU of the box (one rank's 500
    workgroups of 75 KB LDS are resident together, eight ranks' are not — the in-launch hand-off of this shape is covered at 2 x
    200 robots in test_gpu_sharded.py): bit-identical to the single-world oracle, every belief finite, two ticks."""
    sc = S.junction_scenario(4000, 32, tiles=20)
    assert len(sc["robots"]) == 4000 and sc["K"] == 32 and sc["params"]["enable_mask"] == 15
    assert len(sc["ir_late"]) > 4000 and not sc["ir"] and sc["connect_after_ticks"] == 1
    cluster = sharded.LocalCluster(sc, 8, World)
    owners = cluster.ranks[0].plan.owner
    assert len(cluster.ranks) == 8 and sorted(np.bincount(owners)) == [500] * 8
    assert all(sw.plan.ghosts for sw in cluster.ranks) and not any(sw.late_pending for sw in cluster.ranks)
    ref = oracle.OracleWorld(sc["params"], threads=ORACLE_THREADS)
    S.populate(ref, sc)
    tick = S.tick_inputs(sc)
    for t in range(2):
        cluster.tick(steps=sc["steps"], **tick)
        ref.tick(steps=sc["steps"], **tick)
        assert_identical(cluster, ref, what=f"configs[4] 4000 x 32 junction, connected after the first tick, 8 ranks, tick {t}")
        assert all(np.isfinite(x).all() for x in cluster.read_beliefs())


def test_config4_everything_at_once_leaves_the_finite_range_identically():
    """the robustness variant: inter-robot AND tracking factors on robots that have never iterated — the reference's arithmetic
    itself goes through 1e+100 to inf / NaN within a tick; engine and oracle agree wherever the oracle holds a number"""
    sc = S.junction_scenario(4000, 32, tiles=20, connect_after_ticks=0)
    assert len(sc["ir"]) > 4000
    eng, ref = _two_ticks(sc)
    assert not all(np.isfinite(x).all() for x in ref.read_beliefs())


def test_config4_full_size_grid():
    sc = S.grid_scenario(4000, 32, interrobot=True, tracking=True)
    assert len(sc["ir"]) / 4000 > 7.0
    _two_ticks(sc)


def test_config4_full_size_junction_tiles_tracking_without_interrobot_stays_finite():
    """the same 4000 x 32 lanes with dynamic + obstacle + TRACKING factors (the K = 32 lane maps, the BIG tracking
    lanes) where the reference's arithmetic stays finite: bit-identical, every belief a number"""
    sc = S.junction_scenario(4000, 32, tiles=20, interrobot=False)
    assert sc["params"]["enable_mask"] == 13 and not sc["ir"]
    _two_ticks(sc, finite=True)


@pytest.mark.parametrize("K,n", [(16, 400), (12, 150), (10, 64)])
def test_resident_launch_through_non_finite_beliefs(K, n):
    """Inter-robot + tracking factors on the grid leave the finite range within a tick in the reference's own arithmetic (DESIGN.md
    §2) — here at horizons of at most 16 variables, where a resident schedule launch runs the external and the internal variable
    sweep side by side and a belief update that finds its precision "zero", singular or its covariance non-finite has to fall back
    on what the other sweep left (variable.rs:273-297): identical wherever the oracle holds a number, tick after tick."""
    sc = S.grid_scenario(n, K, interrobot=True, tracking=True)
    eng, ref = World(sc["params"]), oracle.OracleWorld(sc["params"], threads=ORACLE_THREADS)
    assert S.populate(eng, sc) == S.populate(ref, sc)
    tick = S.tick_inputs(sc)
    worst = 0.0
    for t in range(6):
        for w in (eng, ref):
            w.tick(steps=sc["steps"], **tick)
        assert eng.last_launch_count() == 1 or not RESIDENT  # the resident path
        worst = max(worst, assert_identical_where_finite(eng, ref, what=f"{sc['name']} K={K} tick {t}", max_nan_only_mismatch=5e-3))
    if K == 16:
        assert worst > 0.0  # this one does leave the finite range: the fall-back was exercised


@pytest.mark.parametrize("K,tracking", [(32, False), (32, True), (21, False)])
def test_resident_launch_beyond_64_kb_of_lds_and_128_edges(K, tracking):
    """Long horizons on the crowded grid: an interior robot has 8 x (K - 1) > 128 inter-robot edges (a second round of the
    edge lanes, gathered by quads like the first) and, at K = 32, 75 KB of LDS per resident workgroup (the kernel is told to
    accept more than 64 KB) — the whole schedule still runs as one launch and matches the oracle."""
    sc = S.grid_scenario(144, K, interrobot=True, tracking=tracking)
    assert max(np.bincount([b for _, b, _ in sc["ir"]])) * (K - 1) > 128
    eng, ref = World(sc["params"]), oracle.OracleWorld(sc["params"], threads=ORACLE_THREADS)
    assert S.populate(eng, sc) == S.populate(ref, sc)
    tick = S.tick_inputs(sc)
    for t in range(3):
        for w in (eng, ref):
            w.tick(steps=sc["steps"], **tick)
        assert eng.last_launch_count() == 1 or not RESIDENT
        if tracking:
            assert_identical_where_finite(eng, ref, what=f"{sc['name']} tick {t}", max_nan_only_mismatch=5e-3)
        else:
            assert_identical(eng, ref, what=f"{sc['name']} tick {t}")


def test_resident_launch_with_one_directional_connections():
    """Connections that exist in one direction only (the reference's bookkeeping can leave them so): the robot whose records are
    read has no edge of its own towards the reader, which it must still wait for before it overwrites the buffer the reader
    uses (the peer table is the symmetric closure) — whole schedules as one launch, bit-identical."""
    sc = S.grid_scenario(400, 16, interrobot=True)
    sc = dict(sc, ir=[(a, b, n0) for a, b, n0 in sc["ir"] if (a < b) == ((a + b) % 3 != 0)])
    eng, ref = World(sc["params"]), oracle.OracleWorld(sc["params"], threads=ORACLE_THREADS)
    assert S.populate(eng, sc) == S.populate(ref, sc)
    tick = S.tick_inputs(sc)
    for t in range(5):
        for w in (eng, ref):
            w.tick(steps=sc["steps"], **tick)
        assert eng.last_launch_count() == 1 or not RESIDENT
        assert_identical(eng, ref, what=f"one-directional connections, tick {t}")


def test_resident_launch_next_to_another_tenant_of_the_gpu():
    """Every workgroup of a resident schedule launch waits for its neighbours INSIDE the launch, and a plain launch promises
    co-residency to nobody: here a second stream of the process keeps the CUs busy with f64 matrix products while a 1000-robot
    world ticks, so that part of a resident launch's grid cannot start (measured: without the residency census such a launch
    waits until its 2 s bound and leaves the world invalid — the workgroups that spin hold what the other tenant's next
    workgroups need, and the tail of the grid queues behind those).  With the census the launch notices before it has written
    anything, returns, and the host runs that schedule launch by launch (and the next few, backing off): the beliefs are the
    oracle's, bit for bit, and no error is raised."""
    import torch
    sc = S.grid_scenario(1000, 16, interrobot=True)
    eng, ref = World(sc["params"]), oracle.OracleWorld(sc["params"], threads=ORACLE_THREADS)
    assert S.populate(eng, sc) == S.populate(ref, sc)
    tick = S.tick_inputs(sc)
    hog = torch.cuda.Stream()
    a = torch.randn(3072, 3072, dtype=torch.float64, device="cuda")
    with torch.cuda.stream(hog):
        for _ in range(6):  # ~0.1 s of other work in flight on the second stream
            a = (a @ a) * 1e-3
    for t in range(3):
        eng.tick(steps=sc["steps"], **tick)
        with torch.cuda.stream(hog):
            a = (a @ a) * 1e-3
        assert eng.last_launch_count() in (1, len(sc["steps"]) + 1)  # resident, or sent back and run launch by launch
    eng.synchronize()  # raises if a wait inside a resident launch gave up
    hog.synchronize()
    for t in range(3):
        ref.tick(steps=sc["steps"], **tick)
    assert_identical(eng, ref, what="resident launches next to a second stream's matrix products")


def test_resident_launch_queries_and_the_declining_world():
    """mgx_resident_ready / _outcome / _stats on a world of its own: which form a schedule takes is known before it is issued, what
    became of its launch after; a world told to decline (mgx_set_resident_launches(w, 2)) runs launch by launch like one told not
    to try — beliefs of the oracle either way."""
    from magics_amd import hostlib
    sc = S.grid_scenario(144, 16, interrobot=True)
    eng, ref = World(sc["params"]), oracle.OracleWorld(sc["params"])
    assert S.populate(eng, sc) == S.populate(ref, sc)
    assert eng.resident_ready(sc["steps"]) and not eng.resident_ready([1]) and not eng.resident_ready([])  # (one segment: nothing to keep resident for)
    assert eng.resident_outcome() == hostlib.RESIDENT_NONE
    eng.iterate(sc["steps"])
    assert eng.resident_outcome() == hostlib.RESIDENT_RAN and eng.resident_outcome() == hostlib.RESIDENT_NONE
    assert eng.resident_stats() == (1, 0, 0) and eng.last_launch_count() == 1
    eng.set_resident_launches("decline")
    assert not eng.resident_ready(sc["steps"])
    eng.iterate(sc["steps"])
    assert eng.resident_outcome() == hostlib.RESIDENT_NONE and eng.last_launch_count() == len(sc["steps"]) + 1
    eng.set_resident_launches(False)
    assert not eng.resident_ready(sc["steps"])
    eng.iterate(sc["steps"])
    eng.set_resident_launches(True)
    eng.iterate(sc["steps"])
    assert eng.resident_stats() == (2, 0, 0) and eng.last_launch_count() == 1
    for _ in range(4):
        ref.iterate(sc["steps"])
    eng.synchronize()
    assert_identical(eng, ref, what="resident, declining, switched off, resident again")
