"""Sharded execution on the GPU: several ranks of a sharded world driven inside one process
(LocalCluster: ghosts + halo pack / all-to-all-v / unpack per external iteration) must give the
beliefs of the single-world CPU oracle bit for bit."""
import os

import numpy as np
import pytest

from magics_amd import World, scenarios as S, sharded

import oracle
from parity import assert_identical

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world_size", [2, 4])
def test_local_cluster_equals_oracle(world_size):
    sc = S.grid_scenario(64, 10, interrobot=True, pitch=2.5, comm_radius=5.0)
    cluster = sharded.LocalCluster(sc, world_size, World)
    assert any(sw.plan.ghosts for sw in cluster.ranks)
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    steps = sc["steps"] + [1, 1, 2, 3, 2]
    for tick in range(2):
        cluster.iterate(steps)
        ref.iterate(steps)
        assert_identical(cluster, ref, what=f"{world_size} ranks, tick {tick}")


def test_local_cluster_gating_and_prior_changes():
    sc = S.grid_scenario(36, 10, interrobot=True, pitch=2.5, comm_radius=5.0)
    cluster = sharded.LocalCluster(sc, 3, World)
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    boundary = sorted({g for sw in cluster.ranks for g in sw.plan.ghosts})
    def script(w):
        w.iterate([3, 3, 3])
        w.set_antenna(boundary[0], False)
        w.set_idle(boundary[1], True)
        w.change_prior(boundary[2], 9, np.array([0.5, 0.25, 1.0, -1.0]))
        w.iterate([3, 3, 3])
        w.set_antenna(boundary[0], True)
        w.set_idle(boundary[1], False)
        w.iterate([3, 3])
    script(cluster)
    script(ref)
    assert_identical(cluster, ref, what="3 ranks, gating + change_prior on boundary robots")


_RANK_STREAMS = []  # one stream per rank, created once and reused by every test of this module


def _own_stream_factory():
    """Every rank of a direct-exchange cluster needs its own stream (see LocalCluster) — and the ranks of a cluster whose
    ghost records travel inside resident launches need streams the device runs SIDE BY SIDE: HIP maps streams onto a few
    hardware queues, two kernels of one queue run one after the other, and a resident launch that waits for a rank behind it in
    its own queue waits until its bound (seen in a soak run that took fresh streams from torch's pool for every script: 3 of 56
    clusters).  So the same four streams — the first ones this process creates — serve every test.  (One process per GPU, the
    real deployment, has no such coupling: every process has queues of its own.)"""
    import torch
    used = []

    def make(params):
        if len(used) == len(_RANK_STREAMS):
            _RANK_STREAMS.append(torch.cuda.Stream())
        st = _RANK_STREAMS[len(used)]
        used.append(st)
        return World(params, stream=st.cuda_stream)
    return make, used


@pytest.mark.parametrize("world_size", [2, 3])
def test_direct_exchange_equals_oracle(world_size):
    """Peer-mapped stores + device-side arrival counters instead of the all-to-all-v."""
    sc = S.grid_scenario(64, 10, interrobot=True, pitch=2.5, comm_radius=5.0)
    make, streams = _own_stream_factory()
    cluster = sharded.LocalCluster(sc, world_size, make, direct=True)
    assert all(sw.direct for sw in cluster.ranks)
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    steps = sc["steps"] + [1, 1, 2, 3, 2]
    n_ext = sum(1 for s in steps if s & 2)
    for tick in range(3):
        cluster.iterate(steps)
        ref.iterate(steps)
        assert_identical(cluster, ref, what=f"direct exchange, {world_size} ranks, tick {tick}")
    for sw in cluster.ranks:
        assert sw.world.halo_direct_status() == 3 * n_ext


def test_direct_exchange_gating_and_prior_changes():
    sc = S.grid_scenario(36, 10, interrobot=True, pitch=2.5, comm_radius=5.0)
    make, streams = _own_stream_factory()
    cluster = sharded.LocalCluster(sc, 3, make, direct=True)
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    boundary = sorted({g for sw in cluster.ranks for g in sw.plan.ghosts})

    def script(w):
        w.iterate([3, 3, 3])
        w.set_antenna(boundary[0], False)
        w.set_idle(boundary[1], True)
        w.change_prior(boundary[2], 9, np.array([0.5, 0.25, 1.0, -1.0]))
        w.iterate([3, 3, 3])
        w.set_antenna(boundary[0], True)
        w.set_idle(boundary[1], False)
        w.iterate([3, 3])
    script(cluster)
    script(ref)
    assert_identical(cluster, ref, what="direct exchange, gating + change_prior on boundary robots")


def test_direct_exchange_reports_a_missing_peer(monkeypatch):
    """A producer that never shows up must end in a reported timeout, not in a hung GPU."""
    monkeypatch.setenv("MGX_HALO_TIMEOUT_MS", "200")
    sc = S.grid_scenario(36, 10, interrobot=True, pitch=2.5, comm_radius=5.0)
    make, streams = _own_stream_factory()
    cluster = sharded.LocalCluster(sc, 2, make, direct=True)
    lonely = cluster.ranks[0]
    lonely.iterate([3])            # rank 1 never runs its side of the exchange
    with pytest.raises(RuntimeError, match="timed out"):
        lonely.world.halo_direct_status()


def test_rccl_transport_loads_and_runs_with_one_rank():
    """The in-library RCCL exchange on the one GPU of the box: the library resolves RCCL, creates a
    one-rank communicator and runs the (peerless) exchange in front of every external phase."""
    from magics_amd import hostlib
    sc = S.grid_scenario(16, 10, interrobot=True, pitch=2.5, comm_radius=5.0)
    w = World(sc["params"])
    S.populate(w, sc)
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    uid = hostlib.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    w.halo_plan([], [])
    w.halo_rccl_connect(uid, 1, 0, [], [0], [0])
    for x in (w, ref):
        x.iterate(sc["steps"])
    assert_identical(w, ref, what="one-rank RCCL communicator")
    w.halo_rccl_disconnect()


def test_sharded_junction_scenario_rasterises_on_every_rank():
    """BASELINE configs[4] in small, sharded: every rank rasterises the crossroads itself
    (mgx_world_set_environment) and the cluster's beliefs equal the single-world oracle's."""
    sc = S.junction_scenario(60, 12, tiles=2, connect_after_ticks=0)  # a sharded world plans its ghosts from sc["ir"]
    cluster = sharded.LocalCluster(sc, 3, World)
    assert any(sw.plan.ghosts for sw in cluster.ranks)
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    for tick in range(2):
        cluster.iterate(sc["steps"])
        ref.iterate(sc["steps"])
        assert_identical(cluster, ref, what=f"junction tiles on 3 ranks, tick {tick}")


@pytest.mark.parametrize("world_size,mode,shape", [(3, "collective", (60, 12, 2)), (2, "direct", (60, 12, 2)), (2, "resident", (60, 12, 2)),
                                                   (2, "resident", (400, 32, 6))])
def test_sharded_junction_robots_meet_after_their_first_tick(world_size, mode, shape):
    """BASELINE configs[4] in the regime it is quoted on, sharded (scenarios.junction_scenario, connect_after_ticks = 1): the robots
    run one driver tick on their own, THEN create_interrobot_factors hooks them up (sc["ir_late"]).  A sharded world plans its
    ghosts and exchange lists for those pairs from the start, every rank ticks at the same time, the ghosts' delivery counts a new
    factor remembers (robot.rs:1549-1585) are their owners' — beliefs of the single-world oracle bit for bit, every one finite,
    over the collective transport, the direct one, and with the ghost records inside one resident launch per rank and tick (also
    at configs[4]'s own horizon: 2 x 200 robots x 32 variables, 75 KB of LDS per workgroup)."""
    n, K, tiles = shape
    sc = S.junction_scenario(n, K, tiles=tiles)
    assert sc["ir_late"] and not sc["ir"] and sc["connect_after_ticks"] == 1
    make, kw = World, {}
    if mode != "collective":
        make, _streams = _own_stream_factory()
        kw = dict(direct=True, resident=mode == "resident")
    cluster = sharded.LocalCluster(sc, world_size, make, **kw)
    assert any(sw.plan.ghosts for sw in cluster.ranks) and not any(sw.late_pending for sw in cluster.ranks)
    if mode == "resident":
        assert cluster.resident
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    tick = S.tick_inputs(sc)
    for t in range(3):
        cluster.tick(steps=sc["steps"], **tick)
        ref.tick(steps=sc["steps"], **tick)
        assert_identical(cluster, ref, what=f"junction, connected after the first tick, {world_size} ranks ({mode}), tick {t}")
        if mode == "resident" and t > 0:
            assert all(sw.world.last_launch_count() == 1 for sw in cluster.ranks) or getattr(cluster, "declined", 0)
    assert all(np.isfinite(x).all() for x in cluster.read_beliefs())


def _circle_driver(w, sc, n, K, **kw):
    from magics_amd.driver import Driver
    return Driver(w, n, K, waypoints=[[tuple(rb["goal"])] for rb in sc["robots"]], radii=[rb["radius"] for rb in sc["robots"]],
                  t0=[rb["t0"] for rb in sc["robots"]], steps=sc["steps"], comms_radius=12.0, target_speed=sc["target_speed"], **kw)


@pytest.mark.parametrize("world_size,direct", [(2, False), (3, False), (2, True), (3, True), (2, "resident"), (3, "resident")])
def test_sharded_world_follows_its_topology(world_size, direct):
    """A whole mission on a sharded world that follows its topology: robots cross a circle, connect and
    disconnect across rank boundaries, arrive and despawn.  Every rank replays the connection
    bookkeeping on all positions; exchange lists follow the connections.  Topology events, robot
    numbers, trajectories and beliefs equal the single-world oracle's, tick by tick.
    direct: the exchange lives in the engines (peer-mapped stores into one record slot per ghost robot, wired ONCE and re-aimed
    whenever the lists change — mgx_halo_direct_setup_slots / _connect_slots): no host-driven all-to-all in any tick.
    "resident": on top of that the ghosts' exchange records travel INSIDE one resident launch per schedule and rank, the push
    tables re-aimed with the lists (mgx_halo_resident_connect_peers / _aim)."""
    n, K = 9, 10
    sc = S.circle_scenario(n, K, circle_radius=12.0, n_internal=10, n_external=10)
    sc["ir"] = []
    owner = np.arange(n) % world_size  # interleaved ownership: every neighbour pair crosses a rank boundary sooner or later
    if direct:
        make, _streams = _own_stream_factory()
        cluster = sharded.LocalCluster(sc, world_size, make, owner=owner, dynamic=True, direct=True, resident=direct == "resident")
        assert all(sw.direct and sw.transport == ("direct+resident" if direct == "resident" else "direct") for sw in cluster.ranks)
        assert cluster.resident == (direct == "resident")
    else:
        cluster = sharded.LocalCluster(sc, world_size, World, owner=owner, dynamic=True)
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    dc, dr = _circle_driver(cluster, sc, n, K), _circle_driver(ref, sc, n, K)
    events, plans = [], set()
    for tick in range(400):
        if not (dc.finished_at < 0).any():
            break
        ec, er = dc.tick(), dr.tick()
        assert ec == er, (tick, ec, er)
        events.append(ec)
        assert dc.next_number == dr.next_number
        assert np.array_equal(dc.translation, dr.translation) and np.array_equal(dc.finished_at, dr.finished_at), tick
        plans.add(tuple(tuple(sw.send_counts) for sw in cluster.ranks))
        if tick % 10 == 9:
            assert_identical(cluster, ref, what=f"{world_size} ranks following their topology, tick {tick + 1}")
            for r in range(n):
                if dc.alive[r]:
                    assert list(cluster.connections(r)) == list(ref.connections(r))
    assert (dc.finished_at >= 0).all() and np.array_equal(dc.finished_at, dr.finished_at)
    assert sum(c for c, _ in events) > n and sum(d for _, d in events) > 0   # connections came and went
    assert len(plans) > 3                                                      # and the exchange lists with them
    assert dc.summary()["messages"] == dr.summary()["messages"]               # MessageCount of every graph, from its owner's rank
    if direct:
        for sw in cluster.ranks:
            assert sw.world.halo_direct_status() > (20 if direct == "resident" else 100)  # exchanges inside the engine, none timed out
    if direct == "resident":
        st = [tuple(int(x) for x in sw.world.resident_stats()) for sw in cluster.ranks]
        print("resident launches / declined / back-off left per rank:", st, "schedules declined by the cluster:", getattr(cluster, "declined", 0))
        assert all(x[0] > 50 for x in st), st  # most ticks ran as ONE launch per rank (robots that are alone on their rank vote no)


@pytest.mark.parametrize("world_size,direct", [(2, False), (3, False), (3, True), (2, "resident")])
def test_robots_migrate_between_ranks(world_size, direct):
    """Re-balancing a sharded world that follows its topology: every ten ticks the robots are dealt out again in strips of
    where they ARE (mgx_shard_partition on the current positions), and the ones whose strip changed move to their new rank —
    graph state, counters and the inter-robot factors attached to their variables as one record (mgx_robot_export /
    _import / _release), the in-engine transports wired again.  Robots crossing a circle swap sides, so every robot moves
    at least once; events, trajectories, beliefs and MessageCounts stay the single-world oracle's, tick by tick."""
    n, K = 9, 10
    sc = S.circle_scenario(n, K, circle_radius=12.0, n_internal=10, n_external=10)
    sc["ir"] = []
    if direct:
        make, _streams = _own_stream_factory()
        cluster = sharded.LocalCluster(sc, world_size, make, dynamic=True, direct=True, resident=direct == "resident")
    else:
        cluster = sharded.LocalCluster(sc, world_size, World, dynamic=True)
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    dc, dr = _circle_driver(cluster, sc, n, K), _circle_driver(ref, sc, n, K)
    moved, movers = 0, set()
    for tick in range(400):
        if not (dc.finished_at < 0).any():
            break
        assert dc.tick() == dr.tick(), tick
        assert np.array_equal(dc.translation, dr.translation) and np.array_equal(dc.finished_at, dr.finished_at), tick
        if tick % 10 == 9:
            assert_identical(cluster, ref, what=f"{world_size} ranks, robots migrating, tick {tick + 1}")
            old = cluster.ranks[0].plan.owner
            live = np.nonzero(dc.alive)[0]
            new = old.copy()
            if len(live) >= world_size:
                new[live] = sharded.partition_strips(cluster.read_variable_means(0)[live, :2], world_size)
            movers |= set(int(g) for g in np.nonzero(new != old)[0])
            moved += cluster.migrate(new)
            for sw in cluster.ranks:
                assert np.array_equal(sw.plan.owner, new) and sw.plan.local == [int(g) for g in np.nonzero(new == sw.plan.rank)[0]]
            assert_identical(cluster, ref, what=f"{world_size} ranks, right after the migration of tick {tick + 1}")
            for r in live:
                assert list(cluster.connections(r)) == list(ref.connections(r))
    assert (dc.finished_at >= 0).all() and np.array_equal(dc.finished_at, dr.finished_at)
    print(f"{moved} migrations of {len(movers)} robots over {tick} ticks")
    assert moved >= n // 2 and len(movers) >= n // 2
    assert dc.summary()["messages"] == dr.summary()["messages"]  # the counters travelled with the graphs
    if direct:
        for sw in cluster.ranks:
            sw.world.halo_direct_status()  # raises if an exchange timed out
    if direct == "resident":
        st = [tuple(int(x) for x in sw.world.resident_stats()) for sw in cluster.ranks]
        print("resident launches / declined / back-off left per rank:", st)


def test_migration_carries_frozen_inboxes_flags_and_fresh_factors():
    """What else lives on the owner's rank only: the inboxes factor kinds froze with when they were switched off
    (mgx_set_enabled), the records inter-robot factors kept, the first-update marks of kinds that came back — and factors
    created by the topology pass of the very tick before.  Robots change ranks in every one of these states, with radios
    failing at random; beliefs, connections and MessageCounts stay the single-world oracle's."""
    n, K, world_size = 10, 10, 3
    sc = S.circle_scenario(n, K, circle_radius=10.0, n_internal=10, n_external=10)   # comes with its initial connections
    assert sc["ir"]
    cluster = sharded.LocalCluster(sc, world_size, World, dynamic=True)
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    rng = np.random.default_rng(5)
    draws = rng.random((80, n)) > 0.2

    def failures(tick, k):
        return draws[tick, :k]
    dc, dr = (_circle_driver(w, sc, n, K, failure_draws=failures, despawn_when_finished=False) for w in (cluster, ref))
    dc.next_number = dr.next_number = 1 + (K - 1) * len(sc["ir"])
    full = sc["params"]["enable_mask"] if isinstance(sc["params"], dict) and "enable_mask" in sc["params"] else 7
    masks = {8: full & ~2, 16: full, 24: full & ~1, 30: full & ~4, 36: full, 44: full & ~2, 50: full}  # tick -> kinds enabled from then on
    moved = 0
    for tick in range(64):
        if tick in masks:
            for w in (cluster, ref):
                w.set_enabled(masks[tick])
        assert dc.tick() == dr.tick(), tick
        if tick % 4 == 3:  # every fourth tick a third of the robots moves one rank on: in every state the script goes through
            new = cluster.ranks[0].plan.owner.copy()
            sel = np.arange(n) % 3 == (tick // 4) % 3
            new[sel] = (new[sel] + 1) % world_size
            moved += cluster.migrate(new)
            assert_identical(cluster, ref, what=f"migration in state {tick}")
    assert moved > 40
    assert_identical(cluster, ref, what="3 ranks, kinds switched, radios failing, robots migrating")
    for r in range(n):
        assert list(cluster.connections(r)) == list(ref.connections(r))
    assert dc.summary()["messages"] == dr.summary()["messages"]


def test_sharded_topology_with_comms_failures_and_initial_connections():
    n, K, world_size = 10, 10, 2
    sc = S.circle_scenario(n, K, circle_radius=10.0, n_internal=10, n_external=10)   # comes with its initial connections
    assert sc["ir"]
    cluster = sharded.LocalCluster(sc, world_size, World, dynamic=True)
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    rng = np.random.default_rng(11)
    draws = rng.random((80, n)) > 0.25

    def failures(tick, k):
        return draws[tick, :k]
    first = 1 + (K - 1) * len(sc["ir"])
    dc, dr = (_circle_driver(w, sc, n, K, failure_draws=failures, despawn_when_finished=False) for w in (cluster, ref))
    dc.next_number = dr.next_number = first
    for tick in range(80):
        assert dc.tick() == dr.tick(), tick
    assert_identical(cluster, ref, what="2 ranks, comms failures, initial connections")
    assert dc.summary()["messages"] == dr.summary()["messages"]


def test_sharded_world_switches_factor_kinds():
    """change_factor_enabled on a sharded world (all ranks together, after an exchange): inter-robot factors
    go off and come back across rank boundaries, with schedules that run external iterations before the
    owners' next internal sweep, next to obstacle factors switched the same way."""
    sc = S.grid_scenario(36, 10, interrobot=True, pitch=2.2, comm_radius=5.0)
    cluster = sharded.LocalCluster(sc, 3, World)
    assert any(sw.plan.ghosts for sw in cluster.ranks)
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    script = [(7, [3] * 4), (5, [3] * 3), (5, [1, 3, 3]), (7, [2, 2, 3, 3]), (3, [3] * 3), (7, [2, 3, 3]), (7, [3] * 3)]
    for t, (mask, steps) in enumerate(script):
        for w in (cluster, ref):
            w.set_enabled(mask)
            if t == 3:
                w.change_prior(7, 9, np.array([0.3, -0.2, 1.0, 0.5]))
            w.iterate(steps)
        assert_identical(cluster, ref, what=f"sharded kind switching, step {t} (mask {mask})")


def test_rank_that_owns_no_robot_yet():
    """A rank of a world that follows its topology may hold ghosts only (formations spawn where they spawn): every
    per-tick call still works there — zero-sized reads and uploads included — and the run equals the oracle's."""
    n, K, world_size = 6, 10, 3
    sc = S.circle_scenario(n, K, circle_radius=9.0, n_internal=10, n_external=10)
    sc["ir"] = []
    owner = np.arange(n) % 2   # rank 2 owns nothing
    cluster = sharded.LocalCluster(sc, world_size, World, owner=owner, dynamic=True)
    assert len(cluster.ranks[2].plan.local) == 0 and all(len(cluster.ranks[q].plan.local) == 3 for q in (0, 1))
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    dc, dr = _circle_driver(cluster, sc, n, K), _circle_driver(ref, sc, n, K)
    for tick in range(60):
        assert dc.tick() == dr.tick(), tick
        assert np.array_equal(dc.translation, dr.translation), tick
    assert np.array_equal(cluster.read_variable_means(0), ref.read_variable_means(0))
    assert_identical(cluster, ref, what="a rank with ghosts only")


def test_message_counts_on_a_sharded_world():
    """MessageCount (factorgraph/mod.rs:29-137) of every robot, kept by the rank that owns it: the rank's mirror holds every
    connection its robots take part in (those towards other ranks as bookkeeping only) and is told the prior changes other
    ranks apply to its ghosts — same counts as the single-world oracle through gating, prior changes and driver ticks."""
    n = 36
    sc = S.grid_scenario(n, 10, interrobot=True, pitch=2.5, comm_radius=5.0)
    cluster = sharded.LocalCluster(sc, 3, World)
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    boundary = sorted({g for sw in cluster.ranks for g in sw.plan.ghosts})
    tick = S.tick_inputs(sc)

    def same(what):
        for r in range(n):
            assert cluster.message_counts(r) == ref.message_counts(r), (what, r, cluster.message_counts(r), ref.message_counts(r))

    def both(fn):
        fn(cluster)
        fn(ref)
    same("after construction")
    both(lambda w: w.iterate([3, 3, 1, 2, 3]))
    same("first iterations")
    both(lambda w: (w.set_antenna(boundary[0], False), w.set_idle(boundary[1], True),
                    w.change_prior(boundary[2], 9, np.array([0.5, 0.25, 1.0, -1.0])), w.iterate([3, 3, 3])))
    same("gating + prior change on boundary robots")
    both(lambda w: (w.set_antenna(boundary[0], True), w.set_idle(boundary[1], False)))
    for t in range(2):
        both(lambda w: w.tick(steps=sc["steps"], **tick))
        same(f"driver tick {t}")
    assert_identical(cluster, ref, what="sharded world with message counts")


# ---- resident schedule launches on sharded worlds: ghost records travel INSIDE the launches -----------------------------
@pytest.mark.parametrize("world_size,n,K", [(2, 64, 10), (3, 96, 16), (4, 400, 16)])
def test_resident_launches_on_a_sharded_world(world_size, n, K):
    """Every rank runs its schedule as ONE launch (mgx_last_launch_count == 1): boundary robots store their snapshot records and
    progress words straight into the other ranks' ghost areas at the end of every segment, and the workgroups there poll those
    words like a local neighbour's.  Several ticks (the parities and segment counts carry over), schedules that open with an
    internal and with an external iteration, and a schedule too short to be resident in between: beliefs of the single-world
    oracle, bit for bit."""
    sc = S.grid_scenario(n, K, interrobot=True, pitch=2.5, comm_radius=5.0)
    make, streams = _own_stream_factory()
    cluster = sharded.LocalCluster(sc, world_size, make, direct=True, resident=True)
    assert cluster.resident and all(sw.resident for sw in cluster.ranks)
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    scripts = [sc["steps"], sc["steps"], [2, 3, 3, 1, 3], [3], sc["steps"] + [1, 1, 2, 3, 2], [2, 2, 3]]
    for tick, steps in enumerate(scripts):
        declined = getattr(cluster, "declined", 0)
        backing_off = cluster.ranks[0].world.resident_stats()[2] > 0
        cluster.iterate(steps)
        ref.iterate(steps)
        for sw in cluster.ranks:
            sw.synchronize()  # raises if a wait inside a launch gave up
            want = 1 if len(sharded.segments(steps)) >= 2 else len(sharded.segments(steps))
            # (the ranks' launches run side by side only if the device takes them so — one process, streams that may share a
            # hardware queue: if not, they find out, agree on "no" and the schedule runs launch by launch: counted, not failed)
            if getattr(cluster, "declined", 0) == declined and not backing_off:
                assert sw.world.last_launch_count() == want, (tick, sw.plan.rank, sw.world.last_launch_count())
        assert_identical(cluster, ref, what=f"resident launches, {world_size} ranks, tick {tick}")
    print(f"{world_size} ranks: {getattr(cluster, 'declined', 0)} of {len(scripts)} schedules declined")


def test_resident_launches_on_eight_ranks_in_one_process(tmp_path):
    """BASELINE configs[3]'s rank count with every schedule as ONE launch per rank: 8 x 90 robots x 16, twelve neighbours each
    (the 728 workgroups of the eight launches are on the device together — three per CU at this density; 8 x 1000 are not:
    tests/test_gpu_fullsize.py runs that on the launch-per-segment transports, and 8 x 120 here are declined by the ranks'
    agreement and run launch by launch).  A process of its own with twelve hardware queues: with the default four, ranks share
    queues and the launches are declined as well."""
    import json
    import subprocess
    import sys
    out = str(tmp_path / "out.json")
    env = dict(os.environ, GPU_MAX_HW_QUEUES="12")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "resident_cluster_worker.py"), "8", "720", "16", out],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]
    d = json.load(open(out))
    assert d["resident"] and all(g > 0 for g in d["ghosts"])
    assert d["one_launch"] >= 1, d  # bit-identical either way (checked in the worker); at least one schedule side by side
    print(d)


def test_resident_sharded_rank_declines():
    """The ranks' launches of one schedule go ahead together or not at all: one rank that says no (here: told to; in the field: its
    workgroups do not all get onto the device, or its launch has not started when the others have waited long enough) says it on
    the word all ranks look at before they write anything.  Every world is then as it was, the schedule runs launch by launch,
    the following ones too (the back-off, the same on every rank) — and when that has run out they are resident launches again.
    Beliefs of the single-world oracle throughout."""
    sc = S.grid_scenario(48, 10, interrobot=True, pitch=2.5, comm_radius=5.0)
    make, streams = _own_stream_factory()
    cluster = sharded.LocalCluster(sc, 3, make, direct=True, resident=True)
    assert cluster.resident
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)

    def both(steps, what):
        cluster.iterate(steps)
        ref.iterate(steps)
        for sw in cluster.ranks:
            sw.synchronize()
        assert_identical(cluster, ref, what=what)
    both(sc["steps"], "before")
    before = [sw.world.resident_stats() for sw in cluster.ranks]
    cluster.ranks[1].world.set_resident_launches("decline")
    both([2, 3, 3, 1, 3], "the declined schedule (opens with an external iteration: its exchange ran in front of the launch)")
    after = [sw.world.resident_stats() for sw in cluster.ranks]
    assert cluster.declined >= 1
    assert all(a[1] == b[1] + 1 for a, b in zip(after, before)), (before, after)  # every rank counts the same declined launch
    assert len({a[2] for a in after}) == 1 and after[0][2] > 0                      # ... and backs off alike
    cluster.ranks[1].world.set_resident_launches(True)
    n = 0
    while cluster.ranks[0].world.resident_stats()[2] > 0:
        both(sc["steps"], f"backing off, schedule {n}")
        assert all(sw.world.resident_stats()[0] == a[0] for sw, a in zip(cluster.ranks, after))  # no resident launch meanwhile
        n += 1
        assert n < 40
    declined = cluster.declined
    both(sc["steps"], "resident again")
    if cluster.declined == declined:
        assert all(sw.world.resident_stats()[0] == a[0] + 1 and sw.world.last_launch_count() == 1 for sw, a in zip(cluster.ranks, after))


def test_resident_sharded_gating_and_prior_changes():
    sc = S.grid_scenario(36, 10, interrobot=True, pitch=2.5, comm_radius=5.0)
    make, streams = _own_stream_factory()
    cluster = sharded.LocalCluster(sc, 3, make, direct=True, resident=True)
    assert cluster.resident
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    boundary = sorted({g for sw in cluster.ranks for g in sw.plan.ghosts})

    def script(w):
        w.iterate([3, 3, 3])
        w.set_antenna(boundary[0], False)
        w.set_idle(boundary[1], True)
        w.change_prior(boundary[2], 9, np.array([0.5, 0.25, 1.0, -1.0]))
        w.iterate([3, 3, 3])
        w.set_antenna(boundary[0], True)
        w.set_idle(boundary[1], False)
        w.iterate([3, 3])
        w.iterate([2, 3, 3])
    script(cluster)
    script(ref)
    for sw in cluster.ranks:
        sw.synchronize()
    assert_identical(cluster, ref, what="resident sharded launches, gating + change_prior on boundary robots")


def test_resident_sharded_survives_a_missing_rank():
    """A rank that never launches its side: the launch of the other waits for it on the ranks' agreement word — before it has
    written anything — gives up after the bound (MGX_RESIDENT_CENSUS_SHARDED_US) and returns: DECLINED, the world as it was."""
    from magics_amd import hostlib
    sc = S.grid_scenario(36, 10, interrobot=True, pitch=2.5, comm_radius=5.0)
    make, streams = _own_stream_factory()
    cluster = sharded.LocalCluster(sc, 2, make, direct=True, resident=True)
    assert cluster.resident
    lonely = cluster.ranks[0]
    before = lonely.read_beliefs()
    lonely.world.iterate([1, 3, 3])  # rank 1 never runs (a schedule that opens with an internal iteration: no exchange in front)
    assert lonely.world.resident_outcome() == hostlib.RESIDENT_DECLINED
    assert lonely.world.resident_outcome() == hostlib.RESIDENT_NONE
    lonely.synchronize()
    after = lonely.read_beliefs()
    assert all(np.array_equal(a, b) for a, b in zip(before[1:], after[1:]))
    assert lonely.world.resident_stats()[1] == 1


def test_resident_sharded_without_agreement_reports_a_missing_rank(monkeypatch):
    """Wired without a coordinator (mgx_halo_resident_connect: coordinator_area NULL) there is no agreement: a rank whose peer
    never launches gives up after the bound on its waits and the world says so."""
    monkeypatch.setenv("MGX_RESIDENT_TIMEOUT_MS", "300")
    from magics_amd import hostlib
    sc = S.grid_scenario(36, 10, interrobot=True, pitch=2.5, comm_radius=5.0)
    make, streams = _own_stream_factory()
    cluster = sharded.LocalCluster(sc, 2, make, direct=True, resident=True, agree=False)
    assert cluster.resident
    lonely = cluster.ranks[0]
    lonely.world.iterate([1, 3, 3])  # rank 1 never runs (a schedule that opens with an internal iteration: no exchange in front)
    with pytest.raises(hostlib.MgxError):
        lonely.synchronize()


def test_resident_sharded_switching_of_inter_robot_factors():
    """Inter-robot factors switched off and on again on a sharded world whose ghost records travel inside resident launches: what
    the factors freeze with and thaw against are the ghosts' CURRENT records (the engine exchanges them when the kind is switched:
    the ghosts' plain copies are as old as the last exchange kernel), and the schedules that thaw run launch by launch."""
    sc = S.grid_scenario(64, 10, interrobot=True, pitch=2.5, comm_radius=5.0)
    make, streams = _own_stream_factory()
    cluster = sharded.LocalCluster(sc, 3, make, direct=True, resident=True)
    assert cluster.resident
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    on = sc["params"]["enable_mask"]

    def script(w):
        w.iterate(sc["steps"])
        w.iterate([1, 3, 1])          # ends with an internal iteration: the last exchange is one sweep old
        w.set_enabled(on & ~S.EN_IR)
        w.iterate([3, 3, 1])
        w.set_enabled(on)
        w.iterate([3, 3])
        w.iterate(sc["steps"])
    script(cluster)
    script(ref)
    for sw in cluster.ranks:
        sw.synchronize()
    assert_identical(cluster, ref, what="resident sharded launches, inter-robot factors switched off and on")
