"""Sharded execution on the GPU: several ranks of a sharded world driven inside one process
(LocalCluster: ghosts + halo pack / all-to-all-v / unpack per external iteration) must give the
beliefs of the single-world CPU oracle bit for bit."""
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


def _own_stream_factory():
    """Every rank of a direct-exchange cluster needs its own stream (see LocalCluster)."""
    import torch
    streams = []

    def make(params):
        st = torch.cuda.Stream()
        streams.append(st)
        return World(params, stream=st.cuda_stream)
    return make, streams


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
    sc = S.junction_scenario(60, 12, tiles=2)
    cluster = sharded.LocalCluster(sc, 3, World)
    assert any(sw.plan.ghosts for sw in cluster.ranks)
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    for tick in range(2):
        cluster.iterate(sc["steps"])
        ref.iterate(sc["steps"])
        assert_identical(cluster, ref, what=f"junction tiles on 3 ranks, tick {tick}")
