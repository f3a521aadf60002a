"""Multi-rank host logic on CPU: the shard plan is consistent across ranks and the per-external-
iteration all-to-all-v (torch.distributed, gloo, world_size 2) delivers every boundary robot's
snapshot record to the rank that evaluates its inter-robot factors, before each external phase."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

from magics_amd import scenarios as S
from magics_amd import sharded

WORDS_PER_VAR = 25


class FakeWorld:
    """Stands in for magics_amd.World in the CPU tests of the sharding driver: it implements only
    topology + halo_pack / halo_unpack and checks, at every external phase, that each ghost carries
    the snapshot of the right robot at the right version.  No GBP arithmetic."""

    def __init__(self, params):
        self.robots, self.conns = [], []
        self.version = 0          # number of internal sweeps this rank has run
        self.checked = 0

    def set_sdf(self, *a):
        pass

    def add_robot(self, mean0, prior_diag, dt, radius, path=None, order_key=None, ghost=False):
        self.K = len(mean0)
        self.robots.append(dict(key=order_key, ghost=ghost, version=-1 if ghost else 0))
        return len(self.robots) - 1

    def ir_connect(self, a, b, n0):
        if self.robots[b]["ghost"]:          # owned here, evaluated on the target's rank: bookkeeping only (message counters)
            assert not self.robots[a]["ghost"], "a connection between two ghosts is nobody's business here"
            self.bookkeeping = getattr(self, "bookkeeping", 0) + 1
            return
        self.conns.append((a, b))

    @staticmethod
    def halo_words(K):
        return WORDS_PER_VAR * K

    def halo_plan(self, send, recv):
        self.send, self.recv = list(send), list(recv)
        assert all(not self.robots[r]["ghost"] for r in self.send)
        assert all(self.robots[r]["ghost"] for r in self.recv)

    def _view(self, ptr, n):
        return np.ctypeslib.as_array((C.c_double * n).from_address(ptr))

    def halo_pack(self, ptr):
        w = self.halo_words(self.K)
        buf = self._view(ptr, max(1, len(self.send) * w))
        for j, r in enumerate(self.send):
            buf[j * w:(j + 1) * w] = self.robots[r]["key"]
            buf[j * w + 1] = self.robots[r]["version"]

    def halo_unpack(self, ptr):
        w = self.halo_words(self.K)
        buf = self._view(ptr, max(1, len(self.recv) * w))
        for j, r in enumerate(self.recv):
            assert buf[j * w] == self.robots[r]["key"], "record landed in the wrong ghost"
            assert (buf[j * w + 2:(j + 1) * w] == self.robots[r]["key"]).all()
            self.robots[r]["version"] = int(buf[j * w + 1])

    def sweep(self, ext, internal, n_int, hints=0):
        if ext:
            for a, b in self.conns:  # every factor evaluated here sees its owner's latest snapshot
                assert self.robots[a]["version"] == self.version, (self.robots[a], self.version)
                self.checked += 1
        for _ in range(n_int):
            self.version += 1
            for r in self.robots:
                if not r["ghost"]:
                    r["version"] = self.version

    def synchronize(self):
        pass


def _scenario():
    return S.grid_scenario(36, 10, interrobot=True, obstacles=False, pitch=3.0, comm_radius=5.0)


def test_segments_group_external_then_internal():
    assert sharded.segments([3, 3, 3]) == [(False, 1), (True, 1), (True, 1), (True, 0)]
    assert sharded.segments([1, 1, 3, 2, 1]) == [(False, 3), (True, 0), (True, 1)]
    assert sharded.segments([2, 2]) == [(True, 0), (True, 0)]
    assert sharded.segments([]) == []


@pytest.mark.parametrize("world_size", [2, 3, 4])
def test_shard_plans_agree_across_ranks(world_size):
    sc = _scenario()
    plans = [sharded.ShardPlan(sc, r, world_size) for r in range(world_size)]
    n = len(sc["robots"])
    assert sorted(sum((p.local for p in plans), [])) == list(range(n))
    sizes = [len(p.local) for p in plans]
    assert max(sizes) - min(sizes) <= 1
    for a in plans:
        for b in plans:
            if a.rank != b.rank:
                assert a.send_lists[b.rank] == b.recv_lists[a.rank]
    # every inter-robot factor is evaluated exactly once, on its target's rank
    assert sorted(sum((p.connections for p in plans), [])) == sorted(sc["ir"])
    # strips: only neighbouring ranks talk
    for p in plans:
        for q in range(world_size):
            if abs(q - p.rank) > 1:
                assert not p.send_lists[q] and not p.recv_lists[q]
    assert any(p.ghosts for p in plans)


def _worker(rank, world_size, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        sc = _scenario()
        sw = sharded.ShardedWorld(sc, rank, world_size, FakeWorld, comm=sharded.TorchDistComm(),
                                  tensor_factory=lambda n: torch.zeros(n, dtype=torch.float64))
        # the in-engine transports need the real engine: on this (fake, CPU) world every rank fails to wire them, the
        # ranks agree on that, and all of them fall back to the host-driven collective
        assert sharded.connect(sw, sw.comm) == "collective" and sw.transport == "collective"
        steps = [3, 3, 1, 1, 3, 2, 3]
        sw.iterate(steps)
        sw.iterate(steps)
        n_ext = sum(1 for s in steps if s & 2) * 2
        assert sw.world.checked == n_ext * len(sw.plan.connections) and sw.world.checked > 0
        q.put((rank, "ok", sw.world.checked, len(sw.plan.ghosts)))
    except Exception as e:  # pragma: no cover
        q.put((rank, f"error: {e!r}", 0, 0))
    finally:
        dist.destroy_process_group()


def test_all_to_all_exchange_gloo_world_size_2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert [r[1] for r in res] == ["ok", "ok"], res
    assert all(r[3] > 0 for r in res)  # both ranks really had ghosts to fill


class DirectFakeWorld(FakeWorld):
    """FakeWorld + a host model of the direct halo exchange (include/mgx.h): every rank's receive
    area is a numpy array in a shared registry keyed by a fake address; `connect` records where this
    rank's segments go; `iterate` pushes through those addresses and unpacks from its own area."""
    registry = {}
    next_addr = [1 << 20]

    def halo_direct_setup(self, n_sources):
        words = self.halo_words(self.K)
        self.area = np.full((2, max(1, len(self.recv)) * words), -1.0)
        self.flags = np.zeros(max(1, n_sources), dtype=np.int64)
        a_recv, a_flag = self.next_addr[0], self.next_addr[0] + (1 << 16)
        self.next_addr[0] += 1 << 17
        self.registry[a_recv] = ("area", self)
        for j in range(max(1, n_sources)):
            self.registry[a_flag + 8 * j] = ("flag", self, j)
        self.n_sources, self.seq = n_sources, 0
        return a_recv, a_flag

    def halo_direct_connect(self, first, base, nrec, off, slot):
        assert len(base) == self.n_sources, "producers and consumers of a rank must be the same peers"
        assert first[0] == 0 and first[-1] == len(self.send)
        self.peers = []
        for p in range(len(base)):
            kind, peer = self.registry[base[p]]
            assert kind == "area" and nrec[p] == len(peer.recv)
            fk = self.registry[slot[p]]
            assert fk[0] == "flag" and fk[1] is peer
            self.peers.append((peer, first[p], first[p + 1], off[p], fk[2]))

    def direct_push(self):
        self.seq += 1
        w = self.halo_words(self.K)
        for peer, lo, hi, off, fslot in self.peers:
            for k, r in enumerate(self.send[lo:hi]):
                rec = np.full(w, float(self.robots[r]["key"]))
                rec[1] = self.robots[r]["version"]
                peer.area[self.seq & 1, (off + k) * w:(off + k + 1) * w] = rec
            peer.flags[fslot] = self.seq

    def direct_wait_unpack(self):
        assert (self.flags[:self.n_sources] >= self.seq).all(), "a producer has not pushed this exchange"
        w = self.halo_words(self.K)
        for j, r in enumerate(self.recv):
            rec = self.area[self.seq & 1, j * w:(j + 1) * w]
            assert rec[0] == self.robots[r]["key"], "record landed in the wrong ghost"
            self.robots[r]["version"] = int(rec[1])


@pytest.mark.parametrize("world_size", [2, 3, 4])
def test_direct_exchange_wiring(world_size):
    """ShardedWorld.direct_setup / direct_connect: segments, offsets and counter slots agree across
    ranks, so that every ghost receives the snapshot of the right robot at the right version."""
    DirectFakeWorld.registry.clear()
    sc = _scenario()
    ranks = [sharded.ShardedWorld(sc, r, world_size, DirectFakeWorld, tensor_factory=lambda n: None) for r in range(world_size)]
    infos = {sw.plan.rank: sw.direct_setup(export_ipc=False) for sw in ranks}
    for sw in ranks:
        sw.direct_connect(infos)
    for it in range(4):
        for sw in ranks:
            sw.world.direct_push()
        for sw in ranks:
            sw.world.direct_wait_unpack()
            sw.world.sweep(3, 3, 1)
    assert sum(sw.world.checked for sw in ranks) > 0
    assert any(sw.plan.ghosts for sw in ranks)


def _plan_reference(sc, owner, rank, world_size):
    """the plan as the round-1 Python launcher derived it (kept here as the checker of mgx_shard_plan_*)"""
    local = [r for r in range(len(sc["robots"])) if owner[r] == rank]
    ghosts, send, conns = set(), [set() for _ in range(world_size)], []
    for a, b, n0 in sc["ir"]:
        if owner[b] == rank:
            conns.append((a, b, n0))
            if owner[a] != rank:
                ghosts.add(a)
        elif owner[a] == rank:
            send[owner[b]].add(a)
    ghosts = sorted(ghosts)
    return local, ghosts, conns, [sorted(s) for s in send], [[g for g in ghosts if owner[g] == p] for p in range(world_size)]


@pytest.mark.parametrize("world_size", [1, 2, 3, 8])
def test_shard_plan_of_the_c_abi_equals_the_reference_derivation(world_size):
    sc = S.grid_scenario(400, 10, interrobot=True, pitch=2.5, comm_radius=5.0)
    pos = np.asarray(sc["positions"])
    owner = sharded.partition_strips(pos, world_size)
    # equal-count strips in (y, x) order
    order = np.lexsort((pos[:, 0], pos[:, 1]))
    want = np.empty(len(pos), dtype=np.int64)
    for k in range(world_size):
        want[order[k * len(pos) // world_size:(k + 1) * len(pos) // world_size]] = k
    assert np.array_equal(owner, want)
    for rank in range(world_size):
        plan = sharded.ShardPlan(sc, rank, world_size)
        local, ghosts, conns, send, recv = _plan_reference(sc, owner, rank, world_size)
        assert plan.local == local and plan.ghosts == ghosts and plan.connections == conns
        assert plan.send_lists == send and plan.recv_lists == recv


def test_shard_plan_rejects_bad_input():
    from magics_amd import hostlib
    with pytest.raises(hostlib.MgxError):
        hostlib.shard_plan([0, 1, 5], [0], [1], 0, 2)       # owner rank out of range
    with pytest.raises(hostlib.MgxError):
        hostlib.shard_plan([0, 1], [0], [7], 0, 2)          # connection names a robot that does not exist
    with pytest.raises(hostlib.MgxError):
        hostlib.shard_plan([0, 1], [0], [1], 2, 2)          # rank out of range


def test_connections_made_after_the_first_tick_are_planned_from_the_start():
    """sc["ir_late"] (scenarios.junction_scenario, connect_after_ticks = 1): the ghosts and the exchange lists of a sharded
    world are those of the connections as they END UP, the factors come with ShardedWorld.connect_late()"""
    sc = _scenario()
    late = dict(sc, ir=[], ir_late=sc["ir"], connect_after_ticks=1)
    for r in range(3):
        a, b = sharded.ShardPlan(sc, r, 3), sharded.ShardPlan(late, r, 3)
        assert (a.ghosts, a.send_lists, a.recv_lists, a.connections) == (b.ghosts, b.send_lists, b.recv_lists, b.connections)
        assert a.ghosts
    ranks = [sharded.ShardedWorld(late, r, 3, FakeWorld, tensor_factory=lambda n: None) for r in range(3)]
    assert all(sw.late_pending and not sw.world.conns for sw in ranks)  # (in-process ranks: the cluster ticks them in lockstep first)
    for sw in ranks:
        sw.connect_late(ticked=True)
    want = [sharded.ShardedWorld(sc, r, 3, FakeWorld, tensor_factory=lambda n: None) for r in range(3)]
    assert all(not sw.late_pending and sw.world.conns == w0.world.conns and sw.world.conns for sw, w0 in zip(ranks, want))
