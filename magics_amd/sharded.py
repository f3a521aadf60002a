"""Robots sharded over the GPUs of one node, one process per GPU.

Robots are independent during internal sweeps; they couple only through inter-robot factors,
once per external iteration (robot.rs:1803-1859).  Every inter-robot factor F_AB (owned by A,
attached to B's variable) is evaluated on the rank that owns its only consumer B
(factorgraph.rs:745-754 discards its message to A), which needs nothing from A but the
belief A's variable sends to its own factors — A's *snapshot record*.  So the whole exchange is
ONE all-to-all-v of snapshot records of boundary robots per external iteration, placed right
before the launch that runs the external phase; no other collective exists on the path.

``ShardedWorld`` builds the rank-local world (local robots + ghost copies of remote neighbours),
the send / receive lists and drives ``mgx_sweep`` segments with the exchange in between.  The
communicator is injected: ``TorchDistComm`` (RCCL via torch.distributed "nccl"; "gloo" in the CPU
tests) or a ``LocalCluster`` that runs several ranks inside one process (single-GPU tests).

``connect_direct`` replaces the collective by peer-mapped stores (include/mgx.h, "direct halo
exchange"): producers write their boundary records straight into the consumers' receive areas
(hipIpc-mapped across processes) and the engine waits for them on the device, so an external
iteration costs no host work beyond the launches and ``iterate`` is one C call per tick.
"""
import numpy as np

from . import hostlib

PH_FACTOR, PH_VARIABLE = 1, 2


def partition_strips(positions, world_size):
    """Owner rank of every robot: contiguous strips in (y, x) order with equal robot counts —
    spatial blocks keep cross-rank neighbour pairs few.  mgx_shard_partition."""
    return hostlib.shard_partition(np.asarray(positions)[:, :2], world_size).astype(np.int64)


def segments(steps):
    """Schedule steps -> launches [(external?, n_internal)]: phases in reference order (internal
    then external per step, robot.rs:1787-1860) grouped as an optional external phase followed by
    the internal iterations that follow it (the same grouping as mgx_iterate)."""
    ph = []
    for s in steps:
        if s & hostlib.STEP_INTERNAL:
            ph.append("I")
        if s & hostlib.STEP_EXTERNAL:
            ph.append("E")
    out, i = [], 0
    while i < len(ph):
        ext = ph[i] == "E"
        if ext:
            i += 1
        n = 0
        while i < len(ph) and ph[i] == "I":
            n += 1
            i += 1
        out.append((ext, n))
    return out


class ShardPlan:
    """Who owns what, which robots are ghosts here, what is sent where, in which order: the C ABI's
    mgx_shard_partition / mgx_shard_plan_* (host logic of the library, identical on every rank, no
    communication); this class only holds the result for the launcher."""

    def __init__(self, sc, rank, world_size, owner=None):
        self.rank, self.world_size = rank, world_size
        n = len(sc["robots"])
        if owner is not None:
            self.owner = np.asarray(owner)
        elif n:
            self.owner = partition_strips(sc["positions"], world_size)
        else:
            self.owner = np.zeros(0, dtype=np.int64)  # an empty world: robots join later (ShardedWorld.add_robot)
        assert len(self.owner) == n
        # (connections made after the robots' first ticks on their own — sc["ir_late"], scenarios.junction_scenario — are planned
        # for from the start: their ghosts exist, and exchange records, before the factors do)
        ir = list(sc["ir"]) + list(sc.get("ir_late") or [])
        p = hostlib.shard_plan(self.owner, [c[0] for c in ir], [c[1] for c in ir], rank, world_size)
        self.local = [int(r) for r in p["local"]]
        self.ghosts = [int(g) for g in p["ghosts"]]
        self.connections = [ir[int(c)] for c in p["connections"]]  # evaluated here (target is local)
        sf, rf = p["send_first"], p["recv_first"]
        self.send_lists = [[int(g) for g in p["send_robots"][sf[q]:sf[q + 1]]] for q in range(world_size)]
        self.recv_lists = [[int(g) for g in p["recv_robots"][rf[q]:rf[q + 1]]] for q in range(world_size)]
        self.K = sc.get("K")


class ShardedWorld:
    def __init__(self, sc, rank, world_size, world_factory, comm=None, owner=None, tensor_factory=None, dynamic=False):
        """dynamic=True: the world FOLLOWS its topology (robots move, connections come and go).  Every
        rank then holds every robot of the scenario — its own ones and ghost copies of all the others,
        under the same ids everywhere — so that `update_topology` on all positions replays the
        reference's connection bookkeeping (robot.rs:1386-1586: connection sets, robot numbers, node
        slots) identically on every rank; only the factors whose target is local exist on the device,
        and the exchange lists are re-derived from the connections after every pass that changed
        something (`mgx_halo_plan_from_connections`).  Host-driven exchange only."""
        self.sc, self.plan, self.comm = sc, ShardPlan(sc, rank, world_size, owner), comm
        plan = self.plan
        self.dynamic = dynamic
        self.world = world_factory(sc["params"])
        w = self.world
        if dynamic:
            self._init_dynamic(sc, tensor_factory)
            return
        if sc.get("env") is not None:
            w.set_environment(sc["env"])  # every rank rasterises the same environment itself
        else:
            w.set_sdf(sc["sdf"]["rgb"], sc["sdf"]["world_w"], sc["sdf"]["world_h"])
        self.lid = {}
        for g in plan.local:
            rb = sc["robots"][g]
            self.lid[g] = w.add_robot(rb["mean0"], rb["prior_diag"], rb["dt"], rb["radius"], path=rb["path"],
                                      order_key=rb["order_key"])
        # ghosts: the owners of the connections evaluated here (their records arrive by the exchange) and — bookkeeping only,
        # for the MessageCount of the local graphs — the targets of connections local robots own towards other ranks
        ir_all = list(sc["ir"]) + list(sc.get("ir_late") or [])
        out_targets = sorted({b for a, b, _ in ir_all if plan.owner[a] == rank and plan.owner[b] != rank} - set(plan.ghosts))
        for g in list(plan.ghosts) + out_targets:
            rb = sc["robots"][g]
            self.lid[g] = w.add_robot(rb["mean0"], rb["prior_diag"], rb["dt"], rb["radius"], path=None,
                                      order_key=rb["order_key"], ghost=True)
        # every connection a local robot takes part in, in the order create_interrobot_factors made them (the node slots of
        # an owner's graph follow it): the ones whose target is local get device edges, the others exist on the host mirror only
        for a, b, n0 in sc["ir"]:
            if plan.owner[b] == rank or plan.owner[a] == rank:
                w.ir_connect(self.lid[a], self.lid[b], n0)
        words = w.halo_words(plan.K)
        self.send_counts = [len(l) * words for l in plan.send_lists]
        self.recv_counts = [len(l) * words for l in plan.recv_lists]
        send_flat = [self.lid[g] for l in plan.send_lists for g in l]
        recv_flat = [self.lid[g] for l in plan.recv_lists for g in l]
        w.halo_plan(send_flat, recv_flat)
        make = tensor_factory or _torch_tensor_factory
        self.send_buf = make(max(1, sum(self.send_counts)))
        self.recv_buf = make(max(1, sum(self.recv_counts)))

        self.direct = False
        self.transport = "collective" if plan.world_size > 1 else "none"  # sharded.connect() moves it to an in-engine one
        self.late_pending = bool(sc.get("ir_late"))
        if self.late_pending and (comm is not None or plan.world_size == 1):
            self.connect_late()  # (the ranks of a LocalCluster are ticked and connected by the cluster, in lockstep)

    def connect_late(self, ticked=False):
        """scenarios.populate's second half on a sharded world: the robots run sc["connect_after_ticks"] driver ticks on their own
        (every rank at the same time — the exchanges of those ticks carry the ghosts' records although nothing reads them yet),
        one more exchange so that every ghost's delivery counts are its owner's current ones (a factor created now remembers
        them, robot.rs:1549-1585), then create_interrobot_factors for the pairs of sc["ir_late"] this rank takes part in.
        ticked=True: the caller has run the ticks and the exchange (LocalCluster)."""
        from . import scenarios
        sc, plan, rank = self.sc, self.plan, self.plan.rank
        if not self.late_pending:
            return
        if not ticked:
            tick = scenarios.tick_inputs(sc)
            steps = tick.pop("steps", None) or sc["steps"]
            for _ in range(sc["connect_after_ticks"]):
                self.update_priors(**tick)
                self.iterate(steps)
            self.exchange()
        for a, b, n0 in sc["ir_late"]:
            if plan.owner[b] == rank or plan.owner[a] == rank:
                self.world.ir_connect(self.lid[a], self.lid[b], n0)
        self.late_pending = False

    # -- a world that follows its topology ---------------------------------------------------------------
    def _init_dynamic(self, sc, tensor_factory):
        w, plan = self.world, self.plan
        if sc.get("env") is not None:
            w.set_environment(sc["env"])
        elif sc.get("sdf") is not None:
            w.set_sdf(sc["sdf"]["rgb"], sc["sdf"]["world_w"], sc["sdf"]["world_h"])
        self.lid = {}
        for g, rb in enumerate(sc["robots"]):
            local = plan.owner[g] == plan.rank
            self.lid[g] = w.add_robot(rb["mean0"], rb["prior_diag"], rb["dt"], rb["radius"], path=rb["path"] if local else None,
                                      order_key=rb["order_key"], ghost=not local)
            assert self.lid[g] == g
        for a, b, n0 in sc["ir"]:
            w.ir_connect(a, b, n0)  # bookkeeping on every rank; device edges only where b is local
        self._make = tensor_factory or _torch_tensor_factory
        self.send_buf = self.recv_buf = None
        self.send_counts = self.recv_counts = [0] * plan.world_size
        self.direct = False
        self.transport = "collective" if plan.world_size > 1 else "none"  # sharded.connect() moves it to an in-engine one
        self.replan()

    def add_robot(self, mean0, prior_diag, dt, radius, path=None, owner=None, order_key=None):
        """A robot joins a world that follows its topology (a formation spawns): every rank calls this with the
        same arguments; the robot is real on its owner's rank (default: round robin) and a ghost everywhere else.
        order_key: the graph's place in the Entity order (the same on every rank); default: behind every key so far."""
        assert self.dynamic
        plan = self.plan
        g = len(plan.owner)
        owner = g % plan.world_size if owner is None else int(owner)
        local = owner == plan.rank
        # order key: above every key the scenario handed out (they need not be 0 .. n-1), the same on every rank
        self._next_key = max(getattr(self, "_next_key", 0), max((rb["order_key"] for rb in self.sc["robots"]), default=-1) + 1, g)
        if order_key is None:
            key, self._next_key = self._next_key, self._next_key + 1
        else:
            key, self._next_key = int(order_key), max(self._next_key, int(order_key) + 1)
        rid = self.world.add_robot(mean0, prior_diag, dt, radius, path=path if local else None, order_key=key, ghost=not local)
        assert rid == g
        plan.owner = np.append(plan.owner, owner)
        if local:
            plan.local.append(g)
        plan.K = np.asarray(mean0).shape[0]
        self.lid[g] = g
        self.replan()
        if getattr(self, "_slot_wiring", None) is not None and self.comm is not None:
            # A multi-process world whose exchange lives in the engines: the robot needs a slot in every other rank's receive
            # (and ghost) area and the push tables name the old lists and device indices — the engine refuses an exchange until
            # mgx_halo_direct_connect_slots has run again.  Collective, like the call itself: nobody may still be pushing into an
            # area that is about to be closed.  (The ranks of a LocalCluster are wired again by the cluster.)
            resident = bool(getattr(self, "resident", False))
            self.synchronize()
            self.comm.barrier()
            self.direct_close()
            connect(self, self.comm, "direct", resident=resident)
        return g

    def set_environment(self, env):
        self.world.set_environment(env)

    # -- migration: robots change their owning rank (re-balancing a world that follows its topology) -----------
    def _migrate_out(self, moves):
        """records of the robots this rank gives away (mgx_robot_export), which become ghosts here (mgx_robot_release)"""
        out = {g: self.world.robot_export(g) for g, a, _ in moves if a == self.plan.rank}
        for g in out:
            self.world.robot_release(g)
        return out

    def _migrate_in(self, moves, records):
        plan = self.plan
        for g, _, b in moves:
            if b == plan.rank:
                self.world.robot_import(g, records[g])
        plan.owner = plan.owner.copy()
        for g, _, b in moves:
            plan.owner[g] = b
        plan.local = [int(g) for g in np.nonzero(plan.owner == plan.rank)[0]]  # (device order of the locals: by id)
        self._slot_wiring = self._res_wiring = None  # device indices changed: whatever was wired names the old ones
        self.replan()

    def migrate(self, new_owner):
        """Re-balance: `new_owner[robot]` = the rank that is to own each robot from now on (the same table on every rank).
        Collective, BETWEEN ticks (after the sweeps that followed the last topology pass).  A robot's graph state, its
        counters and the factors attached to its variables travel as one record (include/mgx.h: mgx_robot_export /
        _import / _release) over the control-plane channel; the replicated bookkeeping stays; exchange lists and in-engine
        transports are made again.  Returns the number of robots that moved.  The results stay bit-identical to the
        unsharded world's."""
        assert self.dynamic
        plan, comm = self.plan, self.comm
        new_owner = np.asarray(new_owner)
        assert new_owner.shape == plan.owner.shape and ((0 <= new_owner) & (new_owner < plan.world_size)).all()
        moves = [(int(g), int(plan.owner[g]), int(new_owner[g])) for g in np.nonzero(new_owner != plan.owner)[0]]
        if not moves:
            return 0
        in_engine, resident = self.direct and comm is not None, bool(getattr(self, "resident", False))
        if in_engine:
            self.synchronize()
            comm.barrier()  # nobody may still be pushing into an area that is about to be closed
            self.direct_close()
        out = self._migrate_out(moves)
        records = {}
        for part in ([out] if comm is None else comm.all_gather_object(out)):
            records.update(part)
        self._migrate_in(moves, records)
        if in_engine:
            connect(self, comm, "direct", resident=resident)
        return len(moves)

    def replan(self):
        """Exchange lists for the connections now held (same result on both ends of every exchange)."""
        if self.plan.K is None:
            return
        plan, words = self.plan, self.world.halo_words(self.plan.K)
        sc_, rc_ = self.world.halo_plan_from_connections(plan.owner, plan.rank, plan.world_size)
        self.send_counts, self.recv_counts = [c * words for c in sc_], [c * words for c in rc_]
        self._send_counts_robots = [int(c) for c in sc_]
        if getattr(self, "_slot_wiring", None) is not None:
            if len(plan.owner) == self._slot_robots:
                self._aim_slots()  # the exchange lives in the engine: only the pushes' destinations follow the lists
                if getattr(self, "_res_wiring", None) is not None:
                    self._aim_resident()
            # robots joined: the areas are wired again by whoever drives the ranks (add_robot over a communicator, LocalCluster);
            # until then the engine refuses every exchange over the old tables (MGX_ERR_STATE), it does not run them
            return
        for name, need in (("send_buf", sum(self.send_counts)), ("recv_buf", sum(self.recv_counts))):
            buf = getattr(self, name)
            if buf is None or buf.numel() < max(1, need):
                setattr(self, name, self._make(max(1, need) * 2))

    def update_topology(self, positions_all, radius, next_number, method=hostlib.NEIGHBOURS_AUTO):
        """One topology pass on ALL robots' positions (the caller gathers them); then the new exchange
        lists and one exchange, so that the factors the pass created find their owners' current records
        when the next sweep lays them out.  Collective: every rank calls it with the same arguments."""
        assert self.dynamic
        out = self.world.update_topology(positions_all, radius, next_number, method=method)
        if out[1] or out[2]:
            if getattr(self, "_res_wiring", None) is not None and self.comm is not None:
                self.world.synchronize()  # (mgx_halo_resident_aim: with every rank's launches through)
                self.comm.barrier()
            self.replan()
            if getattr(self, "_res_wiring", None) is not None and self.comm is not None:
                self.comm.barrier()
            if self.comm is not None:
                if self.direct:
                    self.world.halo_direct_exchange()  # (in the engine: push, then wait for every peer's)
                else:
                    self.exchange()
        return out

    # per-tick calls of a driver over global robot ids: every rank runs the same driver (replicated
    # control plane, magics_amd/driver.py) and applies to its world what concerns it
    def read_variable_means(self, var_ix):
        assert self.dynamic
        n = len(self.plan.owner)
        out = np.zeros((n, 4))
        mine = (list(self.plan.local), self.world.read_variable_means(var_ix))
        for ids, rows in ([mine] if self.comm is None else self.comm.all_gather_object(mine)):
            out[ids] = rows
        return out

    def update_priors(self, robots, waypoints_xy, time_scale, what, max_speed, delta_t):
        robots = np.asarray(robots)
        mine = np.nonzero(self.plan.owner[robots] == self.plan.rank)[0]
        self._note_foreign_prior_changes(robots, what)
        if len(mine):
            local_ids = np.array([self.lid[int(g)] for g in robots[mine]], dtype=np.int32)
            self.world.update_priors(robots=local_ids, waypoints_xy=np.asarray(waypoints_xy)[mine],
                                     time_scale=np.asarray(time_scale)[mine], what=np.asarray(what)[mine], max_speed=max_speed,
                                     delta_t=delta_t)

    def _note_foreign_prior_changes(self, robots, what):
        """the prior updates other ranks apply to robots that are ghosts here deliver to factors local robots own: counters"""
        rs, vs = [], []
        for g, wh in zip(np.asarray(robots), np.asarray(what)):
            g = int(g)
            if self.plan.owner[g] != self.plan.rank and g in self.lid:
                if wh & 1:
                    rs.append(self.lid[g]); vs.append(self.plan.K - 1)
                if wh & 2:
                    rs.append(self.lid[g]); vs.append(0)
        if rs and hasattr(self.world, "note_change_priors"):
            self.world.note_change_priors(rs, vs)

    def set_antennas(self, robots, active):
        self.world.set_antennas(robots, active)  # flags of every robot live on every rank

    def remove_robot(self, robot):
        self.world.remove_robot(robot)

    def connections(self, robot):
        return self.world.connections(robot)

    def message_counts(self, robot):
        """MessageCount of a robot this rank owns (None for the others: their owner counts them)"""
        if self.plan.owner[robot] != self.plan.rank:
            return None
        return self.world.message_counts(self.lid[robot])

    # -- direct exchange wiring ----------------------------------------------------------------------
    def direct_setup(self, export_ipc):
        """Allocate this rank's receive area; returns what the peers need to know about it."""
        plan, ws = self.plan, self.plan.world_size
        sources = [p for p in range(ws) if plan.recv_lists[p]]
        recv, flags = self.world.halo_direct_setup(len(sources))
        offsets, acc = [], 0
        for p in range(ws):
            offsets.append(acc)
            acc += len(plan.recv_lists[p])
        info = dict(rank=plan.rank, n_records=acc, offsets=offsets, slot={p: j for j, p in enumerate(sources)})
        if export_ipc:
            info["recv_handle"], info["flags_handle"] = hostlib.ipc_export(recv), hostlib.ipc_export(flags)
        else:
            info["recv_ptr"], info["flags_ptr"] = recv, flags
        return info

    def direct_connect(self, infos):
        """infos[q]: what rank q published in direct_setup (handles are opened here)."""
        plan, ws = self.plan, self.plan.world_size
        consumers = [q for q in range(ws) if plan.send_lists[q]]
        first, base, nrec, off, slot = [0], [], [], [], []
        self._opened = getattr(self, "_opened", [])
        for q in consumers:
            inf = infos[q]
            if "recv_ptr" in inf:
                r, f = inf["recv_ptr"], inf["flags_ptr"]
            else:
                r, f = hostlib.ipc_open(inf["recv_handle"]), hostlib.ipc_open(inf["flags_handle"])
                self._opened += [r, f]
            first.append(first[-1] + len(plan.send_lists[q]))
            base.append(r)
            nrec.append(inf["n_records"])
            off.append(inf["offsets"][plan.rank])
            slot.append(f + 8 * inf["slot"][plan.rank])
        self.world.halo_direct_connect(first, base, nrec, off, slot)
        self.direct = True

    # -- the direct exchange of a world that FOLLOWS ITS TOPOLOGY: wired once, re-aimed when the lists change -----
    def direct_setup_slots(self, export_ipc, spare=64):
        """Allocate a receive area with one record slot per ghost robot (every robot another rank owns: they are all ghosts on a
        world that follows its topology) + `spare` for robots that join; returns what the peers need: where it is, its capacity
        and the slot of every robot in it."""
        assert self.dynamic
        plan, ws = self.plan, self.plan.world_size
        n = len(plan.owner)
        slots = self.world.halo_ghost_slots(np.arange(n, dtype=np.int32))
        cap = int((slots >= 0).sum()) + int(spare)
        recv, flags = self.world.halo_direct_setup_slots(ws - 1, cap)
        sources = [p for p in range(ws) if p != plan.rank]
        info = dict(rank=plan.rank, capacity=cap, slots=slots, slot={p: j for j, p in enumerate(sources)}, n_robots=n)
        if export_ipc:
            info["recv_handle"], info["flags_handle"] = hostlib.ipc_export(recv), hostlib.ipc_export(flags)
        else:
            info["recv_ptr"], info["flags_ptr"] = recv, flags
        return info

    def direct_connect_slots(self, infos):
        """infos[q]: what rank q published in direct_setup_slots (handles are opened here, once)."""
        plan = self.plan
        self._opened = getattr(self, "_opened", [])
        wiring = {}
        for q in range(plan.world_size):
            if q == plan.rank:
                continue
            inf = infos[q]
            if "recv_ptr" in inf:
                r, f = inf["recv_ptr"], inf["flags_ptr"]
            else:
                r, f = hostlib.ipc_open(inf["recv_handle"]), hostlib.ipc_open(inf["flags_handle"])
                self._opened += [r, f]
            wiring[q] = dict(recv=r, flag=f + 8 * inf["slot"][plan.rank], capacity=inf["capacity"], slots=np.asarray(inf["slots"]))
        self._slot_wiring = wiring
        self._slot_robots = len(plan.owner)
        self.direct = True
        self.transport = "direct"
        self.replan()  # (aims the pushes)

    def _aim_slots(self):
        """send list -> (peer segments, slot of every entry in its consumer's area): after every change of the lists"""
        plan, wiring = self.plan, self._slot_wiring
        if len(plan.owner) != self._slot_robots:
            raise hostlib.MgxError("robots joined since the direct exchange was wired: wire it again (direct_setup_slots / direct_connect_slots)")
        peers = [q for q in range(plan.world_size) if q != plan.rank]
        send = self.world.halo_send_list()
        first, slot = [0], []
        k = 0
        for q, cnt in ((q, self._send_counts_robots[q]) for q in range(plan.world_size)):
            if q == plan.rank:
                assert cnt == 0
                continue
            for g in send[k:k + cnt]:
                sl = int(wiring[q]["slots"][g])
                assert sl >= 0, (g, q)
                slot.append(sl)
            k += cnt
            first.append(k)
        assert k == len(send)
        self.world.halo_direct_connect_slots(first, [wiring[q]["recv"] for q in peers], [wiring[q]["capacity"] for q in peers], slot,
                                             [wiring[q]["flag"] for q in peers])

    def resident_setup_slots(self, export_ipc):
        """(a world that follows its topology, behind direct_connect_slots) allocate this rank's ghost area — a slot per ghost
        robot as it is — and say what the peers need: where, how many slots, parity, segment count, every robot's slot."""
        assert self.dynamic and getattr(self, "_slot_wiring", None) is not None
        plan = self.plan
        area, ng, par, seg, _, ok = self.world.halo_resident_setup(self.world.halo_n_recv())
        self._own_area = area
        info = dict(rank=plan.rank, n_ghosts=ng, parity=par, segments=seg, eligible=ok,
                    slots=self.world.halo_ghost_slots(np.arange(len(plan.owner), dtype=np.int32)))
        if export_ipc:
            info["area_handle"] = hostlib.ipc_export(area)
        else:
            info["area_ptr"] = area
        return info

    def resident_connect_peers(self, infos, agree=True):
        """infos[q]: what rank q published in resident_setup_slots.  Every other rank is a peer, once; which records go where
        follows the lists (_aim_resident, from replan)."""
        plan = self.plan
        peers = [q for q in range(plan.world_size) if q != plan.rank]
        areas = {}
        for q in set(peers) | {0}:
            inf = infos[q]
            if q == plan.rank:
                areas[q] = self._own_area
            elif "area_ptr" in inf:
                areas[q] = inf["area_ptr"]
            else:
                areas[q] = hostlib.ipc_open(inf["area_handle"])
                self._opened_areas = getattr(self, "_opened_areas", []) + [areas[q]]
        self.world.halo_resident_connect_peers([areas[q] for q in peers], [infos[q]["n_ghosts"] for q in peers],
                                               [infos[q]["parity"] for q in peers], [infos[q]["segments"] for q in peers],
                                               coordinator_area=areas[0] if agree else None, n_ranks=plan.world_size if agree else 0)
        self._res_wiring = {q: (j, np.asarray(infos[q]["slots"])) for j, q in enumerate(peers)}
        self.resident = True
        self.transport = "direct+resident"
        self._aim_resident()

    def _aim_resident(self):
        """which local robot's exchange records go into which peer's ghost slot: from the send list as it stands"""
        plan = self.plan
        send = self.world.halo_send_list()
        robots, peer, slot, k = [], [], [], 0
        for q in range(plan.world_size):
            cnt = self._send_counts_robots[q]
            if q != plan.rank:
                j, slots = self._res_wiring[q]
                for g in send[k:k + cnt]:
                    robots.append(self.lid[g]); peer.append(j); slot.append(int(slots[g]))
            k += cnt
        self.world.halo_resident_aim(robots, peer, slot)

    # -- resident schedule launches: ghost records travel INSIDE the launches ---------------------------
    def resident_setup(self, export_ipc):
        """Allocate this rank's ghost area (after the direct exchange is wired); returns what the peers need to know:
        where it is, how many ghost slots it has, the slot of every robot this rank receives, this rank's buffer parity
        and segment count, and whether this rank can run resident launches at all (the ranks go all or none)."""
        plan = self.plan
        flat = [g for l in plan.recv_lists for g in l]
        area, ng, par, seg, slots, ok = self.world.halo_resident_setup(len(flat))
        self._own_area = area
        info = dict(rank=plan.rank, n_ghosts=ng, parity=par, segments=seg, eligible=ok, slots=dict(zip(flat, slots)))
        if export_ipc:
            info["area_handle"] = hostlib.ipc_export(area)
        else:
            info["area_ptr"] = area
        return info

    def resident_connect(self, infos, agree=True):
        """infos[q]: what rank q published in resident_setup (handles are opened here).  Rank 0's area doubles as the place
        where the ranks agree on every schedule's launches (include/mgx.h: coordinator_area), so every rank maps it."""
        plan = self.plan
        self._opened = getattr(self, "_opened", [])
        mapped = {}

        def area_of(q):
            if q not in mapped:
                inf = infos[q]
                if "area_ptr" in inf:
                    mapped[q] = inf["area_ptr"]
                elif q == plan.rank:
                    mapped[q] = self._own_area
                else:
                    mapped[q] = hostlib.ipc_open(inf["area_handle"])
                    self._opened_areas = getattr(self, "_opened_areas", []) + [mapped[q]]
            return mapped[q]
        robots, area, ngs, slot, par, seg = [], [], [], [], [], []
        for q in range(plan.world_size):
            if not plan.send_lists[q]:
                continue
            inf = infos[q]
            a = area_of(q)
            for g in plan.send_lists[q]:
                robots.append(self.lid[g]); area.append(a); ngs.append(inf["n_ghosts"]); slot.append(inf["slots"][g])
                par.append(inf["parity"]); seg.append(inf["segments"])
        self.world.halo_resident_connect(robots, area, ngs, slot, par, seg, coordinator_area=area_of(0) if agree else None,
                                         n_ranks=plan.world_size if agree else 0)
        self.resident = True

    def direct_close(self):
        """Call on every rank, after a barrier: nobody may still be pushing into a closed area."""
        if getattr(self, "resident", False):
            self.world.halo_resident_disconnect()
            self.resident = False
        if getattr(self, "rccl", False):
            self.world.halo_rccl_disconnect()
            self.direct = self.rccl = False
        if self.direct:
            self.world.halo_direct_disconnect()
            self.direct = False
        for ptr in getattr(self, "_opened", []) + getattr(self, "_opened_areas", []):
            hostlib.ipc_close(ptr)
        self._opened, self._opened_areas = [], []
        self._slot_wiring = self._res_wiring = None

    def resident_close(self):
        """The ghost areas only (the direct exchange stays wired).  Call on every rank, after a barrier."""
        self.world.halo_resident_disconnect()
        self.resident = False
        for ptr in getattr(self, "_opened_areas", []):
            hostlib.ipc_close(ptr)
        self._opened_areas = []

    # -- exchange pieces (a LocalCluster drives them itself) ---------------------------------------
    def pack(self):
        self.world.halo_pack(self.send_buf.data_ptr())

    def unpack(self):
        self.world.halo_unpack(self.recv_buf.data_ptr())

    def exchange(self):
        if self.plan.world_size == 1 or self.direct:  # direct: the engine exchanges inside the launch sequence
            return
        self.pack()
        foreign = self._collective_on_another_stream()
        if foreign:      # pack ran on the world's stream, the collective runs on torch's current one: order them
            self.world.synchronize()
        self.comm.all_to_all(self.recv_buf, self.send_buf, self.recv_counts, self.send_counts)
        if foreign:
            import torch
            torch.cuda.current_stream().synchronize()
        self.unpack()

    def _collective_on_another_stream(self):
        """the device buffers travel through torch.distributed on torch's CURRENT stream; pack / unpack run on the world's
        (decided once per world stream: this sits in the per-iteration path of the host-driven exchange)"""
        key = int(getattr(self.world, "stream_handle", 0) or 0)
        cached = getattr(self, "_foreign_stream", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        buf = self.send_buf
        foreign = False
        if buf is not None and getattr(buf, "is_cuda", False):
            import torch
            foreign = int(torch.cuda.current_stream().cuda_stream) != key
        self._foreign_stream = (key, foreign)
        return foreign

    def sweep_segment(self, ext, n_int, next_ext=False):
        hints = hostlib.HINT_NEXT_STARTS_EXTERNAL if (ext and next_ext) else 0
        self.world.sweep(PH_FACTOR | PH_VARIABLE if ext else 0, PH_FACTOR | PH_VARIABLE if n_int else 0, n_int, hints=hints)

    # -- World-like interface over global robot ids -------------------------------------------------
    def iterate(self, steps):
        if self.direct or self.plan.world_size == 1:
            if getattr(self, "_thaw_watch", False):
                self._settle_resident()
            self.world.iterate(steps)  # one C call: launches (and exchanges) are sequenced by the engine
            return
        segs = segments(steps)
        for k, (ext, n_int) in enumerate(segs):
            if ext:
                self.exchange()
            self.sweep_segment(ext, n_int, next_ext=k + 1 < len(segs) and segs[k + 1][0])

    def batch(self):
        """`with sw.batch(): for ...: sw.iterate(steps)` — World.batch for a rank whose exchange lives in the engine (every rank
        brackets alike: the merged launches wait for each other's records like any schedule's); a no-op where the exchange is
        driven from the host (its collectives sit between the launches)"""
        if self.direct or self.plan.world_size == 1:
            return self.world.batch()
        import contextlib
        return contextlib.nullcontext()

    def set_antenna(self, robot, active):
        if robot in self.lid:
            self.world.set_antenna(self.lid[robot], active)

    def set_idle(self, robot, idle):
        if robot in self.lid:
            self.world.set_idle(self.lid[robot], idle)


    def change_prior(self, robot, var_ix, mean):
        if self.plan.owner[robot] == self.plan.rank:
            self.world.change_prior(self.lid[robot], var_ix, mean)
        elif robot in self.lid and hasattr(self.world, "note_change_priors"):
            self.world.note_change_priors([self.lid[robot]], [var_ix])  # counters of the local factors attached to it

    def set_enabled(self, mask):
        """change_factor_enabled on every rank (collective): an exchange first, so that the ghosts' records the
        inter-robot factors freeze with / thaw against are their owners' current ones."""
        if not self.direct:
            self.exchange()
        self.world.set_enabled(mask)
        self._settle_resident()

    def _settle_resident(self):
        """Whether a schedule may run as one resident launch has to come out the same on every rank (a rank that went resident
        next to one that did not would wait for records that never come), and "factors are still thawing" depends on the flags of
        the robots a rank holds.  So after a switch of factor kinds resident launches are off everywhere until NO rank is thawing
        any more (asked over the control plane in front of every schedule while that lasts)."""
        if not getattr(self, "resident", False):
            return
        mine = self.world.is_thawing()
        anyone = any(self.comm.all_gather_object(mine)) if self.comm is not None else mine
        self.world.set_resident_launches(not anyone)
        self._thaw_watch = anyone

    def read_beliefs(self):
        """(global robot ids of the local robots, eta, lam, means) of this rank."""
        eta, lam, mu = self.world.read_beliefs()
        return list(self.plan.local), eta, lam, mu

    def synchronize(self):
        self.world.synchronize()

    def flush(self):
        """everything issued so far is enqueued (World.flush: a lingering launch is told to end, nothing is waited for)"""
        if hasattr(self.world, "flush"):
            self.world.flush()


def _torch_tensor_factory(n):
    """device buffers of the halo exchange (torch is the allocator / collective plumbing)"""
    import torch
    if not torch.cuda.is_available():
        raise hostlib.MgxError("halo buffers need a GPU (pass tensor_factory explicitly for host-side tests)")
    return torch.zeros(n, dtype=torch.float64, device="cuda")


def connect_rccl(sw, comm):
    """The all-to-all-v inside the library: grouped ncclSend / ncclRecv enqueued by mgx_iterate itself
    (include/mgx.h).  `comm` is only the control plane that spreads the RCCL unique id."""
    plan = sw.plan
    if plan.world_size == 1:
        return
    uid = hostlib.rccl_unique_id() if plan.rank == 0 else None
    uid = comm.all_gather_object(uid)[0]
    peers = [q for q in range(plan.world_size) if plan.send_lists[q] or plan.recv_lists[q]]
    send_first, recv_first = [0], [0]
    for q in peers:
        send_first.append(send_first[-1] + len(plan.send_lists[q]))
        recv_first.append(recv_first[-1] + len(plan.recv_lists[q]))
    sw.world.halo_rccl_connect(uid, plan.world_size, plan.rank, peers, send_first, recv_first)
    sw.direct = sw.rccl = True  # same driving mode: the engine exchanges inside its launch sequence
    comm.barrier()


def connect_direct(sw, comm):
    """Wire the direct exchange of a multi-process run: one all-gather of the (tiny) area
    descriptions over the control-plane communicator, then every rank maps its consumers' areas."""
    if sw.plan.world_size == 1:
        return
    infos = comm.all_gather_object(sw.direct_setup(export_ipc=True))
    sw.direct_connect({inf["rank"]: inf for inf in infos})
    comm.barrier()


def _connect_resident(sw, comm, all_ok):
    """On top of a wired direct exchange: the ghost areas for resident schedule launches.  Every rank reports whether it can
    run them; only if ALL can (and all succeed in mapping their consumers' areas) are they switched on — a rank that ran
    resident launches next to one that did not would wait for records that never come."""
    info = None
    try:
        info = sw.resident_setup(export_ipc=True)
    except Exception:  # noqa: BLE001
        info = None
    infos = comm.all_gather_object(info)
    if any(i is None or i["eligible"] != 1 for i in infos):
        return False
    try:
        sw.resident_connect({i["rank"]: i for i in infos})
        ok = True
    except Exception:  # noqa: BLE001
        ok = False
    if all_ok(ok):
        comm.barrier()
        sw.transport = "direct+resident"
        return True
    if ok:
        sw.world.halo_resident_disconnect()
        sw.resident = False
    comm.barrier()
    return False


def _connect_resident_slots(sw, comm, all_ok):
    """The same for a world that follows its topology (behind direct_connect_slots): every other rank is a peer once, the push
    tables follow the exchange lists (ShardedWorld._aim_resident).  A rank without inter-robot factors yet is welcome (eligible 2):
    every schedule is decided where the ranks agree."""
    info = None
    try:
        sw.world.sweep(0, 0, 0)
        info = sw.resident_setup_slots(export_ipc=True)
    except Exception:  # noqa: BLE001
        info = None
    infos = comm.all_gather_object(info)
    if any(i is None or i["eligible"] < 1 for i in infos):
        return False
    try:
        sw.resident_connect_peers({i["rank"]: i for i in infos})
        ok = True
    except Exception:  # noqa: BLE001
        ok = False
    if all_ok(ok):
        comm.barrier()
        return True
    if ok:
        sw.world.halo_resident_disconnect()
        sw.resident, sw._res_wiring, sw.transport = False, None, "direct"
    comm.barrier()
    return False


def rewire_resident(sw, comm):
    """Collective: wire the resident launches of a multi-process world again (after a change of some rank's layout switched
    them off there).  The order matters — a rank's ghost area (rank 0's also holds the agreement word) is stored into by its
    peers' launches for as long as they hold it mapped: barrier (nobody is inside a launch: every rank has synchronised),
    disconnect + close the mappings everywhere, barrier, then set up and connect as the first time."""
    sw.synchronize()
    comm.barrier()
    sw.resident_close()
    comm.barrier()
    if sw.transport == "direct+resident":
        sw.transport = "direct"
    return _connect_resident(sw, comm, lambda ok: all(comm.all_gather_object(bool(ok))))


def connect(sw, comm, transport="auto", resident=True):
    """Wire the exchange of a multi-process sharded world, in-engine transports first (one C call per tick, no host
    work per exchange): "direct" (peer-mapped stores through hipIpc) -> "rccl" (grouped ncclSend / ncclRecv enqueued by
    the engine) -> "collective" (pack / all_to_all_single / unpack driven from the host: the fallback that only needs
    `comm`).  With "direct" wired and `resident`, the ranks then try to agree on resident schedule launches ("direct+resident":
    ghost records and progress words travel inside ONE launch per schedule and rank, include/mgx.h).
    Collective: every rank calls it with the same arguments; after every step the ranks agree on whether ALL
    of them succeeded, so a failure on one rank moves every rank to the next transport instead of leaving the others
    waiting.  Returns the transport in use (also `sw.transport`)."""
    if sw.plan.world_size == 1 or comm is None:
        sw.transport = "none"
        return sw.transport
    order = {"auto": ["direct", "rccl", "collective"], "direct": ["direct", "collective"], "rccl": ["rccl", "collective"],
             "collective": ["collective"]}[transport]

    def all_ok(ok):
        return all(comm.all_gather_object(bool(ok)))
    for t in order:
        if t == "collective":
            break
        try:
            if t == "direct":
                slots = bool(getattr(sw, "dynamic", False))  # a world that follows its topology: one slot per ghost robot
                info, err = None, None
                try:
                    import os
                    if os.environ.get("MGX_TEST_FAIL_DIRECT_SETUP_RANK") == str(sw.plan.rank):  # (fault injection for the fallback tests)
                        raise hostlib.MgxError("injected: this rank cannot set up its receive area")
                    info = sw.direct_setup_slots(export_ipc=True) if slots else sw.direct_setup(export_ipc=True)
                except Exception as e:  # noqa: BLE001
                    err = e
                infos = comm.all_gather_object(info)
                if any(i is None for i in infos):
                    comm.barrier()
                    sw.direct_close()  # the ranks whose set-up succeeded free their areas (nobody has mapped them yet)
                    continue
                try:
                    if slots:
                        sw.direct_connect_slots({i["rank"]: i for i in infos})
                    else:
                        sw.direct_connect({i["rank"]: i for i in infos})
                    ok = True
                except Exception:  # noqa: BLE001
                    ok = False
                if all_ok(ok):
                    comm.barrier()
                    sw.transport = "direct"
                    if resident:
                        (_connect_resident_slots if slots else _connect_resident)(sw, comm, all_ok)
                    return sw.transport
                comm.barrier()
                sw.direct_close()
            else:
                uid = None
                try:
                    uid = hostlib.rccl_unique_id() if sw.plan.rank == 0 else b""
                except Exception:  # noqa: BLE001
                    uid = None
                uids = comm.all_gather_object(uid)
                if uids[0] is None:
                    continue
                plan = sw.plan
                peers = [q for q in range(plan.world_size) if plan.send_lists[q] or plan.recv_lists[q]]
                send_first, recv_first = [0], [0]
                for q in peers:
                    send_first.append(send_first[-1] + len(plan.send_lists[q]))
                    recv_first.append(recv_first[-1] + len(plan.recv_lists[q]))
                try:
                    sw.world.halo_rccl_connect(uids[0], plan.world_size, plan.rank, peers, send_first, recv_first)
                    ok = True
                except Exception:  # noqa: BLE001
                    ok = False
                if all_ok(ok):
                    sw.direct = sw.rccl = True
                    comm.barrier()
                    sw.transport = "rccl"
                    return sw.transport
                if ok:
                    sw.world.halo_rccl_disconnect()
        except Exception:  # noqa: BLE001
            raise
    sw.transport = "collective"
    return sw.transport


class TorchDistComm:
    """all-to-all-v over torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" on CPU)."""

    def __init__(self, group=None, stage_through_host=False):
        """stage_through_host: device buffers travel through host copies (a backend without device
        support, i.e. gloo with GPU worlds: dry runs only)."""
        import torch.distributed as dist
        self.dist, self.group, self.stage = dist, group, stage_through_host

    def all_to_all(self, recv, send, recv_counts, send_counts):
        n_r, n_s = sum(recv_counts), sum(send_counts)
        if self.stage and recv.is_cuda:
            r, s_ = recv[:n_r].cpu(), send[:n_s].cpu()
            self.dist.all_to_all_single(r, s_, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts),
                                        group=self.group)
            recv[:n_r].copy_(r)
            return
        self.dist.all_to_all_single(recv[:n_r], send[:n_s], output_split_sizes=list(recv_counts),
                                    input_split_sizes=list(send_counts), group=self.group)

    def all_gather_object(self, obj):
        out = [None] * self.dist.get_world_size(self.group)
        self.dist.all_gather_object(out, obj, group=self.group)
        return out

    def barrier(self):
        self.dist.barrier(group=self.group)


class LocalCluster:
    """All ranks of a sharded world inside ONE process (one GPU): used by the tests to check the
    ghost / halo numerics against the unsharded world without a multi-GPU node."""

    def __init__(self, sc, world_size, world_factory, owner=None, tensor_factory=None, direct=False, dynamic=False, resident=False,
                 agree=True):
        """direct=True: the ranks exchange through peer-mapped stores (same address space, no IPC);
        `world_factory` must then give every rank its OWN stream — a rank's wait kernel would block
        a shared stream before the other rank's stores are even enqueued.
        dynamic=True: worlds that follow their topology (see ShardedWorld); the cluster then also
        offers the per-tick calls of magics_amd.driver.Driver over global robot ids."""
        self.ranks = [ShardedWorld(sc, r, world_size, world_factory, comm=None, owner=owner,
                                   tensor_factory=tensor_factory, dynamic=dynamic) for r in range(world_size)]
        self.n_robots, self.K = len(sc["robots"]), sc.get("K")
        self.resident = False
        self.agree = agree
        if sc.get("ir_late") and not dynamic:
            # robots that iterate on their own before they meet (ShardedWorld.connect_late): the ticks in lockstep over the
            # collective transport, the factors, THEN the in-engine transports are wired (for the exchange lists as they end up)
            from . import scenarios
            self.direct_slots, self._want_resident = False, False
            tick = scenarios.tick_inputs(sc)
            for _ in range(sc["connect_after_ticks"]):
                self.tick(steps=sc["steps"], **tick)
            if world_size > 1:
                self._exchange()
            for sw in self.ranks:
                sw.connect_late(ticked=True)
        self.direct_slots = bool(direct and dynamic and world_size > 1)
        self._want_resident = bool(resident and self.direct_slots)
        if self.direct_slots:
            if self.K is not None:
                self._wire_slots()
        elif direct and world_size > 1:
            infos = {sw.plan.rank: sw.direct_setup(export_ipc=False) for sw in self.ranks}
            for sw in self.ranks:
                sw.direct_connect(infos)
            if resident:
                # resident=True: every rank's workgroups have to be on the device TOGETHER (each rank a stream of its own on
                # a hardware queue of its own, and all ranks' robots within the device's resident slots)
                for sw in self.ranks:
                    sw.world.sweep(0, 0, 0)
                infos = {sw.plan.rank: sw.resident_setup(export_ipc=False) for sw in self.ranks}
                if all(i["eligible"] == 1 for i in infos.values()):
                    for sw in self.ranks:
                        sw.resident_connect(infos, agree=agree)
                    self.resident = True
                    self.agree = agree

    def _wire_slots(self):
        """the direct exchange of a cluster that follows its topology: one record slot per ghost robot, aimed again by every
        rank's replan (ShardedWorld.direct_setup_slots / direct_connect_slots); wired again when robots join"""
        for sw in self.ranks:
            sw.synchronize()
        for sw in self.ranks:
            if getattr(sw, "resident", False):
                sw.world.halo_resident_disconnect()
                sw.resident, sw._res_wiring = False, None
        self.resident = False
        infos = {sw.plan.rank: sw.direct_setup_slots(export_ipc=False) for sw in self.ranks}
        for sw in self.ranks:
            sw.direct_connect_slots(infos)
        if self._want_resident:
            for sw in self.ranks:
                sw.world.sweep(0, 0, 0)
            infos = {sw.plan.rank: sw.resident_setup_slots(export_ipc=False) for sw in self.ranks}
            if all(i["eligible"] >= 1 for i in infos.values()):
                for sw in self.ranks:
                    sw.resident_connect_peers(infos, agree=self.agree)
                self.resident = True

    def _exchange(self):
        if self.direct_slots and self.ranks[0].direct:  # in the engines: all pushes before the first wait (one thread drives all ranks)
            for sw in self.ranks:
                sw.world.halo_direct_exchange(hostlib.HALO_PUSH)
            for sw in self.ranks:
                sw.world.halo_direct_exchange(hostlib.HALO_WAIT)
            return
        for sw in self.ranks:
            sw.pack()
        for sw in self.ranks:
            sw.synchronize()
        for dst in self.ranks:  # the all-to-all-v, by hand
            off_r = 0
            for p, src in enumerate(self.ranks):
                cnt = dst.recv_counts[p]
                off_s = sum(src.send_counts[:dst.plan.rank])
                assert cnt == src.send_counts[dst.plan.rank]
                if cnt:
                    dst.recv_buf[off_r:off_r + cnt].copy_(src.send_buf[off_s:off_s + cnt])
                off_r += cnt
        _sync_tensors(self.ranks)
        for sw in self.ranks:
            sw.unpack()

    def iterate(self, steps):
        segs = segments(steps)
        if getattr(self, "_thaw_watch", False):
            self._settle_resident()
        # (a schedule the engines would run launch by launch — too short, inter-robot factors off, backing off after a declined
        # launch — is driven from here, in lockstep: see below)
        if self.resident and not getattr(self, "_thaw_watch", False) and self.ranks[0].world.resident_ready(steps):
            if segs[0][0]:  # a schedule that opens with an external iteration: its exchange, all pushes first
                for sw in self.ranks:
                    sw.world.halo_direct_exchange(hostlib.HALO_PUSH)
            for sw in self.ranks:  # one launch per rank, each on its own stream: they run side by side
                sw.world.iterate(steps)
            # ... if the device takes them side by side: the launches find out by themselves and agree on one answer
            # (include/mgx.h: mgx_halo_resident_connect).  After a "no" every world is as it was, and the schedule runs launch
            # by launch below — from here, not inside each rank's next call: one thread drives all ranks, and their exchanges
            # must be enqueued in lockstep (all pushes before the first wait).
            if not self.agree:
                return
            outcomes = [sw.world.resident_outcome() for sw in self.ranks]
            assert all(o == outcomes[0] for o in outcomes), f"the ranks' launches disagree: {outcomes}"
            if outcomes[0] != hostlib.RESIDENT_DECLINED:
                return
            self.declined = getattr(self, "declined", 0) + 1
        for k, (ext, n_int) in enumerate(segs):
            if ext and len(self.ranks) > 1:
                if self.ranks[0].direct:
                    # all pushes before the first wait: the ranks' streams may share a hardware queue
                    # here, and a waiting kernel must never sit in front of the stores it waits for
                    for sw in self.ranks:
                        sw.world.halo_direct_exchange(hostlib.HALO_PUSH)
                else:
                    self._exchange()
            for sw in self.ranks:
                sw.sweep_segment(ext, n_int, next_ext=k + 1 < len(segs) and segs[k + 1][0])

    # -- what a scenario runner needs on top (magics_amd/sim.py on a dynamic cluster) --------------------------
    def set_environment(self, env):
        for sw in self.ranks:
            sw.set_environment(env)

    def add_robot(self, mean0, prior_diag, dt, radius, path=None, owner=None, order_key=None):
        ids = [sw.add_robot(mean0, prior_diag, dt, radius, path=path, owner=owner, order_key=order_key) for sw in self.ranks]
        assert all(i == ids[0] for i in ids)
        self.n_robots, self.K = ids[0] + 1, np.asarray(mean0).shape[0]
        if self.direct_slots:
            self._wire_slots()  # (a robot more: every rank's area has a slot more, or the robot a slot in the others')
        return ids[0]

    def migrate(self, new_owner):
        """ShardedWorld.migrate for all ranks of the cluster (the records change hands in-process)"""
        new_owner = np.asarray(new_owner)
        old = self.ranks[0].plan.owner
        moves = [(int(g), int(old[g]), int(new_owner[g])) for g in np.nonzero(new_owner != old)[0]]
        if not moves:
            return 0
        for sw in self.ranks:
            sw.synchronize()
        records = {}
        for sw in self.ranks:
            records.update(sw._migrate_out(moves))
        for sw in self.ranks:
            sw._migrate_in(moves, records)
        if self.direct_slots:
            self._wire_slots()
        return len(moves)

    def tick(self, robots, waypoints_xy, time_scale, what, max_speed, delta_t, steps):
        self.update_priors(robots, waypoints_xy, time_scale, what, max_speed, delta_t)
        self.iterate(steps)

    # -- per-tick calls of a driver, over global robot ids (dynamic clusters) ------------------------------
    def update_topology(self, positions_all, radius, next_number, method=hostlib.NEIGHBOURS_AUTO):
        if self.resident and self.direct_slots:
            for sw in self.ranks:
                sw.synchronize()  # (the aims behind a pass that changed the lists: with every rank's launches through)
        outs = [sw.update_topology(positions_all, radius, next_number, method=method) for sw in self.ranks]
        assert all(o == outs[0] for o in outs), "the replicated bookkeeping diverged"
        if (outs[0][1] or outs[0][2]) and len(self.ranks) > 1:
            self._exchange()
        return outs[0]

    def read_variable_means(self, var_ix):
        out = np.zeros((self.n_robots, 4))
        for sw in self.ranks:
            out[sw.plan.local] = sw.world.read_variable_means(var_ix)
        return out

    def update_priors(self, robots, waypoints_xy, time_scale, what, max_speed, delta_t):
        robots = np.asarray(robots)
        for sw in self.ranks:
            mine = np.nonzero(sw.plan.owner[robots] == sw.plan.rank)[0]
            sw._note_foreign_prior_changes(robots, what)
            if len(mine):
                local_ids = np.array([sw.lid[int(g)] for g in robots[mine]], dtype=np.int32)  # rank-local ids (equal to the global ones on a dynamic world)
                sw.world.update_priors(robots=local_ids, waypoints_xy=np.asarray(waypoints_xy)[mine],
                                       time_scale=np.asarray(time_scale)[mine], what=np.asarray(what)[mine], max_speed=max_speed,
                                       delta_t=delta_t)

    def set_antennas(self, robots, active):
        for sw in self.ranks:  # flags of every robot are kept on every rank (they gate both ends of a connection)
            sw.world.set_antennas(robots, active)

    def remove_robot(self, robot):
        for sw in self.ranks:
            sw.world.remove_robot(robot)

    def set_enabled(self, mask):
        if len(self.ranks) > 1:
            if self.ranks[0].direct:
                # the engine exchanges by itself when inter-robot factors are switched (the ghosts' plain copies are as old as the
                # last exchange kernel): all pushes before the first wait, as in iterate
                for sw in self.ranks:
                    sw.world.halo_direct_exchange(hostlib.HALO_PUSH)
            else:
                self._exchange()
        for sw in self.ranks:
            sw.world.set_enabled(mask)
        self._settle_resident()

    def _settle_resident(self):
        """see ShardedWorld._settle_resident: resident launches are off on every rank while any rank is thawing"""
        if not self.resident:
            return
        anyone = any(sw.world.is_thawing() for sw in self.ranks)
        for sw in self.ranks:
            sw.world.set_resident_launches(not anyone)
        self._thaw_watch = anyone

    def connections(self, robot):
        outs = [sw.world.connections(robot) for sw in self.ranks]
        assert all(list(o) == list(outs[0]) for o in outs)
        return outs[0]

    def message_counts(self, robot):
        for sw in self.ranks:
            c = sw.message_counts(robot)
            if c is not None:
                return c
        raise KeyError(robot)

    def set_antenna(self, robot, active):
        for sw in self.ranks:
            if sw.dynamic:
                sw.world.set_antenna(robot, active)
            else:
                sw.set_antenna(robot, active)

    def set_idle(self, robot, idle):
        for sw in self.ranks:
            if sw.dynamic:
                sw.world.set_idle(robot, idle)
            else:
                sw.set_idle(robot, idle)

    def change_prior(self, robot, var_ix, mean):
        for sw in self.ranks:
            sw.change_prior(robot, var_ix, mean)

    def read_beliefs(self):
        K = self.K
        eta, lam, mu = np.zeros((self.n_robots * K, 4)), np.zeros((self.n_robots * K, 4, 4)), np.zeros((self.n_robots * K, 4))
        for sw in self.ranks:
            ids, e, l, m = sw.read_beliefs()
            for j, g in enumerate(ids):
                eta[g * K:(g + 1) * K] = e[j * K:(j + 1) * K]
                lam[g * K:(g + 1) * K] = l[j * K:(j + 1) * K]
                mu[g * K:(g + 1) * K] = m[j * K:(j + 1) * K]
        return eta, lam, mu


def _sync_tensors(ranks):
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except ImportError:
        pass
