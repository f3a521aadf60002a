"""Host-side mirror of the reference's FactorGraph API over the C ABI (include/mgx.h).

``World`` owns every robot's factor graph on the device; the method names follow
crates/magics/src/factorgraph/factorgraph.rs and crates/magics/src/planner/robot.rs
(``internal_factor_iteration``, ``change_prior`` ...).  ``FactorGraph`` is the per-robot
handle a reference user would hold (a Bevy component there, ``(world, robot_id)`` here).
"""
import ctypes as C

import numpy as np

from . import hostlib
from .hostlib import Params, RobotDesc, c_double_p
from .hostlib import check as _check


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


def make_params(p):
    return Params(
        float(p["sigma_dynamics"]), float(p["sigma_interrobot"]), float(p["sigma_obstacle"]),
        float(p["sigma_tracking"]), float(p["safety_multiplier"]),
        float(p.get("tracking_switch_padding", 1.0)), float(p.get("tracking_attraction_distance", 2.0)),
        int(p.get("enable_mask", 7)), 0)


class World:
    def _chk(self, rc):
        return _check(rc, self._L)

    def __init__(self, params, stream=None, fma=None):
        self._L = hostlib.lib(fma)
        self._p = make_params(params)
        h = C.c_void_p()
        self._chk(self._L.mgx_world_create(C.byref(self._p), C.byref(h)))
        self._w = h
        self._next_key = 0
        self._args = {}  # array arguments seen before: id -> (the array, its address), see _arg
        self.stream_handle = 0  # raw HIP stream every launch of this world goes to (0: the default stream)
        if stream is not None:
            self.set_stream(stream)

    # -- lifecycle -----------------------------------------------------------------------
    def close(self):
        if getattr(self, "_w", None):
            self._L.mgx_world_destroy(self._w)
            self._w = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream):
        """``stream``: raw hipStream_t value (e.g. ``torch.cuda.current_stream().cuda_stream``)."""
        self._chk(self._L.mgx_set_stream(self._w, C.c_void_p(int(stream))))
        self.stream_handle = int(stream)

    def synchronize(self):
        self._chk(self._L.mgx_synchronize(self._w))

    def flush(self):
        """everything issued so far is enqueued: a batch is submitted, a lingering launch told to end — no wait (mgx_flush)"""
        self._chk(self._L.mgx_flush(self._w))

    def set_linger(self, microseconds):
        """how long a resident launch waits for the next schedule before it ends (0: never; None: the default) — mgx_set_linger"""
        self._chk(self._L.mgx_set_linger(self._w, -1 if microseconds is None else int(microseconds)))

    def linger_stats(self):
        """(launches that lingered, schedules posted into them, posts taken back and re-run, launches that ended by themselves)"""
        v = [C.c_uint64() for _ in range(4)]
        self._chk(self._L.mgx_linger_stats(self._w, *[C.byref(x) for x in v]))
        return tuple(int(x.value) for x in v)

    # -- topology --------------------------------------------------------------------------
    def set_sdf(self, rgb, world_w, world_h):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        if rgb.ndim != 3 or rgb.shape[2] != 3:
            raise ValueError("rgb must be HxWx3 u8")
        h, w, _ = rgb.shape
        self._chk(self._L.mgx_world_set_sdf(self._w, rgb.ctypes.data, w, h, float(world_w), float(world_h)))

    def set_environment(self, env):
        """simulation_loader.rs:154-162 + robot.rs:1259-1264: rasterise + blur the environment on the
        device with its own sdf settings and use the result as the obstacle image."""
        from .environment import _Desc
        d = _Desc(env)
        self._chk(self._L.mgx_world_set_environment(self._w, C.byref(d.desc)))

    def add_robot(self, mean0, prior_diag, dt, radius, path=None, order_key=None, ghost=False):
        mean0 = _f64(mean0)
        K = mean0.shape[0]
        prior_diag = _f64(prior_diag, (K,))
        dt = _f64(dt, (K - 1,))
        d = RobotDesc()
        d.K = K
        d.mean0, d.prior_diag, d.dt = _dp(mean0), _dp(prior_diag), _dp(dt)
        d.radius = float(radius)
        if path is not None:
            path = np.ascontiguousarray(path, dtype=np.float32)
            d.n_path = path.shape[0]
            d.path_xy = path.ctypes.data_as(C.POINTER(C.c_float))
        if order_key is None:
            order_key = self._next_key
        self._next_key = max(self._next_key, int(order_key) + 1)
        d.order_key = int(order_key)
        d.ghost = 1 if ghost else 0
        rid = C.c_int32(-1)
        self._chk(self._L.mgx_robot_add(self._w, C.byref(d), C.byref(rid)))
        return rid.value

    def ir_connect(self, owner, other, first_robot_number):
        self._chk(self._L.mgx_ir_connect(self._w, owner, other, int(first_robot_number)))

    def remove_robot(self, robot):
        """Entity despawn (robot.rs:2172): the robot leaves every query for good."""
        self._chk(self._L.mgx_robot_remove(self._w, robot))

    def ir_disconnect(self, a, b):
        self._chk(self._L.mgx_ir_disconnect(self._w, a, b))

    def set_enabled(self, mask):
        """change_factor_enabled for every graph (factorgraph.rs:1529-1539): MGX_FACTOR_* bits."""
        self._chk(self._L.mgx_set_enabled(self._w, int(mask)))

    def set_antenna(self, robot, active):
        self._chk(self._L.mgx_set_antenna(self._w, robot, int(bool(active))))

    def set_idle(self, robot, idle):
        self._chk(self._L.mgx_set_idle(self._w, robot, int(bool(idle))))

    def set_antennas(self, robots, active):
        """update_failed_comms (robot.rs:1593-1601): one write per antenna per tick."""
        robots = np.ascontiguousarray(robots, dtype=np.int32)
        active = np.ascontiguousarray(active, dtype=np.uint8)
        assert robots.shape == active.shape
        self._chk(self._L.mgx_set_antennas(self._w, robots.size, robots.ctypes.data, active.ctypes.data))

    # -- dynamic inter-robot topology (robot.rs:1362-1586) --------------------------------------
    def neighbours(self, positions, radius, method=hostlib.NEIGHBOURS_AUTO):
        """update_robot_neighbours: CSR (row_ptr, neighbours) of robots_within_comms_range."""
        pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        ptr = np.zeros(pos.shape[0] + 1, dtype=np.int32)
        need = C.c_uint64()
        self._chk(self._L.mgx_neighbours(self._w, pos.ctypes.data, float(radius), method, ptr.ctypes.data, None, 0, C.byref(need)))
        idx = np.zeros(max(need.value, 1), dtype=np.int32)
        self._chk(self._L.mgx_neighbours(self._w, pos.ctypes.data, float(radius), method, ptr.ctypes.data, idx.ctypes.data,
                                         need.value, C.byref(need)))
        return ptr, idx[:need.value]

    def update_topology(self, positions, radius, next_number, method=hostlib.NEIGHBOURS_AUTO):
        """update_robot_neighbours + delete_/create_interrobot_factors; returns (next robot number,
        connections created, pairs deleted)."""
        pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        nxt = C.c_uint64(int(next_number))
        stats = (C.c_uint32 * 2)()
        self._chk(self._L.mgx_update_topology(self._w, pos.ctypes.data, float(radius), method, C.byref(nxt), stats))
        return nxt.value, stats[0], stats[1]

    def connections(self, robot):
        n = C.c_uint32()
        self._chk(self._L.mgx_connections(self._w, robot, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.int32)
        self._chk(self._L.mgx_connections(self._w, robot, out.ctypes.data, n.value, C.byref(n)))
        return out[:n.value].tolist()

    # -- hot path ----------------------------------------------------------------------------
    def iterate(self, steps):
        # (a tick's schedule is issued over and over: its byte string is made once — building it costs more host time than the call)
        key = steps if isinstance(steps, (bytes, tuple)) else tuple(steps)
        b = _STEP_BYTES.get(key)
        if b is None:
            if len(_STEP_BYTES) > 256:
                _STEP_BYTES.clear()
            b = _STEP_BYTES[key] = bytes(bytearray(int(s) for s in steps))
        self._chk(self._L.mgx_iterate(self._w, b, len(b)))

    def batch_begin(self):
        self._chk(self._L.mgx_batch_begin(self._w))

    def batch_end(self):
        """-> (schedules recorded since batch_begin, sweep-kernel launches they were submitted as)"""
        ns, nl = C.c_uint32(), C.c_uint32()
        self._chk(self._L.mgx_batch_end(self._w, C.byref(ns), C.byref(nl)))
        return ns.value, nl.value

    def batch(self):
        """`with world.batch(): for ...: world.iterate(steps)` — the schedules are submitted together, merged into as few
        resident launches as their segments fit (include/mgx.h: mgx_batch_begin); same results, bit for bit"""
        return _Batch(self)

    def sweep(self, external_phases, internal_phases, n_internal=1, robot=-1, hints=0):
        self._chk(self._L.mgx_sweep(self._w, robot, external_phases, internal_phases, n_internal, hints))

    def internal_factor_iteration(self, robot=-1):
        self._chk(self._L.mgx_internal_factor_iteration(self._w, robot))

    def internal_variable_iteration(self, robot=-1):
        self._chk(self._L.mgx_internal_variable_iteration(self._w, robot))

    def external_factor_iteration(self, robot=-1):
        self._chk(self._L.mgx_external_factor_iteration(self._w, robot))

    def external_variable_iteration(self, robot=-1):
        self._chk(self._L.mgx_external_variable_iteration(self._w, robot))

    def change_prior(self, robot, var_ix, mean):
        mean = _f64(mean, (4,))
        self._chk(self._L.mgx_change_prior(self._w, robot, var_ix, _dp(mean)))

    def reset_variables(self, robot, means, first_last_sigma=1e30, inbetween_sigma=float("inf")):
        """FactorGraph::reset_variables (factorgraph.rs:1541-1564); the defaults are the reference's call (robot.rs:768)."""
        m = _f64(means).reshape(-1, 4)
        self._chk(self._L.mgx_reset_variables(self._w, robot, _dp(m), len(m), float(first_last_sigma), float(inbetween_sigma)))

    def reset_tracking_factors(self, robot):
        """FactorGraph::reset_tracking_factors (factorgraph.rs:1566-1590)."""
        self._chk(self._L.mgx_reset_tracking_factors(self._w, robot))

    def change_priors(self, robots, var_ix, means):
        robots = np.ascontiguousarray(robots, dtype=np.int32)
        var_ix = np.ascontiguousarray(var_ix, dtype=np.uint32)
        means = _f64(means, (len(robots), 4))
        self._chk(self._L.mgx_change_priors(self._w, len(robots), robots.ctypes.data_as(C.POINTER(C.c_int32)),
                                        var_ix.ctypes.data_as(C.POINTER(C.c_uint32)), _dp(means)))

    def update_priors(self, robots, waypoints_xy, time_scale, what, max_speed, delta_t):
        """update_prior_of_horizon_state (what & 1) / update_prior_of_current_state_v3 (what & 2) for the
        listed robots (robot.rs:2182-2338)."""
        robots = np.ascontiguousarray(robots, dtype=np.int32)
        n = len(robots)
        wp = _f64(waypoints_xy, (n, 2))
        ts = _f64(time_scale, (n,))
        what = np.ascontiguousarray(what, dtype=np.uint8)
        self._chk(self._L.mgx_update_priors(self._w, n, robots.ctypes.data_as(C.POINTER(C.c_int32)), _dp(wp), _dp(ts),
                                            what.ctypes.data_as(C.POINTER(C.c_uint8)), float(max_speed), float(delta_t)))

    def _arg(self, a, dtype, shape=None):
        """(keep-alive, address) of an array argument.  A driver hands the same arrays tick after tick: an array that is already
        what the C side reads (dtype, C-contiguous) is remembered with its address — preparing four pointers costs more host
        time than the call they go into (15 us of a 130 us tick)."""
        e = self._args.get(id(a))
        if e is not None and e[0] is a:
            return e
        b = np.ascontiguousarray(a, dtype=dtype)
        if shape is not None and b.shape != shape:
            raise ValueError(f"expected shape {shape}, got {b.shape}")
        e = (b, b.ctypes.data)
        if b is a:  # (held here: its id cannot be handed to another object, its buffer cannot be resized)
            if len(self._args) >= 16:
                self._args.pop(next(iter(self._args)))
            self._args[id(a)] = e
        return e

    def tick(self, robots, waypoints_xy, time_scale, what, max_speed, delta_t, steps):
        """One FixedUpdate tick of the planner chain (robot.rs:86-103): the prior updates of `update_priors`
        for the listed robots, then `iterate(steps)` — one C call (mgx_tick)."""
        r = self._arg(robots, np.int32)
        n = len(r[0])
        wp = self._arg(waypoints_xy, np.float64, (n, 2))
        ts = self._arg(time_scale, np.float64, (n,))
        wh = self._arg(what, np.uint8)
        key = steps if isinstance(steps, (bytes, tuple)) else tuple(steps)
        b = _STEP_BYTES.get(key)
        if b is None:
            if len(_STEP_BYTES) > 256:
                _STEP_BYTES.clear()
            b = _STEP_BYTES[key] = bytes(bytearray(int(s) for s in steps))
        self._chk(self._L.mgx_tick(self._w, n, r[1], wp[1], ts[1], wh[1], float(max_speed), float(delta_t), b, len(b)))

    # -- read-back -----------------------------------------------------------------------------
    def get_belief(self, robot, var_ix):
        eta, lam, mean, cov = np.zeros(4), np.zeros((4, 4)), np.zeros(4), np.zeros((4, 4))
        valid = C.c_int32()
        self._chk(self._L.mgx_get_belief(self._w, robot, var_ix, _dp(eta), _dp(lam), _dp(mean), _dp(cov), C.byref(valid)))
        return {"eta": eta, "lam": lam, "mean": mean, "cov": cov, "valid": bool(valid.value)}

    def num_robots(self):
        a, b = C.c_uint32(), C.c_uint32()
        self._chk(self._L.mgx_num_robots(self._w, C.byref(a), C.byref(b)))
        return a.value, b.value

    # -- whole driver ticks on the device (include/mgx.h, mgx_mission_*) ---------------------------------------------
    def mission_set(self, robot, waypoints_xy, reach_var, finish_var, reach_dist2, finish_dist2, translation, time_scale):
        wp = _f64(waypoints_xy).reshape(-1, 2)
        d = hostlib.MissionDesc(len(wp), 0, _dp(wp), int(reach_var), int(finish_var), float(reach_dist2), float(finish_dist2),
                                (C.c_float * 3)(*[float(x) for x in translation]), 0.0, float(time_scale))
        self._chk(self._L.mgx_mission_set(self._w, int(robot), C.byref(d)))

    def mission_tick(self, comms_radius, next_number, steps, max_speed, delta_t, despawn_finished=True, antennas=None,
                     method=hostlib.NEIGHBOURS_AUTO):
        """One FixedUpdate of the planner chain with the mission state on the device.  Returns
        (next robot number, connections created, pairs deleted, missions completed this tick)."""
        nn, st = C.c_uint64(int(next_number)), (C.c_uint32 * 3)()
        ant = None
        if antennas is not None:
            ant = np.ascontiguousarray(antennas, dtype=np.uint8)
        self._chk(self._L.mgx_mission_tick(self._w, float(comms_radius), int(method), C.byref(nn), 1 if despawn_finished else 0,
                                           None if ant is None else ant.ctypes.data, float(max_speed), float(delta_t),
                                           bytes(bytearray(steps)), len(steps), st))
        return nn.value, st[0], st[1], st[2]

    def mission_tick_begin(self, comms_radius, next_number, despawn_finished=True, method=hostlib.NEIGHBOURS_AUTO):
        """reached_waypoint + the topology pass (the tick's one synchronisation).  Returns (next robot number, connections
        created, pairs deleted, ids of the robots whose mission completed in this tick)."""
        nn, st = C.c_uint64(int(next_number)), (C.c_uint32 * 3)()
        self._chk(self._L.mgx_mission_tick_begin(self._w, float(comms_radius), int(method), C.byref(nn), 1 if despawn_finished else 0, st))
        fin = np.zeros(st[2], dtype=np.int32)
        if st[2]:
            k = C.c_uint32()
            self._chk(self._L.mgx_mission_finished(self._w, fin.ctypes.data, st[2], C.byref(k)))
        return nn.value, st[0], st[1], fin

    def mission_tick_end(self, steps, max_speed, delta_t, antennas=None):
        ant = None if antennas is None else np.ascontiguousarray(antennas, dtype=np.uint8)
        self._chk(self._L.mgx_mission_tick_end(self._w, None if ant is None else ant.ctypes.data, float(max_speed), float(delta_t),
                                               bytes(bytearray(steps)), len(steps)))

    def mission_run(self, n_ticks, comms_radius, next_number, steps, max_speed, delta_t, despawn_finished=True, method=hostlib.NEIGHBOURS_AUTO,
                    failure_rate=0.0, wyrand_state=None, stop_when_all_finished=False, want_translations=True, want_antennas=False):
        """Up to n_ticks whole driver ticks in ONE call (mgx_mission_run): per tick what mission_tick_begin / _end would have
        given.  Returns dict(ticks, next_number, wyrand_state, created [t], deleted [t], finished (list of id arrays per tick),
        translations [t, n, 3] f32, antennas [t, n] u8)."""
        n, _ = self.num_robots()
        T = int(n_ticks)
        d = hostlib.MissionRunDesc()
        created, deleted, nfin = (np.zeros(T, np.uint32) for _ in range(3))
        fin = np.zeros(max(n, 1), np.int32)
        tr = np.zeros((T, n, 3), np.float32) if want_translations else None
        ant = np.zeros((T, n), np.uint8) if want_antennas else None
        nn, ws = C.c_uint64(int(next_number)), C.c_uint64(0 if wyrand_state is None else int(wyrand_state))
        sb = bytes(bytearray(int(x) for x in steps))
        d.n_ticks, d.comms_radius, d.method, d.despawn_finished = T, float(comms_radius), int(method), 1 if despawn_finished else 0
        d.stop_when_all_finished, d.n_steps, d.steps = (1 if stop_when_all_finished else 0), len(sb), sb
        d.max_speed, d.delta_t, d.failure_rate = float(max_speed), float(delta_t), float(failure_rate)
        d.wyrand_state = None if wyrand_state is None else C.pointer(ws)
        d.robot_number_next = C.pointer(nn)
        u32p = C.POINTER(C.c_uint32)
        d.created, d.deleted, d.n_finished = created.ctypes.data_as(u32p), deleted.ctypes.data_as(u32p), nfin.ctypes.data_as(u32p)
        d.finished, d.finished_capacity = fin.ctypes.data_as(C.POINTER(C.c_int32)), len(fin)
        d.translations = None if tr is None else tr.ctypes.data_as(C.POINTER(C.c_float))
        d.antennas = None if ant is None else ant.ctypes.data_as(C.POINTER(C.c_uint8))
        self._chk(self._L.mgx_mission_run(self._w, C.addressof(d)))
        t = int(d.ticks_done)
        cuts = np.concatenate([[0], np.cumsum(nfin[:t])]).astype(int)
        return dict(ticks=t, next_number=int(nn.value), wyrand_state=None if wyrand_state is None else int(ws.value), created=created[:t],
                    deleted=deleted[:t], finished=[fin[cuts[i]:cuts[i + 1]].copy() for i in range(t)],
                    translations=None if tr is None else tr[:t], antennas=None if ant is None else ant[:t])

    def mission_translations(self):
        """Transforms [n, 3] as of the end of the last tick; valid once the stream has been synchronised since (no sync here)."""
        n, _ = self.num_robots()
        out, k = np.zeros((n, 3), np.float32), C.c_uint32()
        self._chk(self._L.mgx_mission_translations(self._w, out.ctypes.data, n, C.byref(k)))
        return out[:min(n, k.value)]

    def mission_read(self):
        """(Transform translations [n, 3] f32, next waypoint index per robot, completion tick per robot)"""
        n, _ = self.num_robots()
        tr, tg, fin = np.zeros((n, 3), np.float32), np.zeros(n, np.int32), np.zeros(n, np.int64)
        self._chk(self._L.mgx_mission_read(self._w, tr.ctypes.data, tg.ctypes.data, fin.ctypes.data))
        return tr, tg, fin

    def set_resident_launches(self, enabled):
        """per-world switch: False keeps every schedule on the launch-per-segment path (mgx_set_resident_launches)"""
        self._chk(self._L.mgx_set_resident_launches(self._w, 2 if enabled == "decline" else 1 if enabled else 0))

    def resident_outcome(self):
        """what became of the resident launch of the last iterate (hostlib.RESIDENT_NONE / _RAN / _DECLINED: issue it again)"""
        o = C.c_int32()
        self._chk(self._L.mgx_resident_outcome(self._w, C.byref(o)))
        return o.value

    def resident_ready(self, steps):
        """whether iterate(steps) would go out as a resident launch now (mgx_resident_ready)"""
        b, r = bytes(int(x) for x in steps), C.c_int32()
        self._chk(self._L.mgx_resident_ready(self._w, b, len(b), C.byref(r)))
        return bool(r.value)

    def resident_stats(self):
        """(resident launches so far, declined ones, what is left of the back-off after the last declined one)"""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint32()
        self._chk(self._L.mgx_resident_stats(self._w, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def is_thawing(self):
        """factors switched back on are still resuming from their frozen inboxes (mgx_is_thawing)"""
        t = C.c_int32()
        self._chk(self._L.mgx_is_thawing(self._w, C.byref(t)))
        return bool(t.value)

    def last_launch_count(self):
        """sweep-kernel launches of the last iterate / tick call (1: the whole schedule ran as one resident launch)"""
        n = C.c_uint32()
        self._chk(self._L.mgx_last_launch_count(self._w, C.byref(n)))
        return n.value

    def note_change_priors(self, robots, var_ix):
        """counters only: prior changes another rank applied to robots that are ghosts here (mgx_note_change_priors)"""
        r, v = np.ascontiguousarray(robots, dtype=np.int32), np.ascontiguousarray(var_ix, dtype=np.uint32)
        if len(r):
            self._chk(self._L.mgx_note_change_priors(self._w, len(r), r.ctypes.data_as(C.POINTER(C.c_int32)),
                                                     v.ctypes.data_as(C.POINTER(C.c_uint32))))

    def message_counts(self, robot):
        """(sent internal, sent external, received internal, received external) of one graph."""
        out = (C.c_uint64 * 4)()
        self._chk(self._L.mgx_message_counts(self._w, robot, out))
        return tuple(int(x) for x in out)

    def read_variable_means(self, var_ix):
        """Belief mean of one variable of every robot, [n_robots, 4]."""
        nr, _ = self.num_robots()
        out = np.zeros((nr, 4))
        self._chk(self._L.mgx_read_variable_means(self._w, var_ix, _dp(out)))
        return out

    def read_means(self):
        _, nv = self.num_robots()
        means = np.zeros((nv, 4))
        self._chk(self._L.mgx_read_means(self._w, _dp(means)))
        return means

    def read_beliefs(self):
        _, nv = self.num_robots()
        eta, lam, means = np.zeros((nv, 4)), np.zeros((nv, 4, 4)), np.zeros((nv, 4))
        self._chk(self._L.mgx_read_beliefs(self._w, _dp(eta), _dp(lam), _dp(means)))
        return eta, lam, means

    # -- halo ------------------------------------------------------------------------------------
    @staticmethod
    def halo_words(K):
        return hostlib.lib().mgx_halo_words(K)

    # halo exchange through RCCL inside the library (see include/mgx.h)
    def halo_rccl_connect(self, unique_id, n_ranks, rank, peer_rank, send_first, recv_first):
        a = np.ascontiguousarray(peer_rank, dtype=np.uint32)
        b = np.ascontiguousarray(send_first, dtype=np.uint32)
        c = np.ascontiguousarray(recv_first, dtype=np.uint32)
        assert b.size == c.size == (a.size + 1 if a.size else 0) or (a.size == 0)
        self._chk(self._L.mgx_halo_rccl_connect(self._w, unique_id, n_ranks, rank, a.size, a.ctypes.data, b.ctypes.data, c.ctypes.data))

    def halo_rccl_disconnect(self):
        self._chk(self._L.mgx_halo_rccl_disconnect(self._w))

    # direct halo exchange (peer-mapped stores; see include/mgx.h)
    def halo_direct_setup(self, n_sources):
        recv, flags = C.c_void_p(), C.c_void_p()
        self._chk(self._L.mgx_halo_direct_setup(self._w, n_sources, C.byref(recv), C.byref(flags)))
        return recv.value, flags.value

    def halo_direct_connect(self, send_first, peer_recv_base, peer_recv_records, peer_record_offset, peer_flag_slot):
        n = len(peer_recv_base)
        a = np.ascontiguousarray(send_first, dtype=np.uint32)
        b = np.ascontiguousarray(peer_recv_base, dtype=np.uint64)
        c = np.ascontiguousarray(peer_recv_records, dtype=np.uint64)
        d = np.ascontiguousarray(peer_record_offset, dtype=np.uint64)
        e = np.ascontiguousarray(peer_flag_slot, dtype=np.uint64)
        assert a.size == n + 1 and b.size == c.size == d.size == e.size == n
        self._chk(self._L.mgx_halo_direct_connect(self._w, n, a.ctypes.data, b.ctypes.data, c.ctypes.data, d.ctypes.data, e.ctypes.data))

    def halo_direct_setup_slots(self, n_sources, slot_capacity):
        """a receive area with one record slot per ghost robot (a wiring that outlives the exchange lists)"""
        recv, flags = C.c_void_p(), C.c_void_p()
        self._chk(self._L.mgx_halo_direct_setup_slots(self._w, n_sources, slot_capacity, C.byref(recv), C.byref(flags)))
        return recv.value, flags.value

    def halo_ghost_slots(self, robots):
        r = np.ascontiguousarray(robots, dtype=np.int32)
        out = np.zeros(r.size, dtype=np.int32)
        self._chk(self._L.mgx_halo_ghost_slots(self._w, r.size, r.ctypes.data, out.ctypes.data))
        return out

    def robot_export(self, robot):
        """the record of a robot this rank owns, for the rank that is to own it (mgx_robot_export)"""
        n = C.c_uint64()
        self._chk(self._L.mgx_robot_export(self._w, int(robot), None, 0, C.byref(n)))
        buf = np.zeros(n.value, dtype=np.uint8)
        self._chk(self._L.mgx_robot_export(self._w, int(robot), buf.ctypes.data, n.value, C.byref(n)))
        return buf.tobytes()

    def robot_import(self, robot, record):
        buf = np.frombuffer(record, dtype=np.uint8)
        self._chk(self._L.mgx_robot_import(self._w, int(robot), buf.ctypes.data, buf.size))

    def robot_release(self, robot):
        self._chk(self._L.mgx_robot_release(self._w, int(robot)))

    def halo_send_list(self):
        """robot ids of the send list as it stands (by consumer rank)"""
        ns, nr = C.c_uint32(), C.c_uint32()
        self._chk(self._L.mgx_halo_get_lists(self._w, None, 0, None, 0, C.byref(ns), C.byref(nr)))
        out = np.zeros(max(ns.value, 1), dtype=np.int32)
        self._chk(self._L.mgx_halo_get_lists(self._w, out.ctypes.data, ns.value, None, 0, C.byref(ns), C.byref(nr)))
        return [int(x) for x in out[:ns.value]]

    def halo_direct_connect_slots(self, send_first, peer_recv_base, peer_slot_capacity, entry_slot, peer_flag_slot):
        n = len(peer_recv_base)
        a = np.ascontiguousarray(send_first, dtype=np.uint32)
        b = np.ascontiguousarray(peer_recv_base, dtype=np.uint64)
        c = np.ascontiguousarray(peer_slot_capacity, dtype=np.uint64)
        d = np.ascontiguousarray(entry_slot, dtype=np.uint32)
        e = np.ascontiguousarray(peer_flag_slot, dtype=np.uint64)
        assert a.size == n + 1 and b.size == c.size == e.size == n and d.size == int(a[-1])
        self._chk(self._L.mgx_halo_direct_connect_slots(self._w, n, a.ctypes.data, b.ctypes.data, c.ctypes.data, d.ctypes.data, e.ctypes.data))

    def halo_direct_exchange(self, what=hostlib.HALO_PUSH | hostlib.HALO_WAIT):
        self._chk(self._L.mgx_halo_direct_exchange(self._w, what))

    def halo_direct_status(self):
        """Number of exchanges so far; raises if one of them timed out waiting for a peer."""
        n, bad = C.c_uint64(), C.c_uint64()
        self._chk(self._L.mgx_halo_direct_status(self._w, C.byref(n), C.byref(bad)))
        return n.value

    def halo_direct_disconnect(self):
        self._chk(self._L.mgx_halo_direct_disconnect(self._w))

    # resident schedule launches on sharded worlds (ghost records travel inside the launches; see include/mgx.h)
    def halo_resident_setup(self, n_recv):
        """Allocates this rank's ghost area.  Returns (area address, ghost slots, parity, segment count, slot of every entry of
        the receive list, eligible)."""
        area, ng, par, seg, ok = C.c_void_p(), C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_int32()
        slots = np.zeros(max(int(n_recv), 1), dtype=np.int32)
        self._chk(self._L.mgx_halo_resident_setup(self._w, C.byref(area), C.byref(ng), C.byref(par), C.byref(seg), slots.ctypes.data, C.byref(ok)))
        # (eligible: 1 yes; 2 everything but inter-robot factors — a world that follows its topology may get them later; 0 no)
        return area.value, ng.value, par.value, seg.value, slots[:n_recv].tolist(), int(ok.value)

    def halo_n_recv(self):
        ns, nr = C.c_uint32(), C.c_uint32()
        self._chk(self._L.mgx_halo_get_lists(self._w, None, 0, None, 0, C.byref(ns), C.byref(nr)))
        return nr.value

    def halo_resident_connect_peers(self, peer_area, peer_ghost_slots, peer_parity, peer_segment_count, coordinator_area=None, n_ranks=0):
        n = len(peer_area)
        b = np.ascontiguousarray(peer_area, dtype=np.uint64)
        c = np.ascontiguousarray(peer_ghost_slots, dtype=np.uint32)
        e = np.ascontiguousarray(peer_parity, dtype=np.uint32)
        f = np.ascontiguousarray(peer_segment_count, dtype=np.uint64)
        assert b.size == c.size == e.size == f.size == n
        self._chk(self._L.mgx_halo_resident_connect_peers(self._w, n, b.ctypes.data, c.ctypes.data, e.ctypes.data, f.ctypes.data,
                                                          coordinator_area, int(n_ranks)))

    def halo_resident_aim(self, robots, peer_index, peer_slot):
        a = np.ascontiguousarray(robots, dtype=np.int32)
        b = np.ascontiguousarray(peer_index, dtype=np.uint32)
        c = np.ascontiguousarray(peer_slot, dtype=np.uint32)
        assert a.size == b.size == c.size
        self._chk(self._L.mgx_halo_resident_aim(self._w, a.size, a.ctypes.data, b.ctypes.data, c.ctypes.data))

    def halo_resident_connect(self, robots, peer_area, peer_ghost_slots, peer_slot, peer_parity, peer_segment_count,
                              coordinator_area=None, n_ranks=0):
        """coordinator_area: rank 0's ghost area as this rank maps it — where the ranks agree on every schedule's launches"""
        n = len(robots)
        a = np.ascontiguousarray(robots, dtype=np.int32)
        b = np.ascontiguousarray(peer_area, dtype=np.uint64)
        c = np.ascontiguousarray(peer_ghost_slots, dtype=np.uint32)
        d = np.ascontiguousarray(peer_slot, dtype=np.uint32)
        e = np.ascontiguousarray(peer_parity, dtype=np.uint32)
        f = np.ascontiguousarray(peer_segment_count, dtype=np.uint64)
        assert b.size == c.size == d.size == e.size == f.size == n
        self._chk(self._L.mgx_halo_resident_connect(self._w, n, a.ctypes.data, b.ctypes.data, c.ctypes.data, d.ctypes.data, e.ctypes.data,
                                                    f.ctypes.data, coordinator_area, int(n_ranks)))

    def halo_resident_disconnect(self):
        self._chk(self._L.mgx_halo_resident_disconnect(self._w))

    def halo_plan(self, send_robots, recv_ghosts):
        a = np.ascontiguousarray(send_robots, dtype=np.int32)
        b = np.ascontiguousarray(recv_ghosts, dtype=np.int32)
        ip = C.POINTER(C.c_int32)
        self._chk(self._L.mgx_halo_plan(self._w, len(a), a.ctypes.data_as(ip), len(b), b.ctypes.data_as(ip)))

    def halo_plan_from_connections(self, rank_of, my_rank, n_ranks):
        """Exchange lists derived from the connections held (mgx_halo_plan_from_connections):
        returns (send_counts, recv_counts) in records per peer rank."""
        rank_of = np.ascontiguousarray(rank_of, dtype=np.int32)
        sc, rc = (C.c_uint32 * n_ranks)(), (C.c_uint32 * n_ranks)()
        self._chk(self._L.mgx_halo_plan_from_connections(self._w, rank_of.ctypes.data_as(C.POINTER(C.c_int32)), rank_of.size,
                                                         int(my_rank), int(n_ranks), sc, rc))
        return list(sc), list(rc)

    def halo_pack(self, dev_ptr):
        self._chk(self._L.mgx_halo_pack(self._w, C.c_void_p(dev_ptr)))

    def halo_unpack(self, dev_ptr):
        self._chk(self._L.mgx_halo_unpack(self._w, C.c_void_p(dev_ptr)))

    def factorgraph(self, robot):
        return FactorGraph(self, robot)


class FactorGraph:
    """Per-robot handle with the method names of the reference ``FactorGraph``
    (factorgraph.rs:688-826,494-528)."""

    def __init__(self, world, robot_id):
        self.world, self.id = world, robot_id

    def internal_factor_iteration(self):
        self.world.internal_factor_iteration(self.id)

    def internal_variable_iteration(self):
        self.world.internal_variable_iteration(self.id)

    def change_prior_of_variable(self, variable_index, new_mean):
        self.world.change_prior(self.id, variable_index, new_mean)
        return []  # routing to external factors happens on the device

    def variable(self, variable_index):
        return self.world.get_belief(self.id, variable_index)


_STEP_BYTES = {}


class _Batch:
    def __init__(self, world):
        self.world, self.schedules, self.launches = world, 0, 0

    def __enter__(self):
        self.world.batch_begin()
        return self

    def __exit__(self, exc_type, exc, tb):
        self.schedules, self.launches = self.world.batch_end()
        return False
