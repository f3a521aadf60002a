"""Headless run of a reference scenario directory (config.toml + environment.yaml + formation.yaml,
unmodified) on a World-like backend — the HIP engine (`magics_amd.World`) in the product, the CPU
oracle in the tests.  One `tick()` is one FixedUpdate of the reference at `simulation.hz`, preceded
by the frame's Update systems that matter to the path (the formation spawners):

    advance_time + spawn_formation          crates/magics/src/planner/spawner.rs:386-649   (Update)
    reached_waypoint                        planner/robot.rs:2080-2176                     \\
    update_robot_neighbours / delete_ / create_interrobot_factors   robot.rs:1362-1586      |
    update_failed_comms                     robot.rs:1593-1601                               > FixedUpdate
    update_prior_of_horizon_state / update_prior_of_current_state_v3   robot.rs:2182-2338   |
    iterate_gbp_v2                          robot.rs:1769-1861                              /
    export                                  crates/magics/src/export.rs:249-262,283-620

The reference runs its Update systems at the display's frame rate against virtual time, so the
interleaving of spawns with fixed ticks is not reproducible there; here a frame is exactly one
fixed tick long.  Rendering, picking, RRT* planning (`planning-strategy: rrt-star` is rejected),
collision bookkeeping with parry2d and goal areas are outside the path (SURVEY §8 out of scope).
"""
import json

import numpy as np

from . import config as _config
from . import environment as _environment
from . import hostlib, spawner
from .prng import WyRand
from .scenarios import robot_initial_state

F = np.float32


class Simulation:
    def __init__(self, scenario, world, neighbours_method=hostlib.NEIGHBOURS_AUTO, device_missions=None):
        """scenario: `config.load_scenario(dir)` (or a dict of the same shape); world: a fresh
        World-like object created with `config.world_params(scenario["config"])`.
        device_missions (default: whenever the world offers them, i.e. the engine): routes, reached-when rules and
        Transforms live on the device (include/mgx.h, mgx_mission_*) and a tick is two calls around the comms draws with
        ONE synchronisation and no belief read-back; otherwise this loop does all of it on the host (the CPU oracle's
        path, and the checker of the other one: tests/test_gpu_sim.py)."""
        self.dev = hasattr(world, "mission_tick_begin") if device_missions is None else bool(device_missions)
        self._pending_track = None
        self._trk_log = []  # what _track noted since _tracks last ran
        self.name = scenario.get("name", "")
        self.cfg, self.env, self.formations = scenario["config"], scenario["environment"], scenario["formation"]["formations"]
        self.w = world
        sim, gbp = self.cfg["simulation"], self.cfg["gbp"]
        self.hz = sim["hz"]
        self.dt_ns = int(round(1e9 / self.hz))                   # Time::<Fixed>::from_hz
        self.dt32 = F(self.dt_ns * 1e-9)                         # Time<Fixed>::delta_seconds
        self.rng = WyRand(sim["prng-seed"])                      # simulation_loader.rs:651-652
        self.world_dims = _environment.world_size(self.env)      # spawner.rs:437-442
        sch = gbp["iteration-schedule"]
        self.steps = hostlib.schedule(_config.SCHEDULE_KINDS[sch["schedule"]], sch["internal"], sch["external"])
        self.comms_radius = self.cfg["robot"]["communication"]["radius"]
        self.failure_rate = float(F(self.cfg["robot"]["communication"]["failure-rate"]))
        self.max_speed = self.cfg["robot"]["target-speed"]
        self.despawn = sim["despawn-robot-when-final-waypoint-reached"]
        self.method = neighbours_method
        world.set_environment(self.env)                          # simulation_loader.rs:154-162
        self.spawners = [spawner.FormationSpawner(i, f) for i, f in enumerate(self.formations)]
        self.robots = []           # per robot: dict (see _add_robot)
        self._translation = np.zeros((0, 3), dtype=F)
        self.tick_no, self.next_number, self.K = 0, 1, None
        self.events = []           # (tick, connections created, pairs deleted)
        self.collisions = {}       # (robot a, robot b), a < b -> {"colliding": bool, "times": int, "aabbs": [...]}
        self.entities = spawner.EntityAllocator()  # the robots' Entity bits = their graphs' order (id.rs:19-54)

    # -- spawn_formation (spawner.rs:415-649) + RobotBundle::new (robot.rs:1134-1356) -------------------
    def _add_robot(self, desc, formation_index):
        if desc["planning-strategy"] != "only-local":
            raise NotImplementedError("planning-strategy rrt-star: the global planner is outside the hot path")
        rb = self.cfg["robot"]
        states = desc["waypoints"]
        mean0, prior, dt = robot_initial_state(states[0], states[1], desc["timesteps"], desc["radius"], rb["target-speed"], rb["planning-horizon"])
        if self.K is None:
            self.K = len(desc["timesteps"])
        path = np.array([s[:2] for s in states], dtype=np.float32)  # the route's positions (robot.rs:1310-1315)
        key = self.entities.alloc()
        rid = self.w.add_robot(mean0, prior, dt, float(desc["radius"]), path=path, order_key=key)
        assert rid == len(self.robots)
        t0 = F(desc["radius"]) / F(2.0) / F(rb["target-speed"])
        self.robots.append({"id": rid, "formation": formation_index, "radius": F(desc["radius"]), "waypoints": [s.copy() for s in states],
                            "target": 1, "alive": True, "completed": False, "started_at": self.elapsed(), "finished_at": None,
                            "time_scale": float(self.dt32 / t0), "colour": desc["colour"], "rng": desc["rng"],
                            "strategy": desc["planning-strategy"], "reach": desc["waypoint-reached-when-intersects"],
                            "finish": desc["finished-when-intersects"], "positions": [], "velocities": [], "travelled": 0.0,
                            "trk_elapsed": 0, "trk_prev": None, "entity": key})
        self._translation = np.vstack([self._translation, np.array([[states[0][0], -1.5, states[0][1]]], dtype=F)])  # spawner.rs:548
        if self.dev:
            me = self.robots[-1]

            def rule(when):
                kind, n = when["intersects-with"]
                var = 0 if kind == "current" else (self.K - 1 if kind == "horizon" or n >= self.K else n)
                dkind, meter = when["distance"]
                lim = me["radius"] * me["radius"] if dkind == "robot-radius" else F(meter) * F(meter)
                return var, float(lim)
            (rv, rd), (fv, fd) = rule(me["reach"]), rule(me["finish"])
            self.w.mission_set(rid, np.array([s_[:2] for s_ in states[1:]], dtype=np.float64), rv, fv, rd, fd,
                               self._translation[-1], me["time_scale"])

    @property
    def translation(self):
        """Transform::translation of every robot [n, 3] f32 (fetched from the device when the missions live there)"""
        if self.dev and self.robots:
            tr = self.w.mission_read()[0]
            self._translation[:len(tr)] = tr[:len(self._translation)]
        return self._translation

    def _spawn(self):
        for sp in self.spawners:
            sp.tick(self.dt_ns)
            if sp.ready_to_spawn():
                sp.spawn()
                descs = spawner.spawn_formation(self.formations[sp.index], self.cfg, self.world_dims, self.rng)
                for d in descs or []:  # None: "failed to spawn formation", the reference logs and skips
                    self._add_robot(d, sp.index)

    def elapsed(self):
        return self.tick_no * self.dt_ns * 1e-9

    # -- reached_waypoint (robot.rs:2080-2176) --------------------------------------------------------------
    def _reached_waypoint(self):
        todo = [r for r in self.robots if r["alive"] and not r["completed"]]
        if not todo:
            return
        means = {}

        def mean_of(var):
            if var not in means:
                means[var] = self.w.read_variable_means(var)
            return means[var]
        for r in todo:
            last = r["target"] == len(r["waypoints"]) - 1
            when = r["finish"] if last else r["reach"]
            kind, n = when["intersects-with"]
            var = 0 if kind == "current" else (self.K - 1 if kind == "horizon" or n >= self.K else n)
            est = mean_of(var)[r["id"], :2].astype(F)
            dkind, meter = when["distance"]
            limit = r["radius"] * r["radius"] if dkind == "robot-radius" else F(meter) * F(meter)
            d = est - r["waypoints"][r["target"]][:2]
            if F(d[0] * d[0] + d[1] * d[1]) < limit:
                r["target"] += 1                                   # Route::advance
                if r["target"] >= len(r["waypoints"]):             # Mission::next_route: the only route is done
                    r["completed"], r["finished_at"] = True, self.elapsed()
                    if self.despawn:
                        self.w.remove_robot(r["id"])
                        self.entities.free(r["entity"])
                        r["alive"] = False

    # PositionTracker / VelocityTracker (planner/tracking.rs:104-122,189-218; 100 ms timers, spawner.rs:620-621):
    # FixedUpdate systems over Changed<Transform>, i.e. the robots that moved this tick; sampled after the move
    def _track(self, moving, translation, now):
        """track_robots (export.rs / tracking: a sample of every moving robot's Transform every 100 ms of simulated time).  The tick
        only NOTES what the samples are taken from — the moving robots' ids and their rows of the Transforms; the per-robot lists
        the export holds (positions, velocities with their measuring spans) are made from the notes when somebody reads them
        (_tracks): a dictionary per robot per tick was a fifth of a scenario's wall time."""
        if moving:
            ids = np.fromiter((r["id"] for r in moving), dtype=np.int64, count=len(moving))
            self._trk_log.append((ids, np.asarray(translation)[ids].copy(), now))

    def _tracks(self):
        """bring every robot's positions / velocities up to the last tick noted by _track"""
        log, self._trk_log = self._trk_log, []
        for ids, rows, now in log:
            for k, rid in enumerate(ids):
                r = self.robots[int(rid)]
                r["trk_elapsed"] += self.dt_ns
                if r["trk_elapsed"] < 100_000_000:
                    continue
                r["trk_elapsed"] %= 100_000_000
                pos = rows[k]
                r["positions"].append([float(pos[0]), float(pos[2])])
                if r["trk_prev"] is not None:
                    dt = now - r["trk_prev"][1]
                    v = (pos - r["trk_prev"][0]) / F(dt)
                    r["velocities"].append({"velocity": [float(v[0]), float(v[1]), float(v[2])], "timestamp": now,
                                            "measured_over": {"secs": int(dt), "nanos": int(round((dt - int(dt)) * 1e9))}})
                r["trk_prev"] = (pos, now)

    # update_robot_robot_collisions (planner/collisions.rs:72-140, FixedUpdate): every pair of live robots, bounding
    # spheres of their Ball(radius) at the Transform's (x, z) — parry2d 0.13 BoundingSphere::intersects, restated from its
    # published source (third party, absent from /root/reference): |c_b - c_a|^2 <= (r_a + r_b)^2 in f32 — through the
    # Free / Colliding state machine of CollisionHistory (:455-495): a Free -> Colliding edge is one collision, recorded with
    # the intersection of the two balls' AABBs (:113-119).  Robot - environment collisions need the colliders of the
    # reference's 3-D map generator (environment/map_generator.rs) and parry2d's shape queries: not built (counted 0).
    def _collide(self, alive, translation):
        if len(alive) < 2:
            return
        ids = np.array([r["id"] for r in alive])
        rad = np.array([r["radius"] for r in alive], dtype=F)
        pos = np.ascontiguousarray(translation[ids][:, [0, 2]], dtype=F)
        d = pos[None, :, :] - pos[:, None, :]                        # d[i, j] = c_j - c_i (f32)
        rs = rad[:, None] + rad[None, :]
        hit = np.triu(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] <= rs * rs, k=1)
        now = {(int(ids[i]), int(ids[j])): (i, j) for i, j in zip(*np.nonzero(hit))}
        for key, h in self.collisions.items():                      # Colliding -> Free
            if h["colliding"] and key not in now:
                h["colliding"] = False
        for key, (i, j) in now.items():                             # Free -> Colliding: one collision, with the AABBs' intersection
            h = self.collisions.setdefault(key, {"colliding": False, "times": 0, "aabbs": []})
            if not h["colliding"]:
                h["colliding"] = True
                h["times"] += 1
                lo = np.maximum(pos[i] - rad[i], pos[j] - rad[j])
                hi = np.minimum(pos[i] + rad[i], pos[j] + rad[j])
                h["aabbs"].append({"mins": [float(lo[0]), float(lo[1])], "maxs": [float(hi[0]), float(hi[1])]})

    def _flush_trackers(self, synchronise=False):
        """device missions: the samples of the last tick, taken from the Transforms that tick sent to the host behind its
        launches (complete after the next synchronisation — the next tick's own, or an explicit one here)"""
        if self._pending_track is None:
            return
        if synchronise:
            self.w.synchronize()
        moving, now, alive = self._pending_track
        self._pending_track = None
        tr = self.w.mission_translations()
        self._track(moving, tr, now)
        self._collide(alive, tr)

    def _tick_device(self):
        w = self.w
        self.next_number, created, deleted, fin = w.mission_tick_begin(self.comms_radius, self.next_number, despawn_finished=self.despawn,
                                                                       method=self.method)
        self._flush_trackers()
        if created or deleted:
            self.events.append((self.tick_no, created, deleted))
        for rid in fin:                                            # reached_waypoint completed the mission (robot.rs:2150-2176)
            r = self.robots[int(rid)]
            r["target"] = len(r["waypoints"])
            r["completed"], r["finished_at"] = True, self.elapsed()
            if self.despawn:
                r["alive"] = False
                self.entities.free(r["entity"])
        live = [r for r in self.robots if r["alive"]]
        ant = None
        if live:
            active = np.array([not self.rng.gen_bool(self.failure_rate) for _ in live], dtype=np.uint8)  # robot.rs:1599
            if self.failure_rate > 0.0:
                ant = np.ones(len(self.robots), dtype=np.uint8)
                ant[[r["id"] for r in live]] = active
        moving = [r for r in live if not r["completed"]]
        w.mission_tick_end(self.steps, float(self.max_speed), float(self.dt32), antennas=ant)
        self._pending_track = (moving, (self.tick_no + 1) * self.dt_ns * 1e-9, live)

    def tick(self):
        w = self.w
        self._spawn()
        if self.robots and self.dev:
            self._tick_device()
        elif self.robots:
            self._reached_waypoint()
            self.next_number, created, deleted = w.update_topology(self._translation, self.comms_radius, self.next_number, method=self.method)
            if created or deleted:
                self.events.append((self.tick_no, created, deleted))
            live = [r for r in self.robots if r["alive"]]
            if live:
                active = np.array([not self.rng.gen_bool(self.failure_rate) for _ in live], dtype=np.uint8)  # robot.rs:1599
                if self.failure_rate > 0.0:
                    w.set_antennas(np.array([r["id"] for r in live], dtype=np.int32), active)
            moving = [r for r in live if not r["completed"]]
            if moving:
                ids = np.array([r["id"] for r in moving], dtype=np.int32)
                m0, m1 = w.read_variable_means(0), w.read_variable_means(1)
                ts = np.array([r["time_scale"] for r in moving])
                change = ts[:, None] * (m1[ids] - m0[ids])                                   # robot.rs:2314
                self._translation[ids, 0] += change[:, 0].astype(F)                          # robot.rs:2328-2329
                self._translation[ids, 2] += change[:, 1].astype(F)
                for r, c in zip(moving, change):
                    r["travelled"] += float(np.hypot(c[0], c[1]))
                w.tick(robots=ids, waypoints_xy=np.array([r["waypoints"][r["target"]][:2] for r in moving], dtype=np.float64),
                       time_scale=ts, what=np.full(len(moving), 3, dtype=np.uint8), max_speed=float(self.max_speed),
                       delta_t=float(self.dt32), steps=self.steps)                          # prior updates + iterate_gbp_v2
            else:
                w.iterate(self.steps)
            self._track(moving, self._translation, (self.tick_no + 1) * self.dt_ns * 1e-9)
            self._collide(live, self._translation)
        self.tick_no += 1

    def finished(self):
        """AllFormationsFinished: every spawner is exhausted and every spawned robot completed its mission."""
        return all(sp.exhausted() for sp in self.spawners) and all(r["completed"] for r in self.robots)

    # -- many ticks per call (device missions): what lies between two spawns is ONE mgx_mission_run ------------------
    def _quiet_ticks(self, cap):
        """how many ticks, the coming one included, pass before a spawner acts again (becomes ready, or runs out): the
        spawners' timers run ahead on copies"""
        sps = [sp.clone() for sp in self.spawners]
        n = 0
        while n < cap:
            was = [sp.exhausted() for sp in sps]
            for sp in sps:
                sp.tick(self.dt_ns)
            if any(sp.ready_to_spawn() for sp in sps) or [sp.exhausted() for sp in sps] != was:
                break
            n += 1
        return n

    def _run_chunk(self, cap):
        """One frame's Update (the spawners) and then every FixedUpdate up to the next frame in which a spawner acts, in one
        engine call: identical to tick() that many times — the per-tick bookkeeping (events, completions, trackers,
        collisions) is replayed from what the call returns per tick."""
        self._spawn()
        if not self.robots:
            self.tick_no += 1
            return
        self._flush_trackers(synchronise=True)
        m = 1 + (self._quiet_ticks(cap - 1) if cap > 1 else 0)
        for _ in range(m - 1):  # the real spawners live through the same frames (nothing happens in them)
            for sp in self.spawners:
                sp.tick(self.dt_ns)
        out = self.w.mission_run(m, self.comms_radius, self.next_number, self.steps, float(self.max_speed), float(self.dt32),
                                 despawn_finished=self.despawn, method=self.method, failure_rate=self.failure_rate,
                                 wyrand_state=self.rng.state, stop_when_all_finished=all(sp.exhausted() for sp in self.spawners))
        self.next_number, self.rng.state = out["next_number"], out["wyrand_state"]
        for j in range(out["ticks"]):
            if out["created"][j] or out["deleted"][j]:
                self.events.append((self.tick_no, int(out["created"][j]), int(out["deleted"][j])))
            for rid in out["finished"][j]:
                r = self.robots[int(rid)]
                r["target"] = len(r["waypoints"])
                r["completed"], r["finished_at"] = True, self.elapsed()
                if self.despawn:
                    r["alive"] = False
                    self.entities.free(r["entity"])
            live = [r for r in self.robots if r["alive"]]
            moving = [r for r in live if not r["completed"]]
            tr = out["translations"][j]
            self._track(moving, tr, (self.tick_no + 1) * self.dt_ns * 1e-9)
            self._collide(live, tr)
            self.tick_no += 1

    def run(self, max_ticks=None, max_time=None, chunk=256):
        """chunk: device missions — up to that many ticks per engine call (mgx_mission_run) where no spawner acts; 1: tick by tick"""
        limit = self.cfg["simulation"]["max-time"] if max_time is None else max_time
        while not self.finished() and self.elapsed() < limit and (max_ticks is None or self.tick_no < max_ticks):
            if self.dev and chunk > 1 and hasattr(self.w, "mission_run"):
                cap = 0  # ticks the loop's own conditions allow from here
                while cap < chunk and (self.tick_no + cap) * self.dt_ns * 1e-9 < limit and (max_ticks is None or self.tick_no + cap < max_ticks):
                    cap += 1
                self._run_chunk(max(cap, 1))
            else:
                self.tick()
        if self.dev:
            self._flush_trackers(synchronise=True)
        return self

    # -- export (export.rs:249-262, 283-620) ------------------------------------------------------------------
    def export(self):
        if self.dev:
            self._flush_trackers(synchronise=True)
        self._tracks()
        sch = self.cfg["gbp"]["iteration-schedule"]
        robots = {}
        for r in self.robots:
            sent_i, sent_e, recv_i, recv_e = self.w.message_counts(r["id"])
            wps = [[float(s[0]), float(s[1])] for s in r["waypoints"]]
            fin = r["finished_at"] if r["finished_at"] is not None else self.elapsed()
            robots[str(r["id"])] = {
                "radius": float(r["radius"]), "positions": r["positions"], "velocities": r["velocities"],
                "collisions": {"robots": sum(h["times"] for k, h in self.collisions.items() if r["id"] in k), "environment": 0},
                "messages": {"sent": {"internal": sent_i, "external": sent_e}, "received": {"internal": recv_i, "external": recv_e}},
                "mission": {"waypoints": [wps[0], wps[-1]], "started_at": r["started_at"], "finished_at": fin,
                            "routes": [{"waypoints": wps, "started_at": r["started_at"], "finished_at": fin}]},
                "planning_strategy": r["strategy"], "color": r["colour"]}
        return {"scenario": self.name, "makespan": self.elapsed(), "delta_t": float(self.dt32),
                "gbp": {"iterations": {"internal": sch["internal"], "external": sch["external"]}}, "robots": robots,
                "prng_seed": self.cfg["simulation"]["prng-seed"], "config": self.cfg, "obstacles": {},
                "collisions": {"robots": [{"robot_a": a, "robot_b": b, "aabbs": h["aabbs"]} for (a, b), h in sorted(self.collisions.items())],
                               "environment": []},
                # goal_areas: the reference registers GoalAreaPlugin but its only system that SPAWNS goal areas
                # (setup_goal_areas_for_junction_scenario, goal_area.rs:105-119) is commented out of the plugin (:8-11):
                # the exported map is empty in the reference itself
                "goal_areas": {}}

    def export_json(self, path):
        with open(path, "w", encoding="utf-8") as f:
            json.dump(self.export(), f)


def run_scenario(directory, world_factory, max_ticks=None, max_time=None):
    """Load a scenario directory and run it to the end (or the given bound)."""
    sc = _config.load_scenario(directory)
    sim = Simulation(sc, world_factory(_config.world_params(sc["config"])))
    return sim.run(max_ticks=max_ticks, max_time=max_time)
