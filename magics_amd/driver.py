"""Headless mirror of the reference's per-tick system chain (crates/magics/src/planner/robot.rs:
86-103): what a host does around the engine every FixedUpdate tick.  Works with any World-like
object (the HIP engine or the CPU oracle), which is how the tests check whole missions.

    reached_waypoint                         robot.rs:2080-2176  (FixedUpdate, not ordered against the chain)
    update_robot_neighbours                  robot.rs:1362-1384  \\
    delete_interrobot_factors                robot.rs:1386-1439   > World.update_topology
    create_interrobot_factors                robot.rs:1441-1586  /
    update_failed_comms                      robot.rs:1593-1601  World.set_antennas (draws injected)
    update_prior_of_horizon_state            robot.rs:2182-2283  \\ World.update_priors
    update_prior_of_current_state_v3         robot.rs:2286-2338  /  (+ the Transform increment, here)
    iterate_gbp_v2                           robot.rs:1769-1861  World.iterate

Missions are plain waypoint lists (the reference's `Mission` with one route, planning strategy
"only-local"); global planning, spawners, timers and rendering stay outside.
"""
import numpy as np

F32 = np.float32


class Driver:
    def __init__(self, world, n_robots, K, waypoints, radii, t0, steps, comms_radius, target_speed, hz=10.0,
                 height=0.5, despawn_when_finished=True, failure_draws=None):
        """waypoints[r]: list of (x, y) still to visit (the first one is the next waypoint);
        radii[r]: robot radius (reached-when-intersects distance, formation.yaml `robot-radius`);
        t0[r]: the robot's T0 component (f32); failure_draws: callable(tick, n) -> bool array of
        antennas that stay ON this tick (None: no comms failures)."""
        self.w, self.n, self.K = world, n_robots, K
        self.way = [list(map(tuple, wp)) for wp in waypoints]
        self.radii = np.asarray(radii, dtype=np.float64)
        self.dt32 = F32(1.0) / F32(hz)                                   # Time<Fixed>::delta_seconds
        self.time_scale = np.array([float(self.dt32 / F32(t)) for t in t0])   # robot.rs:2309 (f32 quotient)
        self.steps, self.comms_radius = steps, float(comms_radius)
        self.max_speed, self.delta_t = float(F32(target_speed)), float(self.dt32)
        self.despawn, self.failure_draws = despawn_when_finished, failure_draws
        cur = world.read_variable_means(0)
        # Transform::translation (f32; Bevy's y is up, the plane is x-z)
        self.translation = np.stack([cur[:, 0], np.full(n_robots, height), cur[:, 1]], axis=1).astype(F32)
        self.alive = np.ones(n_robots, dtype=bool)
        self.finished_at = np.full(n_robots, -1, dtype=np.int64)
        self.next_number, self.tick_no = 1, 0
        self.travelled = np.zeros(n_robots)

    # reached_waypoint (robot.rs:2080-2176): intermediate waypoints are checked against the horizon
    # variable, the last one against the current variable (formation.yaml of the Circle Experiment)
    def _reached_waypoint(self):
        todo = [r for r in range(self.n) if self.alive[r] and self.way[r]]
        if not todo:
            return
        horizon, current = self.w.read_variable_means(self.K - 1), self.w.read_variable_means(0)
        for r in todo:
            last = len(self.way[r]) == 1
            est = (current if last else horizon)[r, :2].astype(F32)           # estimated_position_vec2 (f32)
            wp = np.array(self.way[r][0], dtype=F32)
            d = est - wp
            if F32(d[0] * d[0] + d[1] * d[1]) < F32(self.radii[r]) * F32(self.radii[r]):
                self.way[r].pop(0)
                if not self.way[r]:
                    self.finished_at[r] = self.tick_no
                    if self.despawn:
                        self.w.remove_robot(r)
                        self.alive[r] = False

    def tick(self):
        w = self.w
        self._reached_waypoint()
        self.next_number, created, deleted = w.update_topology(self.translation, self.comms_radius, self.next_number)
        live = np.nonzero(self.alive)[0]
        if self.failure_draws is not None and len(live):
            w.set_antennas(live.astype(np.int32), self.failure_draws(self.tick_no, len(live)))
        moving = np.array([r for r in live if self.way[r]], dtype=np.int32)   # a next waypoint exists (robot.rs:2216-2228)
        if len(moving):
            m0, m1 = w.read_variable_means(0), w.read_variable_means(1)
            change = self.time_scale[moving, None] * (m1[moving] - m0[moving])   # change_in_state (robot.rs:2314)
            args = dict(robots=moving, waypoints_xy=np.array([self.way[r][0] for r in moving], dtype=np.float64),
                        time_scale=self.time_scale[moving], what=np.full(len(moving), 3, dtype=np.uint8),
                        max_speed=self.max_speed, delta_t=self.delta_t)
            self.translation[moving, 0] += change[:, 0].astype(F32)             # robot.rs:2328-2329
            self.translation[moving, 2] += change[:, 1].astype(F32)
            self.travelled[moving] += np.sqrt(change[:, 0] * change[:, 0] + change[:, 1] * change[:, 1])
            if hasattr(w, "tick"):
                w.tick(steps=self.steps, **args)                                 # prior updates + schedule, one call
            else:
                w.update_priors(**args)
                w.iterate(self.steps)
        else:
            w.iterate(self.steps)
        self.tick_no += 1
        return created, deleted

    def run(self, max_ticks):
        """Ticks until every robot has finished (or max_ticks).  Returns the summary the reference's
        export carries per robot (export.rs:249-262: makespan, distance travelled, message counts)."""
        while self.tick_no < max_ticks and (self.finished_at < 0).any():
            self.tick()
        return self.summary()

    def summary(self):
        done = self.finished_at >= 0
        return {"ticks": self.tick_no, "finished": int(done.sum()), "makespan_s": float(self.finished_at.max() * self.delta_t) if done.all() else None,
                "finished_at_tick": self.finished_at.tolist(), "distance_travelled": self.travelled.tolist(),
                "messages": [self.w.message_counts(r) for r in range(self.n)]}


class DeviceDriver:
    """The same chain with the mission state ON THE DEVICE (include/mgx.h, mgx_mission_*): routes, next waypoints,
    reached-when rules and Transforms are handed over once, and a tick is ONE call — reached_waypoint, the topology
    pass on the device's Transforms, the comms draws, both prior updates with the Transform increment and the GBP
    schedule — with one synchronisation inside (the neighbour rows of the connection bookkeeping) and no belief
    read-back at all.  Engine worlds only; `Driver` above on the CPU oracle is its checker (tests/test_gpu_driver.py)."""

    def __init__(self, world, n_robots, K, waypoints, radii, t0, steps, comms_radius, target_speed, hz=10.0,
                 height=0.5, despawn_when_finished=True, failure_draws=None):
        self.w, self.n, self.K = world, n_robots, K
        self.dt32 = F32(1.0) / F32(hz)
        self.steps, self.comms_radius = steps, float(comms_radius)
        self.max_speed, self.delta_t = float(F32(target_speed)), float(self.dt32)
        self.despawn, self.failure_draws = despawn_when_finished, failure_draws
        cur = world.read_variable_means(0)
        for r in range(n_robots):
            r2 = float(F32(radii[r]) * F32(radii[r]))
            # intermediate waypoints against the horizon variable, the last one against the current variable, both
            # within the robot's radius (the Circle Experiment's formation.yaml)
            world.mission_set(r, np.asarray(waypoints[r], dtype=np.float64), K - 1, 0, r2, r2,
                              (F32(cur[r, 0]), F32(height), F32(cur[r, 1])), float(self.dt32 / F32(t0[r])))
        self.n_way = np.array([len(wp) for wp in waypoints])
        self.next_number, self.tick_no = 1, 0
        self._alive = np.ones(n_robots, dtype=bool)

    def tick(self):
        if self.failure_draws is None:
            self.next_number, created, deleted, finished = self.w.mission_tick(
                self.comms_radius, self.next_number, self.steps, self.max_speed, self.delta_t, despawn_finished=self.despawn, antennas=None)
        else:
            # the reference draws one value per robot alive AFTER this tick's despawns, in id order (robot.rs:1593-1601 runs
            # behind the despawn of the robots that reached their last waypoint): the tick's first half says who completed,
            # the draws are made for the robots that are left, the second half applies them (as sim.py does)
            self.next_number, created, deleted, fin = self.w.mission_tick_begin(self.comms_radius, self.next_number,
                                                                                despawn_finished=self.despawn)
            if self.despawn and len(fin):
                self._alive[np.asarray(fin, dtype=np.int64)] = False
            live = np.nonzero(self._alive)[0]
            d = np.asarray(self.failure_draws(self.tick_no, len(live)), dtype=np.uint8)
            ant = np.ones(self.n, dtype=np.uint8)
            ant[live] = d
            self.w.mission_tick_end(self.steps, self.max_speed, self.delta_t, antennas=ant)
        self.tick_no += 1
        self._finished_dirty = True
        return created, deleted

    def state(self):
        """(translation [n, 3] f32, remaining waypoints per robot, completion tick per robot) — synchronises"""
        tr, tg, fin = self.w.mission_read()
        if self.despawn:
            self._alive = fin < 0
        return tr, self.n_way - tg, fin

    @property
    def translation(self):
        return self.state()[0]

    @property
    def finished_at(self):
        return self.state()[2]

    def run(self, max_ticks, check_every=16):
        """Ticks until every robot has finished (or max_ticks), looking at the missions every `check_every` ticks."""
        while self.tick_no < max_ticks:
            self.tick()
            if self.tick_no % check_every == 0 and not (self.finished_at < 0).any():
                break
        return self.summary()

    def summary(self):
        tr, left, fin = self.state()
        done = fin >= 0
        return {"ticks": self.tick_no, "finished": int(done.sum()), "makespan_s": float(fin.max() * self.delta_t) if done.all() else None,
                "finished_at_tick": fin.tolist(), "messages": [self.w.message_counts(r) for r in range(self.n)]}
