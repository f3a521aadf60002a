"""Formations -> robots: the reference's spawner as host logic (no device work here).

    FormationSpawner / RepeatingTimer     crates/magics/src/planner/spawner.rs:178-312,350-367
    spawn_formation                       spawner.rs:415-649
    Formation::as_positions               crates/gbp_config/src/formation.rs:304-455
    {randomly,evenly}_place_nonoverlapping_circles_along_line_segment   formation.rs:546-639
    get_variable_timesteps                crates/magics/src/utils.rs:35-75 (mgx_variable_timesteps)

Everything is f32 like the reference (glam Vec2 arithmetic, `as f32` casts); sin / cos go through
the C library.  The random draws come from `prng.WyRand` in the order spawn_formation makes them:
the radii of the formation, the placement attempts, then per robot the display colour and the
forked per-robot generator.
"""
import ctypes
import ctypes.util

import numpy as np

from . import hostlib

F = np.float32
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.sinf.restype = _libm.cosf.restype = ctypes.c_float
_libm.sinf.argtypes = _libm.cosf.argtypes = [ctypes.c_float]
PI32 = F(np.pi)
N_DISPLAY_COLOURS = 14  # DisplayColour (crates/magics/src/theme.rs:72-87)


def _v(x, y):
    return np.array([x, y], dtype=F)


def _length(v):  # glam Vec2::length: sqrt(x x + y y)
    return np.sqrt(v[0] * v[0] + v[1] * v[1])


def _normalize(v):  # glam Vec2::normalize: self * length_recip()
    return v * (F(1.0) / _length(v))


def _normalize_or_zero(v):  # glam Vec2::normalize_or_zero
    with np.errstate(divide="ignore"):
        rcp = F(1.0) / _length(v)
    return v * rcp if np.isfinite(rcp) and rcp > 0.0 else _v(0.0, 0.0)


def _lerp(a, b, s):  # glam Vec2::lerp: self + (rhs - self) * s
    return a + (b - a) * F(s)


def _polar(angle, magnitude):  # formation.rs:459-463
    return _v(F(_libm.cosf(float(angle))) * F(magnitude), F(_libm.sinf(float(angle))) * F(magnitude))


def point_to_world_position(p, world_dims):  # WorldDimensions::point_to_world_position (formation.rs:512-518)
    return _v(F((p[0] - 0.5) * world_dims[0]), F((p[1] - 0.5) * world_dims[1]))


def randomly_place_along_line_segment(a, b, radii, max_attempts, rng):  # formation.rs:546-590
    for _ in range(max_attempts):
        placed, lerp_amounts = [], []
        for radius in radii:
            amount = rng.gen_range_f32(0.0, 1.0)
            pos = _lerp(a, b, amount)
            if all(_length(pos - q) >= other + F(radius) for q, other in placed):
                lerp_amounts.append(amount)
                placed.append((pos, F(radius)))
                if len(placed) == len(radii):
                    return lerp_amounts
    return None


def evenly_place_along_line_segment(a, b, radii):  # formation.rs:594-639
    radii = [F(r) for r in radii]
    lo, hi = min(radii), max(radii)
    dist = _length(a - b)
    if dist / hi < lo:
        return None
    direction = _normalize(b - a)
    extra = dist / hi
    center = a + radii[0] * direction
    placed = []
    for r1, r2 in zip(radii, radii[1:] + [F(0.0)]):
        diff = r2 - r1
        placed.append(_length(center - a) / dist)
        center = center + ((r1 + diff) * F(2.0) + (extra - diff) * direction)  # f32 + Vec2 adds to both components
    return placed


def as_positions(formation, world_dims, radii, rng):
    """Formation::as_positions -> (initial positions [n][2], waypoint positions [n_wp][n][2]) or None."""
    ip = formation["initial-position"]
    shape, (strategy, attempts) = ip["shape"], ip["placement-strategy"]
    n = formation["robots"]
    if shape["kind"] == "line-segment":
        a, b = (point_to_world_position(p, world_dims) for p in shape["points"])
        if strategy == "random":
            amounts = randomly_place_along_line_segment(a, b, radii, attempts, rng)
        else:
            amounts = evenly_place_along_line_segment(a, b, radii)
        if amounts is None:
            return None
        assert len(amounts) == n
        initial = [_lerp(a, b, t) for t in amounts]
        waypoints = []
        for wp in formation["waypoints"]:
            if wp["shape"]["kind"] != "line-segment":
                raise NotImplementedError("no time for the other combinations sadly :(")  # formation.rs:357
            wa, wb = (point_to_world_position(p, world_dims) for p in wp["shape"]["points"])
            order = amounts if wp["projection-strategy"] == "identity" else amounts[::-1]
            waypoints.append([_lerp(wa, wb, t) for t in order])
        return initial, waypoints
    if shape["kind"] == "circle":
        if strategy != "equal":
            raise NotImplementedError("todo!() in the reference (formation.rs:401-405)")
        center = point_to_world_position(shape["center"], world_dims)
        step = F(2.0) * PI32 / F(n)
        angles = [F(i) * step for i in range(n)]
        initial = [center + _polar(t, shape["radius"]) for t in angles]
        waypoints = []
        for wp in formation["waypoints"]:
            if wp["shape"]["kind"] != "circle":
                raise NotImplementedError("no time for the other combinations sadly :(")
            if wp["projection-strategy"] != "cross":
                raise ValueError("does not make sense for a circle")  # formation.rs:432
            c = point_to_world_position(wp["shape"]["center"], world_dims)
            waypoints.append([c + _polar(t + PI32, wp["shape"]["radius"]) for t in angles])
        return initial, waypoints
    raise NotImplementedError("Shape::Polygon: todo!() in the reference (formation.rs:453)")


class RepeatingTimer:  # spawner.rs:178-219 over bevy's Timer (TimerMode::Repeating)
    def __init__(self, every_ns, times):
        self.duration, self.times = every_ns, times  # times: None = infinite
        self.elapsed, self._just = 0, False

    def tick(self, delta_ns):
        self.elapsed += delta_ns
        if self.duration == 0:
            self._just = True  # a zero-length repeating timer finishes on every tick
            self.elapsed = 0
        elif self.elapsed >= self.duration:
            self._just = True
            self.elapsed %= self.duration
        else:
            self._just = False

    def exhausted(self):
        return self.times is not None and self.times == 0

    def just_finished(self):
        finished = self._just and not self.exhausted()
        if finished and self.times is not None and self.times > 0:
            self.times -= 1
        return finished


class FormationSpawner:  # spawner.rs:221-312
    INACTIVE, READY, COOLDOWN, FINISHED = range(4)

    def __init__(self, index, formation):
        rep = formation["repeat"]
        self.index = index
        self.delay, self.delay_elapsed = formation["delay"], 0
        self.timer = RepeatingTimer(rep["every"], rep["times"]) if rep is not None else RepeatingTimer(0, 1)  # spawner.rs:350-353
        self.spawned, self.state = 0, self.INACTIVE

    def tick(self, delta_ns):
        if self.state == self.INACTIVE:
            self.delay_elapsed += delta_ns
            if self.delay_elapsed >= self.delay:  # Timer(Once)::just_finished
                self.state = self.READY
        elif self.state == self.COOLDOWN:
            self.timer.tick(delta_ns)
            if self.timer.just_finished():
                self.state = self.FINISHED if self.timer.exhausted() else self.READY

    def clone(self):
        """a copy whose timers run independently (all state is scalars; the formation itself is not copied)"""
        import copy
        c = copy.copy(self)
        c.timer = copy.copy(self.timer)
        return c

    def ready_to_spawn(self):
        return self.state == self.READY

    def spawn(self):
        if self.state == self.READY:
            self.state = self.COOLDOWN
            self.spawned += 1

    def exhausted(self):
        return self.state == self.FINISHED


def spawn_formation(formation, config, world_dims, rng):
    """spawn_formation (spawner.rs:415-649) -> list of robot descriptions (dicts), or None when
    the placement failed.  Consumes `rng` exactly as the reference's system does."""
    rb, n = config["robot"], formation["robots"]
    target_speed = F(rb["target-speed"])
    radii = [rng.gen_range_f32_inclusive(rb["radius"]["min"], rb["radius"]["max"]) for _ in range(n)]
    placed = as_positions(formation, world_dims, radii, rng)
    if placed is None:
        return None
    initial, waypoint_positions = placed

    def pose(a, b):
        v = _normalize_or_zero(b - a) * target_speed
        return np.array([a[0], a[1], v[0], v[1]], dtype=F)
    initial_poses = [pose(a, b) for a, b in zip(initial, waypoint_positions[0])]
    legs = waypoint_positions + [waypoint_positions[-1]]  # .chain(last).tuple_windows()
    waypoint_poses = [[pose(a, b) for a, b in zip(legs[k], legs[k + 1])] for k in range(len(waypoint_positions))]
    if not radii:
        return []
    lookahead_horizon = int(F(rb["target-speed"]) * F(rb["planning-horizon"]))  # `as u32`
    timesteps = hostlib.variable_timesteps(lookahead_horizon, config["gbp"]["lookahead-multiple"])
    robots = []
    for i in range(n):
        states = [initial_poses[i]] + [wps[i].copy() for wps in waypoint_poses]
        states[-1][2:] = states[-2][2:]  # last.update_velocity(second_last.velocity())
        colour = rng.gen_index(N_DISPLAY_COLOURS)  # DisplayColour::iter().choose(prng)
        child = rng.fork()                          # prng.fork_rng()
        robots.append({"radius": radii[i], "waypoints": states, "timesteps": timesteps, "colour": colour, "rng": child,
                       "planning-strategy": formation["planning-strategy"],
                       "waypoint-reached-when-intersects": formation["waypoint-reached-when-intersects"],
                       "finished-when-intersects": formation["finished-when-intersects"]})
    return robots


class EntityAllocator:
    """Bevy 0.13's `Entities` allocator restricted to the robots (bevy_ecs/src/entity/mod.rs: `alloc` pops the `pending` list
    of freed indices last-freed-first and hands the index out with the generation `free` raised; a fresh index starts at
    generation 1).  An Entity orders by `to_bits()` = generation << 32 | index — GENERATION first — and that order is the order
    of the factor graphs (FactorGraphId = Entity, id.rs:19-54: inbox order, slot order of an inter-robot factor).  So a robot
    spawned into a reused index sorts behind every first-generation robot, and two robots of the same later generation sort
    by index, i.e. possibly against their spawn order.  The reference's other entities (meshes, waypoints, UI) take indices
    from the same allocator and are not modelled: between robots of one generation the real application may order differently;
    a host that has the real entities passes `Entity::to_bits()` as `order_key`."""

    def __init__(self):
        self.generation, self.pending = [], []

    def alloc(self):
        if self.pending:
            i = self.pending.pop()
        else:
            i = len(self.generation)
            self.generation.append(1)
        return (self.generation[i] << 32) | i

    def free(self, bits):
        i = int(bits) & 0xffffffff
        assert self.generation[i] == int(bits) >> 32, "stale entity"
        self.generation[i] += 1
        self.pending.append(i)
