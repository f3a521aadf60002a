"""Synthetic N-robot x K-horizon factor graphs (host logic, no device needed).

Restates the graph-construction rules of the reference spawner so that the engine and
the CPU oracle are fed identical inputs:

* ``RobotBundle::new`` — crates/magics/src/planner/robot.rs:1134-1356 (f32 interpolation of
  the initial means, 1e30 / +inf prior precisions, delta_t = t0 * (ts[i+1]-ts[i]) in f32);
* ``get_variable_timesteps`` — crates/magics/src/utils.rs:35-75 (via ``mgx_variable_timesteps``);
* ``create_interrobot_factors`` numbering — robot.rs:1490-1541 (robot_number in
  (robot asc, neighbour asc, i = 1..K-1) order);
* parameter values — config/scenarios/Junction Twoway/config.toml (f32 literals widened).

A scenario is a plain dict; ``populate(world, scenario)`` feeds it to any object with the
``World`` interface (``magics_amd.World``; the tests also pass the CPU oracle's world).
"""
import math
import os

import numpy as np

from . import hostlib

F32 = np.float32
MASK64 = (1 << 64) - 1


class SplitMix64:
    """SplitMix64 PRNG (public-domain algorithm by S. Vigna); seed 805 = the reference's
    circle-scenario seed (config/scenarios/Circle Experiment/config.toml:75)."""

    def __init__(self, seed=805):
        self.s = seed & MASK64

    def next_u64(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def uniform(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * ((self.next_u64() >> 11) * (1.0 / (1 << 53)))


def f32w(x):
    """an f32 config literal widened to f64 (``Float::from(f32)``)."""
    return float(F32(x))


JUNCTION_PARAMS = dict(
    sigma_dynamics=f32w(0.1), sigma_interrobot=f32w(0.01), sigma_obstacle=f32w(0.01),
    sigma_tracking=f32w(0.01), safety_multiplier=f32w(2.5),
    tracking_switch_padding=f32w(1.0), tracking_attraction_distance=f32w(2.0),
)
CIRCLE_PARAMS = dict(
    sigma_dynamics=f32w(1.0), sigma_interrobot=f32w(0.005), sigma_obstacle=f32w(0.005),
    sigma_tracking=f32w(0.1), safety_multiplier=f32w(2.2),
    tracking_switch_padding=f32w(1.0), tracking_attraction_distance=f32w(2.0),
)

EN_DYN, EN_IR, EN_OBS, EN_TRK = 1, 2, 4, 8

# lookahead horizons reaching the BASELINE.json horizon lengths with lookahead_multiple 3
HORIZON_FOR_K = {10: 18, 12: 25, 16: 45, 21: 75, 32: 176, 33: 180, 34: 188, 35: 199, 40: 261, 45: 331}


def timesteps_for_K(K, multiple=3):
    ts = hostlib.variable_timesteps(HORIZON_FOR_K[K], multiple)
    assert len(ts) == K, (K, ts)
    return ts


def robot_initial_state(start, goal, timesteps, radius, target_speed, planning_horizon):
    """Initial means / priors / delta_t of one robot — robot.rs:1156-1255 in f32."""
    start = np.asarray(start, dtype=F32)
    goal = np.asarray(goal, dtype=F32)
    s2g = goal - start
    length = F32(math.sqrt(float(np.sum(s2g.astype(np.float64) ** 2))))
    length = F32(np.sqrt(np.sum(s2g * s2g, dtype=F32)))
    direction = s2g / length if length > 0 else s2g
    reach = min(length, F32(planning_horizon) * F32(target_speed))
    horizon = start + F32(reach) * direction
    last = F32(timesteps[-1])
    K = len(timesteps)
    mean0 = np.zeros((K, 4), dtype=np.float64)
    for i, ts in enumerate(timesteps):
        m = start + (horizon - start) * (F32(ts) / last)
        mean0[i] = m.astype(np.float64)
    prior = np.full(K, np.inf)
    prior[0] = prior[-1] = 1e30
    t0 = F32(radius) / F32(2.0) / F32(target_speed)
    dt = np.array([float(F32(t0 * F32(timesteps[i + 1] - timesteps[i]))) for i in range(K - 1)])
    return mean0, prior, dt


def gaussian_blur_u8(img, sigma):
    """Separable Gaussian blur of a single-channel float image (mimics the env_to_png blur,
    crates/env_to_png/src/lib.rs:156-161, for synthetic inputs only)."""
    r = max(1, int(math.ceil(3 * sigma)))
    xs = np.arange(-r, r + 1, dtype=np.float64)
    k = np.exp(-0.5 * (xs / sigma) ** 2)
    k /= k.sum()
    out = img.astype(np.float64)
    tmp = np.empty_like(out)
    for axis in (0, 1):
        pad = [(0, 0), (0, 0)]
        pad[axis] = (r, r)
        p = np.pad(out, pad, mode="edge")
        acc = np.zeros_like(out)
        for j, kv in enumerate(k):  # acc += kv * window, without a temporary per tap
            sl = [slice(None), slice(None)]
            sl[axis] = slice(j, j + out.shape[axis])
            np.multiply(p[tuple(sl)], kv, out=tmp)
            acc += tmp
        out = acc
    return out


def synthetic_sdf(rng, world_w, world_h, px_per_m=10, disc_area_frac=0.02, blur_sigma_px=2.0):
    """White RGB u8 image with seeded black discs (radius 1-3 m), Gaussian blurred.  Large images are
    cached on disk (the ranks and child processes of one bench run all build the same one)."""
    w, h = int(round(world_w * px_per_m)), int(round(world_h * px_per_m))
    cache = None
    if w * h >= 4_000_000:
        import hashlib
        import tempfile
        key = hashlib.sha1(repr((rng.s, world_w, world_h, px_per_m, disc_area_frac, blur_sigma_px, "v1")).encode()).hexdigest()[:16]
        cache = os.path.join(os.environ.get("MGX_SDF_CACHE", tempfile.gettempdir()), f"mgx_sdf_{key}.npz")
        try:
            with np.load(cache) as z:
                red, state = z["red"], int(z["state"])
            if red.shape == (h, w):
                rng.s = state  # the generator ends where drawing the discs would have left it
                return np.repeat(red[:, :, None], 3, axis=2)
        except (OSError, ValueError, KeyError):
            pass
    img = np.full((h, w), 255.0)
    n_discs = int(round(disc_area_frac * world_w * world_h / (math.pi * 4.0)))
    yy, xx = None, None
    for _ in range(n_discs):
        cx = rng.uniform(-world_w / 2, world_w / 2)
        cy = rng.uniform(-world_h / 2, world_h / 2)
        rad = rng.uniform(1.0, 3.0)
        # pixel bbox (image y axis is flipped: py = (-y + H/2) * scale)
        px, py, pr = (cx + world_w / 2) * px_per_m, (-cy + world_h / 2) * px_per_m, rad * px_per_m
        x0, x1 = max(0, int(px - pr) - 1), min(w, int(px + pr) + 2)
        y0, y1 = max(0, int(py - pr) - 1), min(h, int(py + pr) + 2)
        if x0 >= x1 or y0 >= y1:
            continue
        yy, xx = np.mgrid[y0:y1, x0:x1]
        mask = (xx + 0.5 - px) ** 2 + (yy + 0.5 - py) ** 2 <= pr * pr
        img[y0:y1, x0:x1][mask] = 0.0
    img = gaussian_blur_u8(img, blur_sigma_px)
    red = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    if cache is not None:
        try:
            tmp = f"{cache}.{os.getpid()}.tmp.npz"
            np.savez(tmp, red=red, state=np.uint64(rng.s))
            os.replace(tmp, cache)
        except OSError:
            pass
    return np.repeat(red[:, :, None], 3, axis=2)


def grid_scenario(n_robots, K=16, interrobot=False, comm_radius=8.0, obstacles=True, seed=805,
                  pitch=5.0, target_speed=5.0, radius=1.0, n_internal=10, n_external=None,
                  tracking=False, schedule_kind=hostlib.SCHEDULE_INTERLEAVE_EVENLY, origin=(0.0, 0.0),
                  grid_side=None, first_index=0, total_robots=None):
    """BASELINE.json configs 2-4: robots on a ceil(sqrt(N))^2 grid.

    ``first_index`` / ``total_robots`` select a contiguous slice of the global robot list
    (used when sharding: every rank generates the same global layout and keeps its part).
    """
    rng = SplitMix64(seed)
    total = total_robots if total_robots is not None else n_robots
    G = grid_side or int(math.ceil(math.sqrt(total)))
    ts = timesteps_for_K(K)
    planning_horizon = HORIZON_FOR_K[K] / target_speed
    # all robots of the global layout are drawn so slices agree across ranks
    starts, goals, pos = [], [], []
    for r in range(total):
        gx, gy = r % G, r // G
        x = (gx - (G - 1) / 2) * pitch + rng.uniform(-0.5, 0.5) + origin[0]
        y = (gy - (G - 1) / 2) * pitch + rng.uniform(-0.5, 0.5) + origin[1]
        th = rng.uniform(0.0, 2 * math.pi)
        vx, vy = target_speed * math.cos(th), target_speed * math.sin(th)
        starts.append((x, y, vx, vy))
        far = 1000.0
        goals.append((x + far * math.cos(th), y + far * math.sin(th), vx, vy))
        pos.append((x, y))
    pos = np.array(pos)
    world = G * pitch + 50.0
    sdf = None
    if obstacles:
        sdf = dict(rgb=synthetic_sdf(rng, world, world), world_w=world, world_h=world)
    else:
        sdf = dict(rgb=np.full((16, 16, 3), 255, dtype=np.uint8), world_w=world, world_h=world)
    enable = EN_DYN | EN_OBS | (EN_IR if interrobot else 0) | (EN_TRK if tracking else 0)
    params = dict(JUNCTION_PARAMS, enable_mask=enable)
    robots = []
    for r in range(total):
        mean0, prior, dt = robot_initial_state(starts[r], goals[r], ts, radius, target_speed, planning_horizon)
        path = None
        if tracking:
            s, g = np.array(starts[r][:2]), np.array(goals[r][:2])
            d = (g - s) / np.linalg.norm(g - s)
            path = np.array([s, s + 30.0 * d, s + 60.0 * d], dtype=F32)
        robots.append(dict(mean0=mean0, prior_diag=prior, dt=dt, radius=radius, path=path, order_key=r,
                           pos=pos[r], goal=np.array(goals[r][:2]), t0=F32(radius) / F32(2.0) / F32(target_speed)))
    pairs = []
    if interrobot:
        pairs = neighbour_pairs(pos, comm_radius)
    ir = number_ir_pairs(pairs, K)
    n_ext = n_external if n_external is not None else (n_internal if interrobot else 0)
    steps = hostlib.schedule(schedule_kind, n_internal, n_ext)
    return dict(params=params, sdf=sdf, robots=robots, ir=ir, steps=steps, K=K, positions=pos, target_speed=target_speed,
                name=f"grid{total}x{K}{'+ir' if interrobot else ''}{'+trk' if tracking else ''}")


def neighbour_pairs(pos, radius):
    """Ordered (a, b) pairs, a != b, |p_a - p_b| <= radius, sorted (a asc, b asc) — the order in
    which `create_interrobot_factors` walks robots (robot.rs:1490-1541). Cell-list search."""
    pos = np.asarray(pos)
    cell = {}
    inv = 1.0 / radius
    for i, (x, y) in enumerate(pos):
        cell.setdefault((int(math.floor(x * inv)), int(math.floor(y * inv))), []).append(i)
    pairs = []
    r2 = radius * radius
    for i, (x, y) in enumerate(pos):
        cx, cy = int(math.floor(x * inv)), int(math.floor(y * inv))
        nb = []
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for j in cell.get((cx + dx, cy + dy), ()):
                    if j != i and (pos[j, 0] - x) ** 2 + (pos[j, 1] - y) ** 2 <= r2:
                        nb.append(j)
        for j in sorted(nb):
            pairs.append((i, j))
    return pairs


def number_ir_pairs(pairs, K):
    """(owner, other, first_robot_number): a global counter starting at 1 hands K-1 numbers to
    each directed pair (RobotNumberGenerator, robot.rs:121-140,1527)."""
    out, n = [], 1
    for a, b in pairs:
        out.append((a, b, n))
        n += K - 1
    return out


def circle_scenario(n_robots=10, K=10, circle_radius=50.0, seed=805, n_internal=50, n_external=10):
    """BASELINE.json config 1: robots equally spaced on a circle, heading to the antipode
    (crates/gbp_config/src/formation.rs:391-450), all-white image, comms radius = circle
    radius, Circle-Experiment sigmas and 50/10 interleave-evenly schedule."""
    rng = SplitMix64(seed)
    target_speed = HORIZON_FOR_K[K] / 3.0  # planning horizon 3 s
    ts = timesteps_for_K(K)
    robots, pos = [], []
    for r in range(n_robots):
        a = 2 * math.pi * r / n_robots
        x, y = circle_radius * math.cos(a), circle_radius * math.sin(a)
        vx, vy = -target_speed * math.cos(a), -target_speed * math.sin(a)
        radius = f32w(rng.uniform(2.0, 3.0))
        mean0, prior, dt = robot_initial_state((x, y, vx, vy), (-x, -y, vx, vy), ts, radius, target_speed, 3.0)
        robots.append(dict(mean0=mean0, prior_diag=prior, dt=dt, radius=radius, path=None, order_key=r,
                           pos=np.array([x, y]), goal=np.array([-x, -y]), t0=F32(radius) / F32(2.0) / F32(target_speed)))
        pos.append((x, y))
    pos = np.array(pos)
    world = 4 * circle_radius
    sdf = dict(rgb=np.full((200, 200, 3), 255, dtype=np.uint8), world_w=world, world_h=world)
    params = dict(CIRCLE_PARAMS, enable_mask=EN_DYN | EN_OBS | EN_IR)
    ir = number_ir_pairs(neighbour_pairs(pos, circle_radius), K)
    steps = hostlib.schedule(hostlib.SCHEDULE_INTERLEAVE_EVENLY, n_internal, n_external)
    return dict(params=params, sdf=sdf, robots=robots, ir=ir, steps=steps, K=K, positions=pos,
                name=f"circle{n_robots}x{K}", target_speed=target_speed)


def junction_environment(tiles):
    """`tiles` x `tiles` crossroads with the settings of config/scenarios/Junction Twoway/environment.yaml
    (tile-size 100, path-width 0.16, sdf: 200 px / tile, expansion 0.01, blur 0.01)."""
    from . import environment
    return environment.new(["┼" * tiles] * tiles, 0.16, 2.0, 100.0, sdf={"resolution": 200, "expansion": 0.01, "blur": 0.01})


def junction_scenario(n_robots, K=32, tiles=None, comm_radius=20.0, seed=805, target_speed=5.0, radius=1.0,
                      n_internal=10, n_external=10, tracking=True, interrobot=True, connect_after_ticks=1):
    """BASELINE.json configs[4] (SURVEY.md §8d "Config 5"): a `tiles` x `tiles` grid of two-way
    crossroads rasterised by the env_to_png rule ON THE DEVICE (World.set_environment), robots on
    the lanes of config/scenarios/Junction Twoway/formation.yaml (4 m off the centre line, entering
    from the four sides, turning left / right or driving across), each with its 2-3 point lane
    polyline as tracking path, Junction Twoway sigmas and communication radius.

    connect_after_ticks (default 1): the robots run that many driver ticks on their own before the inter-robot factors are
    created — the order the reference's driver works in (create_interrobot_factors, robot.rs:1441-1586, hooks up robots that
    have been iterating since they spawned; a factor starts from its target's current belief, :1549-1585).  `populate` does
    it: the pairs are in sc["ir_late"], sc["ir"] (connected before the first sweep) is empty.  With inter-robot AND tracking
    factors on robots that have never iterated, the reference's own arithmetic leaves the finite range within a tick (the first
    inter-robot messages are Schur complements of rank-1 blocks next to the rounding residue of the dynamics factors, DESIGN.md
    §2); after one tick on their own the beliefs are formed and the same 4000 x 32 world stays finite.  0: everything at once
    (what a sharded world, which plans its ghosts from sc["ir"], is built from)."""
    rng = SplitMix64(seed)
    per_tile = 10
    if tiles is None:
        tiles = max(1, int(math.ceil(math.sqrt(n_robots / per_tile))))
    per_tile = int(math.ceil(n_robots / (tiles * tiles)))
    ts = timesteps_for_K(K)
    planning_horizon = HORIZON_FOR_K[K] / target_speed
    half = 50.0

    def turn(p, k):  # k quarter turns clockwise about the tile centre: entries W -> N -> E -> S
        x, y = p
        for _ in range(k):
            x, y = y, -x
        return x, y
    robots, pos = [], []
    for r in range(n_robots):
        tile, q = divmod(r, per_tile)
        row, col = divmod(tile, tiles)
        cx, cy = (col - (tiles - 1) / 2) * 100.0, ((tiles - 1) / 2 - row) * 100.0
        entry, slot = q % 4, q // 4
        lane = 4.0 + rng.uniform(-1.0, 1.0)
        dist = 5.0 + 11.0 * slot + rng.uniform(-1.5, 1.5)
        manoeuvre = int(rng.uniform(0.0, 3.0))
        route = [(-half + dist, lane)]                      # entering from the west, heading east
        if manoeuvre == 0:
            route += [(-2.5 + rng.uniform(-1.0, 1.0), lane), (-2.5, half + 15.0)]     # left: north on x = -2.5
        elif manoeuvre == 1:
            route += [(lane, lane), (lane, -half - 15.0)]                             # right: south on x = +lane
        else:
            route += [(half + 15.0, lane)]                                            # across
        route = [turn(p, entry) for p in route]
        route = [(cx + x, cy + y) for x, y in route]
        hx, hy = turn((1.0, 0.0), entry)
        start = (route[0][0], route[0][1], target_speed * hx, target_speed * hy)
        nxt = (route[1][0], route[1][1], target_speed * hx, target_speed * hy)
        mean0, prior, dt = robot_initial_state(start, nxt, ts, radius, target_speed, planning_horizon)
        robots.append(dict(mean0=mean0, prior_diag=prior, dt=dt, radius=radius, path=np.array(route, dtype=F32) if tracking else None,
                           order_key=r, pos=np.array(route[0]), goal=np.array(route[1]), t0=F32(radius) / F32(2.0) / F32(target_speed)))
        pos.append(route[0])
    pos = np.array(pos)
    enable = EN_DYN | EN_OBS | (EN_IR if interrobot else 0) | (EN_TRK if tracking else 0)
    params = dict(JUNCTION_PARAMS, enable_mask=enable)
    ir = number_ir_pairs(neighbour_pairs(pos, comm_radius) if interrobot else [], K)
    steps = hostlib.schedule(hostlib.SCHEDULE_CENTERED, n_internal, n_external if interrobot else 0)
    late = connect_after_ticks > 0 and bool(ir)
    return dict(params=params, env=junction_environment(tiles), sdf=None, robots=robots, ir=[] if late else ir, ir_late=ir if late else [],
                connect_after_ticks=connect_after_ticks if late else 0, steps=steps, K=K, positions=pos,
                target_speed=target_speed, name=f"junction{n_robots}x{K}+{tiles}x{tiles}tiles")


def tick_inputs(sc, hz=10.0):
    """Arguments of `update_priors` for one driver tick at `hz` (FixedUpdate, config.simulation.hz):
    every robot heads for its goal; time_scale = fixed_dt / t0 as an f32 quotient (robot.rs:2309)."""
    n = len(sc["robots"])
    dt32 = F32(1.0) / F32(hz)
    return dict(robots=np.arange(n, dtype=np.int32), waypoints_xy=np.array([rb["goal"] for rb in sc["robots"]], dtype=np.float64),
                time_scale=np.array([float(dt32 / rb["t0"]) for rb in sc["robots"]]), what=np.full(n, 3, dtype=np.uint8),
                max_speed=f32w(sc["target_speed"]), delta_t=float(dt32))


def populate(world, sc, robots=None):
    """Feed a scenario to a ``World``-like object. Returns the list of robot ids."""
    if sc.get("env") is not None:
        world.set_environment(sc["env"])  # rasterised by the backend itself (device / CPU oracle)
    else:
        world.set_sdf(sc["sdf"]["rgb"], sc["sdf"]["world_w"], sc["sdf"]["world_h"])
    ids = []
    for rb in sc["robots"]:
        ids.append(world.add_robot(rb["mean0"], rb["prior_diag"], rb["dt"], rb["radius"], path=rb["path"],
                                   order_key=rb["order_key"]))
    for a, b, n0 in sc["ir"]:
        world.ir_connect(ids[a], ids[b], n0)
    if sc.get("ir_late"):  # robots that have iterated on their own before they meet (junction_scenario, connect_after_ticks)
        tick = tick_inputs(sc)
        for _ in range(sc["connect_after_ticks"]):
            world.tick(steps=sc["steps"], **tick)
        for a, b, n0 in sc["ir_late"]:
            world.ir_connect(ids[a], ids[b], n0)
    return ids


def algorithmic_bytes_per_robot_iter(K, D=0.0, tracking=False):
    """SURVEY.md §8d / BASELINE.md §4 traffic model (f64): bytes per robot per iteration."""
    T = 1 if tracking else 0
    E = 2 * (K - 1) + (K - 2) * (1 + T)
    R = D * (K - 1)
    internal = E * 352 + (E * 160 + R * 160 + K * 160) + K * 320 + (E + R) * 192
    external = (R * 544 + (E * 160 + R * 160 + K * 160) + K * 320 + R * 192) if R > 0 else 0
    return internal + external
