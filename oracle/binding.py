"""ctypes binding of oracle/gbp_oracle.c (test infrastructure, see package docstring)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_double_p = C.POINTER(C.c_double)


class Params(C.Structure):
    _fields_ = [
        ("sigma_dynamics", C.c_double),
        ("sigma_interrobot", C.c_double),
        ("sigma_obstacle", C.c_double),
        ("sigma_tracking", C.c_double),
        ("safety_multiplier", C.c_double),
        ("tracking_switch_padding", C.c_double),
        ("tracking_attraction_distance", C.c_double),
        ("enable_mask", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class RobotDesc(C.Structure):
    _fields_ = [
        ("K", C.c_uint32),
        ("n_path", C.c_uint32),
        ("mean0", c_double_p),
        ("prior_diag", c_double_p),
        ("dt", c_double_p),
        ("path_xy", C.POINTER(C.c_float)),
        ("radius", C.c_double),
        ("order_key", C.c_uint64),
        ("ghost", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


FLAVOURS = ("fma", "lu", "seq", "fma_lu")  # arithmetic variants of the same restatement (gbp_oracle.c header)


def build_flavour(flavour):
    """The oracle with one of the reference's unknowable arithmetic choices swapped (gbp_oracle.c,
    FLAVOURS): 'fma' fused GEMM steps, 'lu' pivoting 4x4 inverse, 'seq' no unrolled_dot pairing."""
    assert flavour in FLAVOURS, flavour
    out = os.path.join(_HERE, f"libgbp_oracle_{flavour}.so")
    src = os.path.join(_HERE, "gbp_oracle.c")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, os.path.basename(out)], check=True, capture_output=True)
    return out


def build(native=False, out_dir=None):
    """Compile gbp_oracle.c with gcc. ``native=True`` builds a -march=native variant
    (used by bench.py's cpu_baseline on the box it runs on)."""
    if native:
        out_dir = out_dir or _HERE
        out = os.path.join(out_dir, "libgbp_oracle_native.so")
        cmd = ["gcc", "-O3", "-fno-tree-slp-vectorize", "-march=native", "-ffp-contract=off", "-fno-fast-math", "-fopenmp",
               "-fPIC", "-shared", "-o", out, os.path.join(_HERE, "gbp_oracle.c"), "-lm"]
        subprocess.run(cmd, check=True, capture_output=True)
        return out
    subprocess.run(["make", "-C", _HERE, "libgbp_oracle.so", "variants"], check=True, capture_output=True)
    return os.path.join(_HERE, "libgbp_oracle.so")


def _bind(path):
    L = C.CDLL(path)
    L.orc_world_create.restype = C.c_void_p
    L.orc_world_create.argtypes = [C.POINTER(Params)]
    L.orc_world_destroy.argtypes = [C.c_void_p]
    L.orc_set_threads.argtypes = [C.c_void_p, C.c_int]
    L.orc_world_set_sdf.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, C.c_double]
    L.orc_robot_add.argtypes = [C.c_void_p, C.POINTER(RobotDesc), C.POINTER(C.c_int32)]
    L.orc_robot_remove.argtypes = [C.c_void_p, C.c_int32]
    L.orc_ir_connect.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_uint64]
    L.orc_ir_disconnect.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.orc_set_antenna.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.orc_set_idle.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.orc_iterate.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
    L.orc_neighbours.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.orc_update_topology.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    L.orc_connections.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    for n in ("internal_factor", "internal_variable", "external_factor", "external_variable"):
        getattr(L, f"orc_{n}_iteration").argtypes = [C.c_void_p, C.c_int32]
    L.orc_change_prior.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, c_double_p]
    L.orc_reset_variables.argtypes = [C.c_void_p, C.c_int32, c_double_p, C.c_double, C.c_double]
    L.orc_reset_tracking_factors.argtypes = [C.c_void_p, C.c_int32]
    L.orc_update_priors.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int32), c_double_p, c_double_p, C.POINTER(C.c_uint8),
                                    C.c_double, C.c_double]
    L.orc_get_belief.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, c_double_p, c_double_p, c_double_p,
                                 c_double_p, C.POINTER(C.c_int32)]
    L.orc_read_beliefs.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p]
    L.orc_message_counts.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64)]
    L.orc_num_robots.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.orc_variable_inbox.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_int32, C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32), c_double_p, c_double_p]
    L.orc_schedule.argtypes = [C.c_int32, C.c_uint8, C.c_uint8, C.c_char_p, C.c_uint32]
    L.orc_variable_timesteps.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32]
    L.orc_inv4.argtypes = [c_double_p, c_double_p]
    L.orc_marginalise.argtypes = [c_double_p, c_double_p, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p]
    L.orc_euclidean_norm.restype = C.c_double
    L.orc_euclidean_norm.argtypes = [c_double_p, C.c_int]
    L.orc_l1_norm.restype = C.c_double
    L.orc_l1_norm.argtypes = [c_double_p, C.c_int]
    L.orc_normalize.argtypes = [c_double_p, C.c_int]
    L.orc_obstacle_measure.restype = C.c_double
    L.orc_obstacle_measure.argtypes = [C.c_void_p, c_double_p]
    return L


def lib(path=None):
    """Load (building if necessary) the oracle shared library."""
    global _LIB
    if path is not None:
        return _bind(path)
    if _LIB is None:
        p = os.path.join(_HERE, "libgbp_oracle.so")
        src = os.path.join(_HERE, "gbp_oracle.c")
        if not os.path.exists(p) or os.path.getmtime(p) < os.path.getmtime(src):
            build()
        _LIB = _bind(p)
    return _LIB


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        assert a.shape == shape, (a.shape, shape)
    return a


def make_params(p):
    return Params(
        float(p["sigma_dynamics"]), float(p["sigma_interrobot"]), float(p["sigma_obstacle"]),
        float(p["sigma_tracking"]), float(p["safety_multiplier"]),
        float(p.get("tracking_switch_padding", 1.0)), float(p.get("tracking_attraction_distance", 2.0)),
        int(p.get("enable_mask", 7)), 0)


class OracleWorld:
    """Same interface as ``magics_amd.World`` over the CPU oracle."""

    def __init__(self, params, threads=1, lib_path=None):
        self._L = lib(lib_path)
        self._p = make_params(params)
        self._w = self._L.orc_world_create(C.byref(self._p))
        self._L.orc_set_threads(self._w, int(threads))
        self._keep = []

    def close(self):
        if self._w:
            self._L.orc_world_destroy(self._w)
            self._w = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise RuntimeError(f"oracle call failed: {rc}")
        return rc

    def set_sdf(self, rgb, world_w, world_h):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        h, w, c = rgb.shape
        assert c == 3
        self._chk(self._L.orc_world_set_sdf(self._w, rgb.ctypes.data, w, h, float(world_w), float(world_h)))

    def set_environment(self, env):
        """simulation_loader.rs:154-162 + robot.rs:1259-1264 through the CPU rasteriser (oracle/env.py)."""
        from . import env as _env
        red = _env.env_to_sdf_image(env)
        nrows, ncols = len(env["tiles"]["grid"]), len(env["tiles"]["grid"][0])
        ts = float(np.float32(env["tiles"]["settings"]["tile-size"]))
        self.set_sdf(np.repeat(red[:, :, None], 3, axis=2), ts * ncols, ts * nrows)

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
        n = C.c_uint32()
        self._L.orc_num_robots(self._w, C.byref(n), None)
        d.order_key = int(order_key) if order_key is not None else self._next_key()
        d.ghost = 1 if ghost else 0
        rid = C.c_int32(-1)
        self._chk(self._L.orc_robot_add(self._w, C.byref(d), C.byref(rid)))
        return rid.value

    def _next_key(self):
        self._k = getattr(self, "_k", -1) + 1
        return self._k

    def ir_connect(self, owner, other, first_robot_number):
        self._chk(self._L.orc_ir_connect(self._w, owner, other, int(first_robot_number)))

    def remove_robot(self, robot):
        self._chk(self._L.orc_robot_remove(self._w, robot))

    def ir_disconnect(self, a, b):
        self._chk(self._L.orc_ir_disconnect(self._w, a, b))

    def set_enabled(self, mask):
        self._L.orc_set_enabled.argtypes = [C.c_void_p, C.c_uint32]
        self._chk(self._L.orc_set_enabled(self._w, int(mask)))

    def set_antenna(self, robot, active):
        self._chk(self._L.orc_set_antenna(self._w, robot, int(bool(active))))

    def set_idle(self, robot, idle):
        self._chk(self._L.orc_set_idle(self._w, robot, int(bool(idle))))

    def set_antennas(self, robots, active):
        for r, a in zip(robots, active):
            self.set_antenna(int(r), bool(a))

    def neighbours(self, positions, radius, method=0):
        pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        n = pos.shape[0]
        ptr = np.zeros(n + 1, dtype=np.int32)
        need = C.c_uint64()
        self._chk(self._L.orc_neighbours(self._w, pos.ctypes.data, float(radius), ptr.ctypes.data, None, 0, C.byref(need)))
        idx = np.zeros(max(need.value, 1), dtype=np.int32)
        self._chk(self._L.orc_neighbours(self._w, pos.ctypes.data, float(radius), ptr.ctypes.data, idx.ctypes.data,
                                         need.value, C.byref(need)))
        return ptr, idx[:need.value]

    def update_topology(self, positions, radius, next_number, method=0):
        pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        nxt = C.c_uint64(int(next_number))
        stats = (C.c_uint32 * 2)()
        self._chk(self._L.orc_update_topology(self._w, pos.ctypes.data, float(radius), C.byref(nxt), stats))
        return nxt.value, stats[0], stats[1]

    def connections(self, robot):
        n = C.c_uint32()
        self._chk(self._L.orc_connections(self._w, robot, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.int32)
        self._chk(self._L.orc_connections(self._w, robot, out.ctypes.data, n.value, C.byref(n)))
        return out[:n.value].tolist()

    def iterate(self, steps):
        steps = bytes(bytearray(int(s) for s in steps))
        self._chk(self._L.orc_iterate(self._w, steps, len(steps)))

    def internal_factor_iteration(self, robot=-1):
        self._chk(self._L.orc_internal_factor_iteration(self._w, robot))

    def internal_variable_iteration(self, robot=-1):
        self._chk(self._L.orc_internal_variable_iteration(self._w, robot))

    def external_factor_iteration(self, robot=-1):
        self._chk(self._L.orc_external_factor_iteration(self._w, robot))

    def external_variable_iteration(self, robot=-1):
        self._chk(self._L.orc_external_variable_iteration(self._w, robot))

    def change_prior(self, robot, var_ix, mean):
        mean = _f64(mean, (4,))
        self._chk(self._L.orc_change_prior(self._w, robot, var_ix, _dp(mean)))

    def reset_variables(self, robot, means, first_last_sigma=1e30, inbetween_sigma=float("inf")):
        m = _f64(means)
        self._chk(self._L.orc_reset_variables(self._w, robot, _dp(m), float(first_last_sigma), float(inbetween_sigma)))

    def reset_tracking_factors(self, robot):
        self._chk(self._L.orc_reset_tracking_factors(self._w, robot))

    def change_priors(self, robots, var_ix, means):
        means = _f64(means)
        for r, v, m in zip(robots, var_ix, means):
            self.change_prior(int(r), int(v), m)

    def update_priors(self, robots, waypoints_xy, time_scale, what, max_speed, delta_t):
        robots = np.ascontiguousarray(robots, dtype=np.int32)
        n = len(robots)
        wp, ts = _f64(waypoints_xy, (n, 2)), _f64(time_scale, (n,))
        what = np.ascontiguousarray(what, dtype=np.uint8)
        self._chk(self._L.orc_update_priors(self._w, n, robots.ctypes.data_as(C.POINTER(C.c_int32)), _dp(wp), _dp(ts),
                                            what.ctypes.data_as(C.POINTER(C.c_uint8)), float(max_speed), float(delta_t)))

    def tick(self, robots, waypoints_xy, time_scale, what, max_speed, delta_t, steps):
        """The reference's chain: both prior updates for the listed robots, then the schedule (robot.rs:86-103)."""
        self.update_priors(robots, waypoints_xy, time_scale, what, max_speed, delta_t)
        self.iterate(steps)

    def get_belief(self, robot, var_ix):
        eta, lam, mean, cov = np.zeros(4), np.zeros((4, 4)), np.zeros(4), np.zeros((4, 4))
        valid = C.c_int32()
        self._chk(self._L.orc_get_belief(self._w, robot, var_ix, _dp(eta), _dp(lam), _dp(mean), _dp(cov),
                                         C.byref(valid)))
        return {"eta": eta, "lam": lam, "mean": mean, "cov": cov, "valid": bool(valid.value)}

    def num_robots(self):
        a, b = C.c_uint32(), C.c_uint32()
        self._L.orc_num_robots(self._w, C.byref(a), C.byref(b))
        return a.value, b.value

    def read_means(self):
        return self.read_beliefs()[2]

    def read_variable_means(self, var_ix):
        mu = self.read_beliefs()[2]
        nr = self.num_robots()[0]
        return mu.reshape(nr, -1, 4)[:, var_ix, :].copy()

    def message_counts(self, robot):
        """(sent internal, sent external, received internal, received external) of one graph."""
        out = (C.c_uint64 * 4)()
        self._chk(self._L.orc_message_counts(self._w, robot, out))
        return tuple(int(x) for x in out)

    def read_beliefs(self):
        _, nv = self.num_robots()
        eta, lam, means = np.zeros((nv, 4)), np.zeros((nv, 4, 4)), np.zeros((nv, 4))
        self._chk(self._L.orc_read_beliefs(self._w, _dp(eta), _dp(lam), _dp(means)))
        return eta, lam, means

    def synchronize(self):
        pass

    def variable_inbox(self, robot, var_ix):
        """White-box: list of (from_robot, from_index, present, eta, lam) of a variable's inbox."""
        out, j = [], 0
        while True:
            fr, fi = C.c_int32(), C.c_int32()
            eta, lam = np.zeros(4), np.zeros((4, 4))
            p = self._L.orc_variable_inbox(self._w, robot, var_ix, j, C.byref(fr), C.byref(fi), _dp(eta), _dp(lam))
            if p < 0:
                return out
            out.append((fr.value, fi.value, bool(p), eta, lam))
            j += 1


    def variable_inbox_graphs(self, robot, var_ix):
        """White-box: the sending graph (robot id) of every inbox slot of a variable."""
        return [e[0] for e in self.variable_inbox(robot, var_ix)]


def schedule(kind, n_internal, n_external):
    buf = C.create_string_buffer(256)
    n = lib().orc_schedule(int(kind), n_internal, n_external, buf, 256)
    if n < 0:
        raise ValueError("bad schedule arguments")
    return list(buf.raw[:n])


def variable_timesteps(horizon, multiple):
    buf = (C.c_uint32 * 1024)()
    n = lib().orc_variable_timesteps(horizon, multiple, buf, 1024)
    if n < 0:
        raise ValueError("bad arguments")
    return list(buf[:n])
