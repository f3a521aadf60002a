"""ctypes loader of the C-ABI library ``libmgx.so`` (include/mgx.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  Loading it
needs no GPU; every *compute* entry point fails loudly (``MgxError``) when no HIP device is
usable — there is no CPU path in the product.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmgx.so")
# The product library is built with -ffp-contract=off: GBP on these graphs is numerically chaotic
# while beliefs form (rank-deficient precisions inverted next to rounding noise), so the kernels
# keep the reference's scalar f64 operation order and never fuse a*b+c — results are then bit
# identical to the CPU restatement (tests/test_gpu_parity.py).  libmgx_fma.so is the same source
# with FMA contraction, kept only to measure what that costs (MGX_FMA=1 selects it).
FMA_LIB_PATH = os.path.join(_HERE, "lib", "libmgx_fma.so")

SCHEDULE_CENTERED = 0
SCHEDULE_SOON_AS_POSSIBLE = 1
SCHEDULE_LATE_AS_POSSIBLE = 2
SCHEDULE_INTERLEAVE_EVENLY = 3
SCHEDULE_HALF_BEGINNING_HALF_END = 4

STEP_INTERNAL = 1
STEP_EXTERNAL = 2
NEIGHBOURS_AUTO, NEIGHBOURS_PAIRS, NEIGHBOURS_GRID = 0, 1, 2
RESIDENT_NONE, RESIDENT_RAN, RESIDENT_DECLINED = 0, 1, 2
HALO_PUSH, HALO_WAIT = 1, 2
HINT_NEXT_STARTS_EXTERNAL = 1

c_double_p = C.POINTER(C.c_double)


class MgxError(RuntimeError):
    pass


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


SHAPE_CIRCLE, SHAPE_TRIANGLE, SHAPE_REGULAR_POLYGON, SHAPE_POLYGON, SHAPE_RECTANGLE = range(5)


class EnvObstacle(C.Structure):
    _fields_ = [
        ("shape", C.c_int32),
        ("tile_row", C.c_int32),
        ("tile_col", C.c_int32),
        ("sides", C.c_uint32),
        ("n_points", C.c_uint32),
        ("points_xy", c_double_p),
        ("radius", C.c_double),
        ("angle_a", C.c_double),
        ("angle_b", C.c_double),
        ("width", C.c_double),
        ("height", C.c_double),
        ("rotation", C.c_double),
        ("translation_x", C.c_double),
        ("translation_y", C.c_double),
    ]


class EnvDesc(C.Structure):
    _fields_ = [
        ("n_rows", C.c_uint32),
        ("n_cols", C.c_uint32),
        ("tiles", C.POINTER(C.c_uint32)),
        ("tile_size", C.c_float),
        ("path_width", C.c_float),
        ("sdf_resolution", C.c_uint32),
        ("sdf_expansion", C.c_float),
        ("sdf_blur", C.c_float),
        ("n_obstacles", C.c_uint32),
        ("obstacles", C.POINTER(EnvObstacle)),
    ]


# every symbol include/mgx.h declares: name -> (restype, argtypes)
_V = C.c_void_p
SYMBOLS = {
    "mgx_world_create": (C.c_int, [C.POINTER(Params), C.POINTER(_V)]),
    "mgx_world_destroy": (C.c_int, [_V]),
    "mgx_last_error": (C.c_char_p, []),
    "mgx_set_stream": (C.c_int, [_V, _V]),
    "mgx_synchronize": (C.c_int, [_V]),
    "mgx_world_set_sdf": (C.c_int, [_V, _V, C.c_uint32, C.c_uint32, C.c_double, C.c_double]),
    "mgx_env_image_size": (C.c_int, [C.POINTER(EnvDesc), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mgx_env_to_image": (C.c_int, [C.POINTER(EnvDesc), C.c_uint32, C.c_float, C.c_void_p]),
    "mgx_env_to_sdf_image": (C.c_int, [C.POINTER(EnvDesc), C.c_uint32, C.c_float, C.c_float, C.c_void_p]),
    "mgx_world_set_environment": (C.c_int, [_V, C.POINTER(EnvDesc)]),
    "mgx_robot_add": (C.c_int, [_V, C.POINTER(RobotDesc), C.POINTER(C.c_int32)]),
    "mgx_robot_remove": (C.c_int, [_V, C.c_int32]),
    "mgx_ir_connect": (C.c_int, [_V, C.c_int32, C.c_int32, C.c_uint64]),
    "mgx_ir_disconnect": (C.c_int, [_V, C.c_int32, C.c_int32]),
    "mgx_set_enabled": (C.c_int, [_V, C.c_uint32]),
    "mgx_set_antenna": (C.c_int, [_V, C.c_int32, C.c_int32]),
    "mgx_set_idle": (C.c_int, [_V, C.c_int32, C.c_int32]),
    "mgx_set_antennas": (C.c_int, [_V, C.c_uint32, C.c_void_p, C.c_void_p]),
    "mgx_neighbours": (C.c_int, [_V, C.c_void_p, C.c_float, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64,
                                 C.POINTER(C.c_uint64)]),
    "mgx_update_topology": (C.c_int, [_V, C.c_void_p, C.c_float, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "mgx_connections": (C.c_int, [_V, C.c_int32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "mgx_iterate": (C.c_int, [_V, C.c_char_p, C.c_uint32]),
    "mgx_batch_begin": (C.c_int, [_V]),
    "mgx_batch_end": (C.c_int, [_V, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mgx_sweep": (C.c_int, [_V, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "mgx_internal_factor_iteration": (C.c_int, [_V, C.c_int32]),
    "mgx_internal_variable_iteration": (C.c_int, [_V, C.c_int32]),
    "mgx_external_factor_iteration": (C.c_int, [_V, C.c_int32]),
    "mgx_external_variable_iteration": (C.c_int, [_V, C.c_int32]),
    "mgx_change_prior": (C.c_int, [_V, C.c_int32, C.c_uint32, c_double_p]),
    "mgx_change_priors": (C.c_int, [_V, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), c_double_p]),
    "mgx_update_priors": (C.c_int, [_V, C.c_uint32, C.POINTER(C.c_int32), c_double_p, c_double_p, C.POINTER(C.c_uint8), C.c_double,
                          C.c_double]),
    "mgx_tick": (C.c_int, [_V, C.c_uint32, _V, _V, _V, _V, C.c_double, C.c_double, C.c_char_p, C.c_uint32]),  # (the four arrays as addresses: world.py _arg)
    "mgx_get_belief": (C.c_int, [_V, C.c_int32, C.c_uint32, c_double_p, c_double_p, c_double_p, c_double_p,
                                 C.POINTER(C.c_int32)]),
    "mgx_read_beliefs": (C.c_int, [_V, c_double_p, c_double_p, c_double_p]),
    "mgx_message_counts": (C.c_int, [_V, C.c_int32, C.POINTER(C.c_uint64)]),
    "mgx_note_change_priors": (C.c_int, [_V, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]),
    "mgx_read_means": (C.c_int, [_V, c_double_p]),
    "mgx_read_variable_means": (C.c_int, [_V, C.c_uint32, c_double_p]),
    "mgx_rccl_unique_id": (C.c_int, [C.c_char_p]),
    "mgx_halo_rccl_connect": (C.c_int, [_V, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_halo_rccl_disconnect": (C.c_int, [_V]),
    "mgx_halo_direct_setup": (C.c_int, [_V, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "mgx_halo_direct_connect": (C.c_int, [_V, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_halo_direct_exchange": (C.c_int, [_V, C.c_uint32]),
    "mgx_halo_direct_status": (C.c_int, [_V, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "mgx_halo_direct_disconnect": (C.c_int, [_V]),
    "mgx_halo_direct_setup_slots": (C.c_int, [_V, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "mgx_halo_ghost_slots": (C.c_int, [_V, C.c_uint32, C.c_void_p, C.c_void_p]),
    "mgx_robot_export": (C.c_int, [_V, C.c_int32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "mgx_robot_import": (C.c_int, [_V, C.c_int32, C.c_void_p, C.c_uint64]),
    "mgx_robot_release": (C.c_int, [_V, C.c_int32]),
    "mgx_halo_get_lists": (C.c_int, [_V, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mgx_halo_direct_connect_slots": (C.c_int, [_V, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_halo_resident_setup": (C.c_int, [_V, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                          C.c_void_p, C.POINTER(C.c_int32)]),
    "mgx_halo_resident_connect": (C.c_int, [_V, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_uint32]),
    "mgx_resident_outcome": (C.c_int, [_V, C.POINTER(C.c_int32)]),
    "mgx_resident_ready": (C.c_int, [_V, C.c_char_p, C.c_uint32, C.POINTER(C.c_int32)]),
    "mgx_resident_stats": (C.c_int, [_V, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "mgx_halo_resident_disconnect": (C.c_int, [_V]),
    "mgx_halo_resident_connect_peers": (C.c_int, [_V, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "mgx_halo_resident_aim": (C.c_int, [_V, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_ipc_export": (C.c_int, [C.c_void_p, C.c_char_p]),
    "mgx_ipc_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "mgx_ipc_close": (C.c_int, [C.c_void_p]),
    "mgx_reset_variables": (C.c_int, [_V, C.c_int32, c_double_p, C.c_uint32, C.c_double, C.c_double]),
    "mgx_reset_tracking_factors": (C.c_int, [_V, C.c_int32]),
    "mgx_mission_set": (C.c_int, [_V, C.c_int32, C.c_void_p]),
    "mgx_mission_tick": (C.c_int, [_V, C.c_float, C.c_uint32, C.POINTER(C.c_uint64), C.c_int32, C.c_void_p, C.c_double, C.c_double,
                                   C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "mgx_mission_tick_begin": (C.c_int, [_V, C.c_float, C.c_uint32, C.POINTER(C.c_uint64), C.c_int32, C.POINTER(C.c_uint32)]),
    "mgx_mission_run": (C.c_int, [_V, C.c_void_p]),
    "mgx_mission_tick_end": (C.c_int, [_V, C.c_void_p, C.c_double, C.c_double, C.c_char_p, C.c_uint32]),
    "mgx_mission_finished": (C.c_int, [_V, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "mgx_mission_translations": (C.c_int, [_V, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "mgx_mission_read": (C.c_int, [_V, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_num_robots": (C.c_int, [_V, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mgx_last_launch_count": (C.c_int, [_V, C.POINTER(C.c_uint32)]),
    "mgx_flush": (C.c_int, [_V]),
    "mgx_set_linger": (C.c_int, [_V, C.c_int32]),
    "mgx_linger_stats": (C.c_int, [_V, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "mgx_set_resident_launches": (C.c_int, [_V, C.c_int32]),
    "mgx_is_thawing": (C.c_int, [_V, C.POINTER(C.c_int32)]),
    "mgx_halo_words": (C.c_uint32, [C.c_uint32]),
    "mgx_halo_plan": (C.c_int, [_V, C.c_uint32, C.POINTER(C.c_int32), C.c_uint32, C.POINTER(C.c_int32)]),
    "mgx_halo_plan_from_connections": (C.c_int, [_V, C.POINTER(C.c_int32), C.c_uint32, C.c_int32, C.c_uint32, C.POINTER(C.c_uint32),
                                                 C.POINTER(C.c_uint32)]),
    "mgx_shard_partition": (C.c_int, [c_double_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32)]),
    "mgx_shard_plan_create": (C.c_int, [C.POINTER(C.c_int32), C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_uint32, C.c_int32,
                                        C.c_uint32, C.POINTER(_V)]),
    "mgx_shard_plan_destroy": (None, [_V]),
    "mgx_shard_plan_counts": (C.c_int, [_V] + [C.POINTER(C.c_uint32)] * 5),
    "mgx_shard_plan_get": (C.c_int, [_V, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]),
    "mgx_halo_pack": (C.c_int, [_V, _V]),
    "mgx_halo_unpack": (C.c_int, [_V, _V]),
    "mgx_euclidean_norm": (C.c_double, [c_double_p, C.c_uint32]),
    "mgx_l1_norm": (C.c_double, [c_double_p, C.c_uint32]),
    "mgx_normalize": (None, [c_double_p, C.c_uint32]),
    "mgx_det": (C.c_double, [c_double_p, C.c_uint32]),
    "mgx_inverse": (C.c_int, [c_double_p, C.c_uint32, c_double_p]),
    "mgx_mvn_from_information_and_precision": (C.c_int, [c_double_p, C.c_uint32, c_double_p, C.c_uint32, C.c_uint32, C.POINTER(_V)]),
    "mgx_mvn_from_mean_and_covariance": (C.c_int, [c_double_p, C.c_uint32, c_double_p, C.c_uint32, C.c_uint32, C.POINTER(_V)]),
    "mgx_mvn_destroy": (None, [_V]),
    "mgx_mvn_len": (C.c_uint32, [_V]),
    "mgx_mvn_get": (C.c_int, [_V, c_double_p, c_double_p, c_double_p]),
    "mgx_mvn_covariance": (C.c_int, [_V, c_double_p]),
    "mgx_mvn_update_information_vector": (C.c_int, [_V, c_double_p]),
    "mgx_mvn_update_precision_matrix": (C.c_int, [_V, c_double_p]),
    "mgx_mvn_set_information_vector": (C.c_int, [_V, c_double_p]),
    "mgx_mvn_set_precision_matrix": (C.c_int, [_V, c_double_p]),
    "mgx_mvn_add_assign_information_vector": (C.c_int, [_V, c_double_p]),
    "mgx_mvn_add_assign_precision_matrix": (C.c_int, [_V, c_double_p]),
    "mgx_mvn_update": (C.c_int, [_V]),
    "mgx_mvn_combine": (C.c_int, [_V, _V, C.c_int32, C.POINTER(_V)]),
    "mgx_mvn_combine_assign": (C.c_int, [_V, _V, C.c_int32]),
    "mgx_schedule": (C.c_int, [C.c_int32, C.c_uint8, C.c_uint8, C.c_char_p, C.c_uint32]),
    "mgx_variable_timesteps": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32]),
}

_libs = {}


def _preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  If
    libmgx.so pulled in the system copy first, a later `import torch` would load a SECOND HIP
    runtime into the process (its DT_NEEDED is spelled without the version, so the loader does
    not match it to the loaded SONAME) and one of the two then sees no GPU.  Loading torch's
    copy first makes both libmgx.so and torch resolve to the same runtime.  Without torch the
    system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib(fma=None):
    """The loaded C-ABI library; raises ``MgxError`` when it has not been built."""
    if fma is None:
        fma = os.environ.get("MGX_FMA", "0") == "1"
    key = bool(fma)
    if key not in _libs:
        path = FMA_LIB_PATH if key else os.environ.get("MGX_LIB", LIB_PATH)  # MGX_LIB: experiments only
        if not os.path.exists(path):
            raise MgxError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _preload_hip_runtime()
        L = C.CDLL(path)
        experiment = not key and "MGX_LIB" in os.environ  # an older / diagnostic build named by MGX_LIB may lack newer entry points
        for name, (res, args) in SYMBOLS.items():
            if experiment and not hasattr(L, name):
                continue
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _libs[key] = L
    return _libs[key]


def check(rc, L=None):
    if rc < 0:
        raise MgxError(f"mgx error {rc}: {(L or lib()).mgx_last_error().decode(errors='replace')}")
    return rc


def rccl_unique_id():
    buf = C.create_string_buffer(128)
    check(lib().mgx_rccl_unique_id(buf))
    return buf.raw


def ipc_export(dev_ptr):
    """64-byte hipIpc handle of a device allocation (mgx_ipc_export)."""
    buf = C.create_string_buffer(64)
    check(lib().mgx_ipc_export(C.c_void_p(dev_ptr), buf))
    return buf.raw


def ipc_open(handle):
    out = C.c_void_p()
    check(lib().mgx_ipc_open(handle, C.byref(out)))
    return out.value


def ipc_close(dev_ptr):
    check(lib().mgx_ipc_close(C.c_void_p(dev_ptr)))


def schedule(kind, n_internal, n_external):
    """GbpSchedule::schedule — list of step bytes (STEP_INTERNAL | STEP_EXTERNAL)."""
    buf = C.create_string_buffer(256)
    n = lib().mgx_schedule(int(kind), int(n_internal), int(n_external), buf, 256)
    if n < 0:
        raise ValueError("bad schedule arguments")
    return list(buf.raw[:n])


def variable_timesteps(lookahead_horizon, lookahead_multiple):
    """get_variable_timesteps (crates/magics/src/utils.rs:35-75)."""
    buf = (C.c_uint32 * 4096)()
    n = lib().mgx_variable_timesteps(int(lookahead_horizon), int(lookahead_multiple), buf, 4096)
    if n < 0:
        raise ValueError("bad arguments")
    return list(buf[:n])


class MissionRunDesc(C.Structure):
    """mgx_mission_run_desc (include/mgx.h)"""
    _fields_ = [("n_ticks", C.c_uint32), ("comms_radius", C.c_float), ("method", C.c_uint32), ("despawn_finished", C.c_int32),
                ("stop_when_all_finished", C.c_int32), ("n_steps", C.c_uint32), ("steps", C.c_char_p), ("max_speed", C.c_double),
                ("delta_t", C.c_double), ("failure_rate", C.c_double), ("wyrand_state", C.POINTER(C.c_uint64)),
                ("robot_number_next", C.POINTER(C.c_uint64)), ("created", C.POINTER(C.c_uint32)), ("deleted", C.POINTER(C.c_uint32)),
                ("n_finished", C.POINTER(C.c_uint32)), ("finished", C.POINTER(C.c_int32)), ("finished_capacity", C.c_uint32),
                ("finished_total", C.c_uint32), ("translations", C.POINTER(C.c_float)), ("antennas", C.POINTER(C.c_uint8)),
                ("ticks_done", C.c_uint32), ("reserved", C.c_uint32)]


class MissionDesc(C.Structure):
    """mgx_mission_desc (include/mgx.h)"""
    _fields_ = [("n_waypoints", C.c_uint32), ("reserved", C.c_uint32), ("waypoints_xy", c_double_p), ("reach_var", C.c_uint32),
                ("finish_var", C.c_uint32), ("reach_dist2", C.c_float), ("finish_dist2", C.c_float), ("translation", C.c_float * 3),
                ("reserved2", C.c_float), ("time_scale", C.c_double)]


def shard_partition(positions_xy, n_ranks):
    """mgx_shard_partition: owner rank of every robot (equal-count strips in (y, x) order)."""
    import numpy as np
    pos = np.ascontiguousarray(np.asarray(positions_xy, dtype=np.float64).reshape(-1, 2))
    owner = np.zeros(len(pos), dtype=np.int32)
    check(lib().mgx_shard_partition(pos.ctypes.data_as(c_double_p), len(pos), int(n_ranks), owner.ctypes.data_as(C.POINTER(C.c_int32))))
    return owner


def shard_plan(owner, conn_owner, conn_other, rank, n_ranks):
    """mgx_shard_plan_create / _get: dict(local, ghosts, connections, send_first, send_robots, recv_first, recv_robots)."""
    import numpy as np
    L = lib()
    i32p, u32p = C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
    owner = np.ascontiguousarray(owner, dtype=np.int32)
    ca, cb = np.ascontiguousarray(conn_owner, dtype=np.int32), np.ascontiguousarray(conn_other, dtype=np.int32)
    h = C.c_void_p()
    check(L.mgx_shard_plan_create(owner.ctypes.data_as(i32p), len(owner), ca.ctypes.data_as(i32p), cb.ctypes.data_as(i32p), len(ca), int(rank),
                                  int(n_ranks), C.byref(h)))
    try:
        n = [C.c_uint32() for _ in range(5)]
        check(L.mgx_shard_plan_counts(h, *[C.byref(x) for x in n]))
        nl, ng, nc, ns, nr = (x.value for x in n)
        out = dict(local=np.zeros(nl, np.int32), ghosts=np.zeros(ng, np.int32), connections=np.zeros(nc, np.uint32),
                   send_first=np.zeros(n_ranks + 1, np.uint32), send_robots=np.zeros(ns, np.int32),
                   recv_first=np.zeros(n_ranks + 1, np.uint32), recv_robots=np.zeros(nr, np.int32))
        check(L.mgx_shard_plan_get(h, out["local"].ctypes.data_as(i32p), out["ghosts"].ctypes.data_as(i32p), out["connections"].ctypes.data_as(u32p),
                                   out["send_first"].ctypes.data_as(u32p), out["send_robots"].ctypes.data_as(i32p),
                                   out["recv_first"].ctypes.data_as(u32p), out["recv_robots"].ctypes.data_as(i32p)))
    finally:
        L.mgx_shard_plan_destroy(h)
    return out
