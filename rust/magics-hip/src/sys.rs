//! Raw bindings of include/mgx.h: the structs and constants below by hand, the `extern "C"` block GENERATED from the
//! header by tools/gen_rust_sys.py (what `bindgen include/mgx.h` emits); tests/test_rust_shim.py fails on any drift.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct mgx_world { _private: [u8; 0] }

#[repr(C)]
#[derive(Clone, Copy)]
pub struct mgx_params {
    pub sigma_dynamics: f64,
    pub sigma_interrobot: f64,
    pub sigma_obstacle: f64,
    pub sigma_tracking: f64,
    pub safety_multiplier: f64,
    pub tracking_switch_padding: f64,
    pub tracking_attraction_distance: f64,
    pub enable_mask: u32,
    pub reserved: u32,
}

#[repr(C)]
pub struct mgx_robot_desc {
    pub k: u32,
    pub n_path: u32,
    pub mean0: *const f64,
    pub prior_diag: *const f64,
    pub dt: *const f64,
    pub path_xy: *const f32,
    pub radius: f64,
    pub order_key: u64,
    pub ghost: u32,
    pub reserved: u32,
}

pub const MGX_SHAPE_CIRCLE: i32 = 0;
pub const MGX_SHAPE_TRIANGLE: i32 = 1;
pub const MGX_SHAPE_REGULAR_POLYGON: i32 = 2;
pub const MGX_SHAPE_POLYGON: i32 = 3;
pub const MGX_SHAPE_RECTANGLE: i32 = 4;

/// gbp_environment::Obstacle as plain data (include/mgx.h)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct mgx_env_obstacle {
    pub shape: i32,
    pub tile_row: i32,
    pub tile_col: i32,
    pub sides: u32,
    pub n_points: u32,
    pub points_xy: *const f64,
    pub radius: f64,
    pub angle_a: f64,
    pub angle_b: f64,
    pub width: f64,
    pub height: f64,
    pub rotation: f64,
    pub translation_x: f64,
    pub translation_y: f64,
}

/// gbp_environment::Environment as plain data (include/mgx.h)
#[repr(C)]
pub struct mgx_env_desc {
    pub n_rows: u32,
    pub n_cols: u32,
    pub tiles: *const u32,
    pub tile_size: f32,
    pub path_width: f32,
    pub sdf_resolution: u32,
    pub sdf_expansion: f32,
    pub sdf_blur: f32,
    pub n_obstacles: u32,
    pub obstacles: *const mgx_env_obstacle,
}

/// a robot's mission handed to the device (include/mgx.h, mgx_mission_*)
#[repr(C)]
pub struct mgx_mission_desc {
    pub n_waypoints: u32,
    pub reserved: u32,
    pub waypoints_xy: *const f64,
    pub reach_var: u32,
    pub finish_var: u32,
    pub reach_dist2: f32,
    pub finish_dist2: f32,
    pub translation: [f32; 3],
    pub reserved2: f32,
    pub time_scale: f64,
}

/// many ticks in one call (include/mgx.h, mgx_mission_run)
#[repr(C)]
pub struct mgx_mission_run_desc {
    pub n_ticks: u32,
    pub comms_radius: f32,
    pub method: u32,
    pub despawn_finished: i32,
    pub stop_when_all_finished: i32,
    pub n_steps: u32,
    pub steps: *const u8,
    pub max_speed: f64,
    pub delta_t: f64,
    pub failure_rate: f64,
    pub wyrand_state: *mut u64,
    pub robot_number_next: *mut u64,
    pub created: *mut u32,
    pub deleted: *mut u32,
    pub n_finished: *mut u32,
    pub finished: *mut i32,
    pub finished_capacity: u32,
    pub finished_total: u32,
    pub translations: *mut f32,
    pub antennas: *mut u8,
    pub ticks_done: u32,
    pub reserved: u32,
}

/// opaque handles (include/mgx.h)
#[repr(C)]
pub struct mgx_mvn { _private: [u8; 0] }
#[repr(C)]
pub struct mgx_shard_plan { _private: [u8; 0] }

// BEGIN generated from include/mgx.h (tools/gen_rust_sys.py)
extern "C" {
    pub fn mgx_world_create(params: *const mgx_params, out: *mut *mut mgx_world) -> c_int;
    pub fn mgx_world_destroy(w: *mut mgx_world) -> c_int;
    pub fn mgx_last_error() -> *const c_char;
    pub fn mgx_set_stream(w: *mut mgx_world, hip_stream: *mut c_void) -> c_int;
    pub fn mgx_synchronize(w: *mut mgx_world) -> c_int;
    pub fn mgx_world_set_sdf(w: *mut mgx_world, rgb: *const u8, width: u32, height: u32, world_w: f64, world_h: f64) -> c_int;
    pub fn mgx_robot_add(w: *mut mgx_world, desc: *const mgx_robot_desc, robot_id: *mut i32) -> c_int;
    pub fn mgx_robot_remove(w: *mut mgx_world, robot: i32) -> c_int;
    pub fn mgx_ir_connect(w: *mut mgx_world, owner: i32, other: i32, first_robot_number: u64) -> c_int;
    pub fn mgx_ir_disconnect(w: *mut mgx_world, a: i32, b: i32) -> c_int;
    pub fn mgx_set_enabled(w: *mut mgx_world, kind_mask: u32) -> c_int;
    pub fn mgx_set_antenna(w: *mut mgx_world, robot: i32, active: i32) -> c_int;
    pub fn mgx_set_idle(w: *mut mgx_world, robot: i32, idle: i32) -> c_int;
    pub fn mgx_set_antennas(w: *mut mgx_world, n: u32, robots: *const i32, active: *const u8) -> c_int;
    pub fn mgx_neighbours(w: *mut mgx_world, positions_xyz: *const f32, radius: f32, method: u32, row_ptr: *mut i32, neighbours_out: *mut i32, capacity: u64, needed: *mut u64) -> c_int;
    pub fn mgx_update_topology(w: *mut mgx_world, positions_xyz: *const f32, radius: f32, method: u32, robot_number_next: *mut u64, stats: *mut u32) -> c_int;
    pub fn mgx_connections(w: *mut mgx_world, robot: i32, others: *mut i32, capacity: u32, n: *mut u32) -> c_int;
    pub fn mgx_iterate(w: *mut mgx_world, steps: *const u8, n: u32) -> c_int;
    pub fn mgx_batch_begin(w: *mut mgx_world) -> c_int;
    pub fn mgx_batch_end(w: *mut mgx_world, n_schedules: *mut u32, n_launches: *mut u32) -> c_int;
    pub fn mgx_last_launch_count(w: *mut mgx_world, n_launches: *mut u32) -> c_int;
    pub fn mgx_flush(w: *mut mgx_world) -> c_int;
    pub fn mgx_set_linger(w: *mut mgx_world, microseconds: i32) -> c_int;
    pub fn mgx_linger_stats(w: *mut mgx_world, launches: *mut u64, posts: *mut u64, reruns: *mut u64, ended_by_device: *mut u64) -> c_int;
    pub fn mgx_set_resident_launches(w: *mut mgx_world, enabled: i32) -> c_int;
    pub fn mgx_is_thawing(w: *mut mgx_world, thawing: *mut i32) -> c_int;
    pub fn mgx_sweep(w: *mut mgx_world, robot: i32, external_phases: u32, internal_phases: u32, n_internal: u32, hints: u32) -> c_int;
    pub fn mgx_internal_factor_iteration(w: *mut mgx_world, robot: i32) -> c_int;
    pub fn mgx_internal_variable_iteration(w: *mut mgx_world, robot: i32) -> c_int;
    pub fn mgx_external_factor_iteration(w: *mut mgx_world, robot: i32) -> c_int;
    pub fn mgx_external_variable_iteration(w: *mut mgx_world, robot: i32) -> c_int;
    pub fn mgx_change_prior(w: *mut mgx_world, robot: i32, var_ix: u32, mean: *const f64) -> c_int;
    pub fn mgx_change_priors(w: *mut mgx_world, n: u32, robots: *const i32, var_ix: *const u32, means: *const f64) -> c_int;
    pub fn mgx_reset_variables(w: *mut mgx_world, robot: i32, means: *const f64, n_means: u32, first_last_sigma: f64, inbetween_sigma: f64) -> c_int;
    pub fn mgx_reset_tracking_factors(w: *mut mgx_world, robot: i32) -> c_int;
    pub fn mgx_update_priors(w: *mut mgx_world, n: u32, robots: *const i32, waypoints_xy: *const f64, time_scale: *const f64, what: *const u8, max_speed: f64, delta_t: f64) -> c_int;
    pub fn mgx_tick(w: *mut mgx_world, n: u32, robots: *const i32, waypoints_xy: *const f64, time_scale: *const f64, what: *const u8, max_speed: f64, delta_t: f64, steps: *const u8, n_steps: u32) -> c_int;
    pub fn mgx_mission_set(w: *mut mgx_world, robot: i32, desc: *const mgx_mission_desc) -> c_int;
    pub fn mgx_mission_tick(w: *mut mgx_world, comms_radius: f32, method: u32, robot_number_next: *mut u64, despawn_finished: i32, antennas: *const u8, max_speed: f64, delta_t: f64, steps: *const u8, n_steps: u32, stats: *mut u32) -> c_int;
    pub fn mgx_mission_tick_begin(w: *mut mgx_world, comms_radius: f32, method: u32, robot_number_next: *mut u64, despawn_finished: i32, stats: *mut u32) -> c_int;
    pub fn mgx_mission_tick_end(w: *mut mgx_world, antennas: *const u8, max_speed: f64, delta_t: f64, steps: *const u8, n_steps: u32) -> c_int;
    pub fn mgx_mission_finished(w: *mut mgx_world, robots: *mut i32, capacity: u32, n: *mut u32) -> c_int;
    pub fn mgx_mission_run(w: *mut mgx_world, desc: *mut mgx_mission_run_desc) -> c_int;
    pub fn mgx_mission_translations(w: *mut mgx_world, translations: *mut f32, capacity_robots: u32, n_robots: *mut u32) -> c_int;
    pub fn mgx_mission_read(w: *mut mgx_world, translations: *mut f32, targets: *mut i32, finished_tick: *mut i64) -> c_int;
    pub fn mgx_get_belief(w: *mut mgx_world, robot: i32, var_ix: u32, eta: *mut f64, lam: *mut f64, mean: *mut f64, cov: *mut f64, valid: *mut i32) -> c_int;
    pub fn mgx_read_beliefs(w: *mut mgx_world, eta: *mut f64, lam: *mut f64, means: *mut f64) -> c_int;
    pub fn mgx_read_means(w: *mut mgx_world, means: *mut f64) -> c_int;
    pub fn mgx_read_variable_means(w: *mut mgx_world, var_ix: u32, means: *mut f64) -> c_int;
    pub fn mgx_num_robots(w: *mut mgx_world, n_robots: *mut u32, n_variables: *mut u32) -> c_int;
    pub fn mgx_message_counts(w: *mut mgx_world, robot: i32, counts: *mut u64) -> c_int;
    pub fn mgx_note_change_priors(w: *mut mgx_world, n: u32, robots: *const i32, var_ix: *const u32) -> c_int;
    pub fn mgx_halo_words(k: u32) -> u32;
    pub fn mgx_halo_plan(w: *mut mgx_world, n_send: u32, send_robots: *const i32, n_recv: u32, recv_ghosts: *const i32) -> c_int;
    pub fn mgx_halo_plan_from_connections(w: *mut mgx_world, rank_of: *const i32, n_robots: u32, my_rank: i32, n_ranks: u32, send_counts: *mut u32, recv_counts: *mut u32) -> c_int;
    pub fn mgx_shard_partition(positions_xy: *const f64, n_robots: u32, n_ranks: u32, owner: *mut i32) -> c_int;
    pub fn mgx_shard_plan_create(owner: *const i32, n_robots: u32, conn_owner: *const i32, conn_other: *const i32, n_conns: u32, rank: i32, n_ranks: u32, out: *mut *mut mgx_shard_plan) -> c_int;
    pub fn mgx_shard_plan_destroy(plan: *mut mgx_shard_plan);
    pub fn mgx_shard_plan_counts(plan: *const mgx_shard_plan, n_local: *mut u32, n_ghosts: *mut u32, n_connections: *mut u32, n_send: *mut u32, n_recv: *mut u32) -> c_int;
    pub fn mgx_shard_plan_get(plan: *const mgx_shard_plan, local: *mut i32, ghosts: *mut i32, connections: *mut u32, send_first: *mut u32, send_robots: *mut i32, recv_first: *mut u32, recv_robots: *mut i32) -> c_int;
    pub fn mgx_halo_pack(w: *mut mgx_world, dev_buf: *mut c_void) -> c_int;
    pub fn mgx_halo_unpack(w: *mut mgx_world, dev_buf: *const c_void) -> c_int;
    pub fn mgx_robot_export(w: *mut mgx_world, robot: i32, buf: *mut c_void, capacity: u64, bytes: *mut u64) -> c_int;
    pub fn mgx_robot_import(w: *mut mgx_world, robot: i32, buf: *const c_void, bytes: u64) -> c_int;
    pub fn mgx_robot_release(w: *mut mgx_world, robot: i32) -> c_int;
    pub fn mgx_rccl_unique_id(id: *mut u8) -> c_int;
    pub fn mgx_halo_rccl_connect(w: *mut mgx_world, id: *const u8, n_ranks: u32, rank: u32, n_peers: u32, peer_rank: *const u32, send_first: *const u32, recv_first: *const u32) -> c_int;
    pub fn mgx_halo_rccl_disconnect(w: *mut mgx_world) -> c_int;
    pub fn mgx_halo_direct_setup(w: *mut mgx_world, n_sources: u32, recv_base: *mut *mut c_void, flag_base: *mut *mut c_void) -> c_int;
    pub fn mgx_halo_direct_connect(w: *mut mgx_world, n_peers: u32, send_first: *const u32, peer_recv_base: *const *mut c_void, peer_recv_records: *const u64, peer_record_offset: *const u64, peer_flag_slot: *const *mut c_void) -> c_int;
    pub fn mgx_halo_direct_exchange(w: *mut mgx_world, what: u32) -> c_int;
    pub fn mgx_halo_direct_status(w: *mut mgx_world, exchanges: *mut u64, failed_exchange: *mut u64) -> c_int;
    pub fn mgx_halo_direct_disconnect(w: *mut mgx_world) -> c_int;
    pub fn mgx_halo_direct_setup_slots(w: *mut mgx_world, n_sources: u32, slot_capacity: u32, recv_base: *mut *mut c_void, flag_base: *mut *mut c_void) -> c_int;
    pub fn mgx_halo_ghost_slots(w: *mut mgx_world, n: u32, robots: *const i32, slots: *mut i32) -> c_int;
    pub fn mgx_halo_get_lists(w: *mut mgx_world, send_robots: *mut i32, send_capacity: u32, recv_robots: *mut i32, recv_capacity: u32, n_send: *mut u32, n_recv: *mut u32) -> c_int;
    pub fn mgx_halo_direct_connect_slots(w: *mut mgx_world, n_peers: u32, send_first: *const u32, peer_recv_base: *const *mut c_void, peer_slot_capacity: *const u64, entry_slot: *const u32, peer_flag_slot: *const *mut c_void) -> c_int;
    pub fn mgx_halo_resident_setup(w: *mut mgx_world, area_base: *mut *mut c_void, n_ghost_slots: *mut u32, parity: *mut u32, segment_count: *mut u64, recv_slots: *mut i32, eligible: *mut i32) -> c_int;
    pub fn mgx_halo_resident_connect(w: *mut mgx_world, n_targets: u32, robots: *const i32, peer_area_base: *const *mut c_void, peer_ghost_slots: *const u32, peer_slot: *const u32, peer_parity: *const u32, peer_segment_count: *const u64, coordinator_area: *mut c_void, n_ranks: u32) -> c_int;
    pub fn mgx_halo_resident_disconnect(w: *mut mgx_world) -> c_int;
    pub fn mgx_halo_resident_connect_peers(w: *mut mgx_world, n_peers: u32, peer_area_base: *const *mut c_void, peer_ghost_slots: *const u32, peer_parity: *const u32, peer_segment_count: *const u64, coordinator_area: *mut c_void, n_ranks: u32) -> c_int;
    pub fn mgx_halo_resident_aim(w: *mut mgx_world, n_targets: u32, robots: *const i32, peer_index: *const u32, peer_slot: *const u32) -> c_int;
    pub fn mgx_resident_outcome(w: *mut mgx_world, outcome: *mut i32) -> c_int;
    pub fn mgx_resident_ready(w: *mut mgx_world, steps: *const u8, n: u32, ready: *mut i32) -> c_int;
    pub fn mgx_resident_stats(w: *mut mgx_world, launches: *mut u64, declined: *mut u64, backoff: *mut u32) -> c_int;
    pub fn mgx_ipc_export(dev_ptr: *const c_void, handle: *mut u8) -> c_int;
    pub fn mgx_ipc_open(handle: *const u8, dev_ptr: *mut *mut c_void) -> c_int;
    pub fn mgx_ipc_close(dev_ptr: *mut c_void) -> c_int;
    pub fn mgx_euclidean_norm(x: *const f64, n: u32) -> f64;
    pub fn mgx_l1_norm(x: *const f64, n: u32) -> f64;
    pub fn mgx_normalize(x: *mut f64, n: u32);
    pub fn mgx_det(a: *const f64, n: u32) -> f64;
    pub fn mgx_inverse(a: *const f64, n: u32, out: *mut f64) -> c_int;
    pub fn mgx_mvn_from_information_and_precision(information: *const f64, len: u32, precision: *const f64, rows: u32, cols: u32, out: *mut *mut mgx_mvn) -> c_int;
    pub fn mgx_mvn_from_mean_and_covariance(mean: *const f64, len: u32, covariance: *const f64, rows: u32, cols: u32, out: *mut *mut mgx_mvn) -> c_int;
    pub fn mgx_mvn_destroy(m: *mut mgx_mvn);
    pub fn mgx_mvn_len(m: *const mgx_mvn) -> u32;
    pub fn mgx_mvn_get(m: *const mgx_mvn, information: *mut f64, precision: *mut f64, mean: *mut f64) -> c_int;
    pub fn mgx_mvn_covariance(m: *const mgx_mvn, covariance: *mut f64) -> c_int;
    pub fn mgx_mvn_update_information_vector(m: *mut mgx_mvn, value: *const f64) -> c_int;
    pub fn mgx_mvn_update_precision_matrix(m: *mut mgx_mvn, value: *const f64) -> c_int;
    pub fn mgx_mvn_set_information_vector(m: *mut mgx_mvn, value: *const f64) -> c_int;
    pub fn mgx_mvn_set_precision_matrix(m: *mut mgx_mvn, value: *const f64) -> c_int;
    pub fn mgx_mvn_add_assign_information_vector(m: *mut mgx_mvn, value: *const f64) -> c_int;
    pub fn mgx_mvn_add_assign_precision_matrix(m: *mut mgx_mvn, value: *const f64) -> c_int;
    pub fn mgx_mvn_update(m: *mut mgx_mvn) -> c_int;
    pub fn mgx_mvn_combine(a: *const mgx_mvn, b: *const mgx_mvn, op: i32, out: *mut *mut mgx_mvn) -> c_int;
    pub fn mgx_mvn_combine_assign(a: *mut mgx_mvn, b: *const mgx_mvn, op: i32) -> c_int;
    pub fn mgx_env_image_size(env: *const mgx_env_desc, resolution: u32, width: *mut u32, height: *mut u32) -> c_int;
    pub fn mgx_env_to_image(env: *const mgx_env_desc, resolution: u32, expansion: f32, rgb: *mut u8) -> c_int;
    pub fn mgx_env_to_sdf_image(env: *const mgx_env_desc, resolution: u32, expansion: f32, blur_percent: f32, rgb: *mut u8) -> c_int;
    pub fn mgx_world_set_environment(w: *mut mgx_world, env: *const mgx_env_desc) -> c_int;
    pub fn mgx_schedule(kind: i32, n_internal: u8, n_external: u8, steps: *mut u8, capacity: u32) -> c_int;
    pub fn mgx_variable_timesteps(lookahead_horizon: u32, lookahead_multiple: u32, timesteps: *mut u32, capacity: u32) -> c_int;
}
// END generated
