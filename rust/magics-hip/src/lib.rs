//! `FactorGraph`-shaped shim over the MI355X GBP engine (libmgx.so).
//!
//! Mirrors the part of `crates/magics/src/factorgraph/factorgraph.rs` that
//! `crates/magics/src/planner/robot.rs` calls.  Graphs live on the device inside one `World`;
//! a `FactorGraph` is `(world, robot id)`.  Messages are routed on the device, so the methods
//! that return `Vec<…Message>` in the reference return empty vectors here.
//! NOTE: delivered as source; the build image has no Rust toolchain.
pub mod sys;

use std::ffi::CStr;
use std::sync::Arc;

#[derive(Debug)]
pub struct MgxError(pub i32, pub String);

fn check(rc: i32) -> Result<(), MgxError> {
    if rc >= 0 {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(sys::mgx_last_error()) }.to_string_lossy().into_owned();
    Err(MgxError(rc, msg))
}

pub struct World {
    raw: *mut sys::mgx_world,
}
unsafe impl Send for World {}
unsafe impl Sync for World {} // thread-compatible: callers serialise access, like Bevy's exclusive `&mut`

impl World {
    pub fn new(params: sys::mgx_params) -> Result<Arc<Self>, MgxError> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { sys::mgx_world_create(&params, &mut raw) })?;
        Ok(Arc::new(Self { raw }))
    }
    pub fn set_sdf(&self, rgb: &[u8], width: u32, height: u32, world_w: f64, world_h: f64) -> Result<(), MgxError> {
        assert_eq!(rgb.len(), (width * height * 3) as usize);
        check(unsafe { sys::mgx_world_set_sdf(self.raw, rgb.as_ptr(), width, height, world_w, world_h) })
    }
    /// `env_to_png::env_to_sdf_image` with the environment's own sdf settings + the `Sdf` resource handed
    /// to every `ObstacleFactor` (simulation_loader.rs:154-162, robot.rs:1259-1285): rasterised and
    /// blurred on the device.  `desc` must point at live tile / obstacle arrays for the call.
    pub fn set_environment(&self, desc: &sys::mgx_env_desc) -> Result<(), MgxError> {
        check(unsafe { sys::mgx_world_set_environment(self.raw, desc) })
    }
    /// `update_robot_neighbours` + `delete_interrobot_factors` + `create_interrobot_factors`
    /// (robot.rs:1362-1586) in one call: `translations` are the robots' `Transform::translation`
    /// in robot-id order, `robot_number` is the `RobotNumberGenerator` state.  Returns
    /// (connections created, pairs deleted).
    pub fn update_topology(&self, translations: &[[f32; 3]], radius: f32, robot_number: &mut u64) -> Result<(u32, u32), MgxError> {
        let mut stats = [0u32; 2];
        check(unsafe { sys::mgx_update_topology(self.raw, translations.as_ptr().cast(), radius, 0, robot_number, stats.as_mut_ptr()) })?;
        Ok((stats[0], stats[1]))
    }
    /// `FactorGraph::change_factor_enabled` for every graph (factorgraph.rs:1529-1539)
    pub fn change_factor_enabled(&self, kind_mask: u32) -> Result<(), MgxError> {
        check(unsafe { sys::mgx_set_enabled(self.raw, kind_mask) })
    }
    /// `update_failed_comms` (robot.rs:1593-1601): the Bernoulli draws stay with the caller's PRNG
    pub fn set_antennas(&self, robots: &[i32], active: &[u8]) -> Result<(), MgxError> {
        assert_eq!(robots.len(), active.len());
        check(unsafe { sys::mgx_set_antennas(self.raw, robots.len() as u32, robots.as_ptr(), active.as_ptr()) })
    }
    /// `RobotBundle::new` (robot.rs:1134-1356)
    #[allow(clippy::too_many_arguments)]
    pub fn add_robot(self: &Arc<Self>, mean0: &[[f64; 4]], prior_diag: &[f64], dt: &[f64], radius: f64,
                     path: Option<&[[f32; 2]]>, order_key: u64) -> Result<FactorGraph, MgxError> {
        let k = mean0.len();
        assert!(prior_diag.len() == k && dt.len() + 1 == k);
        let desc = sys::mgx_robot_desc {
            k: k as u32,
            n_path: path.map_or(0, |p| p.len() as u32),
            mean0: mean0.as_ptr().cast(),
            prior_diag: prior_diag.as_ptr(),
            dt: dt.as_ptr(),
            path_xy: path.map_or(std::ptr::null(), |p| p.as_ptr().cast()),
            radius,
            order_key,
            ghost: 0,
            reserved: 0,
        };
        let mut id = -1;
        check(unsafe { sys::mgx_robot_add(self.raw, &desc, &mut id) })?;
        Ok(FactorGraph { world: self.clone(), robot: id })
    }
    /// `create_interrobot_factors`, one direction (robot.rs:1500-1585)
    pub fn ir_connect(&self, owner: &FactorGraph, other: &FactorGraph, first_robot_number: u64) -> Result<(), MgxError> {
        check(unsafe { sys::mgx_ir_connect(self.raw, owner.robot, other.robot, first_robot_number) })
    }
    /// `delete_interrobot_factors` (robot.rs:1386-1439)
    pub fn ir_disconnect(&self, a: &FactorGraph, b: &FactorGraph) -> Result<(), MgxError> {
        check(unsafe { sys::mgx_ir_disconnect(self.raw, a.robot, b.robot) })
    }
    /// `iterate_gbp_v2` (robot.rs:1769-1861): `steps[i]` = internal | external << 1
    pub fn iterate(&self, steps: &[u8]) -> Result<(), MgxError> {
        check(unsafe { sys::mgx_iterate(self.raw, steps.as_ptr(), steps.len() as u32) })
    }
    /// Several `iterate` calls, one submission (`mgx_batch_begin` / `mgx_batch_end`): the schedules issued inside `f` are recorded and
    /// submitted together, merged into as few resident launches as their segments fit; same results, bit for bit.
    /// Returns (schedules recorded, sweep-kernel launches they were submitted as).
    pub fn batch<F: FnOnce(&Self) -> Result<(), MgxError>>(&self, f: F) -> Result<(u32, u32), MgxError> {
        check(unsafe { sys::mgx_batch_begin(self.raw) })?;
        let r = f(self);
        let (mut schedules, mut launches) = (0u32, 0u32);
        let e = check(unsafe { sys::mgx_batch_end(self.raw, &mut schedules, &mut launches) });
        r?;
        e?;
        Ok((schedules, launches))
    }
    /// How the last `iterate` / `tick` ran: sweep-kernel launches (1 = the whole schedule as one resident launch).
    pub fn last_launch_count(&self) -> Result<u32, MgxError> {
        let mut n = 0u32;
        check(unsafe { sys::mgx_last_launch_count(self.raw, &mut n) })?;
        Ok(n)
    }
    /// Everything issued so far is enqueued: a lingering launch is told to end, without waiting for the stream.
    pub fn flush(&self) -> Result<(), MgxError> {
        check(unsafe { sys::mgx_flush(self.raw) })
    }
    /// How long a resident launch waits for the next schedule before it ends (microseconds; 0: never, negative: the default).
    pub fn set_linger(&self, microseconds: i32) -> Result<(), MgxError> {
        check(unsafe { sys::mgx_set_linger(self.raw, microseconds) })
    }
    /// (resident launches so far, those declined before they wrote anything, what is left of the back-off)
    pub fn resident_stats(&self) -> Result<(u64, u64, u32), MgxError> {
        let (mut launches, mut declined, mut backoff) = (0u64, 0u64, 0u32);
        check(unsafe { sys::mgx_resident_stats(self.raw, &mut launches, &mut declined, &mut backoff) })?;
        Ok((launches, declined, backoff))
    }
    /// One FixedUpdate of the planner chain (robot.rs:86-103): `update_prior_of_horizon_state` +
    /// `update_prior_of_current_state_v3` for the listed robots, then `iterate_gbp_v2` over `steps`.
    /// `what[i]`: bit 0 = horizon prior, bit 1 = current prior.
    #[allow(clippy::too_many_arguments)]
    pub fn tick(&self, robots: &[i32], waypoints_xy: &[[f64; 2]], time_scale: &[f64], what: &[u8],
                max_speed: f64, delta_t: f64, steps: &[u8]) -> Result<(), MgxError> {
        assert!(robots.len() == waypoints_xy.len() && robots.len() == time_scale.len() && robots.len() == what.len());
        check(unsafe {
            sys::mgx_tick(self.raw, robots.len() as u32, robots.as_ptr(), waypoints_xy.as_ptr().cast(), time_scale.as_ptr(),
                          what.as_ptr(), max_speed, delta_t, steps.as_ptr(), steps.len() as u32)
        })
    }
    /// single prior changes batched (`FactorGraph::change_prior_of_variable`, factorgraph.rs:494-528)
    pub fn change_priors(&self, robots: &[i32], vars: &[u32], means: &[[f64; 4]]) -> Result<(), MgxError> {
        assert!(robots.len() == vars.len() && vars.len() == means.len());
        check(unsafe { sys::mgx_change_priors(self.raw, robots.len() as u32, robots.as_ptr(), vars.as_ptr(), means.as_ptr().cast()) })
    }
}

/// Owner rank of every robot for a node of `n_ranks` GPUs (`mgx_shard_partition`: equal-count strips in (y, x) order).
pub fn shard_partition(positions_xy: &[[f64; 2]], n_ranks: u32) -> Result<Vec<i32>, MgxError> {
    let mut owner = vec![0i32; positions_xy.len()];
    check(unsafe { sys::mgx_shard_partition(positions_xy.as_ptr().cast(), positions_xy.len() as u32, n_ranks, owner.as_mut_ptr()) })?;
    Ok(owner)
}

/// One rank's view of a sharded world (`mgx_shard_plan_*`): local robots, ghosts, the connections evaluated here and the
/// per-peer send / receive lists in the order `mgx_halo_plan` and the transports expect.
pub struct ShardPlan {
    pub local: Vec<i32>,
    pub ghosts: Vec<i32>,
    pub connections: Vec<u32>,
    pub send_first: Vec<u32>,
    pub send_robots: Vec<i32>,
    pub recv_first: Vec<u32>,
    pub recv_robots: Vec<i32>,
}
impl ShardPlan {
    pub fn new(owner: &[i32], conn_owner: &[i32], conn_other: &[i32], rank: i32, n_ranks: u32) -> Result<Self, MgxError> {
        assert_eq!(conn_owner.len(), conn_other.len());
        let mut raw = std::ptr::null_mut();
        check(unsafe {
            sys::mgx_shard_plan_create(owner.as_ptr(), owner.len() as u32, conn_owner.as_ptr(), conn_other.as_ptr(), conn_owner.len() as u32,
                                       rank, n_ranks, &mut raw)
        })?;
        let (mut nl, mut ng, mut nc, mut ns, mut nr) = (0u32, 0u32, 0u32, 0u32, 0u32);
        let rc = unsafe { sys::mgx_shard_plan_counts(raw, &mut nl, &mut ng, &mut nc, &mut ns, &mut nr) };
        let mut p = ShardPlan {
            local: vec![0; nl as usize], ghosts: vec![0; ng as usize], connections: vec![0; nc as usize],
            send_first: vec![0; n_ranks as usize + 1], send_robots: vec![0; ns as usize],
            recv_first: vec![0; n_ranks as usize + 1], recv_robots: vec![0; nr as usize],
        };
        let rc2 = if rc >= 0 {
            unsafe {
                sys::mgx_shard_plan_get(raw, p.local.as_mut_ptr(), p.ghosts.as_mut_ptr(), p.connections.as_mut_ptr(), p.send_first.as_mut_ptr(),
                                        p.send_robots.as_mut_ptr(), p.recv_first.as_mut_ptr(), p.recv_robots.as_mut_ptr())
            }
        } else { rc };
        unsafe { sys::mgx_shard_plan_destroy(raw) };
        check(rc2)?;
        Ok(p)
    }
}

impl Drop for World {
    fn drop(&mut self) {
        unsafe { sys::mgx_world_destroy(self.raw) };
    }
}

/// Stand-in for the reference's `FactorGraph` component.
pub struct FactorGraph {
    world: Arc<World>,
    robot: i32,
}

pub struct Belief {
    pub information_vector: [f64; 4],
    pub precision_matrix: [f64; 16],
    pub mean: [f64; 4],
    pub covariance_matrix: [f64; 16],
    pub valid: bool,
}

impl FactorGraph {
    pub fn id(&self) -> i32 { self.robot }
    pub fn internal_factor_iteration(&mut self) -> Result<(), MgxError> {
        check(unsafe { sys::mgx_internal_factor_iteration(self.world.raw, self.robot) })
    }
    pub fn internal_variable_iteration(&mut self) -> Result<(), MgxError> {
        check(unsafe { sys::mgx_internal_variable_iteration(self.world.raw, self.robot) })
    }
    /// factorgraph.rs:494-528; the returned vector is empty because routing happens on the device
    pub fn change_prior_of_variable(&mut self, variable_index: u32, new_mean: [f64; 4]) -> Result<Vec<()>, MgxError> {
        check(unsafe { sys::mgx_change_prior(self.world.raw, self.robot, variable_index, new_mean.as_ptr()) })?;
        Ok(Vec::new())
    }
    /// `FactorGraph::reset_variables(&means, first_last_sigma, inbetween_sigma)` (factorgraph.rs:1541-1564); the
    /// reference's call is `(means, 1e30, Float::INFINITY)` (robot.rs:768)
    pub fn reset_variables(&mut self, means: &[[f64; 4]], first_last_sigma: f64, inbetween_sigma: f64) -> Result<(), MgxError> {
        // the slice's length travels with it: the library rejects a count that is not the graph's number of variables
        // (the reference asserts it, factorgraph.rs:1548), so a short slice is an error, never an out-of-bounds read
        check(unsafe {
            sys::mgx_reset_variables(self.world.raw, self.robot, means.as_ptr().cast(), means.len() as u32, first_last_sigma, inbetween_sigma)
        })
    }
    /// `FactorGraph::reset_tracking_factors` (factorgraph.rs:1566-1590)
    pub fn reset_tracking_factors(&mut self) -> Result<(), MgxError> {
        check(unsafe { sys::mgx_reset_tracking_factors(self.world.raw, self.robot) })
    }
    /// `messages_sent()` / `messages_received()` (factorgraph.rs:876-890): [sent internal, sent external, received
    /// internal, received external]
    pub fn message_counts(&self) -> Result<[u64; 4], MgxError> {
        let mut c = [0u64; 4];
        check(unsafe { sys::mgx_message_counts(self.world.raw, self.robot, c.as_mut_ptr()) })?;
        Ok(c)
    }
    pub fn set_antenna(&mut self, active: bool) -> Result<(), MgxError> {
        check(unsafe { sys::mgx_set_antenna(self.world.raw, self.robot, active as i32) })
    }
    pub fn set_idle(&mut self, idle: bool) -> Result<(), MgxError> {
        check(unsafe { sys::mgx_set_idle(self.world.raw, self.robot, idle as i32) })
    }
    /// `VariableNode.belief` (variable.rs:40-54)
    pub fn belief(&self, variable_index: u32) -> Result<Belief, MgxError> {
        let mut b = Belief { information_vector: [0.0; 4], precision_matrix: [0.0; 16], mean: [0.0; 4], covariance_matrix: [0.0; 16], valid: false };
        let mut valid = 0;
        check(unsafe {
            sys::mgx_get_belief(self.world.raw, self.robot, variable_index, b.information_vector.as_mut_ptr(),
                                b.precision_matrix.as_mut_ptr(), b.mean.as_mut_ptr(), b.covariance_matrix.as_mut_ptr(), &mut valid)
        })?;
        b.valid = valid != 0;
        Ok(b)
    }
}
