/* mgx.h — C ABI of the MI355X-native GBP message-passing engine.
 *
 * Drop-in boundary for the factor-graph inner loop of AU-Master-Thesis/magics.
 * The reference has no FFI seam; the seam is the public Rust API of `FactorGraph`
 * (crates/magics/src/factorgraph/factorgraph.rs:76) as driven by
 * crates/magics/src/planner/robot.rs.  Because GPU execution is batched over all
 * robots, the ABI is world-level: one `mgx_world` owns every robot's factor graph
 * (device resident, SoA) and each entry point names the reference call it replaces.
 *
 * Conventions: plain C, caller owns every buffer it passes, the library owns all
 * device memory.  Every call returns an `int` status (MGX_OK == 0, negative =
 * error, never aborts).  A world is thread-compatible (external synchronisation).
 * All floating point is f64 (crates/gbp_linalg/src/lib.rs:31), DOFS = 4
 * (crates/magics/src/factorgraph/mod.rs:21).  Matrices are row-major.
 *
 * There is NO CPU fallback: every compute entry point fails with
 * MGX_ERR_NO_DEVICE when no HIP device is usable.
 */
#ifndef MGX_H
#define MGX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGX_DOFS 4

/* status codes */
#define MGX_OK 0
#define MGX_ERR_INVALID (-1)   /* bad argument (null pointer, index out of range, K < 2 ...) */
#define MGX_ERR_NO_DEVICE (-2) /* no usable HIP device / HIP runtime error at init */
#define MGX_ERR_HIP (-3)       /* HIP runtime call failed (see mgx_last_error) */
#define MGX_ERR_STATE (-4)     /* call not valid in the current state */
#define MGX_ERR_NOMEM (-5)

/* factor kinds, bit positions of mgx_params.enable_mask
 * (gbp_config FactorsEnabledSection, crates/gbp_config/src/lib.rs:454-494) */
#define MGX_FACTOR_DYNAMIC 1u
#define MGX_FACTOR_INTERROBOT 2u
#define MGX_FACTOR_OBSTACLE 4u
#define MGX_FACTOR_TRACKING 8u

/* one schedule step, bits of the uint8 passed to mgx_iterate
 * (gbp_schedule::GbpScheduleAtIteration, crates/gbp_schedule/src/schedules/mod.rs:59-63) */
#define MGX_STEP_INTERNAL 1u
#define MGX_STEP_EXTERNAL 2u

/* schedule kinds (crates/gbp_schedule/src/schedules/ *.rs) */
#define MGX_SCHEDULE_CENTERED 0
#define MGX_SCHEDULE_SOON_AS_POSSIBLE 1
#define MGX_SCHEDULE_LATE_AS_POSSIBLE 2
#define MGX_SCHEDULE_INTERLEAVE_EVENLY 3
#define MGX_SCHEDULE_HALF_BEGINNING_HALF_END 4

typedef struct mgx_world mgx_world;

/* World-wide GBP parameters: the values `RobotBundle::new` and
 * `create_interrobot_factors` read from `Config` (robot.rs:1134-1356,1441-1586).
 * Sigmas are the f32 config values already widened to f64 (`Float::from(f32)`). */
typedef struct mgx_params {
    double sigma_dynamics;           /* config.gbp.sigma_factor_dynamics   */
    double sigma_interrobot;         /* config.gbp.sigma_factor_interrobot */
    double sigma_obstacle;           /* config.gbp.sigma_factor_obstacle   */
    double sigma_tracking;           /* config.gbp.sigma_factor_tracking   */
    double safety_multiplier;        /* config.robot.inter_robot_safety_distance_multiplier */
    double tracking_switch_padding;  /* config.gbp.tracking.switch_padding (f32 widened)      */
    double tracking_attraction_distance; /* config.gbp.tracking.attraction_distance          */
    uint32_t enable_mask;            /* MGX_FACTOR_* bits: config.gbp.factors_enabled        */
    uint32_t reserved;
} mgx_params;

/* One robot = one reference `FactorGraph` built by `RobotBundle::new`
 * (robot.rs:1134-1356): K variables, K-1 dynamic factors, K-2 obstacle factors and
 * K-2 tracking factors (on variables 1..K-2). */
typedef struct mgx_robot_desc {
    uint32_t K;               /* number of variables (>= 2)                                  */
    uint32_t n_path;          /* tracking polyline points (0 = none)                         */
    const double *mean0;      /* [K][4] initial variable means (robot.rs:1187-1218)          */
    const double *prior_diag; /* [K] diagonal of the prior precision; non-finite => 0
                                 (variable.rs:146-148); the reference uses 1e30 / +inf      */
    const double *dt;         /* [K-1] dynamic-factor delta_t (robot.rs:1232,1240)           */
    const float *path_xy;     /* [n_path][2] f32 waypoints for the tracking factors or NULL  */
    double radius;            /* robot radius; d_safe = safety_multiplier * radius           */
    uint64_t order_key;       /* total order of graphs = Bevy Entity order (id.rs:19-117);
                                 must be unique per world (across ranks when sharded)       */
    uint32_t ghost;           /* 1 = halo copy of a robot owned by another rank: only its
                                 variable->factor snapshots exist, filled by mgx_halo_unpack */
    uint32_t reserved;
} mgx_robot_desc;

/* ---- lifecycle ------------------------------------------------------------------- */
int mgx_world_create(const mgx_params *params, mgx_world **out);
int mgx_world_destroy(mgx_world *w);
/* text of the last error on this thread (never NULL) */
const char *mgx_last_error(void);
/* HIP stream every kernel / copy of this world is enqueued on (NULL = default stream) */
int mgx_set_stream(mgx_world *w, void *hip_stream);
int mgx_synchronize(mgx_world *w);

/* Obstacle "SDF" image sampled by ObstacleFactor::measure (factor/obstacle.rs:141-188):
 * interleaved RGB u8, row-major, `world_w x world_h` world units (robot.rs:1259-1264).
 * Only the red channel is kept on the device. */
int mgx_world_set_sdf(mgx_world *w, const uint8_t *rgb, uint32_t width, uint32_t height,
                      double world_w, double world_h);

/* ---- topology (FactorGraph::add_variable/add_factor/add_*_edge, robot.rs) ---------- */
/* RobotBundle::new (robot.rs:1134-1356). */
int mgx_robot_add(mgx_world *w, const mgx_robot_desc *desc, int32_t *robot_id);
/* Entity despawn (robot.rs:2172, despawn_entity_after): the robot leaves every query — never
 * iterated again, whatever is addressed to it is dropped (robot.rs:1815,1844), invisible to
 * mgx_neighbours.  Ids stay stable; its last beliefs remain readable.  The other robots drop
 * their factors towards it in the following mgx_update_topology passes (or mgx_ir_disconnect). */
int mgx_robot_remove(mgx_world *w, int32_t robot);
/* create_interrobot_factors for ONE direction (robot.rs:1500-1585): `owner` creates
 * K-1 InterRobotFactors towards `other`, variable i <-> variable i, i = 1..K-1, with
 * robot_number = first_robot_number + (i-1) (robot.rs:1527, interrobot.rs:75).
 * The reference calls this for (a,b) and (b,a) in the same system run. */
int mgx_ir_connect(mgx_world *w, int32_t owner, int32_t other, uint64_t first_robot_number);
/* delete_interrobot_factors (robot.rs:1386-1439): removes a->b and b->a factors and
 * both sides' inbox entries (factorgraph.rs:380-436). */
int mgx_ir_disconnect(mgx_world *w, int32_t a, int32_t b);

/* RadioAntenna.active (robot.rs:1593-1601) and Mission.state.idle() gates of
 * iterate_gbp_v2 (robot.rs:1794,1806,1822,1835,1851). */
/* FactorGraph::change_factor_enabled(FactorsEnabledSection) applied to every graph
 * (factorgraph.rs:1529-1539; ui/settings.rs:491-496) and to factors created from now on
 * (robot.rs:1236,1276,1326,1506 read the same config entry): MGX_FACTOR_* bits.  A disabled factor
 * keeps its last message in the variable's inbox and drops what is sent to it; switched on again it
 * resumes from the inbox it froze with (all four kinds).  On a sharded world every rank calls this with
 * the same mask right after a halo exchange (the records inter-robot factors freeze with / thaw against
 * include the ghosts' — they have to be the owners' current ones). */
int mgx_set_enabled(mgx_world *w, uint32_t kind_mask);
int mgx_set_antenna(mgx_world *w, int32_t robot, int32_t active);
int mgx_set_idle(mgx_world *w, int32_t robot, int32_t idle);

/* update_failed_comms (robot.rs:1593-1601) writes every antenna once per tick: bulk form.
 * The Bernoulli draws stay with the caller (the reference uses Bevy's GlobalEntropy<WyRand>). */
int mgx_set_antennas(mgx_world *w, uint32_t n, const int32_t *robots, const uint8_t *active);

/* ---- dynamic inter-robot topology (robot.rs:1362-1586) ----------------------------------- */
#define MGX_NEIGHBOURS_AUTO 0u  /* all-pairs kernel below 512 robots, hash grid above */
#define MGX_NEIGHBOURS_PAIRS 1u /* all-pairs kernel (what the reference does on the CPU) */
#define MGX_NEIGHBOURS_GRID 2u  /* uniform hash grid, 3x3 cells of ~radius (falls back to all
                                   pairs when radius is not positive and finite) */
/* update_robot_neighbours (robot.rs:1362-1384): `positions_xyz` = Transform::translation of
 * every robot of the world in id order (host, n x 3 f32).  Robot j is in range of i (j != i)
 * unless `radius < |p_i - p_j|` in f32 (so a NaN distance is in range).  CSR result on the
 * host: row_ptr[n+1], rows ascending in Entity order (= order_key) like the BTreeSet
 * robots_within_comms_range.  neighbours_out may be NULL to size the call (*needed). */
int mgx_neighbours(mgx_world *w, const float *positions_xyz, float radius, uint32_t method,
                   int32_t *row_ptr, int32_t *neighbours_out, uint64_t capacity,
                   uint64_t *needed);
/* One pass of update_robot_neighbours + delete_interrobot_factors + create_interrobot_factors
 * (robot.rs:86-99,1362-1586), including the reference's bookkeeping quirk: the pairs to delete
 * go through a HashMap<RobotId,RobotId> (robot.rs:1391-1404), so per robot only the
 * largest-id out-of-range peer gets its factors deleted in that pass, the rest only leave
 * robots_connected_with (and get a second set of factors if they come back in range).
 * `robot_number_next` is RobotNumberGenerator (robot.rs:122-140), advanced by one per created
 * factor.  stats (optional) = {connections created (one direction each), pairs deleted}.
 * Existing state survives: the device state is pulled to the host before the tables are
 * rebuilt, only when something changed. */
int mgx_update_topology(mgx_world *w, const float *positions_xyz, float radius, uint32_t method,
                        uint64_t *robot_number_next, uint32_t *stats);
/* RobotConnections::robots_connected_with of one robot (robot.rs:515-531), ascending.
 * others may be NULL to query the count. */
int mgx_connections(mgx_world *w, int32_t robot, int32_t *others, uint32_t capacity, uint32_t *n);

/* ---- the hot path ------------------------------------------------------------------ */
/* iterate_gbp_v2 (robot.rs:1769-1861): runs `n` schedule steps, step i doing the
 * internal phase if steps[i] & MGX_STEP_INTERNAL and the external phase if
 * steps[i] & MGX_STEP_EXTERNAL.  Asynchronous on the world's stream. */
int mgx_iterate(mgx_world *w, const uint8_t *steps, uint32_t n);
/* Batches: several schedules, one submission.  A resident schedule launch pays for itself once — the robots' graphs go
 * HBM -> LDS when it starts and back when it ends, a sixth of a ten-iteration launch at 1000 x 16 — so a caller that issues
 * schedule after schedule with nothing in between (the reference's `iterate_gbp_v2` loop run ahead of the renderer, a planner
 * that looks several ticks ahead, a benchmark loop) brackets the loop: between mgx_batch_begin and mgx_batch_end the schedules
 * handed to mgx_iterate are recorded and submitted TOGETHER, merged into as few launches as their segments fit (32 [external]
 * internal* segments per launch: three ticks of the 10 / 10 schedule) — when the recorded ones fill a launch, at mgx_batch_end,
 * and in front of ANY other call on this world, which therefore finds the world as if every schedule had run when it was issued.
 * Nothing is reordered, nothing skipped: iterate(a); iterate(b) computes what iterate(a ++ b) computes (the phases are
 * flattened either way, robot.rs:1787-1860), bit for bit — the engine's form of capturing a launch-bound loop in a graph.
 * The one visible difference: work recorded in an open batch is on the world's stream only after its submission — a caller that
 * waits on the stream by its own means (events) closes the batch, or calls mgx_synchronize, first.  Works on sharded worlds
 * whose exchange lives in the engine (every rank brackets alike).  n_schedules / n_launches (may be NULL): schedules recorded
 * since mgx_batch_begin and sweep-kernel launches they were submitted as. */
int mgx_batch_begin(mgx_world *w);
int mgx_batch_end(mgx_world *w, uint32_t *n_schedules, uint32_t *n_launches);

/* How the last mgx_iterate / mgx_tick call ran: the number of sweep-kernel launches it enqueued.  A world whose
 * robots all live on this device, with inter-robot factors enabled and few enough robots for every workgroup
 * to be resident at once, runs a whole schedule as ONE launch (robots hand their snapshot records to their
 * neighbours inside it); otherwise one launch per [external iteration] internal* segment.  MGX_PERSISTENT=0 in
 * the environment forces the latter.  Diagnostic (bench.py prices its roofline per launch with it). */
int mgx_last_launch_count(mgx_world *w, uint32_t *n_launches);

/* LINGERING resident launches — schedules issued back to back ride in ONE launch.  The reference's driver runs iterate_gbp_v2
 * tick after tick (robot.rs:85-108, a .chain() in FixedUpdate) with nothing in between but the two prior updates that mgx_tick
 * folds into the launch; a resident launch that ended with its schedule would write every graph back to HBM only for the next
 * one to stage it again (a sixth of a 10-iteration tick at 1000 x 16).  So on an unsharded world a resident launch does not end
 * with its schedule: the robots' workgroups keep their graphs in LDS and wait — at most `microseconds` — for the next mgx_iterate
 * / mgx_tick, which is POSTED into the running launch (a host-mapped box: plan, prior-update records) instead of
 * launched: mgx_last_launch_count reports 1 for it all the same (one submission).  Every other call on the world first ends the
 * launch (it writes back; the call finds the world as after any launch), so nothing observable changes — results are
 * bit-identical, tests/test_gpu_linger.py — except:
 *   * work the CALLER puts on the world's stream behind mgx_iterate / mgx_tick by its own means (events, other kernels) starts
 *     only when the launch has ended: call mgx_flush (ends it, no wait) or mgx_synchronize first;
 *   * a caller that stops issuing schedules without another call leaves the launch waiting out its bound (default 300 us,
 *     MGX_LINGER_US; MGX_LINGER=0 or mgx_set_linger(w, 0): launches end with their schedule).  Every wait inside the launch is
 *     bounded: a host that dies leaves no spinning GPU.
 * A posted schedule the launch did not take any more (it ended first) is run as a launch of its own — nothing lost, nothing
 * twice.  Launches linger only where the calling pattern promises schedules to come (two lingering launches in a row that ended
 * without a post switch it off until schedules are issued back to back again).
 * mgx_linger_stats: launches that lingered, schedules posted into them, posts taken back and re-run, launches that ended by
 * themselves (waited out their bound) while the host was about to post. */
int mgx_flush(mgx_world *w);
int mgx_set_linger(mgx_world *w, int32_t microseconds /* < 0: default (environment) */);
int mgx_linger_stats(mgx_world *w, uint64_t *launches, uint64_t *posts, uint64_t *reruns, uint64_t *ended_by_device);
/* Per-world switch for the above (default: on): 0 keeps every schedule of THIS world on the launch-per-segment path — what
 * MGX_PERSISTENT=0 does for the whole process.  For measuring one against the other; results are identical either way.
 * On a sharded world with resident launches agreed on (mgx_halo_resident_connect) every rank has to switch alike.
 * 2 (diagnostic): on, but this world DECLINES every resident launch — on a sharded world whose ranks agree on every schedule
 * it says no where the others look (mgx_halo_resident_connect), so every rank takes the fall-back; elsewhere like 0. */
int mgx_set_resident_launches(mgx_world *w, int32_t enabled);
/* Whether factors switched back on (mgx_set_enabled) are still taking their first updates from the inboxes they froze with
 * (or inter-robot factors created while their kind was off still lack inbox keys): such schedules run launch by launch.
 * On a sharded world this depends on the flags of the robots THIS rank holds, so the ranks ask each other (any rank still
 * thawing keeps resident launches off on all of them: magics_amd/sharded.py does it after set_enabled). */
int mgx_is_thawing(mgx_world *w, int32_t *thawing);

/* The launch primitive the calls above and below are built on: one device pass per robot
 * that runs the external phases in `external_phases` (bit0 = external factor sweep + routing,
 * bit1 = external variable sweep + routing) and then `n_internal` internal iterations of the
 * phases in `internal_phases` (bit0 = internal factor sweep, bit1 = internal variable sweep;
 * n_internal > 1 needs both).  A sharded driver uses it to place its halo exchange between
 * an internal and the following external phase.  robot = -1: all robots.
 * hints: MGX_HINT_NEXT_STARTS_EXTERNAL — a promise that the next sweep of this world starts with an
 * external factor sweep and that no flag / topology / prior change happens in between; the
 * inter-robot messages computed here are then not stored to HBM (they are recomputed first). */
#define MGX_HINT_NEXT_STARTS_EXTERNAL 1u
int mgx_sweep(mgx_world *w, int32_t robot, uint32_t external_phases, uint32_t internal_phases,
              uint32_t n_internal, uint32_t hints);

/* Fine-grained mirrors of FactorGraph::{internal_factor_iteration,
 * internal_variable_iteration, external_factor_iteration, external_variable_iteration}
 * (factorgraph.rs:688-826).  robot = -1 runs the sweep for every robot.  The external
 * sweeps include the message routing the reference's caller performs
 * (robot.rs:1813-1858) and are only available world-wide (robot must be -1). */
int mgx_internal_factor_iteration(mgx_world *w, int32_t robot);
int mgx_internal_variable_iteration(mgx_world *w, int32_t robot);
int mgx_external_factor_iteration(mgx_world *w, int32_t robot);
int mgx_external_variable_iteration(mgx_world *w, int32_t robot);

/* FactorGraph::change_prior_of_variable + the caller's routing to external factors
 * (factorgraph.rs:494-528, variable.rs:203-230, robot.rs:2262-2282). */
int mgx_change_prior(mgx_world *w, int32_t robot, uint32_t var_ix, const double mean[4]);
/* batched form: n triples (robot[i], var_ix[i], means[i][4]) in one launch */
int mgx_change_priors(mgx_world *w, uint32_t n, const int32_t *robots, const uint32_t *var_ix,
                      const double *means);

/* FactorGraph::reset_variables(&means, first_last_sigma, inbetween_sigma) (factorgraph.rs:1541-1564; VariableNode::reset,
 * variable.rs:350-360; FactorNode::empty_inbox, factor/mod.rs:480-483) for one robot's graph: every variable's belief mean
 * becomes means[i] and its belief precision diag(sigma) — first_last_sigma for variables 0 and K-1, inbetween_sigma for the
 * others, used as the reference uses them (AS the diagonal, +inf allowed) — and every message in the variables' inboxes and
 * in the inboxes of the graph's own factors becomes empty.  means: [n_means][4]; n_means must equal the graph's number of
 * variables K (the reference asserts it, factorgraph.rs:1548: MGX_ERR_INVALID otherwise, nothing read).  mgx_reset_tracking_factors:
 * FactorGraph::reset_tracking_factors (factorgraph.rs:1566-1590) — the graph's tracking factors skip their next ten
 * updates (tracking.rs:153-155,362-371).  The reference calls the pair (means, 1e30, +inf) when a global path arrives
 * (robot.rs:766-769); rare, so the engine re-lays the device state out on the next launch. */
int mgx_reset_variables(mgx_world *w, int32_t robot, const double *means, uint32_t n_means, double first_last_sigma,
                        double inbetween_sigma);
int mgx_reset_tracking_factors(mgx_world *w, int32_t robot);

/* The driver's per-tick prior updates, batched over robots in one launch:
 *   what[i] & 1: update_prior_of_horizon_state (robot.rs:2182-2283) — the last variable of robots[i]
 *                moves towards waypoints_xy[i] at min(max_speed, distance) for delta_t seconds;
 *   what[i] & 2: update_prior_of_current_state_v3 (robot.rs:2286-2338) — variable 0 moves by
 *                time_scale[i] * (mean_1 - mean_0), time_scale = fixed_dt / t0 (an f32 quotient widened).
 * Each ends in change_prior of that variable.  The caller lists the robots the reference's systems would
 * not skip (not idle / not finished / with a next waypoint) and keeps the mission logic. */
int mgx_update_priors(mgx_world *w, uint32_t n, const int32_t *robots, const double *waypoints_xy,
                      const double *time_scale, const uint8_t *what, double max_speed, double delta_t);
/* One FixedUpdate tick of the planner chain in one call: the two prior updates above for the listed
 * robots, then iterate_gbp_v2 over `steps` (robot.rs:86-103).  Same results as mgx_update_priors
 * followed by mgx_iterate; when the schedule opens with an internal iteration (every schedule of the
 * reference does) the prior updates are applied inside the launch that runs it, on the image each
 * robot's workgroup has just staged, instead of by a kernel of their own. */
int mgx_tick(mgx_world *w, uint32_t n, const int32_t *robots, const double *waypoints_xy,
             const double *time_scale, const uint8_t *what, double max_speed, double delta_t,
             const uint8_t *steps, uint32_t n_steps);

/* ---- whole driver ticks on the device -----------------------------------------------------
 * The rest of the reference's FixedUpdate chain around iterate_gbp_v2 (robot.rs:86-103) with its state on the device, so
 * that a tick costs the host ONE synchronisation (the neighbour rows the connection bookkeeping needs) instead of four
 * belief read-backs: a robot's mission — its route, the next waypoint, the reached-when rules of formation.yaml and the
 * Bevy Transform it moves — is handed over once (mgx_mission_set); every mgx_mission_tick then runs
 *   reached_waypoint                         robot.rs:2080-2176  estimated position (belief mean of the rule's variable, as f32)
 *                                                                against the next waypoint, f32 squared distance; the last
 *                                                                waypoint completes the mission (and despawns the robot,
 *                                                                robot.rs:2172, when despawn_finished is set)
 *   update_robot_neighbours + delete_ / create_interrobot_factors   robot.rs:1362-1586  on the device's Transforms, exactly
 *                                                                mgx_update_topology otherwise
 *   update_failed_comms                      robot.rs:1593-1601  `antennas` (one byte per robot id, the caller's draws; NULL: none)
 *   update_prior_of_horizon_state / _current_state_v3 + the Transform increment   robot.rs:2182-2338
 *   iterate_gbp_v2                           robot.rs:1769-1861  over `steps`
 * stats (optional) = {connections created, pairs deleted, missions completed this tick}.  Unsharded worlds. */
typedef struct mgx_mission_desc {
    uint32_t n_waypoints;         /* waypoints still to visit; the first one is the next target (>= 1)                 */
    uint32_t reserved;
    const double *waypoints_xy;   /* [n][2]; compared as f32 (the reference's Vec2), used as f64 by the horizon prior   */
    uint32_t reach_var, finish_var;   /* variable tested against an intermediate / the final waypoint: 0 = current,
                                         K-1 = horizon (waypoint-reached-when-intersects / finished-when-intersects)   */
    float reach_dist2, finish_dist2;  /* squared distance limits in f32 (robot-radius^2 or meter^2)                    */
    float translation[3];         /* Transform::translation at hand-over (x, height, y)                                */
    float reserved2;
    double time_scale;            /* fixed_dt / t0 as the f32 quotient widened (robot.rs:2309)                         */
} mgx_mission_desc;
int mgx_mission_set(mgx_world *w, int32_t robot, const mgx_mission_desc *desc);
int mgx_mission_tick(mgx_world *w, float comms_radius, uint32_t method, uint64_t *robot_number_next,
                     int32_t despawn_finished, const uint8_t *antennas, double max_speed, double delta_t,
                     const uint8_t *steps, uint32_t n_steps, uint32_t *stats);
/* The tick in two halves, for a caller whose comms-failure draws depend on which robots are still alive after this tick's
 * despawns (the reference draws once per live robot, robot.rs:1599): _begin runs reached_waypoint and the topology pass (the
 * tick's one synchronisation) and leaves the robots whose mission completed in mgx_mission_finished (ascending ids);
 * _end applies the antennas, the prior updates with the Transform increment and the schedule.  Robots may be added (and
 * given missions) between the halves.  mgx_mission_tick == _begin + _end. */
int mgx_mission_tick_begin(mgx_world *w, float comms_radius, uint32_t method, uint64_t *robot_number_next,
                           int32_t despawn_finished, uint32_t *stats);
int mgx_mission_tick_end(mgx_world *w, const uint8_t *antennas, double max_speed, double delta_t,
                         const uint8_t *steps, uint32_t n_steps);
int mgx_mission_finished(mgx_world *w, int32_t *robots, uint32_t capacity, uint32_t *n);
/* MANY ticks in one call: what a headless run does between two spawns (the reference's FixedUpdate chain tick after tick,
 * robot.rs:85-108, with update_failed_comms' draws in between) without the caller's interpreter in the loop —
 *   for each tick:  mgx_mission_tick_begin;  one Bernoulli(failure_rate) draw per robot alive after the tick's despawns, id
 *                   order, from the caller's WyRand stream (rand 0.8.5 Bernoulli over wyrand 0.2.0, restated in
 *                   magics_amd/prng.py: one u64 per draw unless failure_rate == 1; wyrand_state NULL: no stream, no draws);
 *                   mgx_mission_tick_end (the draws as antennas when failure_rate > 0).
 * Per tick t < ticks_done the caller gets what it would have seen tick by tick: connections created / pairs deleted, the
 * number of missions completed and, one after the other in `finished`, the robots concerned (ascending within a tick), the
 * Transform::translation of EVERY robot after the tick's move ([n_ticks][n_robots][3], may be NULL) and the draws
 * ([n_ticks][n_robots], 1 = on air, may be NULL).  stop_when_all_finished: returns behind the tick that completed the last
 * mission.  No robot may join during the call.  Identical to the same ticks issued one by one (tests/test_gpu_sim.py). */
typedef struct mgx_mission_run_desc {
    uint32_t n_ticks;
    float comms_radius;
    uint32_t method;                  /* MGX_NEIGHBOURS_* */
    int32_t despawn_finished, stop_when_all_finished;
    uint32_t n_steps;
    const uint8_t *steps;
    double max_speed, delta_t, failure_rate;
    uint64_t *wyrand_state;           /* in / out */
    uint64_t *robot_number_next;      /* in / out */
    uint32_t *created, *deleted, *n_finished; /* out [n_ticks] */
    int32_t *finished;                /* out [finished_capacity] */
    uint32_t finished_capacity;
    uint32_t finished_total;          /* out */
    float *translations;              /* out [n_ticks][n_robots][3] or NULL */
    uint8_t *antennas;                /* out [n_ticks][n_robots] or NULL */
    uint32_t ticks_done;              /* out */
    uint32_t reserved;
} mgx_mission_run_desc;
int mgx_mission_run(mgx_world *w, mgx_mission_run_desc *desc);
/* Transform::translation [n][3] of every robot as of the end of the last tick, WITHOUT synchronising: _end sends them to
 * the host behind its launches, so they are complete once the stream has been synchronised since — which the next
 * mgx_mission_tick_begin does by itself.  (PositionTracker / VelocityTracker samples, planner/tracking.rs:104-218.) */
int mgx_mission_translations(mgx_world *w, float *translations, uint32_t capacity_robots, uint32_t *n_robots);
/* Transform::translation [n][3], next waypoint index (== the route's length once complete; -1: no mission) and the tick
 * at which each mission completed (-1 before) of every robot, id order; any pointer may be NULL.  Synchronises. */
int mgx_mission_read(mgx_world *w, float *translations, int32_t *targets, int64_t *finished_tick);

/* ---- read-back ----------------------------------------------------------------------- */
/* VariableNode.belief (variable.rs:40-54).  Any output pointer may be NULL. */
int mgx_get_belief(mgx_world *w, int32_t robot, uint32_t var_ix, double eta[4], double lam[16],
                   double mean[4], double cov[16], int32_t *valid);
/* bulk: robots in id order, variables in index order; ghosts are skipped.
 * means [sum K][4]; eta [sum K][4]; lam [sum K][16].  Any pointer may be NULL. */
int mgx_read_beliefs(mgx_world *w, double *eta, double *lam, double *means);
/* means only — what reached_waypoint (robot.rs:2125-2136) and the visualisers read per tick */
int mgx_read_means(mgx_world *w, double *means);
/* the mean of ONE variable of every robot, [n_robots][4]: nth_variable(0) / last_variable for
 * reached_waypoint (robot.rs:2125-2136), variables 0 and 1 for the Transform increment of
 * update_prior_of_current_state_v3 (robot.rs:2309-2330) — 32 bytes per robot instead of 32 K */
int mgx_read_variable_means(mgx_world *w, uint32_t var_ix, double *means);
int mgx_num_robots(mgx_world *w, uint32_t *n_robots, uint32_t *n_variables);
/* FactorGraph::messages_sent() / messages_received() (factorgraph.rs:876-890; MessageCount,
 * factorgraph/mod.rs:29-137) of one robot's graph, as exported by export.rs:434-439:
 * counts = {sent internal, sent external, received internal, received external}, summed over the
 * nodes the graph holds now (a deleted inter-robot factor takes its counts with it).  The counts
 * depend only on topology, enabled kinds, antenna / idle flags and iteration counts, so they are
 * kept on the host (launches are logged, no device work). */
int mgx_message_counts(mgx_world *w, int32_t robot, uint64_t counts[4]);
/* Sharded worlds keep the counters too (for the robots a rank OWNS), provided the rank's mirror holds every connection a
 * local robot takes part in — also the ones it owns towards robots of other ranks (mgx_ir_connect with a ghost target:
 * bookkeeping only, no device edge) — and is told the prior changes other ranks apply to its ghosts: */
int mgx_note_change_priors(mgx_world *w, uint32_t n, const int32_t *robots, const uint32_t *var_ix);

/* ---- multi-GPU halo (one exchange per external iteration) ----------------- */
/* Number of f64 words of one robot's halo record with K variables. */
uint32_t mgx_halo_words(uint32_t K);
/* Register the exchange lists once: the local robots whose snapshot records are written
 * to the send buffer (in this order) and the ghost robots filled from the receive buffer. */
int mgx_halo_plan(mgx_world *w, uint32_t n_send, const int32_t *send_robots, uint32_t n_recv,
                  const int32_t *recv_ghosts);
/* The same lists derived from the connections this world holds, for a sharded world that FOLLOWS its
 * topology: every rank holds every robot (its own ones and ghost copies of all the others, ids equal
 * on all ranks), runs mgx_update_topology on all positions — the bookkeeping of robot.rs:1386-1586 is
 * then replicated, identical everywhere — and exchanges the snapshot records of exactly those robots
 * that own a connection into a robot of another rank.  rank_of[robot] = owning rank; the lists are
 * grouped by peer rank, ascending robot id inside a group, and both ends of every exchange derive
 * matching groups.  send_counts / recv_counts [n_ranks]: records per peer.  Call after every topology
 * pass that changed something, then exchange once (pack / all-to-all-v / unpack) BEFORE the next
 * sweep: the factors the pass created take the owner's delivery count of that exchange as their
 * creation epoch.  Host-driven transports only (the direct / RCCL wirings are per plan). */
int mgx_halo_plan_from_connections(mgx_world *w, const int32_t *rank_of, uint32_t n_robots,
                                   int32_t my_rank, uint32_t n_ranks, uint32_t *send_counts,
                                   uint32_t *recv_counts);
/* The sharding plan itself (host only, no device; identical on every rank, no communication):
 * mgx_shard_partition — owner rank of every robot: contiguous strips in (y, x) order of the robots' positions with equal
 *   robot counts (spatial blocks keep cross-rank pairs few); ties broken by robot index.
 * mgx_shard_plan_create — one rank's view, from the owner map and the directed connections (owner robot a, other robot
 *   b) of create_interrobot_factors (robot.rs:1490-1541): the local robots, the ghosts (owners of connections whose
 *   target is local), the indices of the connections evaluated here, and per peer rank q the robots whose snapshot
 *   records go to q / arrive from q — segments [first[q], first[q+1]) of the send / receive lists, ascending robot id
 *   inside a segment, i.e. exactly the order mgx_halo_plan and the transports expect.
 * mgx_shard_plan_counts sizes the arrays of mgx_shard_plan_get (first arrays: n_ranks + 1 entries); any output may be NULL. */
typedef struct mgx_shard_plan mgx_shard_plan;
int mgx_shard_partition(const double *positions_xy, uint32_t n_robots, uint32_t n_ranks, int32_t *owner);
int mgx_shard_plan_create(const int32_t *owner, uint32_t n_robots, const int32_t *conn_owner,
                          const int32_t *conn_other, uint32_t n_conns, int32_t rank, uint32_t n_ranks,
                          mgx_shard_plan **out);
void mgx_shard_plan_destroy(mgx_shard_plan *plan);
int mgx_shard_plan_counts(const mgx_shard_plan *plan, uint32_t *n_local, uint32_t *n_ghosts,
                          uint32_t *n_connections, uint32_t *n_send, uint32_t *n_recv);
int mgx_shard_plan_get(const mgx_shard_plan *plan, int32_t *local, int32_t *ghosts, uint32_t *connections,
                       uint32_t *send_first, int32_t *send_robots, uint32_t *recv_first,
                       int32_t *recv_robots);

/* Pack the planned robots' variable->own-factor snapshots (what their inter-robot factors on
 * other ranks read) into `dev_buf` (device pointer, n_send records of mgx_halo_words(K) f64),
 * resp. unpack n_recv records into the ghost robots.  Asynchronous on the world's stream. */
int mgx_halo_pack(mgx_world *w, void *dev_buf);
int mgx_halo_unpack(mgx_world *w, const void *dev_buf);

/* ---- migration: a robot changes its owning rank (worlds that follow their topology) --------------------------------------
 * The reference has ONE world and no notion of ownership (robot.rs queries run over every entity); a sharded world that
 * follows its topology re-balances by moving a robot's graph between ranks.  Everything that exists on the owner's rank
 * only travels in one flat, self-describing record: the graph's numeric state (priors, beliefs, factor -> variable
 * messages, snapshot records and delivery counts, tracking records, iteration count, path, frozen inboxes), the totals of
 * its MessageCount, and the state of every InterRobotFactor attached to its variables (kept at the target's rank).  The
 * replicated bookkeeping (connection sets, node slots, robot numbers, flags, the connections' counters) stays where it is.
 * Protocol, on every rank at the same point BETWEEN ticks (after the sweeps that followed the last mgx_update_topology):
 *   old owner:  mgx_robot_export (bytes first with buf = NULL, then the record), mgx_robot_release
 *   new owner:  mgx_robot_import with the record (refused with MGX_ERR_STATE when its connections differ from this
 *               rank's bookkeeping: the ranks' topology passes are out of step)
 *   every rank: the new rank table to mgx_halo_plan_from_connections, and the in-engine transports wired again (device
 *               indices change: locals first) — as after mgx_robot_add.
 * What does NOT travel: device-side missions (mgx_mission_*: routes, next waypoints, Transforms, completion ticks are state of
 * unsharded worlds, device index == robot id).  Export, import and release of a robot that has one are refused (MGX_ERR_STATE).
 * Results stay bit-identical to the unsharded world's (tests/test_gpu_sharded.py::test_robots_migrate_between_ranks). */
int mgx_robot_export(mgx_world *w, int32_t robot, void *buf, uint64_t capacity, uint64_t *bytes);
int mgx_robot_import(mgx_world *w, int32_t robot, const void *buf, uint64_t bytes);
int mgx_robot_release(mgx_world *w, int32_t robot);

/* ---- halo exchange through RCCL inside the library -------------------------------------------
 * pack -> grouped ncclSend / ncclRecv with every peer (the all-to-all-v of boundary snapshots over
 * xGMI) -> unpack, enqueued on the world's stream in front of every world-wide launch
 * that starts with the external factor phase: no host work per exchange beyond the enqueue.  RCCL
 * is resolved at run time (dlopen; the copy already in the process if there is one).  One rank
 * creates the id (mgx_rccl_unique_id), all ranks get it over any control plane and call
 * mgx_halo_rccl_connect (collective: ncclCommInitRank) with their peers and the segments of the
 * send / receive lists (mgx_halo_plan) that belong to each. */
int mgx_rccl_unique_id(uint8_t id[128]);
int mgx_halo_rccl_connect(mgx_world *w, const uint8_t id[128], uint32_t n_ranks, uint32_t rank,
                          uint32_t n_peers, const uint32_t *peer_rank, const uint32_t *send_first,
                          const uint32_t *recv_first);
int mgx_halo_rccl_disconnect(mgx_world *w);

/* ---- direct halo exchange: peer-mapped stores instead of the collective -------
 * Same dataflow as pack -> all-to-all-v -> unpack, without a collective and without host work
 * per exchange: each producer's kernel stores its boundary records straight into the consumer
 * rank's receive area over xGMI and then bumps an arrival counter there; the consumer's kernel
 * waits for the counters of all its producers and fills its ghost robots.  Once connected, every
 * world-wide launch that starts with the external factor phase (mgx_iterate, mgx_sweep,
 * mgx_external_factor_iteration) runs the exchange first, asynchronously on the world's stream.
 * All ranks must issue the same sequence of such launches (they do: one schedule).
 *
 * Wiring, after mgx_halo_plan (send list grouped by consumer, receive list grouped by producer):
 *   1. mgx_halo_direct_setup: allocates this rank's receive area (2 x n_recv records, fine-grained
 *      device memory) and n_sources arrival counters; returns their device addresses.
 *   2. the ranks swap these addresses — mgx_ipc_export / mgx_ipc_open across processes (hipIpc*),
 *      raw pointers inside one process — together with, per producer, the record offset of its
 *      segment in the consumer's area and the address of its counter (flag_base + 8 * index of the
 *      producer among the consumer's sources).
 *   3. mgx_halo_direct_connect with, per consumer p (send-list segment send_first[p] ..
 *      send_first[p+1]): the consumer's area, its size in records, the offset of this rank's
 *      segment in it and this rank's counter there.  Producers and consumers of a rank must be the
 *      same set of peers (inter-robot factors come in pairs, robot.rs:1490-1541).
 * A wait that exceeds MGX_HALO_TIMEOUT_MS (default 5000) gives up, leaves the ghosts untouched
 * and is reported by mgx_halo_direct_status (never a hung GPU). */
int mgx_halo_direct_setup(mgx_world *w, uint32_t n_sources, void **recv_base, void **flag_base);
int mgx_halo_direct_connect(mgx_world *w, uint32_t n_peers, const uint32_t *send_first,
                            void *const *peer_recv_base, const uint64_t *peer_recv_records,
                            const uint64_t *peer_record_offset, void *const *peer_flag_slot);
/* The exchange by hand: MGX_HALO_PUSH sends this rank's records for the next exchange,
 * MGX_HALO_WAIT fills the ghosts once every producer has done so; both = what the launches do by
 * themselves.  A launch that finds its exchange already pushed only waits — ranks that share one
 * process (and possibly a hardware queue) push all of them before the first one waits. */
#define MGX_HALO_PUSH 1u
#define MGX_HALO_WAIT 2u
int mgx_halo_direct_exchange(mgx_world *w, uint32_t what);
int mgx_halo_direct_status(mgx_world *w, uint64_t *exchanges, uint64_t *failed_exchange);
int mgx_halo_direct_disconnect(mgx_world *w);
/* The same exchange for a world whose exchange lists CHANGE (one that follows its topology: connections across ranks come and
 * go with robot.rs:1386-1586, mgx_update_topology + mgx_halo_plan_from_connections), wired once:
 *   mgx_halo_direct_setup_slots: the receive area has `slot_capacity` record slots per parity — one per ghost robot of this rank,
 *      slot = the robot's place among the rank's ghosts in device order (mgx_halo_ghost_slots; capacity >= their number, room for
 *      robots that join) — and EVERY other rank is a source (n_sources = ranks - 1), with or without records in an exchange:
 *      a rank that sends nothing still publishes the exchange number, a rank that receives nothing still waits for all of them
 *      (that wait is its flow control: nobody gets two exchanges ahead of anybody).
 *   mgx_halo_direct_connect_slots: after every change of the lists, (re)aims the pushes — peers in the order of the send list's
 *      segments (one per other rank, possibly empty), and for every entry of the send list the slot of that robot in its
 *      consumer's area.  Exchange numbers go on across calls. */
int mgx_halo_direct_setup_slots(mgx_world *w, uint32_t n_sources, uint32_t slot_capacity, void **recv_base, void **flag_base);
int mgx_halo_ghost_slots(mgx_world *w, uint32_t n, const int32_t *robots, int32_t *slots);
/* the exchange lists as they stand (robot ids, by peer rank in the order of mgx_halo_plan_from_connections' counts) */
int mgx_halo_get_lists(mgx_world *w, int32_t *send_robots, uint32_t send_capacity, int32_t *recv_robots, uint32_t recv_capacity,
                       uint32_t *n_send, uint32_t *n_recv);
int mgx_halo_direct_connect_slots(mgx_world *w, uint32_t n_peers, const uint32_t *send_first, void *const *peer_recv_base,
                                  const uint64_t *peer_slot_capacity, const uint32_t *entry_slot, void *const *peer_flag_slot);

/* ---- resident schedule launches on sharded worlds ----------------------------------------------------------------------
 * (replaces, like the exchanges above, the serial external phase and routing of robot.rs:1803-1859 and
 * factorgraph.rs:719-760 — here without ANY launch boundary or exchange kernel between the iterations of a schedule.)
 * A world whose robots all fit the device at once runs a whole mgx_iterate / mgx_tick schedule as ONE launch (one workgroup per
 * robot, resident for the schedule; neighbouring workgroups hand their snapshot records over inside the launch).  With the
 * calls below the same holds for a SHARDED world: every rank owns a GHOST AREA — fine-grained device memory holding, for each
 * ghost robot, its snapshot records and delivery counts for the two buffer parities and one progress word — which the ghosts'
 * owner ranks map (mgx_ipc_export / mgx_ipc_open across processes, raw pointers inside one process) and store into from
 * inside their own resident launch: a boundary robot's workgroup writes its records at the end of every segment into its own
 * rank's buffers AND (system-scope, over xGMI) into the ghost area of every rank that holds it as a ghost, then that ghost's
 * progress word; the workgroups there poll it like the word of a local neighbour.  One exchange per external iteration as
 * before (robot.rs:1803-1859), with no host work and no kernel boundary.  A schedule that OPENS with an external iteration
 * takes the direct exchange (above) in front of the launch, so mgx_halo_direct_* has to be connected too.
 *   1. mgx_halo_resident_setup (after mgx_halo_plan and mgx_halo_direct_setup / _connect): allocates the ghost area; returns its
 *      address, the number of ghost slots, this rank's buffer parity and segment count (both advance in lockstep on all ranks:
 *      every rank issues the same schedules), the slot of every entry of the receive list (recv_slots[n_recv]) and whether
 *      this rank CAN run resident launches (eligible: inter-robot factors staged in LDS, every local robot's workgroup resident
 *      at once).  The ranks agree on that — all or none.
 *   2. mgx_halo_resident_connect with one entry per (local robot of the send list, rank that receives it): that rank's area,
 *      number of ghost slots, the robot's slot there, that rank's parity and segment count as returned by ITS setup.
 *      coordinator_area / n_ranks: the area of rank 0 as this rank maps it (rank 0: its own) and the number of ranks of the
 *      world.  The first word of that area is where the ranks AGREE on every schedule: the launches of one schedule wait for
 *      each other's records, so they go ahead together or not at all.  Each rank's launch signs in there once all its own
 *      workgroups are on the device; the last rank to sign in says go; a rank that has waited MGX_RESIDENT_CENSUS_SHARDED_US
 *      (default 20000) for the others, or whose own workgroups do not all arrive (another tenant holds the CUs), or that cannot
 *      run this schedule as a resident launch at all, says abort — by system-scope compare-and-swap on that one word, so every
 *      rank reads the same answer before any of them has written anything.  After an abort every rank's world is as it was
 *      and the schedule runs launch by launch with the direct exchange (the engine does that by itself, see
 *      mgx_resident_outcome; the next few schedules skip the resident form — the same ones on every rank).
 *      coordinator_area NULL or n_ranks < 2: no agreement; a rank whose peers never start then gives up after
 *      MGX_RESIDENT_TIMEOUT_MS (default 2000) and reports MGX_ERR_STATE with the world invalid — never a hung GPU.
 * From then on mgx_iterate / mgx_tick run eligible schedules as one launch per rank (mgx_last_launch_count == 1); all ranks
 * must have connected.  A change of the world's layout (robots added / removed) switches the resident form off on THAT rank
 * only; the peers may still hold its area mapped and store into it, so wiring again goes: barrier, mgx_halo_resident_disconnect
 * on every rank, the peers close their mappings, barrier, then setup / connect as above.  A setup while the previous wiring
 * has not been disconnected is refused (MGX_ERR_STATE). */
int mgx_halo_resident_setup(mgx_world *w, void **area_base, uint32_t *n_ghost_slots, uint32_t *parity, uint64_t *segment_count,
                            int32_t *recv_slots, int32_t *eligible);
int mgx_halo_resident_connect(mgx_world *w, uint32_t n_targets, const int32_t *robots, void *const *peer_area_base,
                              const uint32_t *peer_ghost_slots, const uint32_t *peer_slot, const uint32_t *peer_parity,
                              const uint64_t *peer_segment_count, void *coordinator_area, uint32_t n_ranks);
int mgx_halo_resident_disconnect(mgx_world *w);
/* The same for a world whose exchange lists CHANGE (one that follows its topology; with mgx_halo_direct_setup_slots for the
 * exchanges in front of launches): mgx_halo_resident_setup as above — `eligible` 2 says "everything but inter-robot factors is
 * there": they may come later, and every schedule is decided where the ranks agree — then ONCE
 *   mgx_halo_resident_connect_peers: every other rank's area, number of ghost slots, parity and segment count (what translates
 *      this rank's counts into each peer's is settled here), the coordinator area and the number of ranks as above;
 * and after every change of the lists, on every rank, with all ranks' launches through (synchronise, barrier):
 *   mgx_halo_resident_aim: (local robot, index of the peer in connect_peers' order, the robot's ghost slot there) per boundary
 *      robot and rank that now holds it as a ghost; the progress words of this rank's own ghosts start over at "through with
 *      everything so far" (a robot that has just become a neighbour across ranks never stored one here). */
int mgx_halo_resident_connect_peers(mgx_world *w, uint32_t n_peers, void *const *peer_area_base, const uint32_t *peer_ghost_slots,
                                    const uint32_t *peer_parity, const uint64_t *peer_segment_count, void *coordinator_area, uint32_t n_ranks);
int mgx_halo_resident_aim(mgx_world *w, uint32_t n_targets, const int32_t *robots, const uint32_t *peer_index, const uint32_t *peer_slot);
/* What became of the resident launch the last mgx_iterate enqueued (it decides within microseconds of its start whether it goes
 * ahead: residency census, and on sharded worlds the ranks' agreement above).  Waits for that decision.
 *   MGX_RESIDENT_NONE      nothing was pending (the call ran launch by launch, or its outcome has been taken already)
 *   MGX_RESIDENT_RAN       the launch goes ahead
 *   MGX_RESIDENT_DECLINED  the launch returned without touching the world, and the schedule has NOT been run: the caller
 *                          issues the same mgx_iterate again (it now takes the launch-by-launch path).
 * Nobody has to call this: every other entry point looks at the pending decision first and, after a declined launch, runs the
 * schedule launch by launch itself.  It exists for hosts that drive SEVERAL ranks from one thread (magics_amd/sharded.py:
 * LocalCluster): there the ranks' launch-by-launch exchanges must be enqueued in lockstep — all pushes before the first wait
 * — which a re-run inside one rank's call cannot do.  Only after mgx_iterate (a declined mgx_tick is re-run by the engine). */
#define MGX_RESIDENT_NONE 0
#define MGX_RESIDENT_RAN 1
#define MGX_RESIDENT_DECLINED 2
int mgx_resident_outcome(mgx_world *w, int32_t *outcome);
/* Whether mgx_iterate(w, steps, n) would be issued as a resident launch now — on a sharded world whose ranks agree on every
 * schedule (mgx_halo_resident_connect): whether every rank takes it to that agreement (the same answer on every rank) — rather
 * than launch by launch.  For hosts that have to know in advance which form a schedule takes (LocalCluster, as above). */
int mgx_resident_ready(mgx_world *w, const uint8_t *steps, uint32_t n, int32_t *ready);
/* Resident launches of this world so far, how many of them were declined (see above; each is followed by a back-off during
 * which schedules skip the resident form), and what is left of the current back-off, in world-wide external iterations run
 * launch by launch (0: the next eligible schedule is tried as a resident launch).  Any pointer may be NULL. */
int mgx_resident_stats(mgx_world *w, uint64_t *launches, uint64_t *declined, uint32_t *backoff);
/* hipIpcGetMemHandle / hipIpcOpenMemHandle / hipIpcCloseMemHandle on the addresses above
 * (64-byte handles), so the host side needs no HIP binding of its own. */
int mgx_ipc_export(const void *dev_ptr, uint8_t handle[64]);
int mgx_ipc_open(const uint8_t handle[64], void **dev_ptr);
int mgx_ipc_close(void *dev_ptr);

/* ---- gbp_linalg + gbp_multivariate_normal value types (host only, no device) ---------------
 * crates/gbp_linalg/src/lib.rs:47-128: euclidean_norm = sqrt(fold(0, acc + x*x)), l1_norm,
 * normalize (untouched when the norm is 0 or infinite). */
double mgx_euclidean_norm(const double *x, uint32_t n);
double mgx_l1_norm(const double *x, uint32_t n);
void mgx_normalize(double *x, uint32_t n);
/* ndarray-inverse's contract as used by the reference (variable.rs:153,278; marginalise..:79;
 * gbp_multivariate_normal): det(); inverse = "None" (returns 0) exactly when det() == 0, else 1.
 * Row-major n x n. */
double mgx_det(const double *a, uint32_t n);
int mgx_inverse(const double *a, uint32_t n, double *out);

/* MultivariateNormal in information form (crates/gbp_multivariate_normal/src/lib.rs:38-410).
 * Errors mirror MultivariateNormalError (lib.rs:9-30); mgx_last_error() carries the variant text,
 * e.g. "VectorLengthNotEqualMatrixShape(3, 2, 2)". */
typedef struct mgx_mvn mgx_mvn;
#define MGX_MVN_ERR_NON_SQUARE (-16)          /* NonSquarePrecisionMatrix(rows, cols)            */
#define MGX_MVN_ERR_LENGTH (-17)              /* VectorLengthNotEqualMatrixShape(len, rows, cols) */
#define MGX_MVN_ERR_SINGULAR_PRECISION (-18)  /* NonInvertiblePrecisionMatrix                     */
#define MGX_MVN_ERR_SINGULAR_COVARIANCE (-19) /* NonInvertibleCovarianceMatrix                    */
#define MGX_MVN_ADD 0
#define MGX_MVN_SUB 1
#define MGX_MVN_MUL 2 /* product of densities = sum in information form (lib.rs:380-410) */
/* lib.rs:63-93: checks in this order: square, lengths, det == 0; mean = precision . information */
int mgx_mvn_from_information_and_precision(const double *information, uint32_t len,
                                           const double *precision, uint32_t rows, uint32_t cols,
                                           mgx_mvn **out);
/* lib.rs:114-143 */
int mgx_mvn_from_mean_and_covariance(const double *mean, uint32_t len, const double *covariance,
                                     uint32_t rows, uint32_t cols, mgx_mvn **out);
void mgx_mvn_destroy(mgx_mvn *m);
uint32_t mgx_mvn_len(const mgx_mvn *m);
/* information_vector(), precision_matrix(), mean() (any pointer may be NULL); covariance() */
int mgx_mvn_get(const mgx_mvn *m, double *information, double *precision, double *mean);
int mgx_mvn_covariance(const mgx_mvn *m, double *covariance);
/* update_information_vector / update_precision_matrix (lib.rs:158-178); set_* and add_assign_*
 * (the `unsafe` ones, lib.rs:212-263: mean stale until update); update() returns 1 if the mean
 * was recomputed, 0 if it was current (lib.rs:271-279) */
int mgx_mvn_update_information_vector(mgx_mvn *m, const double *value);
int mgx_mvn_update_precision_matrix(mgx_mvn *m, const double *value);
int mgx_mvn_set_information_vector(mgx_mvn *m, const double *value);
int mgx_mvn_set_precision_matrix(mgx_mvn *m, const double *value);
int mgx_mvn_add_assign_information_vector(mgx_mvn *m, const double *value);
int mgx_mvn_add_assign_precision_matrix(mgx_mvn *m, const double *value);
int mgx_mvn_update(mgx_mvn *m);
/* a op b -> new value (Add / Sub / Mul, lib.rs:300-410) and a op= b */
int mgx_mvn_combine(const mgx_mvn *a, const mgx_mvn *b, int32_t op, mgx_mvn **out);
int mgx_mvn_combine_assign(mgx_mvn *a, const mgx_mvn *b, int32_t op);

/* ---- environment rasteriser (scenario front-end) --------------------------- */
/* gbp_environment::Environment (crates/gbp_environment/src/lib.rs:729-735) as plain data, and
 * env_to_png::env_to_image / env_to_sdf_image (crates/env_to_png/src/lib.rs:149-206) computed on
 * the device: what simulation_loader.rs:154-162 runs for each scenario to produce the image the
 * obstacle factors sample.  Images are interleaved RGB u8 (R = G = B), row-major,
 * (n_cols * resolution) x (n_rows * resolution).  Invalid input (a value the reference's
 * Percentage / StrictlyPositiveFinite / Angle / RelativePoint constructors reject, an empty grid)
 * returns MGX_ERR_INVALID where the reference panics or returns Err. */
#define MGX_SHAPE_CIRCLE 0          /* radius                         (lib.rs:114-143) */
#define MGX_SHAPE_TRIANGLE 1        /* angle_a, angle_b, radius       (lib.rs:158-223) */
#define MGX_SHAPE_REGULAR_POLYGON 2 /* sides, radius                  (lib.rs:232-300) */
#define MGX_SHAPE_POLYGON 3         /* n_points, points_xy            (lib.rs:345-415) */
#define MGX_SHAPE_RECTANGLE 4       /* width, height                  (lib.rs:305-340) */
typedef struct mgx_env_obstacle { /* gbp_environment::Obstacle (lib.rs:531-543) */
    int32_t shape;                /* MGX_SHAPE_*                                              */
    int32_t tile_row, tile_col;   /* tile_coordinates                                         */
    uint32_t sides;               /* regular polygon                                          */
    uint32_t n_points;            /* polygon                                                  */
    const double *points_xy;      /* polygon: n_points (x, y) pairs, each in [0, 1]           */
    double radius;                /* circle, triangle (inscribed circle), regular polygon     */
    double angle_a, angle_b;      /* triangle: Angles { A, B } in radians                     */
    double width, height;         /* rectangle                                                */
    double rotation;              /* radians in [0, 2 pi] (angle::Angle)                      */
    double translation_x, translation_y; /* RelativePoint: each in [0, 1]                     */
} mgx_env_obstacle;
typedef struct mgx_env_desc {
    uint32_t n_rows, n_cols;      /* TileGrid::shape (lib.rs:45-63)                           */
    const uint32_t *tiles;        /* n_rows * n_cols Unicode scalar values, row-major         */
    float tile_size, path_width;  /* TileSettings (lib.rs:589-597)                            */
    uint32_t sdf_resolution;      /* SdfSettings (lib.rs:599-617): pixels per tile,           */
    float sdf_expansion, sdf_blur; /*   expansion and blur as fractions of a tile             */
    uint32_t n_obstacles;
    const mgx_env_obstacle *obstacles;
} mgx_env_desc;
int mgx_env_image_size(const mgx_env_desc *env, uint32_t resolution, uint32_t *width, uint32_t *height);
/* env_to_image(env, resolution, expansion) (env_to_png lib.rs:165-206) */
int mgx_env_to_image(const mgx_env_desc *env, uint32_t resolution, float expansion, uint8_t *rgb);
/* env_to_sdf_image(env, resolution, expansion, blur_percent) (lib.rs:149-163): the image above,
 * blurred with image::imageops::blur(sigma = blur_percent * resolution) unless sigma < 1 */
int mgx_env_to_sdf_image(const mgx_env_desc *env, uint32_t resolution, float expansion,
                         float blur_percent, uint8_t *rgb);
/* The call chain of simulation_loader.rs:154-162 + robot.rs:1259-1264: rasterise with the
 * environment's own sdf settings and install the result as the world's obstacle image, world
 * size = tile_size * (n_cols, n_rows). */
int mgx_world_set_environment(mgx_world *w, const mgx_env_desc *env);

/* ---- host helpers (no device needed) --------------------------------------------------- */
/* gbp_schedule: fills steps[max(n_int,n_ext)] with MGX_STEP_* bits. Returns the count
 * or a negative status. (crates/gbp_schedule/src/schedules/ *.rs) */
int mgx_schedule(int32_t kind, uint8_t n_internal, uint8_t n_external, uint8_t *steps,
                 uint32_t capacity);
/* get_variable_timesteps (crates/magics/src/utils.rs:35-75). Returns the count. */
int mgx_variable_timesteps(uint32_t lookahead_horizon, uint32_t lookahead_multiple,
                           uint32_t *timesteps, uint32_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* MGX_H */
