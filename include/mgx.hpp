// mgx.hpp — C++ host side above the C ABI (include/mgx.h), shaped like the reference's Rust API so
// that code written against `magics`' FactorGraph reads the same here: the reference is compiled
// code (Rust, no toolchain in this image), so the native mirror is C++; `rust/magics-hip` carries
// the same thing as Rust source, `magics_amd/world.py` as Python.  Header-only, no dependencies
// beyond the C ABI.  Invariant violations that panic in the reference throw mgx::Error here;
// lookups that return Option return std::optional.
//
// Reference interfaces mirrored (paths under crates/magics/src unless noted):
//   FactorGraph            factorgraph/factorgraph.rs:74-76,190-226,304-353,380-436,494-528,688-826,876-890
//   VariableNode (belief)  factorgraph/variable.rs:40-54,140-166
//   iterate_gbp_v2         planner/robot.rs:1769-1861          GbpSchedule  crates/gbp_schedule
//   topology systems       planner/robot.rs:1362-1601          RobotNumberGenerator planner/robot.rs:122-140
//   MultivariateNormal     crates/gbp_multivariate_normal/src/lib.rs:38-410
#pragma once

#include <array>
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "mgx.h"

namespace mgx {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};
inline int check(int rc) {
    if (rc < 0) throw Error(rc, mgx_last_error());
    return rc;
}

using Vector4 = std::array<double, 4>;
using Matrix4 = std::array<double, 16>;  // row-major

// VariableNode.belief (variable.rs:40-54)
struct Belief {
    Vector4 information_vector, mean;
    Matrix4 precision_matrix, covariance_matrix;
    bool valid;
    std::array<double, 2> estimated_position() const { return {mean[0], mean[1]}; }
    std::array<double, 2> estimated_velocity() const { return {mean[2], mean[3]}; }
};

// MessagesSent / MessagesReceived (factorgraph/mod.rs:29-137)
struct MessageCount {
    uint64_t sent_internal, sent_external, received_internal, received_external;
};

// crates/gbp_schedule: one step of the interleaved schedule
struct GbpScheduleAtIteration {
    bool internal, external;
};
enum class GbpSchedule : int32_t {
    Centered = MGX_SCHEDULE_CENTERED,
    SoonAsPossible = MGX_SCHEDULE_SOON_AS_POSSIBLE,
    LateAsPossible = MGX_SCHEDULE_LATE_AS_POSSIBLE,
    InterleaveEvenly = MGX_SCHEDULE_INTERLEAVE_EVENLY,
    HalfBeginningHalfEnd = MGX_SCHEDULE_HALF_BEGINNING_HALF_END,
};
inline std::vector<GbpScheduleAtIteration> schedule(GbpSchedule kind, uint8_t internal, uint8_t external) {
    uint8_t steps[256];
    const int n = check(mgx_schedule(static_cast<int32_t>(kind), internal, external, steps, 256));
    std::vector<GbpScheduleAtIteration> out((size_t)n);
    for (int i = 0; i < n; i++) out[(size_t)i] = {(steps[i] & MGX_STEP_INTERNAL) != 0, (steps[i] & MGX_STEP_EXTERNAL) != 0};
    return out;
}
// utils.rs:35-75
inline std::vector<uint32_t> get_variable_timesteps(uint32_t lookahead_horizon, uint32_t lookahead_multiple) {
    uint32_t ts[1024];
    const int n = check(mgx_variable_timesteps(lookahead_horizon, lookahead_multiple, ts, 1024));
    return std::vector<uint32_t>(ts, ts + n);
}

class World;

// The reference's `FactorGraph` component: here a handle (world, robot id).  Message routing happens
// on the device, so the calls that return message vectors in the reference return nothing.
class FactorGraph {
public:
    int32_t id() const { return robot_; }
    void internal_factor_iteration();
    void internal_variable_iteration();
    void change_prior_of_variable(uint32_t variable_index, const Vector4 &mean);
    void delete_interrobot_factors_connected_to(const FactorGraph &other);
    std::optional<Belief> get_variable(uint32_t variable_index) const;
    std::optional<Belief> nth_variable(uint32_t i) const { return get_variable(i); }
    std::optional<Belief> first_variable() const { return get_variable(0); }
    std::optional<Belief> last_variable() const;
    MessageCount message_count() const;
    void set_antenna_active(bool active);
    void set_idle(bool idle);

private:
    friend class World;
    FactorGraph(World *w, int32_t robot) : world_(w), robot_(robot) {}
    World *world_;
    int32_t robot_;
};

// planner/robot.rs:122-140
struct RobotNumberGenerator {
    uint64_t next_value = 1;
    uint64_t next() { return next_value++; }
    void reset() { next_value = 1; }
};

class World {
public:
    explicit World(const mgx_params &params) { check(mgx_world_create(&params, &w_)); }
    ~World() { if (w_) mgx_world_destroy(w_); }
    World(const World &) = delete;
    World &operator=(const World &) = delete;
    mgx_world *raw() const { return w_; }

    void set_sdf(const std::vector<uint8_t> &rgb, uint32_t width, uint32_t height, double world_w, double world_h) {
        if (rgb.size() != (size_t)width * height * 3) throw Error(MGX_ERR_INVALID, "rgb size does not match width x height x 3");
        check(mgx_world_set_sdf(w_, rgb.data(), width, height, world_w, world_h));
    }
    // RobotBundle::new (robot.rs:1134-1356): K variables, their initial means, prior diagonal, delta_t
    FactorGraph add_robot(const std::vector<Vector4> &mean0, const std::vector<double> &prior_diag, const std::vector<double> &delta_t,
                          double radius, uint64_t order_key, const std::vector<std::array<float, 2>> &path = {}) {
        if (prior_diag.size() != mean0.size() || delta_t.size() + 1 != mean0.size()) throw Error(MGX_ERR_INVALID, "inconsistent sizes");
        mgx_robot_desc d{};
        d.K = (uint32_t)mean0.size();
        d.n_path = (uint32_t)path.size();
        d.mean0 = mean0[0].data();
        d.prior_diag = prior_diag.data();
        d.dt = delta_t.data();
        d.path_xy = path.empty() ? nullptr : path[0].data();
        d.radius = radius;
        d.order_key = order_key;
        int32_t id = -1;
        check(mgx_robot_add(w_, &d, &id));
        K_ = d.K;
        return FactorGraph(this, id);
    }
    void remove_robot(const FactorGraph &g) { check(mgx_robot_remove(w_, g.id())); }

    // create_interrobot_factors, one direction (robot.rs:1500-1585)
    void connect(const FactorGraph &owner, const FactorGraph &other, RobotNumberGenerator &numbers) {
        check(mgx_ir_connect(w_, owner.id(), other.id(), numbers.next_value));
        numbers.next_value += K_ - 1;
    }
    // update_robot_neighbours + delete_ + create_interrobot_factors (robot.rs:1362-1586);
    // translations: Transform::translation of every robot (x, y, z as f32), id order
    /// One FixedUpdate tick of the planner chain (robot.rs:86-103): both prior updates for the listed robots, then
    /// iterate_gbp_v2 over `steps` — one call, the prior updates ride in the launch that opens the tick
    void tick(const std::vector<int32_t> &robots, const std::vector<double> &waypoints_xy, const std::vector<double> &time_scale,
              const std::vector<uint8_t> &what, double max_speed, double delta_t, const std::vector<uint8_t> &steps) {
        check(mgx_tick(w_, (uint32_t)robots.size(), robots.data(), waypoints_xy.data(), time_scale.data(), what.data(), max_speed,
                       delta_t, steps.data(), (uint32_t)steps.size()));
    }
    /// how the last iterate_gbp_v2 / tick ran: sweep-kernel launches (1 = the whole schedule as one resident launch)
    uint32_t last_launch_count() {
        uint32_t n = 0;
        check(mgx_last_launch_count(w_, &n));
        return n;
    }
    /// everything issued so far is enqueued: a lingering launch is told to end (no wait); how long launches linger (us, 0: never)
    void flush() { check(mgx_flush(w_)); }
    void set_linger(int32_t microseconds) { check(mgx_set_linger(w_, microseconds)); }
    /// FactorGraph::change_factor_enabled (factorgraph.rs:1529-1539) for every graph: MGX_FACTOR_* bits
    void change_factor_enabled(uint32_t kind_mask) { check(mgx_set_enabled(w_, kind_mask)); }
    std::pair<uint32_t, uint32_t> update_topology(const std::vector<std::array<float, 3>> &translations, float comms_radius,
                                                  RobotNumberGenerator &numbers) {
        uint32_t stats[2] = {0, 0};
        check(mgx_update_topology(w_, translations[0].data(), comms_radius, MGX_NEIGHBOURS_AUTO, &numbers.next_value, stats));
        return {stats[0], stats[1]};
    }
    // iterate_gbp_v2 (robot.rs:1769-1861)
    void iterate_gbp_v2(const std::vector<GbpScheduleAtIteration> &sched) {
        std::vector<uint8_t> steps(sched.size());
        for (size_t i = 0; i < sched.size(); i++) steps[i] = (uint8_t)((sched[i].internal ? MGX_STEP_INTERNAL : 0) | (sched[i].external ? MGX_STEP_EXTERNAL : 0));
        check(mgx_iterate(w_, steps.data(), (uint32_t)steps.size()));
    }
    /// several iterate_gbp_v2 calls, one submission (mgx_batch_begin / mgx_batch_end): `{ auto b = world.batch(); for (...) world.iterate_gbp_v2(s); }`
    struct Batch {
        mgx_world *w;
        uint32_t schedules = 0, launches = 0;
        explicit Batch(mgx_world *world) : w(world) { check(mgx_batch_begin(w)); }
        Batch(const Batch &) = delete;
        Batch &operator=(const Batch &) = delete;
        void end() { if (w) { mgx_world *x = w; w = nullptr; check(mgx_batch_end(x, &schedules, &launches)); } }
        ~Batch() { if (w) (void)mgx_batch_end(w, nullptr, nullptr); }
    };
    Batch batch() { return Batch(w_); }
    // update_prior_of_horizon_state + update_prior_of_current_state_v3 (robot.rs:2182-2338)
    void update_priors(const std::vector<int32_t> &robots, const std::vector<std::array<double, 2>> &next_waypoints,
                       const std::vector<double> &time_scale, double max_speed, double delta_t) {
        std::vector<uint8_t> what(robots.size(), 3);
        check(mgx_update_priors(w_, (uint32_t)robots.size(), robots.data(), next_waypoints[0].data(), time_scale.data(), what.data(),
                                max_speed, delta_t));
    }
    std::vector<Vector4> read_variable_means(uint32_t variable_index) {
        uint32_t n = 0;
        check(mgx_num_robots(w_, &n, nullptr));
        std::vector<Vector4> out(n);
        if (n) check(mgx_read_variable_means(w_, variable_index, out[0].data()));
        return out;
    }
    void synchronize() { check(mgx_synchronize(w_)); }
    uint32_t K() const { return K_; }

private:
    mgx_world *w_ = nullptr;
    uint32_t K_ = 0;
};

inline void FactorGraph::internal_factor_iteration() { check(mgx_internal_factor_iteration(world_->raw(), robot_)); }
inline void FactorGraph::internal_variable_iteration() { check(mgx_internal_variable_iteration(world_->raw(), robot_)); }
inline void FactorGraph::change_prior_of_variable(uint32_t ix, const Vector4 &mean) { check(mgx_change_prior(world_->raw(), robot_, ix, mean.data())); }
inline void FactorGraph::delete_interrobot_factors_connected_to(const FactorGraph &other) { check(mgx_ir_disconnect(world_->raw(), robot_, other.robot_)); }
inline std::optional<Belief> FactorGraph::get_variable(uint32_t ix) const {
    if (ix >= world_->K()) return std::nullopt;
    Belief b{};
    int32_t valid = 0;
    check(mgx_get_belief(world_->raw(), robot_, ix, b.information_vector.data(), b.precision_matrix.data(), b.mean.data(),
                         b.covariance_matrix.data(), &valid));
    b.valid = valid != 0;
    return b;
}
inline std::optional<Belief> FactorGraph::last_variable() const { return world_->K() ? get_variable(world_->K() - 1) : std::nullopt; }
inline MessageCount FactorGraph::message_count() const {
    uint64_t c[4];
    check(mgx_message_counts(world_->raw(), robot_, c));
    return {c[0], c[1], c[2], c[3]};
}
inline void FactorGraph::set_antenna_active(bool active) { check(mgx_set_antenna(world_->raw(), robot_, active ? 1 : 0)); }
inline void FactorGraph::set_idle(bool idle) { check(mgx_set_idle(world_->raw(), robot_, idle ? 1 : 0)); }

// crates/gbp_multivariate_normal: errors are the variants of MultivariateNormalError, carried by
// mgx::Error::code (MGX_MVN_ERR_*) and what()
class MultivariateNormal {
public:
    static MultivariateNormal from_information_and_precision(const std::vector<double> &information, const std::vector<double> &precision,
                                                             uint32_t rows, uint32_t cols) {
        mgx_mvn *m = nullptr;
        check(mgx_mvn_from_information_and_precision(information.data(), (uint32_t)information.size(), precision.data(), rows, cols, &m));
        return MultivariateNormal(m);
    }
    static MultivariateNormal from_mean_and_covariance(const std::vector<double> &mean, const std::vector<double> &covariance,
                                                       uint32_t rows, uint32_t cols) {
        mgx_mvn *m = nullptr;
        check(mgx_mvn_from_mean_and_covariance(mean.data(), (uint32_t)mean.size(), covariance.data(), rows, cols, &m));
        return MultivariateNormal(m);
    }
    size_t len() const { return mgx_mvn_len(m_.get()); }
    std::vector<double> information_vector() const { std::vector<double> v(len()); check(mgx_mvn_get(m_.get(), v.data(), nullptr, nullptr)); return v; }
    std::vector<double> precision_matrix() const { std::vector<double> v(len() * len()); check(mgx_mvn_get(m_.get(), nullptr, v.data(), nullptr)); return v; }
    std::vector<double> mean() const { std::vector<double> v(len()); check(mgx_mvn_get(m_.get(), nullptr, nullptr, v.data())); return v; }
    std::vector<double> covariance() const { std::vector<double> v(len() * len()); check(mgx_mvn_covariance(m_.get(), v.data())); return v; }
    bool update() { return check(mgx_mvn_update(m_.get())) == 1; }
    void set_information_vector(const std::vector<double> &v) { check(mgx_mvn_set_information_vector(m_.get(), v.data())); }
    void set_precision_matrix(const std::vector<double> &v) { check(mgx_mvn_set_precision_matrix(m_.get(), v.data())); }
    MultivariateNormal operator+(const MultivariateNormal &o) const { return combine(o, MGX_MVN_ADD); }
    MultivariateNormal operator-(const MultivariateNormal &o) const { return combine(o, MGX_MVN_SUB); }
    MultivariateNormal operator*(const MultivariateNormal &o) const { return combine(o, MGX_MVN_MUL); }
    MultivariateNormal &operator+=(const MultivariateNormal &o) { check(mgx_mvn_combine_assign(m_.get(), o.m_.get(), MGX_MVN_ADD)); return *this; }
    MultivariateNormal &operator-=(const MultivariateNormal &o) { check(mgx_mvn_combine_assign(m_.get(), o.m_.get(), MGX_MVN_SUB)); return *this; }
    MultivariateNormal &operator*=(const MultivariateNormal &o) { check(mgx_mvn_combine_assign(m_.get(), o.m_.get(), MGX_MVN_MUL)); return *this; }

private:
    struct Del { void operator()(mgx_mvn *p) const { mgx_mvn_destroy(p); } };
    explicit MultivariateNormal(mgx_mvn *m) : m_(m) {}
    MultivariateNormal combine(const MultivariateNormal &o, int op) const {
        mgx_mvn *m = nullptr;
        check(mgx_mvn_combine(m_.get(), o.m_.get(), op, &m));
        return MultivariateNormal(m);
    }
    std::unique_ptr<mgx_mvn, Del> m_;
};

}  // namespace mgx
