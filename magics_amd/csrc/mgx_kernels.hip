// mgx_kernels.hip — the small gfx950 kernels of the GBP engine (prior changes, run-time switching of factor kinds, halo
// pack / push / wait, missions, in-place topology change) and their launch wrappers.  The sweep kernel itself is
// mgx_sweep.h, instantiated in mgx_sweep_inst.hip.
//
// Arithmetic: gbp_math.h, compiled with -ffp-contract=off so that results are bit-identical to
// the scalar f64 reference semantics.
#include "mgx_sweep.h"

namespace mgx {

// VariableNode::change_prior + routing (variable.rs:203-230, factorgraph.rs:494-528,
// robot.rs:2262-2282) for variable i of robot r.
__device__ void apply_change_prior(const DevWorld &w, int r, int i, const double (&m)[4]) {
    const int K = w.K, E = w.E, E1 = E + 1;
    const BlobLayout L(K);
    double *b = w.blob + (size_t)r * w.BS;
    const int v = r * K + i;
    // every load first (one round trip), then the stores: the thread is latency-bound
    double pl[16], be[4], bl[16];
#pragma unroll
    for (int c = 0; c < 16; c++) pl[c] = b[L.prior() + (4 + c) * K + i];
#pragma unroll
    for (int c = 0; c < 4; c++) be[c] = b[L.bel() + c * K + i];
#pragma unroll
    for (int c = 0; c < 16; c++) bl[c] = b[L.bel() + (4 + c) * K + i];
    const uint32_t epoch = w.snap_epoch[w.cur][v];
    const int e0 = w.ir_var_ptr[v], e1 = w.ir_var_ptr[v + 1];
#pragma unroll
    for (int a = 0; a < 4; a++)  // prior eta = prior lam . mean (:204)
        b[L.prior() + a * K + i] = ((pl[a * 4 + 0] * m[0] + pl[a * 4 + 1] * m[1]) + pl[a * 4 + 2] * m[2]) + pl[a * 4 + 3] * m[3];
#pragma unroll
    for (int c = 0; c < 4; c++) b[L.mu() + c * K + i] = m[c];  // :206
    // the (stale eta, stale lam, new mean) belief goes to every connected factor (:210-221):
    //   own-graph factors read it from the snapshot record ...
    double *rec = w.snap[w.cur] + (size_t)v * SNAP_W;
#pragma unroll
    for (int c = 0; c < 4; c++) rec[c] = be[c];
#pragma unroll
    for (int c = 0; c < 16; c++) rec[4 + c] = bl[c];
#pragma unroll
    for (int c = 0; c < 4; c++) rec[20 + c] = m[c];
    w.snap_epoch[w.cur][v] = epoch + 1;
    //   ... and foreign inter-robot factors attached to this variable get it in their inbox;
    // every inbox message of the variable becomes empty (:224-227)
    for (int e = e0; e < e1; e++) {
        if (w.enable & 2u) st_soa4(w.ir_bmu, w.NI, e, m);
        w.ir_fv_eta[0 * (size_t)w.NI + e] = 0.0;  // the compact form of the empty message
        w.ir_fv_eta[1 * (size_t)w.NI + e] = 0.0;
        w.ir_fv_lam[0 * (size_t)w.NI + e] = 0.0;
        w.ir_fv_lam[1 * (size_t)w.NI + e] = 0.0;
        w.ir_fv_lam[4 * (size_t)w.NI + e] = 0.0;
        w.ir_fv_lam[5 * (size_t)w.NI + e] = 0.0;
    }
    // factors that are thawing (enabled again, first update still to come) receive this delivery like any
    // enabled factor: it replaces the entry they froze with
    if (w.thaw) {
        const uint32_t tb = w.thaw[r];
        if (tb) {
            double *fz = w.frozen + (size_t)r * frozen_words(K);
            uint8_t *fl = w.frozen_flag + (size_t)r * E;
            if (tb & 1u) {
                const int lanes[2] = {(i >= 1) ? i - 1 : -1, (i <= K - 2) ? (K - 1) + i : -1};  // lanes whose OTHER variable is i
                for (int q = 0; q < 2; q++) {
                    if (lanes[q] < 0) continue;
#pragma unroll
                    for (int c = 0; c < 4; c++) fz[lanes[q] * 20 + c] = be[c];
#pragma unroll
                    for (int c = 0; c < 16; c++) fz[lanes[q] * 20 + 4 + c] = bl[c];
                    fl[lanes[q]] = 1;
                }
            }
            if (i >= 1 && i <= K - 2) {
                if (tb & 4u) {
#pragma unroll
                    for (int c = 0; c < 4; c++) fz[40 * (K - 1) + 4 * (i - 1) + c] = m[c];
                    fl[2 * (K - 1) + (i - 1)] = 1;
                }
                if (tb & 8u) {
#pragma unroll
                    for (int c = 0; c < 4; c++) fz[40 * (K - 1) + 4 * (K - 2) + 4 * (i - 1) + c] = m[c];
                    fl[2 * (K - 1) + (K - 2) + (i - 1)] = 1;
                }
            }
        }
    }
    const int n_dyn = 2 * (K - 1);
    const int es[4] = {(i >= 1) ? (K - 1) + (i - 1) : -1, (i <= K - 2) ? i : -1,
                       (i >= 1 && i <= K - 2) ? n_dyn + (i - 1) : -1,
                       (i >= 1 && i <= K - 2) ? n_dyn + (K - 2) + (i - 1) : -1};
    for (int q = 0; q < 4; q++) {
        if (es[q] < 0) continue;
#pragma unroll
        for (int c = 0; c < 20; c++) b[L.fv() + c * E1 + es[q]] = 0.0;
    }
}

// One thread per (robot, variable, mean) triple.
__global__ void k_change_prior(DevWorld w, int n, const int32_t *robots, const uint32_t *vars, const double *means) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    double m[4];
#pragma unroll
    for (int c = 0; c < 4; c++) m[c] = means[4 * t + c];
    apply_change_prior(w, robots[t], (int)vars[t], m);
}

// The per-tick prior updates of the driver, one thread per (listed robot, update):
//   what & 1: update_prior_of_horizon_state (robot.rs:2182-2283): the last variable moves towards the
//             waypoint at min(max_speed, distance);
//   what & 2: update_prior_of_current_state_v3 (robot.rs:2286-2338): variable 0 moves by
//             time_scale * (mean_1 - mean_0).
// Both end in change_prior of that variable.  The reference runs the horizon system for every robot
// before the current-state system; for K >= 3 the two touch disjoint state of a robot (variable
// K-1 and its factor slots vs variables 0, 1), so the two updates of a robot run side by side.
__global__ void k_update_priors(DevWorld w, int n, const int32_t *robots, const double *waypoints, const double *time_scale,
                                const uint8_t *what, double max_speed, double delta_t) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = g >> 1, which = g & 1;
    if (t >= n) return;
    const int r = robots[t], K = w.K;
    const BlobLayout L(K);
    const double *b = w.blob + (size_t)r * w.BS;
    if (which == 0 && (what[t] & 1u)) {
        const int i = K - 1;
        const double ex = b[L.mu() + 0 * K + i], ey = b[L.mu() + 1 * K + i];  // estimated position (:2242)
        double hx = waypoints[2 * t] - ex, hy = waypoints[2 * t + 1] - ey;     // horizon2waypoint
        const double dist = std::sqrt(hx * hx + hy * hy);                       // euclidean_norm
        double nx = hx, ny = hy;                                                 // .normalized(): unchanged if |.| is 0 / inf
        if (!(dist == 0.0 || std::isinf(dist))) { nx = hx / dist; ny = hy / dist; }
        const double sp = (max_speed < dist || dist != dist) ? max_speed : dist;  // Float::min(max_speed, dist)
        const double vx = sp * nx, vy = sp * ny;                                 // new_velocity
        const double m[4] = {ex + vx * delta_t, ey + vy * delta_t, vx, vy};      // (:2253-2256)
        apply_change_prior(w, r, i, m);
    }
    if (which == 1 && (what[t] & 2u)) {
        double m[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const double m0 = b[L.mu() + c * K + 0], m1 = b[L.mu() + c * K + 1];
            m[c] = m0 + time_scale[t] * (m1 - m0);  // (:2309-2316)
        }
        apply_change_prior(w, r, 0, m);
    }
}

// ---- factor kinds switched off and on at run time (mgx_set_enabled) ------------------------------------
// k_freeze: the inbox of every internal factor of the given kinds as it is NOW (a kind is being switched
// off: from here on these factors receive nothing, factor/mod.rs:307-310), one thread per (robot, edge slot).
__global__ void k_freeze(DevWorld w, uint32_t kinds) {
    const int K = w.K, E = w.E, E1 = E + 1, n_dyn = 2 * (K - 1);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= w.R_local * E) return;
    const int r = t / E, e = t - r * E;
    const uint32_t bit = e < n_dyn ? 1u : (e < n_dyn + (K - 2) ? 4u : 8u);
    if (!(kinds & bit)) return;
    if (w.thaw[r] & bit) return;  // still thawing from an earlier switch: the inbox it froze with then is still its inbox
    const BlobLayout L(K);
    const double *blob = w.blob + (size_t)r * w.BS;
    double *fz = w.frozen + (size_t)r * frozen_words(K);
    const int v0 = r * K;
    if (bit == 1u) {  // what the lane of k_robot_sweep would read: the other variable's snapshot minus our last message to it
        const int f = e % (K - 1), slot = e / (K - 1), o = f + 1 - slot, oe_ix = (1 - slot) * (K - 1) + f;
        const bool present = w.snap_epoch[w.cur][v0 + o] > 0;
        const double *rec = w.snap[w.cur] + (size_t)(v0 + o) * SNAP_W;
        for (int c = 0; c < 20; c++) fz[e * 20 + c] = present ? rec[c] - blob[L.fv() + c * E1 + oe_ix] : 0.0;
        w.frozen_flag[(size_t)r * E + e] = present ? 1 : 0;
    } else {
        const int j = (bit == 4u) ? e - n_dyn : e - n_dyn - (K - 2), var = j + 1;
        // a tracking factor is created WITH the variable's first message in its inbox (FG/factorgraph.rs add_internal_edge
        // hands it variable.prepare_message(), the other kinds get an empty one): before the first delivery its entry is
        // the record the snapshot starts out with (initial belief, mean = initial mean), not an empty message
        const bool present = bit == 8u || w.snap_epoch[w.cur][v0 + var] > 0;
        const double *rec = w.snap[w.cur] + (size_t)(v0 + var) * SNAP_W;
        double *dst = fz + 40 * (K - 1) + (bit == 8u ? 4 * (K - 2) : 0) + 4 * j;
        for (int c = 0; c < 4; c++) dst[c] = present ? rec[20 + c] : 0.0;
        w.frozen_flag[(size_t)r * E + e] = present ? 1 : 0;
    }
}
// thaw[r] = (thaw[r] & keep) | set: a kind is being switched on again (set) / off while still thawing (keep)
__global__ void k_or_bytes(uint8_t *p, int n, uint8_t keep, uint8_t set) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) p[t] = (uint8_t)((p[t] & keep) | set);
}
// k_thaw: runs right before a launch of k_robot_sweep that contains an internal factor sweep.  For every
// robot that takes part (not idle) and has thawing kinds, the first update of those factors is computed
// here from the inbox they froze with — same functions, same operand order as the sweep kernel — and
// written to the robot's blob; skip0[r] then tells the sweep kernel to leave those kinds alone in its first
// internal factor sweep.  One thread per (robot, edge slot).
__global__ void k_thaw(DevWorld w, int robot0, int n_robots, uint32_t ext_mask) {
    const int K = w.K, E = w.E, E1 = E + 1, n_dyn = 2 * (K - 1);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_robots * E) return;
    const int r = robot0 + t / E, e = t % E;
    const uint32_t tb = (uint32_t)w.thaw[r] & w.enable & 13u;
    const bool idle = w.idle[r] != 0;
    if (!tb || idle) return;
    if (e == 0) w.skip0[r] = (uint8_t)tb;
    const uint32_t bit = e < n_dyn ? 1u : (e < n_dyn + (K - 2) ? 4u : 8u);
    if (!(tb & bit)) return;
    const BlobLayout L(K);
    double *blob = w.blob + (size_t)r * w.BS;
    const double *fz = w.frozen + (size_t)r * frozen_words(K);
    const bool present = w.frozen_flag[(size_t)r * E + e] != 0;
    double oe[4], ol[16];
    bool ok = true;
    if (bit == 1u) {
        const int f = e % (K - 1), slot = e / (K - 1);
        const int a2 = 2 * slot, b2 = 2 * (1 - slot), it = r * (K - 1) + f;
        double maa[4], mab[4], mba[4], mbb[4], me[4], ml[16];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                maa[i * 2 + j] = w.dyn_m[(size_t)((a2 + i) * 4 + (a2 + j)) * w.ND + it];
                mab[i * 2 + j] = w.dyn_m[(size_t)((a2 + i) * 4 + (b2 + j)) * w.ND + it];
                mba[i * 2 + j] = w.dyn_m[(size_t)((b2 + i) * 4 + (a2 + j)) * w.ND + it];
                mbb[i * 2 + j] = w.dyn_m[(size_t)((b2 + i) * 4 + (b2 + j)) * w.ND + it];
            }
#pragma unroll
        for (int c = 0; c < 4; c++) me[c] = present ? fz[e * 20 + c] : 0.0;
#pragma unroll
        for (int c = 0; c < 16; c++) ml[c] = present ? fz[e * 20 + 4 + c] : 0.0;
        ok = dynamic_message(maa, mab, mba, mbb, me, ml, oe, ol);
    } else {
        const int j = (bit == 4u) ? e - n_dyn : e - n_dyn - (K - 2);
        const double *src = fz + 40 * (K - 1) + (bit == 8u ? 4 * (K - 2) : 0) + 4 * j;
        double x0[4];
#pragma unroll
        for (int c = 0; c < 4; c++) x0[c] = present ? src[c] : 0.0;
        if (bit == 4u) {
            const SdfView sdf = make_sdf_view(w.sdf, w.sdf_w, w.sdf_h, w.world_w, w.world_h);
            long long idx[4];
            obstacle_taps(sdf, x0[0], x0[1], w.obs_delta, idx);
            double h[4];
#pragma unroll
            for (int q = 0; q < 4; q++) h[q] = (idx[q] >= 0) ? sdf_value(w.sdf[idx[q]]) : 0.0;
            obstacle_message(h, w.obs_delta, w.inv_s2_obs, x0, oe, ol);
        } else {
            const bool radio = (w.antenna[r] != 0) && !idle;
            const int itf = w.iter_factor[r] + (((ext_mask & PH_EXT_FACTOR) && radio) ? 1 : 0);
            if (itf < 10) return;  // factorgraph.rs:701: the sweep kernel skips it too, the entry stays frozen
            const int item = r * (K - 2) + j;
            int rec = w.trk_record[item];
            float lp[2] = {w.trk_last_pos[item], w.trk_last_pos[(size_t)w.NT + item]};
            double lv = w.trk_last_val[item];
            const int p0 = w.path_ptr[r], np = w.path_ptr[r + 1] - p0;
            ok = tracking_update(w.path_xy + 2 * (size_t)p0, np, w.trk_pad, w.trk_attr, w.inv_s2_trk, x0, rec, lp, lv, oe, ol);
            w.trk_record[item] = rec;
            w.trk_last_pos[item] = lp[0];
            w.trk_last_pos[(size_t)w.NT + item] = lp[1];
            w.trk_last_val[item] = lv;
        }
    }
    if (!ok) {
#pragma unroll
        for (int c = 0; c < 4; c++) oe[c] = 0.0;
#pragma unroll
        for (int c = 0; c < 16; c++) ol[c] = 0.0;
    }
#pragma unroll
    for (int c = 0; c < 4; c++) blob[L.fv() + c * E1 + e] = oe[c];
#pragma unroll
    for (int c = 0; c < 16; c++) blob[L.fv() + (4 + c) * E1 + e] = ol[c];
}
// after that launch: a robot that ran an internal variable sweep has delivered fresh messages to every enabled
// factor of its graph, nothing is thawing there any more; skip0 is cleared either way
__global__ void k_thaw_done(DevWorld w, int robot0, int n_robots, int ran_variable_sweep) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_robots) return;
    const int r = robot0 + t;
    if (ran_variable_sweep && !w.idle[r]) w.thaw[r] = 0;
    w.skip0[r] = 0;
}

// Inter-robot factors going off: what every variable has last sent to its own factors is what the inter-robot
// factors it owns keep in their inbox.  A variable that has not delivered since an earlier switch-on keeps
// the record frozen then.
__global__ void k_ir_freeze(DevWorld w, double *frozen_snap, uint32_t *frozen_epoch) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= w.V) return;
    if (w.ir_thaw_epoch && w.snap_epoch[w.cur][v] == w.ir_thaw_epoch[v]) return;
    for (int c = 0; c < SNAP_W; c++) frozen_snap[(size_t)v * SNAP_W + c] = w.snap[w.cur][(size_t)v * SNAP_W + c];
    frozen_epoch[v] = w.snap_epoch[w.cur][v];
}
// ... and coming back: in front of a launch with an external factor sweep, the factors whose owner's variable has
// not delivered since are evaluated here from the frozen record (same function as the sweep kernel's edge lane)
// and marked so that the edge lane leaves them alone.  One thread per (robot, incoming edge).
__global__ void k_thaw_ir(DevWorld w, uint8_t *gate) {
    const int r = blockIdx.x;
    const int v0 = r * w.K, ie0 = w.ir_var_ptr[v0], ne = w.ir_var_ptr[v0 + w.K] - ie0;
    const bool radio = (w.antenna[r] != 0) && (w.idle[r] == 0);
    if (!radio || !(w.enable & 2u)) return;
    for (int j = threadIdx.x; j < ne; j += blockDim.x) {
        const int e = ie0 + j;
        const uint8_t g = gate[e];
        if (!g || g == 3) continue;  // 3: a factor that still lacks inbox keys, settled by k_keyless_ir
        const IrEdgeRec er = w.ir_rec[e];
        if (w.snap_epoch[w.cur][er.src_var] != w.ir_thaw_epoch[er.src_var]) {  // the owner has delivered since: live record
            if (g == 2) gate[e] = 1;
            continue;
        }
        double ao_eta[4], ao_lam[16], a_mu[4], b_mu[4], o6[6];
        ld_soa4(w.ir_bmu, w.NI, e, b_mu);
        const bool a_present = w.ir_frozen_epoch[er.src_var] > er.created;
        const double *rec = w.ir_frozen_snap + (size_t)er.src_var * SNAP_W;
#pragma unroll
        for (int c = 0; c < 4; c++) ao_eta[c] = a_present ? rec[c] : 0.0;
#pragma unroll
        for (int c = 0; c < 16; c++) ao_lam[c] = a_present ? rec[4 + c] : 0.0;
#pragma unroll
        for (int c = 0; c < 4; c++) a_mu[c] = a_present ? rec[20 + c] : 0.0;
        const int dslot = er.dst >> 16;
        double x_lo[4], x_hi[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            x_lo[c] = dslot ? a_mu[c] : b_mu[c];
            x_hi[c] = dslot ? b_mu[c] : a_mu[c];
        }
        const bool ok = interrobot_message_compact(x_lo, x_hi, er.d_safe, er.offset, w.inv_s2_ir, dslot, ao_eta, ao_lam, o6);
        w.ir_fv_eta[0 * (size_t)w.NI + e] = ok ? o6[0] : 0.0;
        w.ir_fv_eta[1 * (size_t)w.NI + e] = ok ? o6[1] : 0.0;
        w.ir_fv_lam[0 * (size_t)w.NI + e] = ok ? o6[2] : 0.0;
        w.ir_fv_lam[1 * (size_t)w.NI + e] = ok ? o6[3] : 0.0;
        w.ir_fv_lam[4 * (size_t)w.NI + e] = ok ? o6[4] : 0.0;
        w.ir_fv_lam[5 * (size_t)w.NI + e] = ok ? o6[5] : 0.0;
        gate[e] = 2;
    }
}

// Inter-robot factors that still lack inbox keys (KeylessRec, mgx_dev.h), in front of a launch with an external factor sweep:
//   no key, or only the owner's variable's: the factor has no entry to answer for the target's variable — nothing is sent,
//     the message the variable holds stays (factor/mod.rs:412-449 iterates the keys it has);
//   only the target's variable's: that variable is the factor's ONLY inbox entry, so it sits in slot 0 of the linearisation
//     point whatever the two graphs' order, the other slot is zeros, and the message to it is marginalised with nothing added
//     (factor/mod.rs:336-349, marginalise_factor_distance.rs:74-127);
//   both: the sweep kernel's edge lane evaluates it like any other factor.
// Edges handled here are marked gate 3 ("on air, settled: neither the edge lane nor k_thaw_ir touches it") for the coming launch.
__global__ void k_keyless_ir(DevWorld w, uint8_t *gate, int n, const KeylessRec *__restrict__ recs) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const KeylessRec kr = recs[t];
    const int e = kr.edge;
    if (!gate[e] || kr.keys == 3u) return;  // owner off the air: nobody evaluates it; complete: the edge lane does
    const bool radio_b = w.antenna[kr.tgt_robot] != 0 && w.idle[kr.tgt_robot] == 0;
    if (!radio_b || !(w.enable & 2u)) return;
    if (kr.keys == 2u) {
        const IrEdgeRec er = w.ir_rec[e];
        double b_mu[4], zero4[4] = {0.0, 0.0, 0.0, 0.0}, zero16[16], oe[4], ol[16];
#pragma unroll
        for (int c = 0; c < 16; c++) zero16[c] = 0.0;
        ld_soa4(w.ir_bmu, w.NI, e, b_mu);
        const bool ok = interrobot_message(b_mu, zero4, er.d_safe, er.offset, w.inv_s2_ir, 0, zero4, zero16, oe, ol);
        w.ir_fv_eta[0 * (size_t)w.NI + e] = ok ? oe[0] : 0.0;
        w.ir_fv_eta[1 * (size_t)w.NI + e] = ok ? oe[1] : 0.0;
        w.ir_fv_lam[0 * (size_t)w.NI + e] = ok ? ol[0] : 0.0;
        w.ir_fv_lam[1 * (size_t)w.NI + e] = ok ? ol[1] : 0.0;
        w.ir_fv_lam[4 * (size_t)w.NI + e] = ok ? ol[4] : 0.0;
        w.ir_fv_lam[5 * (size_t)w.NI + e] = ok ? ol[5] : 0.0;
    }
    gate[e] = 3;
}

// halo: the snapshot records (variables 0..K-1: eta, lam, mu; then the K epochs) of whole robots
__global__ void k_halo_pack(DevWorld w, int n, const int32_t *robots, double *buf) {
    const int words = (SNAP_W + 1) * w.K;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * words) return;
    const int rr = t / words, q = t % words;
    const int v0 = robots[rr] * w.K;
    buf[t] = (q < SNAP_W * w.K) ? w.snap[w.cur][(size_t)v0 * SNAP_W + q] : (double)w.snap_epoch[w.cur][v0 + (q - SNAP_W * w.K)];
}
__global__ void k_halo_unpack(DevWorld w, int n, const int32_t *ghosts, const double *buf) {
    const int words = (SNAP_W + 1) * w.K;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * words) return;
    const int rr = t / words, q = t % words;
    const int v0 = ghosts[rr] * w.K;
    if (q < SNAP_W * w.K)
        w.snap[w.cur][(size_t)v0 * SNAP_W + q] = buf[t];
    else
        w.snap_epoch[w.cur][v0 + (q - SNAP_W * w.K)] = (uint32_t)buf[t];
}

// ---- direct halo exchange: peer-mapped stores over xGMI -----------------------------
// The receive areas and arrival counters are fine-grained device memory of the CONSUMER rank,
// mapped into this process (hipIpc*, or the same address space when the ranks share a process).
// Every access to them is a system-scope atomic, so no cache of either GPU can hold them stale.
//
// push: the snapshot records of this rank's boundary robots go straight into every consumer's
// receive area (dst[rr] = address of the record of sent robot rr, for this exchange's parity);
// the last workgroup to finish then publishes the exchange number in every consumer's counter.
__global__ void __launch_bounds__(256) k_halo_push(DevWorld w, int n, const int32_t *robots, const unsigned long long *dst, int n_peers,
                                                   const unsigned long long *peer_flags, unsigned long long seq, unsigned int *done) {
    const int words = (SNAP_W + 1) * w.K;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n * words) {
        const int rr = t / words, q = t % words;
        const int v0 = robots[rr] * w.K;
        const double val = (q < SNAP_W * w.K) ? w.snap[w.cur][(size_t)v0 * SNAP_W + q] : (double)w.snap_epoch[w.cur][v0 + (q - SNAP_W * w.K)];
        double *d = reinterpret_cast<double *>(dst[rr]);
        __hip_atomic_store(&d[q], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int prev = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {  // every workgroup's records are out
            __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            for (int p = 0; p < n_peers; p++)
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(peer_flags[p]), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// wait + unpack: workgroup 0 waits until all producers have published exchange `seq` (bounded: after
// `timeout_ticks` of the 100 MHz wall clock it records the failure — in the rank's error word and in the host-mapped
// one the host checks after every synchronisation) and then announces the exchange in `ready`; every other workgroup
// waits for that announcement only, so the whole launch follows ONE decision: either every ghost record of the
// exchange is unpacked or none is, and once an exchange has failed no later one unpacks anything (the error words
// are never cleared: the world's beliefs are no longer trusted, mgx_synchronize / mgx_read_* / the next sweep say so).
// by_slot: the record of ghost robot g sits in slot (g's place among this rank's ghosts) of the receive area, not at its place in the
// receive list (mgx_halo_direct_setup_slots: a wiring that outlives the lists)
__global__ void __launch_bounds__(256) k_halo_wait_unpack(DevWorld w, int n, const int32_t *ghosts, const double *recv, int n_sources,
                                                          const unsigned long long *flags, unsigned long long seq,
                                                          unsigned long long *err, long long timeout_ticks, unsigned long long *ready,
                                                          unsigned long long *host_err, int by_slot) {
    if (blockIdx.x == 0) {
        for (int j = threadIdx.x; j < n_sources; j += blockDim.x) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(&flags[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
                if (wall_clock64() - t0 > timeout_ticks) {
                    __hip_atomic_store(err, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (host_err) __hip_atomic_store(host_err, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                __builtin_amdgcn_s_sleep(16);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(ready, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else if (threadIdx.x == 0) {
        while (__hip_atomic_load(ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < seq) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0ull) return;  // this exchange or an earlier one failed
    const int words = (SNAP_W + 1) * w.K;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * words) return;
    const int rr = t / words, q = t % words;
    const int v0 = ghosts[rr] * w.K;
    const size_t at = by_slot ? (size_t)(ghosts[rr] - w.R_local) * (size_t)words + (size_t)q : (size_t)t;
    const double val = __hip_atomic_load(&recv[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (q < SNAP_W * w.K)
        w.snap[w.cur][(size_t)v0 * SNAP_W + q] = val;
    else
        w.snap_epoch[w.cur][v0 + (q - SNAP_W * w.K)] = (uint32_t)val;
}

// belief mean of ONE variable of every local robot -> out[R][4] (the per-tick reads of the driver:
// reached_waypoint, the Transform increment of update_prior_of_current_state_v3)
__global__ void k_gather_variable_means(DevWorld w, int var, double *__restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= w.R_local * 4) return;
    const int r = t >> 2, c = t & 3;
    const BlobLayout L(w.K);
    out[t] = w.blob[(size_t)r * w.BS + L.mu() + c * w.K + var];
}

// ---- missions on the device (mgx_mission_tick) ---------------------------------------------------------------
// reached_waypoint (robot.rs:2080-2176): the estimated position (belief mean of the rule's variable, as f32) against the
// next waypoint, squared distance in f32 against the rule's limit; a robot that reaches its last waypoint is reported in
// the host-mapped event list (ev[0] = count, ev[1 ..] = robot ids) that the host reads at the tick's one synchronisation.
__global__ void k_mission_reached(DevWorld w, DevMission m, int n, long long tick, unsigned int *ev) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n || !m.has[r]) return;
    const int n_wp = m.wp_ptr[r + 1] - m.wp_ptr[r];
    const int t = m.target[r];
    if (t >= n_wp) return;
    const bool last = t == n_wp - 1;
    const uint32_t var = m.vars[2 * r + (last ? 1 : 0)];
    const BlobLayout L(w.K);
    const double *b = w.blob + (size_t)r * w.BS;
    const float ex = (float)b[L.mu() + 0 * w.K + var], ey = (float)b[L.mu() + 1 * w.K + var];
    const float dx = ex - (float)m.wp_xy[2 * (m.wp_ptr[r] + t)], dy = ey - (float)m.wp_xy[2 * (m.wp_ptr[r] + t) + 1];
    if (dx * dx + dy * dy < m.dist2[2 * r + (last ? 1 : 0)]) {
        m.target[r] = t + 1;
        if (last) {
            m.finished_tick[r] = tick;
            const unsigned slot = atomicAdd_system(&ev[0], 1u);
            ev[1 + slot] = (unsigned)r;
        }
    }
}
// Transform::translation of the robots the neighbour search looks at (those the host knew alive when it launched)
__global__ void k_mission_positions(DevMission m, int n, const int32_t *__restrict__ alive, float *__restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * n) return;
    out[t] = m.translation[3 * alive[t / 3] + t % 3];
}
// The inputs of the tick's two prior updates for every robot, in the record form k_robot_sweep applies inside its first
// launch (waypoint x, y, time scale, what) and as the lists k_update_priors takes; and the Transform increment of
// update_prior_of_current_state_v3 (robot.rs:2309-2330: change_in_state = time_scale * (mean_1 - mean_0), its position part
// added to the translation as f32) — from the means as they are BEFORE the prior updates of this tick.
__global__ void k_mission_prepare(DevWorld w, DevMission m, int n, const uint8_t *__restrict__ moving, double *__restrict__ rec,
                                  int32_t *__restrict__ robots, double *__restrict__ waypoints, double *__restrict__ time_scale,
                                  uint8_t *__restrict__ what) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int n_wp = m.has[r] ? m.wp_ptr[r + 1] - m.wp_ptr[r] : 0;
    const int t = m.has[r] ? m.target[r] : 0;
    const bool go = moving[r] && m.has[r] && t < n_wp;  // alive and a next waypoint exists (robot.rs:2216-2228)
    double wx = 0.0, wy = 0.0;
    const double ts = m.time_scale[r];
    if (go) {
        wx = m.wp_xy[2 * (m.wp_ptr[r] + t)];
        wy = m.wp_xy[2 * (m.wp_ptr[r] + t) + 1];
        const BlobLayout L(w.K);
        const double *b = w.blob + (size_t)r * w.BS;
        const double c0 = ts * (b[L.mu() + 0 * w.K + 1] - b[L.mu() + 0 * w.K + 0]);
        const double c1 = ts * (b[L.mu() + 1 * w.K + 1] - b[L.mu() + 1 * w.K + 0]);
        m.translation[3 * r + 0] += (float)c0;  // robot.rs:2328-2329
        m.translation[3 * r + 2] += (float)c1;
    }
    rec[4 * r + 0] = wx; rec[4 * r + 1] = wy; rec[4 * r + 2] = ts; rec[4 * r + 3] = go ? 3.0 : 0.0;
    robots[r] = r;
    waypoints[2 * r] = wx; waypoints[2 * r + 1] = wy;
    time_scale[r] = ts;
    what[r] = go ? 3 : 0;
}

// In-place topology change.  A robot's incoming connections are kept as one sorted list of SLOTS;
// every connection hangs one factor on each of the target's variables 1..K-1, so the edges of
// variable i of robot r are  (K-1) * in_ptr[r] + (i-1) * n_in(r) + q,  q = position in the list.
// k_retopo_robots lays out the new edge arrays from the old ones, one workgroup of two waves per robot: a surviving connection carries
// its message (the six live numbers), response mean and creation epoch over from the arrays being replaced; a new one starts
// empty, created at the owner variable's current delivery count, with the target variable's current belief mean as the response
// it has seen (robot.rs:1549-1585).  The constant record of every edge is derived here too.
// What the host sends are DIFFERENCES (RetopoBlock, one pinned block read over the host link): a 16-byte header per robot — its
// slot range, key split and peer-row start in the new layout — and slot records and peer rows ONLY for the robots whose lists
// changed; everybody else (header.off < 0) keeps the slot records and the peer row the device holds since the last pass
// (slots_old, peers_old) and its edges just move by the shift of its range.  The same workgroup writes the robot's part of
// everything derived: the device copy of the slot ranges (the next pass's "old" ones), the per-variable tables, its peer row,
// the slot records of the new layout.  ONE launch, no copies, no synchronisation; about two reads over the host link per robot
// (the kernel's time is their number, not their bytes: four scattered words per robot were 19 us for a thousand robots).
// gate (may be null): the edge's gate byte — its owner is on air — written along (the flags themselves have not changed)
constexpr int RETOPO_THREADS = 128;  // two waves per robot: a robot's hundred-odd edges are ONE round of dependent loads instead of two
__global__ void __launch_bounds__(RETOPO_THREADS) k_retopo_robots(DevWorld w, RetopoBlock b, const int32_t *__restrict__ in_old, const IrSlotRec *__restrict__ slots_old,
                                                      const int32_t *__restrict__ peers_old, int32_t *__restrict__ in_dst, IrSlotRec *__restrict__ slots_new,
                                                      int32_t *__restrict__ peers_dst, int32_t *__restrict__ var_ptr, int32_t *__restrict__ var_mid,
                                                      int stride_new, IrEdgeRec *__restrict__ recs, double *__restrict__ fv_eta, double *__restrict__ fv_lam,
                                                      double *__restrict__ bmu, uint8_t *__restrict__ gate) {
    const int r = blockIdx.x, lane = threadIdx.x, R = w.R_local, K = w.K, K1 = K - 1;
    // the robot's header and its successor's: 32 bytes, one read over the host link (threads 0 and 1), handed round through LDS
    __shared__ int4 s_h[2];
    if (lane < 2) s_h[lane] = *reinterpret_cast<const int4 *>(&b.hdr[r + lane]);
    __syncthreads();
    const int in0 = s_h[0].x, pp0 = s_h[0].y, mid = s_h[0].z, off = s_h[0].w;
    const int in1 = s_h[1].x, pp1 = s_h[1].y;
    const int n_new = in1 - in0, base_new = K1 * in0;
    const int o0 = in_old[r], n_old = in_old[r + 1] - o0, base_old = K1 * o0;
    const IrSlotRec *mine = off < 0 ? slots_old + o0 : reinterpret_cast<const IrSlotRec *>(b.data + off);  // (its list did not change: n_old == n_new, same positions)
    if (lane == 0) {
        in_dst[r] = in0;
        if (r == R - 1) { in_dst[R] = in1; var_ptr[R * K] = K1 * in1; }
        if (peers_dst) { peers_dst[r] = pp0; if (r == R - 1) peers_dst[R] = pp1; }
    }
    for (int i = lane; i < K; i += RETOPO_THREADS) {
        const int p = (i == 0) ? base_new : base_new + (i - 1) * n_new;  // variable 0 carries no inter-robot factor
        var_ptr[r * K + i] = p;
        var_mid[r * K + i] = (i == 0) ? p : p + mid;
    }
    if (peers_dst) {  // (a row's length is the same in both tables when the robot's lists did not change)
        const int32_t *row = off < 0 ? peers_old + R + 1 + peers_old[r] : reinterpret_cast<const int32_t *>(b.data + off + 2 * n_new);
        for (int q = lane; q < pp1 - pp0; q += RETOPO_THREADS) peers_dst[R + 1 + pp0 + q] = row[q];
    }
    const size_t sn = (size_t)stride_new, so = (size_t)w.NI;
    for (int t = lane; t < n_new * K1; t += RETOPO_THREADS) {
        const int j = t / n_new, q = t - j * n_new;
        IrSlotRec sl = mine[q];
        const int oq = sl.old_slot;  // (its position in the robot's list being replaced — its own, where nothing changed; -1 = created now)
        const int e = base_new + j * n_new + q;
        IrEdgeRec rec;
        rec.src_var = sl.src_robot * K + j + 1;
        rec.src_robot = sl.src_robot;
        rec.dst = (int32_t)(j + 1) | ((sl.flags & 1) ? (1 << 16) : 0);
        rec.d_safe = sl.d_safe;
        rec.offset = (double)1e-6f * (double)(sl.first_number + (unsigned long long)j);  // interrobot.rs:52,75
        if (oq >= 0) {
            const int o = base_old + j * n_old + oq;
            fv_eta[0 * sn + e] = w.ir_fv_eta[0 * so + o];
            fv_eta[1 * sn + e] = w.ir_fv_eta[1 * so + o];
            fv_lam[0 * sn + e] = w.ir_fv_lam[0 * so + o];
            fv_lam[1 * sn + e] = w.ir_fv_lam[1 * so + o];
            fv_lam[4 * sn + e] = w.ir_fv_lam[4 * so + o];
            fv_lam[5 * sn + e] = w.ir_fv_lam[5 * so + o];
#pragma unroll
            for (int c = 0; c < 4; c++) bmu[c * sn + e] = w.ir_bmu[c * so + o];
            rec.created = w.ir_rec[o].created;
        } else {
            fv_eta[0 * sn + e] = 0.0;
            fv_eta[1 * sn + e] = 0.0;
            fv_lam[0 * sn + e] = 0.0;
            fv_lam[1 * sn + e] = 0.0;
            fv_lam[4 * sn + e] = 0.0;
            fv_lam[5 * sn + e] = 0.0;
            const BlobLayout L(K);
#pragma unroll
            for (int c = 0; c < 4; c++)  // the target's belief goes into the new factor — which drops it while its kind is off
                bmu[c * sn + e] = (w.enable & 2u) ? w.blob[(size_t)r * w.BS + L.mu() + c * K + (j + 1)] : 0.0;
            rec.created = w.snap_epoch[w.cur][rec.src_var];
        }
        recs[e] = rec;
        if (gate) gate[e] = (w.antenna[rec.src_robot] && !w.idle[rec.src_robot]) ? 1 : 0;
        if (j == 0) {
            sl.old_slot = q;
            slots_new[in0 + q] = sl;
        }
    }
}
// CSR over variables (and the lower-key / higher-key split) from the per-robot slot lists
__global__ void k_var_tables(int R, int K, const int32_t *__restrict__ in_ptr, const int32_t *__restrict__ in_mid,
                             int32_t *__restrict__ var_ptr, int32_t *__restrict__ var_mid) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > R * K) return;
    if (t == R * K) { var_ptr[t] = (K - 1) * in_ptr[R]; return; }
    const int r = t / K, i = t - r * K;
    const int n_in = in_ptr[r + 1] - in_ptr[r], base = (K - 1) * in_ptr[r];
    const int p = (i == 0) ? base : base + (i - 1) * n_in;  // variable 0 carries no inter-robot factor
    var_ptr[t] = p;
    var_mid[t] = (i == 0) ? p : p + in_mid[r];
}
// gate byte of every edge: its OWNER is on air (antenna on, not idle)
__global__ void k_edge_gates(int n, const IrEdgeRec *__restrict__ recs, const uint8_t *__restrict__ antenna,
                             const uint8_t *__restrict__ idle, uint8_t *__restrict__ gate) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int a = recs[e].src_robot;
    gate[e] = (antenna[a] && !idle[a]) ? 1 : 0;
}

// small byte copy (flag tables from the pinned argument ring into their device arrays)
__global__ void k_copy_bytes(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = src[t];
}

// ---- launch wrappers (called from mgx_world.hip) -------------------------------------------------
size_t sweep_lds_bytes(int K, int ir_edges, bool resident) {  // resident: + the shadow of the factor -> variable messages + 1232 bytes: the parked argument pointer, the plan in force, a posted plan's prior-update record and, per wave, what is derived from the plan (mgx_sweep.h: s_plan, s_urec, s_ps, s_seg)
    const BlobLayout L(K);
    const int io = L.inout_words() + (L.inout_words() & 1);
    return sizeof(double) * (size_t)((SNAP_W + 20 + 20) * K + io + IR_STRIDE * (ir_edges + 1) + (resident ? 20 * L.E1 + 154 : 0)) +
           4 * (size_t)((resident ? 3 : 2) * ((K + 1) & ~1) + ((3 * (K + 1) + 1) & ~1));
}
size_t sweep_lds_bytes(int K, int ir_edges) { return sweep_lds_bytes(K, ir_edges, false); }
bool sweep_supports(int K) { return K >= 3 && 2 * (K - 1) <= 128; }  // beyond 33 variables: two dynamic messages per lane
int blob_words(int K) { const BlobLayout L(K); return (L.words() + 1) & ~1; }

// ---- the sweep kernel's instantiations live in mgx_sweep_inst.hip (several objects, one per flavour and horizon set) ----
// horizon lengths of BASELINE.json / the reference scenarios get constant-K code; 0 / -1: the run-time-K kernels
static int sweep_variant_of(int K) {
    switch (K) {
    case 10: case 11: case 12: case 13: case 16: case 17: case 20: case 21: case 32: case 35: return K;
    default: return 2 * (K - 1) > 64 ? -1 : 0;
    }
}
#define MGX_DECLARE_SET(N)                                                                                                          \
    bool sweep_plain_set##N(int kt, const DevWorld &w, int robot0, int n_robots, uint32_t ext_mask, uint32_t int_mask, int n_int,   \
                            int snap_out, uint32_t hints, hipStream_t stream);                                                      \
    int resident_capacity_set##N(int kt, const DevWorld &w);                                                                        \
    bool resident_launch_set##N(int kt, const DevWorld &w, int n_robots, const SegPlan &plan, bool cooperative, hipStream_t stream, \
                                hipError_t *err);                                                                                   \
    int sharded_capacity_set##N(int kt, const DevWorld &w);                                                                         \
    bool sharded_launch_set##N(int kt, const DevWorld &w, int n_robots, const SegPlan &plan, bool cooperative, hipStream_t stream,  \
                               hipError_t *err);
MGX_DECLARE_SET(0)
MGX_DECLARE_SET(1)
MGX_DECLARE_SET(2)
#undef MGX_DECLARE_SET

hipError_t launch_robot_sweep(const DevWorld &w, int robot0, int n_robots, uint32_t ext_mask, uint32_t int_mask, int n_int,
                              int snap_out, uint32_t hints, hipStream_t stream) {
    if (n_robots <= 0) return hipSuccess;
    const int kt = sweep_variant_of(w.K);
    if (!sweep_plain_set0(kt, w, robot0, n_robots, ext_mask, int_mask, n_int, snap_out, hints, stream) &&
        !sweep_plain_set1(kt, w, robot0, n_robots, ext_mask, int_mask, n_int, snap_out, hints, stream) &&
        !sweep_plain_set2(kt, w, robot0, n_robots, ext_mask, int_mask, n_int, snap_out, hints, stream))
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---- resident schedule launches ------------------------------------------------------------------------
constexpr size_t RESIDENT_LDS_MAX = 160 * 1024;
size_t sweep_resident_lds_max() { return RESIDENT_LDS_MAX; }
// workgroups of the resident kernel the device holds at once (sharded: of the instantiation that takes ghost records in-launch)
int sweep_resident_capacity(const DevWorld &w, bool sharded) {
    const int kt = sweep_variant_of(w.K);
    int cap = sharded ? sharded_capacity_set0(kt, w) : resident_capacity_set0(kt, w);
    if (cap < 0) cap = sharded ? sharded_capacity_set1(kt, w) : resident_capacity_set1(kt, w);
    if (cap < 0) cap = sharded ? sharded_capacity_set2(kt, w) : resident_capacity_set2(kt, w);
    return cap < 0 ? 0 : cap;
}
hipError_t launch_robot_schedule(const DevWorld &w, int n_robots, const SegPlan &plan, bool sharded, bool cooperative, hipStream_t stream) {
    if (n_robots <= 0 || plan.n <= 0) return hipSuccess;
    const int kt = sweep_variant_of(w.K);
    hipError_t e = hipErrorInvalidValue;
    if (sharded) {
        if (!sharded_launch_set0(kt, w, n_robots, plan, cooperative, stream, &e) && !sharded_launch_set1(kt, w, n_robots, plan, cooperative, stream, &e))
            (void)sharded_launch_set2(kt, w, n_robots, plan, cooperative, stream, &e);
    } else {
        if (!resident_launch_set0(kt, w, n_robots, plan, cooperative, stream, &e) && !resident_launch_set1(kt, w, n_robots, plan, cooperative, stream, &e))
            (void)resident_launch_set2(kt, w, n_robots, plan, cooperative, stream, &e);
    }
    return e;
}

// A rank of a sharded world that cannot run the schedule the ranks are about to run as resident launches (SegPlan::agree_seq)
// says so where the others look: abort on the agreement word, and the same in its own decision words, so that its host takes
// the launch-by-launch path through the same door as theirs.
__global__ void k_agree_abort(DevWorld w, unsigned long long launch_seq, unsigned long long agree_seq) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (agree_seq != 0ull) (void)agree_on_launch(w.agree, agree_seq, (unsigned)w.n_ranks, AGREE_ABORT);
    __hip_atomic_store(w.decision, launch_seq * 4ull + RESIDENT_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w.decision_host, launch_seq * 4ull + RESIDENT_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_agree_abort(const DevWorld &w, const SegPlan &plan, hipStream_t stream) {
    hipLaunchKernelGGL(k_agree_abort, dim3(1), dim3(64), 0, stream, w, plan.launch_seq, plan.agree_seq);
    return hipGetLastError();
}

hipError_t launch_change_prior(const DevWorld &w, int n, const int32_t *robots, const uint32_t *vars, const double *means,
                               hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_change_prior, dim3((n + 63) / 64), dim3(64), 0, stream, w, n, robots, vars, means);
    return hipGetLastError();
}
hipError_t launch_update_priors(const DevWorld &w, int n, const int32_t *robots, const double *waypoints, const double *time_scale,
                                const uint8_t *what, double max_speed, double delta_t, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_update_priors, dim3((2 * n + 63) / 64), dim3(64), 0, stream, w, n, robots, waypoints, time_scale, what,
                       max_speed, delta_t);
    return hipGetLastError();
}
hipError_t launch_halo_push(const DevWorld &w, int n, const int32_t *robots, const unsigned long long *dst, int n_peers,
                            const unsigned long long *peer_flags, unsigned long long seq, unsigned int *done, hipStream_t stream, bool always) {
    if (n <= 0 && !always) return hipSuccess;  // (always: a rank with nothing to send still tells every peer the exchange's number)
    const int total = std::max(n, 0) * (SNAP_W + 1) * w.K + (n <= 0 ? 1 : 0);
    hipLaunchKernelGGL(k_halo_push, dim3((total + 255) / 256), dim3(256), 0, stream, w, n, robots, dst, n_peers, peer_flags, seq, done);
    return hipGetLastError();
}
hipError_t launch_halo_wait_unpack(const DevWorld &w, int n, const int32_t *ghosts, const double *recv, int n_sources,
                                   const unsigned long long *flags, unsigned long long seq, unsigned long long *err,
                                   long long timeout_ticks, unsigned long long *ready, unsigned long long *host_err, hipStream_t stream,
                                   bool by_slot) {
    if (n <= 0 && !by_slot) return hipSuccess;  // (slots: a rank that receives nothing still waits for every peer — that is its flow control)
    const int total = std::max(n, 0) * (SNAP_W + 1) * w.K + (n <= 0 ? 1 : 0);
    hipLaunchKernelGGL(k_halo_wait_unpack, dim3((total + 255) / 256), dim3(256), 0, stream, w, std::max(n, 0), ghosts, recv, n_sources, flags, seq,
                       err, timeout_ticks, ready, host_err, by_slot ? 1 : 0);
    return hipGetLastError();
}
hipError_t launch_gather_variable_means(const DevWorld &w, int var, double *out, hipStream_t stream) {
    if (w.R_local <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_gather_variable_means, dim3((unsigned)((w.R_local * 4 + 255) / 256)), dim3(256), 0, stream, w, var, out);
    return hipGetLastError();
}
hipError_t launch_mission_reached(const DevWorld &w, const DevMission &m, int n, long long tick, unsigned int *ev, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_mission_reached, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, stream, w, m, n, tick, ev);
    return hipGetLastError();
}
hipError_t launch_mission_positions(const DevMission &m, int n, const int32_t *alive, float *out, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_mission_positions, dim3((unsigned)((3 * n + 255) / 256)), dim3(256), 0, stream, m, n, alive, out);
    return hipGetLastError();
}
hipError_t launch_mission_prepare(const DevWorld &w, const DevMission &m, int n, const uint8_t *moving, double *rec, int32_t *robots,
                                  double *waypoints, double *time_scale, uint8_t *what, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_mission_prepare, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, stream, w, m, n, moving, rec, robots, waypoints,
                       time_scale, what);
    return hipGetLastError();
}
hipError_t launch_retopo_robots(const DevWorld &w, const RetopoBlock &b, const int32_t *in_old, const IrSlotRec *slots_old, const int32_t *peers_old,
                                int32_t *in_dst, IrSlotRec *slots_new, int32_t *peers_dst, int32_t *var_ptr, int32_t *var_mid, int stride_new,
                                IrEdgeRec *recs, double *fv_eta, double *fv_lam, double *bmu, uint8_t *gate, hipStream_t stream) {
    if (w.R_local <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_retopo_robots, dim3((unsigned)w.R_local), dim3(RETOPO_THREADS), 0, stream, w, b, in_old, slots_old, peers_old, in_dst, slots_new, peers_dst,
                       var_ptr, var_mid, stride_new, recs, fv_eta, fv_lam, bmu, gate);
    return hipGetLastError();
}
hipError_t launch_var_tables(int R, int K, const int32_t *in_ptr, const int32_t *in_mid, int32_t *var_ptr, int32_t *var_mid,
                             hipStream_t stream) {
    hipLaunchKernelGGL(k_var_tables, dim3((unsigned)((R * K + 1 + 255) / 256)), dim3(256), 0, stream, R, K, in_ptr, in_mid, var_ptr, var_mid);
    return hipGetLastError();
}
hipError_t launch_edge_gates(int n, const IrEdgeRec *recs, const uint8_t *antenna, const uint8_t *idle, uint8_t *gate, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_edge_gates, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, recs, antenna, idle, gate);
    return hipGetLastError();
}
hipError_t launch_freeze(const DevWorld &w, uint32_t kinds, hipStream_t stream) {
    const int n = w.R_local * w.E;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_freeze, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, kinds);
    return hipGetLastError();
}
hipError_t launch_or_bytes(uint8_t *p, int n, uint8_t keep, uint8_t set, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_or_bytes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, n, keep, set);
    return hipGetLastError();
}
hipError_t launch_thaw(const DevWorld &w, int robot0, int n_robots, uint32_t ext_mask, hipStream_t stream) {
    const int n = n_robots * w.E;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_thaw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, robot0, n_robots, ext_mask);
    return hipGetLastError();
}
hipError_t launch_thaw_done(const DevWorld &w, int robot0, int n_robots, int clear, hipStream_t stream) {
    if (n_robots <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_thaw_done, dim3((unsigned)((n_robots + 255) / 256)), dim3(256), 0, stream, w, robot0, n_robots, clear);
    return hipGetLastError();
}
hipError_t launch_ir_freeze(const DevWorld &w, double *frozen_snap, uint32_t *frozen_epoch, hipStream_t stream) {
    if (w.V <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_ir_freeze, dim3((unsigned)((w.V + 255) / 256)), dim3(256), 0, stream, w, frozen_snap, frozen_epoch);
    return hipGetLastError();
}
hipError_t launch_keyless_ir(const DevWorld &w, uint8_t *gate, int n, const KeylessRec *recs, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_keyless_ir, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, stream, w, gate, n, recs);
    return hipGetLastError();
}
hipError_t launch_thaw_ir(const DevWorld &w, uint8_t *gate, hipStream_t stream) {
    if (w.R_local <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_thaw_ir, dim3((unsigned)w.R_local), dim3(128), 0, stream, w, gate);
    return hipGetLastError();
}
hipError_t launch_copy_bytes(uint8_t *dst, const uint8_t *src, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_copy_bytes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dst, src, n);
    return hipGetLastError();
}
hipError_t launch_halo_pack(const DevWorld &w, int n, const int32_t *robots, double *buf, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const int total = n * (SNAP_W + 1) * w.K;
    hipLaunchKernelGGL(k_halo_pack, dim3((total + 255) / 256), dim3(256), 0, stream, w, n, robots, buf);
    return hipGetLastError();
}
hipError_t launch_halo_unpack(const DevWorld &w, int n, const int32_t *ghosts, const double *buf, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const int total = n * (SNAP_W + 1) * w.K;
    hipLaunchKernelGGL(k_halo_unpack, dim3((total + 255) / 256), dim3(256), 0, stream, w, n, ghosts, buf);
    return hipGetLastError();
}

}  // namespace mgx