// mgx_kernels.hip — gfx950 kernels of the GBP engine.
//
// k_robot_sweep: ONE WORKGROUP PER ROBOT.  The robot's whole factor graph (variable->factor
// snapshots, current beliefs, factor->variable messages) is staged in LDS once per launch and
// stays there for every phase the launch runs: an optional external phase
// (external_factor_iteration + routing + external_variable_iteration,
// factorgraph.rs:719-760,794-826, robot.rs:1803-1859) followed by `n_int` internal iterations
// (internal_factor_iteration + internal_variable_iteration, factorgraph.rs:688-714,762-790).
// Lanes map to MESSAGES in the factor phases (one Schur complement per lane) and to VARIABLES
// in the variable phases; each phase is a Jacobi sweep (reads only the previous phase's data),
// so phases are separated by workgroup barriers only.  Robots couple only through the
// inter-robot edges, which read the OTHER robot's snapshot buffer `cur` from HBM while this
// launch writes buffer `1 - cur` — no inter-workgroup synchronisation inside a launch.
#include <hip/hip_runtime.h>

#include "gbp_math.h"
#include "mgx_dev.h"

namespace mgx {

__device__ __forceinline__ void ld_soa4(const double *base, int stride, int item, double (&o)[4]) {
#pragma unroll
    for (int c = 0; c < 4; c++) o[c] = base[(size_t)c * stride + item];
}
__device__ __forceinline__ void ld_soa16(const double *base, int stride, int item, double (&o)[16]) {
#pragma unroll
    for (int c = 0; c < 16; c++) o[c] = base[(size_t)c * stride + item];
}
__device__ __forceinline__ void st_soa4(double *base, int stride, int item, const double (&o)[4]) {
#pragma unroll
    for (int c = 0; c < 4; c++) base[(size_t)c * stride + item] = o[c];
}
__device__ __forceinline__ void st_soa16(double *base, int stride, int item, const double (&o)[16]) {
#pragma unroll
    for (int c = 0; c < 16; c++) base[(size_t)c * stride + item] = o[c];
}

// adds the inter-robot messages [e0, e1) of a variable's inbox to (eta, lam), in inbox order
__device__ __forceinline__ void ir_accumulate(const DevWorld &w, int e0, int e1, double (&eta)[4], double (&lam)[16]) {
    for (int e = e0; e < e1; e++) {
#pragma unroll
        for (int c = 0; c < 4; c++) eta[c] += w.ir_fv_eta[(size_t)c * w.NI + e];
#pragma unroll
        for (int c = 0; c < 16; c++) lam[c] += w.ir_fv_lam[(size_t)c * w.NI + e];
    }
}

extern __shared__ double lds[];

__global__ void __launch_bounds__(256) k_robot_sweep(DevWorld w, int robot0, uint32_t ext_mask, uint32_t int_mask,
                                                     int n_int, int snap_out) {
    const int r = robot0 + blockIdx.x;
    const int lane = threadIdx.x, T = blockDim.x;
    const int K = w.K, E = w.E;
    double *s_snap = lds;                       // [24][K]
    double *s_bel = s_snap + SNAP_W * K;        // [24][K]
    double *s_fv = s_bel + SNAP_W * K;          // [20][E]
    uint32_t *s_epoch = (uint32_t *)(s_fv + 20 * E);  // [K]

    const bool idle = w.idle[r] != 0;
    const bool radio = (w.antenna[r] != 0) && !idle;
    const int v0 = r * K;       // first variable of this robot
    const int eb = r * E;       // first internal edge
    const bool is_var = lane < K;

    // ---- per-variable register state ------------------------------------------------------
    double p_eta[4], p_lam[16], mu[4], cov[16];
    int ir_e0 = 0, ir_mid = 0, ir_e1 = 0;  // inbox entries of foreign factors with a lower / higher graph key
    int valid = 0;
    uint32_t epoch = 0;
    if (is_var) {
        const int v = v0 + lane;
        ld_soa4(w.prior_eta, w.V, v, p_eta);
        ld_soa16(w.prior_lam, w.V, v, p_lam);
        ld_soa4(w.bel_mu, w.V, v, mu);
        ld_soa16(w.bel_cov, w.V, v, cov);
        valid = w.bel_valid[v];
        epoch = w.snap_epoch[w.cur][v];
        s_epoch[lane] = epoch;
#pragma unroll
        for (int c = 0; c < 4; c++) s_bel[c * K + lane] = w.bel_eta[(size_t)c * w.V + v];
#pragma unroll
        for (int c = 0; c < 16; c++) s_bel[(4 + c) * K + lane] = w.bel_lam[(size_t)c * w.V + v];
#pragma unroll
        for (int c = 0; c < 4; c++) s_bel[(20 + c) * K + lane] = mu[c];
#pragma unroll
        for (int c = 0; c < SNAP_W; c++) s_snap[c * K + lane] = w.snap[w.cur][(size_t)c * w.V + v];
        ir_e0 = w.ir_var_ptr[v];
        ir_mid = w.ir_var_mid[v];
        ir_e1 = w.ir_var_ptr[v + 1];
    }
    for (int e = lane; e < E; e += T) {
#pragma unroll
        for (int c = 0; c < 4; c++) s_fv[c * E + e] = w.fv_eta[(size_t)c * w.EI + eb + e];
#pragma unroll
        for (int c = 0; c < 16; c++) s_fv[(4 + c) * E + e] = w.fv_lam[(size_t)c * w.EI + eb + e];
    }
    int itf = w.iter_factor[r];

    // ---- factor-lane roles ----------------------------------------------------------------
    const int n_dyn = 2 * (K - 1);
    const bool is_dyn = lane < n_dyn;
    const bool is_obs = lane >= n_dyn && lane < n_dyn + (K - 2);
    const bool is_trk = lane >= n_dyn + (K - 2) && lane < E;
    double maa[4], mab[4], mba[4], mbb[4];
    int dyn_other_var = 0, dyn_other_edge = 0;
    if (is_dyn) {
        const int f = lane % (K - 1), slot = lane / (K - 1);
        const int a2 = 2 * slot, b2 = 2 * (1 - slot);
        const double *M = w.dyn_m;
        const int it = r * (K - 1) + f;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                maa[i * 2 + j] = M[(size_t)((a2 + i) * 4 + (a2 + j)) * w.ND + it];
                mab[i * 2 + j] = M[(size_t)((a2 + i) * 4 + (b2 + j)) * w.ND + it];
                mba[i * 2 + j] = M[(size_t)((b2 + i) * 4 + (a2 + j)) * w.ND + it];
                mbb[i * 2 + j] = M[(size_t)((b2 + i) * 4 + (b2 + j)) * w.ND + it];
            }
        dyn_other_var = f + 1 - slot;
        dyn_other_edge = (1 - slot) * (K - 1) + f;
    }
    const int uvar = (is_obs ? lane - n_dyn : lane - n_dyn - (K - 2)) + 1;  // variable of a unary factor
    int trk_rec = 0;
    float trk_lp[2] = {0.f, 0.f};
    double trk_lv = 0.0;
    const int trk_item = r * (K - 2) + (uvar - 1);
    if (is_trk) {
        trk_rec = w.trk_record[trk_item];
        trk_lp[0] = w.trk_last_pos[trk_item];
        trk_lp[1] = w.trk_last_pos[(size_t)w.NT + trk_item];
        trk_lv = w.trk_last_val[trk_item];
    }
    __syncthreads();

    const int ie0 = w.ir_var_ptr[v0], ie1 = w.ir_var_ptr[v0 + K];
    const bool ir_on = (w.enable & 2u) != 0;

    // ======================= external factor sweep (pull form) ================================
    // Every inter-robot factor F_AB is evaluated by the workgroup of its only consumer B
    // (factorgraph.rs:745-754 keeps only the message to the other graph's variable).
    if (ext_mask & PH_EXT_FACTOR) {
        if (radio && ir_on) {
            for (int e = ie0 + lane; e < ie1; e += T) {
                const int A = w.ir_src_robot[e];
                if (!(w.antenna[A] != 0 && w.idle[A] == 0)) continue;  // A did not run its sweep
                const int s = w.ir_src_var[e];
                double ao_eta[4], ao_lam[16], a_mu[4], b_mu[4];
                const bool a_present = w.snap_epoch[w.cur][s] > w.ir_created[e];
                if (a_present) {
                    ld_soa4(w.snap[w.cur], w.V, s, ao_eta);
                    ld_soa16(w.snap[w.cur] + (size_t)4 * w.V, w.V, s, ao_lam);
                    ld_soa4(w.snap[w.cur] + (size_t)20 * w.V, w.V, s, a_mu);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; c++) { ao_eta[c] = 0.0; a_mu[c] = 0.0; }
#pragma unroll
                    for (int c = 0; c < 16; c++) ao_lam[c] = 0.0;
                }
                ld_soa4(w.ir_bmu, w.NI, e, b_mu);
                const int dslot = w.ir_dst_slot[e];
                double oe[4], ol[16];
                bool ok;
                if (dslot)
                    ok = interrobot_message(a_mu, b_mu, w.ir_dsafe[e], w.ir_off[e], w.inv_s2_ir, 1, ao_eta, ao_lam, oe, ol);
                else
                    ok = interrobot_message(b_mu, a_mu, w.ir_dsafe[e], w.ir_off[e], w.inv_s2_ir, 0, ao_eta, ao_lam, oe, ol);
                if (!ok) {
#pragma unroll
                    for (int c = 0; c < 4; c++) oe[c] = 0.0;
#pragma unroll
                    for (int c = 0; c < 16; c++) ol[c] = 0.0;
                }
                st_soa4(w.ir_fv_eta, w.NI, e, oe);
                st_soa16(w.ir_fv_lam, w.NI, e, ol);
            }
        }
        if (radio) itf += 1;  // iteration_count.factor of B's own external sweep (factorgraph.rs:757)
        __syncthreads();
    }

    // internal-edge slots of variable `lane`
    const int e_left = (lane >= 1) ? (K - 1) + (lane - 1) : -1;        // dynamic factor lane-1 -> slot 1
    const int e_right = (lane <= K - 2) ? lane : -1;                    // dynamic factor lane   -> slot 0
    const int e_obs = (lane >= 1 && lane <= K - 2) ? n_dyn + (lane - 1) : -1;
    const int e_trk = (lane >= 1 && lane <= K - 2) ? n_dyn + (K - 2) + (lane - 1) : -1;

    auto variable_sweep = [&](bool deliver_internal) {
        // VariableNode::update_belief_and_create_factor_responses, variable.rs:251-342
        double eta[4], lam[16];
#pragma unroll
        for (int c = 0; c < 4; c++) eta[c] = p_eta[c];
#pragma unroll
        for (int c = 0; c < 16; c++) lam[c] = p_lam[c];
        // inbox order of the reference (BTreeMap<FactorId, _>, id.rs:19-54): factors of graphs with a
        // lower key, own factors by node index (dynamic i-1, dynamic i, obstacle, tracking; own
        // inter-robot factors are forever empty), then factors of graphs with a higher key
        ir_accumulate(w, ir_e0, ir_mid, eta, lam);
        const int es[4] = {e_left, e_right, e_obs, e_trk};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (es[q] < 0) continue;
#pragma unroll
            for (int c = 0; c < 4; c++) eta[c] += s_fv[c * E + es[q]];
#pragma unroll
            for (int c = 0; c < 16; c++) lam[c] += s_fv[(4 + c) * E + es[q]];
        }
        ir_accumulate(w, ir_mid, ir_e1, eta, lam);
        belief_from_information(eta, lam, mu, cov, valid);
#pragma unroll
        for (int c = 0; c < 4; c++) s_bel[c * K + lane] = eta[c];
#pragma unroll
        for (int c = 0; c < 16; c++) s_bel[(4 + c) * K + lane] = lam[c];
#pragma unroll
        for (int c = 0; c < 4; c++) s_bel[(20 + c) * K + lane] = mu[c];
        if (deliver_internal) {  // responses to own-graph factors (factorgraph.rs:771-786)
#pragma unroll
            for (int c = 0; c < 4; c++) s_snap[c * K + lane] = eta[c];
#pragma unroll
            for (int c = 0; c < 16; c++) s_snap[(4 + c) * K + lane] = lam[c];
#pragma unroll
            for (int c = 0; c < 4; c++) s_snap[(20 + c) * K + lane] = mu[c];
            epoch += 1;
            s_epoch[lane] = epoch;
        }
    };

    // ======================= external variable sweep ==========================================
    if (ext_mask & PH_EXT_VARIABLE) {
        if (radio) {
            if (is_var) variable_sweep(false);
            __syncthreads();
            if (ir_on) {
                // responses to the foreign factors attached to our variables, routed to their
                // inbox (robot.rs:1842-1858): belief - message (variable.rs:308-318)
                for (int e = ie0 + lane; e < ie1; e += T) {
                    const int A = w.ir_src_robot[e];
                    if (!(w.antenna[A] != 0 && w.idle[A] == 0)) continue;  // A cannot receive
                    const int i = w.ir_dst_var[e] - v0;
                    // only the mean of this inbox entry is ever used (it sets the linearisation
                    // point; eta / lam of the target side never reach the kept message)
#pragma unroll
                    for (int c = 0; c < 4; c++) w.ir_bmu[(size_t)c * w.NI + e] = s_bel[(20 + c) * K + i] - 0.0;
                }
            }
        }
        __syncthreads();
    }

    // ======================= internal iterations ==============================================
    if (!idle) {
        const SdfView sdf{w.sdf, w.sdf_w, w.sdf_h, w.world_w, w.world_h};
        for (int it = 0; it < n_int; it++) {
            if (int_mask & PH_INT_FACTOR) {
                double oe[4], ol[16];
                bool store = false;
                if (is_dyn && (w.enable & 1u)) {
                    double me[4], ml[16];
                    const int o = dyn_other_var, oe_ix = dyn_other_edge;
                    if (s_epoch[o] > 0) {  // other variable has answered: belief - our last message
#pragma unroll
                        for (int c = 0; c < 4; c++) me[c] = s_snap[c * K + o] - s_fv[c * E + oe_ix];
#pragma unroll
                        for (int c = 0; c < 16; c++) ml[c] = s_snap[(4 + c) * K + o] - s_fv[(4 + c) * E + oe_ix];
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; c++) me[c] = 0.0;
#pragma unroll
                        for (int c = 0; c < 16; c++) ml[c] = 0.0;
                    }
                    store = true;
                    if (!dynamic_message(maa, mab, mba, mbb, me, ml, oe, ol)) {
#pragma unroll
                        for (int c = 0; c < 4; c++) oe[c] = 0.0;
#pragma unroll
                        for (int c = 0; c < 16; c++) ol[c] = 0.0;
                    }
                } else if (is_obs && (w.enable & 4u)) {
                    double x0[4];
                    const bool pres = s_epoch[uvar] > 0;
#pragma unroll
                    for (int c = 0; c < 4; c++) x0[c] = pres ? s_snap[(20 + c) * K + uvar] : 0.0;
                    long long idx[4];
                    obstacle_taps(sdf, x0[0], x0[1], w.obs_delta, idx);
                    double h[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) h[q] = (idx[q] >= 0) ? sdf_value(w.sdf[idx[q]]) : 0.0;
                    obstacle_message(h, w.obs_delta, w.inv_s2_obs, x0, oe, ol);
                    store = true;
                } else if (is_trk && (w.enable & 8u) && itf >= 10) {  // factorgraph.rs:701
                    double x0[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) x0[c] = s_snap[(20 + c) * K + uvar];
                    const int p0 = w.path_ptr[r], np = w.path_ptr[r + 1] - p0;
                    store = true;
                    if (!tracking_message(w.path_xy + 2 * (size_t)p0, np, w.trk_pad, w.trk_attr, w.inv_s2_trk, x0, trk_rec,
                                          trk_lp, trk_lv, oe, ol)) {
#pragma unroll
                        for (int c = 0; c < 4; c++) oe[c] = 0.0;
#pragma unroll
                        for (int c = 0; c < 16; c++) ol[c] = 0.0;
                    }
                }
                __syncthreads();  // every lane has read the old messages
                if (store) {
#pragma unroll
                    for (int c = 0; c < 4; c++) s_fv[c * E + lane] = oe[c];
#pragma unroll
                    for (int c = 0; c < 16; c++) s_fv[(4 + c) * E + lane] = ol[c];
                }
                itf += 1;
                __syncthreads();
            }
            if (int_mask & PH_INT_VARIABLE) {
                if (is_var) variable_sweep(true);
                __syncthreads();
            }
        }
    }

    // ---- write back --------------------------------------------------------------------------
    for (int e = lane; e < E; e += T) {
#pragma unroll
        for (int c = 0; c < 4; c++) w.fv_eta[(size_t)c * w.EI + eb + e] = s_fv[c * E + e];
#pragma unroll
        for (int c = 0; c < 16; c++) w.fv_lam[(size_t)c * w.EI + eb + e] = s_fv[(4 + c) * E + e];
    }
    if (is_var) {
        const int v = v0 + lane;
#pragma unroll
        for (int c = 0; c < 4; c++) w.bel_eta[(size_t)c * w.V + v] = s_bel[c * K + lane];
#pragma unroll
        for (int c = 0; c < 16; c++) w.bel_lam[(size_t)c * w.V + v] = s_bel[(4 + c) * K + lane];
        st_soa4(w.bel_mu, w.V, v, mu);
        st_soa16(w.bel_cov, w.V, v, cov);
        w.bel_valid[v] = valid;
        if (snap_out >= 0) {
#pragma unroll
            for (int c = 0; c < SNAP_W; c++) w.snap[snap_out][(size_t)c * w.V + v] = s_snap[c * K + lane];
            w.snap_epoch[snap_out][v] = epoch;
        }
    }
    if (is_trk) {
        w.trk_record[trk_item] = trk_rec;
        w.trk_last_pos[trk_item] = trk_lp[0];
        w.trk_last_pos[(size_t)w.NT + trk_item] = trk_lp[1];
        w.trk_last_val[trk_item] = trk_lv;
    }
    if (lane == 0) w.iter_factor[r] = itf;
}

// VariableNode::change_prior + routing (variable.rs:203-230, factorgraph.rs:494-528,
// robot.rs:2262-2282).  One thread per (robot, variable, mean) triple.
__global__ void k_change_prior(DevWorld w, int n, const int32_t *robots, const uint32_t *vars, const double *means) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int r = robots[t], i = (int)vars[t], K = w.K, E = w.E;
    const int v = r * K + i;
    double m[4], pl[16], be[4], bl[16];
#pragma unroll
    for (int c = 0; c < 4; c++) m[c] = means[4 * t + c];
    ld_soa16(w.prior_lam, w.V, v, pl);
#pragma unroll
    for (int a = 0; a < 4; a++)  // prior eta = prior lam . mean (:204)
        w.prior_eta[(size_t)a * w.V + v] = ((pl[a * 4 + 0] * m[0] + pl[a * 4 + 1] * m[1]) + pl[a * 4 + 2] * m[2]) + pl[a * 4 + 3] * m[3];
    st_soa4(w.bel_mu, w.V, v, m);  // :206
    ld_soa4(w.bel_eta, w.V, v, be);
    ld_soa16(w.bel_lam, w.V, v, bl);
    // the (stale eta, stale lam, new mean) belief goes to every connected factor (:210-221):
    //   own-graph factors read it from the snapshot ...
    double *sn = w.snap[w.cur];
    st_soa4(sn, w.V, v, be);
    st_soa16(sn + (size_t)4 * w.V, w.V, v, bl);
    st_soa4(sn + (size_t)20 * w.V, w.V, v, m);
    w.snap_epoch[w.cur][v] += 1;
    //   ... and foreign inter-robot factors attached to this variable get it in their inbox;
    // every inbox message of the variable becomes empty (:224-227)
    for (int e = w.ir_var_ptr[v]; e < w.ir_var_ptr[v + 1]; e++) {
        if (w.enable & 2u) st_soa4(w.ir_bmu, w.NI, e, m);
#pragma unroll
        for (int c = 0; c < 4; c++) w.ir_fv_eta[(size_t)c * w.NI + e] = 0.0;
#pragma unroll
        for (int c = 0; c < 16; c++) w.ir_fv_lam[(size_t)c * w.NI + e] = 0.0;
    }
    const int n_dyn = 2 * (K - 1);
    const int es[4] = {(i >= 1) ? (K - 1) + (i - 1) : -1, (i <= K - 2) ? i : -1,
                       (i >= 1 && i <= K - 2) ? n_dyn + (i - 1) : -1,
                       (i >= 1 && i <= K - 2) ? n_dyn + (K - 2) + (i - 1) : -1};
    for (int q = 0; q < 4; q++) {
        if (es[q] < 0) continue;
        const int e = r * E + es[q];
#pragma unroll
        for (int c = 0; c < 4; c++) w.fv_eta[(size_t)c * w.EI + e] = 0.0;
#pragma unroll
        for (int c = 0; c < 16; c++) w.fv_lam[(size_t)c * w.EI + e] = 0.0;
    }
}

// halo: the snapshot records (variables 0..K-1: eta, lam, mu, epoch) of whole robots
__global__ void k_halo_pack(DevWorld w, int n, const int32_t *robots, double *buf) {
    const int words = (SNAP_W + 1) * w.K;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * words) return;
    const int rr = t / words, q = t % words, i = q / (SNAP_W + 1), c = q % (SNAP_W + 1);
    const int v = robots[rr] * w.K + i;
    buf[t] = (c < SNAP_W) ? w.snap[w.cur][(size_t)c * w.V + v] : (double)w.snap_epoch[w.cur][v];
}
__global__ void k_halo_unpack(DevWorld w, int n, const int32_t *ghosts, const double *buf) {
    const int words = (SNAP_W + 1) * w.K;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * words) return;
    const int rr = t / words, q = t % words, i = q / (SNAP_W + 1), c = q % (SNAP_W + 1);
    const int v = ghosts[rr] * w.K + i;
    if (c < SNAP_W)
        w.snap[w.cur][(size_t)c * w.V + v] = buf[t];
    else
        w.snap_epoch[w.cur][v] = (uint32_t)buf[t];
}

// ---- launch wrappers (called from mgx_world.hip) -------------------------------------------------
size_t sweep_lds_bytes(int K) {
    const int E = 4 * K - 6;
    return sizeof(double) * (size_t)(2 * SNAP_W * K + 20 * E) + sizeof(uint32_t) * (size_t)K;
}
int sweep_block(int K) {
    const int E = 4 * K - 6;
    int need = E > K ? E : K;
    return ((need + 63) / 64) * 64;
}

hipError_t launch_robot_sweep(const DevWorld &w, int robot0, int n_robots, uint32_t ext_mask, uint32_t int_mask, int n_int,
                              int snap_out, hipStream_t stream) {
    if (n_robots <= 0) return hipSuccess;
    const int block = sweep_block(w.K);
    hipLaunchKernelGGL(k_robot_sweep, dim3(n_robots), dim3(block), sweep_lds_bytes(w.K), stream, w, robot0, ext_mask,
                       int_mask, n_int, snap_out);
    return hipGetLastError();
}
hipError_t launch_change_prior(const DevWorld &w, int n, const int32_t *robots, const uint32_t *vars, const double *means,
                               hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_change_prior, dim3((n + 63) / 64), dim3(64), 0, stream, w, n, robots, vars, means);
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
