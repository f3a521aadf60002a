// mgx_kernels.hip — gfx950 kernels of the GBP engine.
//
// k_robot_sweep: ONE 128-THREAD WORKGROUP PER ROBOT, TWO ROLE-SPECIALISED WAVES.
// The robot's whole factor graph (variable->factor snapshots, priors, factor->variable messages,
// the inter-robot messages attached to its variables) is staged in LDS once per launch and stays
// there for every phase the launch runs: an optional external phase
// (external_factor_iteration + routing + external_variable_iteration,
// factorgraph.rs:719-760,794-826, robot.rs:1803-1859) followed by `n_int` internal iterations
// (internal_factor_iteration + internal_variable_iteration, factorgraph.rs:688-714,762-790).
//
//   wave 0 (DYN)  factor phase: one lane per dynamic-factor MESSAGE (2(K-1) lanes), one 4x4
//                 Schur complement each
//   wave 1 (UV)   factor phase: one lane per obstacle / tracking factor;
//                 variable phase: one lane per variable (inbox sum, 4x4 inverse, belief)
//   both waves    external factor sweep: one lane per incoming inter-robot edge ("pull" form:
//                 every factor F_AB is evaluated by the workgroup of its only consumer B)
//
// At 1000 robots the chip holds every workgroup at once (4 per CU = 2 waves per SIMD from
// different robots), so the long dependent f64 chains of one wave are partly hidden behind another
// robot's wave.  All per-variable state lives in LDS, not registers, so each wave stays within the
// 256 VGPRs that two waves per SIMD allow.  Each phase is a Jacobi sweep separated by workgroup
// barriers only.  Robots couple only through the inter-robot edges, which gather the OTHER robot's
// 192-byte snapshot records from buffer `cur` in HBM while this launch writes buffer `1 - cur`:
// no inter-workgroup synchronisation inside a launch.
//
// Arithmetic: gbp_math.h, compiled with -ffp-contract=off so that results are bit-identical to
// the scalar f64 reference semantics.
#include <hip/hip_runtime.h>

#include "gbp_math.h"
#include "mgx_dev.h"

namespace mgx {

constexpr int SWEEP_BLOCK = 128;
enum { ROLE_DYN = 0, ROLE_UV = 1 };

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

extern __shared__ double lds[];

// STAGE_IR: the robot's incoming inter-robot messages are kept in LDS ([20][n_edges]); otherwise
// (a robot with too many edges for LDS) they are read from HBM / L2 in every variable sweep.
// KT: horizon length K as a compile-time constant (0 = read it from the world): with K fixed every
// LDS access is base + immediate offset, which keeps the address arithmetic out of the VGPR budget.
constexpr int IR_STRIDE = 21;  // one staged inter-robot message: 20 f64 + 1 pad (bank spread)

template <int KT, bool STAGE_IR>
__global__ void __launch_bounds__(SWEEP_BLOCK, 2) k_robot_sweep(DevWorld w, int robot0, uint32_t ext_mask, uint32_t int_mask,
                                                             int n_int, int snap_out) {
    const int r = robot0 + blockIdx.x;
    const int tid = threadIdx.x;
    const int role = tid >> 6, lane = tid & 63;
    const int K = KT > 0 ? KT : w.K, E = 4 * K - 6;
    double *s_snap = lds;                              // [24][K] variable -> own-factor snapshots
    double *s_prior = s_snap + SNAP_W * K;             // [20][K] prior eta, lam
    double *s_cov = s_prior + 20 * K;                  // [16][K] belief covariance
    double *s_mu = s_cov + 16 * K;                     // [4][K]  belief mean
    double *s_fv = s_mu + 4 * K;                       // [20][E] factor -> variable messages
    uint32_t *s_epoch = (uint32_t *)(s_fv + 20 * E);   // [K] deliveries
    int32_t *s_valid = (int32_t *)(s_epoch + K);       // [K]
    double *s_ir = (double *)(s_valid + K);            // [ne][IR_STRIDE] inter-robot messages (STAGE_IR)

    const bool idle = w.idle[r] != 0;
    const bool radio = (w.antenna[r] != 0) && !idle;
    const int v0 = r * K, eb = r * E;
    const int ie0 = w.ir_var_ptr[v0], ie1 = w.ir_var_ptr[v0 + K], ne = ie1 - ie0;
    const bool ir_on = (w.enable & 2u) != 0;
    const int n_dyn = 2 * (K - 1);

    // ---- stage the robot in LDS (all 256 threads, coalesced) ----------------------------------
    {
        const double *src = w.snap[w.cur] + (size_t)v0 * SNAP_W;
        for (int t = tid; t < SNAP_W * K; t += SWEEP_BLOCK) s_snap[(t % SNAP_W) * K + (t / SNAP_W)] = src[t];
        for (int t = tid; t < 20 * E; t += SWEEP_BLOCK) {
            const int c = t / E, e = t - c * E;
            s_fv[t] = (c < 4) ? w.fv_eta[(size_t)c * w.EI + eb + e] : w.fv_lam[(size_t)(c - 4) * w.EI + eb + e];
        }
        if (STAGE_IR)
            for (int t = tid; t < 20 * ne; t += SWEEP_BLOCK) {
                const int c = t / ne, j = t - c * ne;
                s_ir[j * IR_STRIDE + c] = (c < 4) ? w.ir_fv_eta[(size_t)c * w.NI + ie0 + j] : w.ir_fv_lam[(size_t)(c - 4) * w.NI + ie0 + j];
            }
        for (int t = tid; t < 20 * K; t += SWEEP_BLOCK) {
            const int c = t / K, i = t - c * K;
            s_prior[t] = (c < 4) ? w.prior_eta[(size_t)c * w.V + v0 + i] : w.prior_lam[(size_t)(c - 4) * w.V + v0 + i];
        }
        for (int t = tid; t < 16 * K; t += SWEEP_BLOCK) s_cov[t] = w.bel_cov[(size_t)(t / K) * w.V + v0 + (t % K)];
        for (int t = tid; t < 4 * K; t += SWEEP_BLOCK) s_mu[t] = w.bel_mu[(size_t)(t / K) * w.V + v0 + (t % K)];
        for (int t = tid; t < K; t += SWEEP_BLOCK) {
            s_epoch[t] = w.snap_epoch[w.cur][v0 + t];
            s_valid[t] = w.bel_valid[v0 + t];
        }
    }
    int itf = w.iter_factor[r];

    // ---- UV wave, variable phase: lane = variable ------------------------------------------------
    const bool is_var = role == ROLE_UV && lane < K;
    int ir_e0 = 0, ir_mid = 0, ir_e1 = 0;
    if (is_var) {
        ir_e0 = w.ir_var_ptr[v0 + lane];
        ir_mid = w.ir_var_mid[v0 + lane];
        ir_e1 = w.ir_var_ptr[v0 + lane + 1];
    }
    // which variable sweep of this launch is the last one (it writes the belief to HBM)
    const bool has_int_var = (int_mask & PH_INT_VARIABLE) && n_int > 0 && !idle;
    const bool any_var_sweep = has_int_var || ((ext_mask & PH_EXT_VARIABLE) && radio);

    // ---- DYN wave: constant potential blocks of this lane's message -----------------------------
    const bool is_dyn = role == ROLE_DYN && lane < n_dyn;
    double maa[4], mab[4], mba[4], mbb[4];
    int dyn_other_var = 0, dyn_other_edge = 0;
    if (is_dyn) {
        const int f = lane % (K - 1), slot = lane / (K - 1);
        const int a2 = 2 * slot, b2 = 2 * (1 - slot);
        const int it = r * (K - 1) + f;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                maa[i * 2 + j] = w.dyn_m[(size_t)((a2 + i) * 4 + (a2 + j)) * w.ND + it];
                mab[i * 2 + j] = w.dyn_m[(size_t)((a2 + i) * 4 + (b2 + j)) * w.ND + it];
                mba[i * 2 + j] = w.dyn_m[(size_t)((b2 + i) * 4 + (a2 + j)) * w.ND + it];
                mbb[i * 2 + j] = w.dyn_m[(size_t)((b2 + i) * 4 + (b2 + j)) * w.ND + it];
            }
        dyn_other_var = f + 1 - slot;
        dyn_other_edge = (1 - slot) * (K - 1) + f;
    }

    // ---- UV wave, factor phase: obstacle lanes [0, K-2), tracking lanes [K-2, 2(K-2)) -----------
    const bool is_obs = role == ROLE_UV && lane < K - 2;
    const bool is_trk = role == ROLE_UV && lane >= K - 2 && lane < 2 * (K - 2);
    const int uvar = (is_trk ? lane - (K - 2) : lane) + 1;          // variable of the unary factor
    const int uedge = n_dyn + lane;                                 // its internal-edge slot
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

    // ======================= external factor sweep (pull form) ================================
    // factorgraph.rs:745-754 keeps only the message to the other graph's variable, so F_AB is
    // evaluated here, at B, from A's snapshot record and B's last response mean.
    if (ext_mask & PH_EXT_FACTOR) {
        if (radio && ir_on) {
            for (int j = tid; j < ne; j += SWEEP_BLOCK) {
                const int e = ie0 + j;
                const int A = w.ir_src_robot[e];
                if (!(w.antenna[A] != 0 && w.idle[A] == 0)) continue;  // A did not run its sweep
                const int s = w.ir_src_var[e];
                double ao_eta[4], ao_lam[16], a_mu[4], b_mu[4];
                const bool a_present = w.snap_epoch[w.cur][s] > w.ir_created[e];
                if (a_present) {
                    const double *rec = w.snap[w.cur] + (size_t)s * SNAP_W;
#pragma unroll
                    for (int c = 0; c < 4; c++) ao_eta[c] = rec[c];
#pragma unroll
                    for (int c = 0; c < 16; c++) ao_lam[c] = rec[4 + c];
#pragma unroll
                    for (int c = 0; c < 4; c++) a_mu[c] = rec[20 + c];
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
                if (STAGE_IR) {
#pragma unroll
                    for (int c = 0; c < 4; c++) s_ir[j * IR_STRIDE + c] = oe[c];
#pragma unroll
                    for (int c = 0; c < 16; c++) s_ir[j * IR_STRIDE + 4 + c] = ol[c];
                }
            }
        }
        if (radio) itf += 1;  // iteration_count.factor of B's own external sweep (factorgraph.rs:757)
        __syncthreads();
    }

    // internal-edge slots of variable `lane`
    const int e_left = (lane >= 1) ? (K - 1) + (lane - 1) : -1;   // dynamic factor lane-1 -> slot 1
    const int e_right = (lane <= K - 2) ? lane : -1;               // dynamic factor lane   -> slot 0
    const int e_obs = (lane >= 1 && lane <= K - 2) ? n_dyn + (lane - 1) : -1;
    const int e_trk = (lane >= 1 && lane <= K - 2) ? n_dyn + (K - 2) + (lane - 1) : -1;

    auto ir_accumulate = [&](int e_from, int e_to, double (&eta)[4], double (&lam)[16]) {
        for (int e = e_from; e < e_to; e++) {
            if (STAGE_IR) {
                const double *m = s_ir + (e - ie0) * IR_STRIDE;
#pragma unroll
                for (int c = 0; c < 4; c++) eta[c] += m[c];
#pragma unroll
                for (int c = 0; c < 16; c++) lam[c] += m[4 + c];
            } else {
#pragma unroll
                for (int c = 0; c < 4; c++) eta[c] += w.ir_fv_eta[(size_t)c * w.NI + e];
#pragma unroll
                for (int c = 0; c < 16; c++) lam[c] += w.ir_fv_lam[(size_t)c * w.NI + e];
            }
        }
    };

    auto variable_sweep = [&](bool deliver_internal, bool last) {
        // VariableNode::update_belief_and_create_factor_responses, variable.rs:251-342
        double b_eta[4], b_lam[16], mu[4], cov[16];
#pragma unroll
        for (int c = 0; c < 4; c++) b_eta[c] = s_prior[c * K + lane];
#pragma unroll
        for (int c = 0; c < 16; c++) b_lam[c] = s_prior[(4 + c) * K + lane];
        // inbox order of the reference (BTreeMap<FactorId, _>, id.rs:19-54): factors of graphs with a
        // lower key, own factors by node index (dynamic i-1, dynamic i, obstacle, tracking; own
        // inter-robot factors are forever empty), then factors of graphs with a higher key
        ir_accumulate(ir_e0, ir_mid, b_eta, b_lam);
        const int es[4] = {e_left, e_right, e_obs, e_trk};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (es[q] < 0) continue;
#pragma unroll
            for (int c = 0; c < 4; c++) b_eta[c] += s_fv[c * E + es[q]];
#pragma unroll
            for (int c = 0; c < 16; c++) b_lam[c] += s_fv[(4 + c) * E + es[q]];
        }
        ir_accumulate(ir_mid, ir_e1, b_eta, b_lam);
#pragma unroll
        for (int c = 0; c < 4; c++) mu[c] = s_mu[c * K + lane];
        int valid = s_valid[lane];
        if (belief_update(b_eta, b_lam, mu, cov, valid)) {  // covariance (and maybe mean) changed
#pragma unroll
            for (int c = 0; c < 16; c++) s_cov[c * K + lane] = cov[c];
#pragma unroll
            for (int c = 0; c < 4; c++) s_mu[c * K + lane] = mu[c];
            s_valid[lane] = valid;
        }
        if (deliver_internal) {  // responses to own-graph factors (factorgraph.rs:771-786)
#pragma unroll
            for (int c = 0; c < 4; c++) s_snap[c * K + lane] = b_eta[c];
#pragma unroll
            for (int c = 0; c < 16; c++) s_snap[(4 + c) * K + lane] = b_lam[c];
#pragma unroll
            for (int c = 0; c < 4; c++) s_snap[(20 + c) * K + lane] = mu[c];
            s_epoch[lane] += 1;
        }
        if (last) {  // the prior is not needed again in this launch: its LDS column carries the
                     // belief (eta, lam) of the last sweep to the coalesced write-back
#pragma unroll
            for (int c = 0; c < 4; c++) s_prior[c * K + lane] = b_eta[c];
#pragma unroll
            for (int c = 0; c < 16; c++) s_prior[(4 + c) * K + lane] = b_lam[c];
        }
    };

    // ======================= external variable sweep ==========================================
    if (ext_mask & PH_EXT_VARIABLE) {
        if (radio) {
            if (is_var) variable_sweep(false, !has_int_var);
            __syncthreads();
            if (ir_on) {
                // responses to the foreign factors attached to our variables, routed to their inbox
                // (robot.rs:1842-1858): only the mean of that inbox entry is ever used (it sets the
                // linearisation point; eta / lam of the target side never reach the kept message)
                for (int j = tid; j < ne; j += SWEEP_BLOCK) {
                    const int e = ie0 + j;
                    const int A = w.ir_src_robot[e];
                    if (!(w.antenna[A] != 0 && w.idle[A] == 0)) continue;  // A cannot receive
                    const int i = w.ir_dst_var[e] - v0;
#pragma unroll
                    for (int c = 0; c < 4; c++) w.ir_bmu[(size_t)c * w.NI + e] = s_mu[c * K + i] - 0.0;
                }
            }
        }
        __syncthreads();
    }

    // ======================= internal iterations ==============================================
    if (!idle) {
        const SdfView sdf = make_sdf_view(w.sdf, w.sdf_w, w.sdf_h, w.world_w, w.world_h);
        for (int it = 0; it < n_int; it++) {
            if (int_mask & PH_INT_FACTOR) {
                // The two messages of one dynamic factor read each other's previous value; both lanes
                // sit in the SAME wave, whose LDS reads all issue before its LDS writes, so no barrier
                // is needed between reading the old and writing the new messages.
                if (is_dyn && (w.enable & 1u)) {
                    double me[4], ml[16], oe[4], ol[16];
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
                    if (!dynamic_message(maa, mab, mba, mbb, me, ml, oe, ol)) {
#pragma unroll
                        for (int c = 0; c < 4; c++) oe[c] = 0.0;
#pragma unroll
                        for (int c = 0; c < 16; c++) ol[c] = 0.0;
                    }
#pragma unroll
                    for (int c = 0; c < 4; c++) s_fv[c * E + lane] = oe[c];
#pragma unroll
                    for (int c = 0; c < 16; c++) s_fv[(4 + c) * E + lane] = ol[c];
                }
                if (is_obs && (w.enable & 4u)) {
                    double x0[4], oe[4], ol[16];
                    const bool pres = s_epoch[uvar] > 0;
#pragma unroll
                    for (int c = 0; c < 4; c++) x0[c] = pres ? s_snap[(20 + c) * K + uvar] : 0.0;
                    long long idx[4];
                    obstacle_taps(sdf, x0[0], x0[1], w.obs_delta, idx);
                    double h[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) h[q] = (idx[q] >= 0) ? sdf_value(w.sdf[idx[q]]) : 0.0;
                    obstacle_message(h, w.obs_delta, w.inv_s2_obs, x0, oe, ol);
#pragma unroll
                    for (int c = 0; c < 4; c++) s_fv[c * E + uedge] = oe[c];
#pragma unroll
                    for (int c = 0; c < 16; c++) s_fv[(4 + c) * E + uedge] = ol[c];
                }
                if (is_trk && (w.enable & 8u) && itf >= 10) {  // factorgraph.rs:701
                    double x0[4], oe[4], ol[16];
#pragma unroll
                    for (int c = 0; c < 4; c++) x0[c] = s_snap[(20 + c) * K + uvar];
                    const int p0 = w.path_ptr[r], np = w.path_ptr[r + 1] - p0;
                    if (!tracking_message(w.path_xy + 2 * (size_t)p0, np, w.trk_pad, w.trk_attr, w.inv_s2_trk, x0, trk_rec,
                                          trk_lp, trk_lv, oe, ol)) {
#pragma unroll
                        for (int c = 0; c < 4; c++) oe[c] = 0.0;
#pragma unroll
                        for (int c = 0; c < 16; c++) ol[c] = 0.0;
                    }
#pragma unroll
                    for (int c = 0; c < 4; c++) s_fv[c * E + uedge] = oe[c];
#pragma unroll
                    for (int c = 0; c < 16; c++) s_fv[(4 + c) * E + uedge] = ol[c];
                }
                itf += 1;
                __syncthreads();
            }
            if (int_mask & PH_INT_VARIABLE) {
                if (is_var) variable_sweep(true, it == n_int - 1);
                __syncthreads();
            }
        }
    }

    // ---- write back (coalesced) ------------------------------------------------------------------
    for (int t = tid; t < 20 * E; t += SWEEP_BLOCK) {
        const int c = t / E, e = t - c * E;
        if (c < 4)
            w.fv_eta[(size_t)c * w.EI + eb + e] = s_fv[t];
        else
            w.fv_lam[(size_t)(c - 4) * w.EI + eb + e] = s_fv[t];
    }
    if (snap_out >= 0) {
        double *dst = w.snap[snap_out] + (size_t)v0 * SNAP_W;
        for (int t = tid; t < SNAP_W * K; t += SWEEP_BLOCK) dst[t] = s_snap[(t % SNAP_W) * K + (t / SNAP_W)];
    }
    if (any_var_sweep) {  // beliefs: s_prior columns now hold (eta, lam) of the last sweep
        for (int t = tid; t < 20 * K; t += SWEEP_BLOCK) {
            const int c = t / K, i = t - c * K;
            if (c < 4)
                w.bel_eta[(size_t)c * w.V + v0 + i] = s_prior[t];
            else
                w.bel_lam[(size_t)(c - 4) * w.V + v0 + i] = s_prior[t];
        }
        for (int t = tid; t < 16 * K; t += SWEEP_BLOCK) w.bel_cov[(size_t)(t / K) * w.V + v0 + (t % K)] = s_cov[t];
        for (int t = tid; t < 4 * K; t += SWEEP_BLOCK) w.bel_mu[(size_t)(t / K) * w.V + v0 + (t % K)] = s_mu[t];
        for (int t = tid; t < K; t += SWEEP_BLOCK) w.bel_valid[v0 + t] = s_valid[t];
    }
    if (snap_out >= 0)
        for (int t = tid; t < K; t += SWEEP_BLOCK) w.snap_epoch[snap_out][v0 + t] = s_epoch[t];
    if (is_trk) {
        w.trk_record[trk_item] = trk_rec;
        w.trk_last_pos[trk_item] = trk_lp[0];
        w.trk_last_pos[(size_t)w.NT + trk_item] = trk_lp[1];
        w.trk_last_val[trk_item] = trk_lv;
    }
    if (tid == 0) w.iter_factor[r] = itf;
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
    //   own-graph factors read it from the snapshot record ...
    double *rec = w.snap[w.cur] + (size_t)v * SNAP_W;
#pragma unroll
    for (int c = 0; c < 4; c++) rec[c] = be[c];
#pragma unroll
    for (int c = 0; c < 16; c++) rec[4 + c] = bl[c];
#pragma unroll
    for (int c = 0; c < 4; c++) rec[20 + c] = m[c];
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

// ---- launch wrappers (called from mgx_world.hip) -------------------------------------------------
size_t sweep_lds_bytes(int K, int ir_edges) {
    const int E = 4 * K - 6;
    return sizeof(double) * (size_t)((SNAP_W + 20 + 16 + 4) * K + 20 * E + IR_STRIDE * ir_edges) + 8 * (size_t)K;
}
bool sweep_supports(int K) { return K >= 3 && 2 * (K - 1) <= 64; }

template <int KT>
static void launch_k(const DevWorld &w, int robot0, int n_robots, uint32_t ext_mask, uint32_t int_mask, int n_int, int snap_out,
                     hipStream_t stream) {
    // 4 workgroups per CU need <= 40 KB each; beyond 48 KB fall back to reading the messages from L2
    const size_t staged = sweep_lds_bytes(w.K, w.ir_max_edges);
    if (staged <= 48 * 1024)
        hipLaunchKernelGGL((k_robot_sweep<KT, true>), dim3(n_robots), dim3(SWEEP_BLOCK), staged, stream, w, robot0, ext_mask,
                           int_mask, n_int, snap_out);
    else
        hipLaunchKernelGGL((k_robot_sweep<KT, false>), dim3(n_robots), dim3(SWEEP_BLOCK), sweep_lds_bytes(w.K, 0), stream, w,
                           robot0, ext_mask, int_mask, n_int, snap_out);
}

hipError_t launch_robot_sweep(const DevWorld &w, int robot0, int n_robots, uint32_t ext_mask, uint32_t int_mask, int n_int,
                              int snap_out, hipStream_t stream) {
    if (n_robots <= 0) return hipSuccess;
    switch (w.K) {  // horizon lengths of BASELINE.json / the reference scenarios get constant-K code
    case 10: launch_k<10>(w, robot0, n_robots, ext_mask, int_mask, n_int, snap_out, stream); break;
    case 12: launch_k<12>(w, robot0, n_robots, ext_mask, int_mask, n_int, snap_out, stream); break;
    case 16: launch_k<16>(w, robot0, n_robots, ext_mask, int_mask, n_int, snap_out, stream); break;
    case 21: launch_k<21>(w, robot0, n_robots, ext_mask, int_mask, n_int, snap_out, stream); break;
    case 32: launch_k<32>(w, robot0, n_robots, ext_mask, int_mask, n_int, snap_out, stream); break;
    default: launch_k<0>(w, robot0, n_robots, ext_mask, int_mask, n_int, snap_out, stream); break;
    }
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
