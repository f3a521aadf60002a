// mgx_kernels.hip — gfx950 kernels of the GBP engine.
//
// k_robot_sweep: ONE 128-THREAD WORKGROUP PER GROUP OF RPB ROBOTS, TWO ROLE-SPECIALISED WAVES.
// The robots' whole factor graphs (variable->factor snapshots, priors, factor->variable messages,
// the inter-robot messages attached to their variables) are staged in LDS once per launch and stay
// there for every phase the launch runs: an optional external phase
// (external_factor_iteration + routing + external_variable_iteration,
// factorgraph.rs:719-760,794-826, robot.rs:1803-1859) followed by `n_int` internal iterations
// (internal_factor_iteration + internal_variable_iteration, factorgraph.rs:688-714,762-790).
//
//   wave 0 (DYN)  factor phase: one lane per dynamic-factor MESSAGE (RPB * 2(K-1) lanes), one 4x4
//                 Schur complement each
//   wave 1 (UV)   factor phase: one lane per obstacle / tracking factor;
//                 variable phase: one lane per variable (inbox sum, 4x4 inverse, belief)
//   both waves    external factor sweep: one lane per incoming inter-robot edge ("pull" form:
//                 every factor F_AB is evaluated by the workgroup of its only consumer B)
//
// RPB (robots per workgroup) packs as many robots as fit the 64 lanes of each role (2 at K = 16):
// the per-message instruction streams are long dependent f64 chains, so a wave costs the same
// whether 30 or 60 of its lanes are live.  All per-variable state lives in LDS, not registers, so
// each wave stays within 256 VGPRs.  Each phase is a Jacobi sweep separated by workgroup barriers
// only.  Robots couple only through the inter-robot edges, which gather the OTHER robot's 192-byte
// snapshot records from buffer `cur` in HBM while this launch writes buffer `1 - cur`: no
// inter-workgroup synchronisation inside a launch.
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

// In-kernel cycle stamps exist only in the diagnostic build (never in libmgx.so): they go to a
// buffer of their own and no output value depends on them.
#ifdef MGX_STAMPS
#define STAMP(var) unsigned long long var = __builtin_readcyclecounter()
#define STAMP_ADD(acc, a, b) acc += (b) - (a)
#else
#define STAMP(var)
#define STAMP_ADD(acc, a, b)
#endif

// STAGE_IR: the robot's incoming inter-robot messages are kept in LDS ([20][n_edges]); otherwise
// (a robot with too many edges for LDS) they are read from HBM / L2 in every variable sweep.
// KT: horizon length K as a compile-time constant (0 = read it from the world): with K fixed every
// LDS access is base + immediate offset, which keeps the address arithmetic out of the VGPR budget.
constexpr int IR_STRIDE = 21;  // one staged inter-robot message: 20 f64 + 1 pad (bank spread)

// robots per workgroup: every role must fit one 64-lane wave
#ifndef MGX_RPB_MAX
#define MGX_RPB_MAX 1
#endif
constexpr int rpb_for(int K) {
    int n = 64 / (2 * (K - 1));
    return n < 1 ? 1 : (n > MGX_RPB_MAX ? MGX_RPB_MAX : n);
}

template <int KT, int RPB, bool STAGE_IR>
__global__ void __launch_bounds__(SWEEP_BLOCK, 2) k_robot_sweep(DevWorld w, int robot0, int robot_end, uint32_t ext_mask,
                                                             uint32_t int_mask, int n_int, int snap_out) {
    STAMP(t_k0);
    const int r0 = robot0 + blockIdx.x * RPB;                       // first robot of this workgroup
    const int n_sub = (robot_end - r0) < RPB ? (robot_end - r0) : RPB;  // robots actually present
    const int tid = threadIdx.x;
    const int role = tid >> 6, lane = tid & 63;
    const int K = KT > 0 ? KT : w.K, E = 4 * K - 6;
    const int KK = RPB * K, EE = RPB * E + 1;                      // LDS column counts (+1: an all-zero
                                                                   // message column for absent edges)
    const int nK = n_sub * K, nE = n_sub * E;                      // live columns
    const int ZCOL = EE - 1;
    double *s_snap = lds;                              // [24][KK] variable -> own-factor snapshots
    double *s_prior = s_snap + SNAP_W * KK;            // [20][KK] prior eta, lam
    double *s_cov = s_prior + 20 * KK;                 // [16][KK] belief covariance
    double *s_mu = s_cov + 16 * KK;                    // [4][KK]  belief mean
    double *s_fv = s_mu + 4 * KK;                      // [20][EE] factor -> variable messages
    uint32_t *s_epoch = (uint32_t *)(s_fv + 20 * EE);  // [KK] deliveries
    int32_t *s_valid = (int32_t *)(s_epoch + KK);      // [KK]
    double *s_ir = (double *)(s_valid + KK);           // [ne][IR_STRIDE] inter-robot messages (STAGE_IR)

    const int v0 = r0 * K, eb = r0 * E;
    const int ie0 = w.ir_var_ptr[v0], ie1 = w.ir_var_ptr[v0 + nK], ne = ie1 - ie0;
    const bool ir_on = (w.enable & 2u) != 0;
    const int n_dyn = 2 * (K - 1), n_una = 2 * (K - 2);

    // ---- stage the robots in LDS (all 128 threads, coalesced; consecutive robots are contiguous) --
    {
        const double *src = w.snap[w.cur] + (size_t)v0 * SNAP_W;
        for (int t = tid; t < SNAP_W * nK; t += SWEEP_BLOCK) s_snap[(t % SNAP_W) * KK + (t / SNAP_W)] = src[t];
        for (int t = tid; t < 20 * nE; t += SWEEP_BLOCK) {
            const int c = t / nE, e = t - c * nE;
            s_fv[c * EE + e] = (c < 4) ? w.fv_eta[(size_t)c * w.EI + eb + e] : w.fv_lam[(size_t)(c - 4) * w.EI + eb + e];
        }
        if (tid < 20) s_fv[tid * EE + ZCOL] = 0.0;
        if (STAGE_IR)
            for (int t = tid; t < 20 * ne; t += SWEEP_BLOCK) {
                const int c = t / ne, j = t - c * ne;
                s_ir[j * IR_STRIDE + c] = (c < 4) ? w.ir_fv_eta[(size_t)c * w.NI + ie0 + j] : w.ir_fv_lam[(size_t)(c - 4) * w.NI + ie0 + j];
            }
        for (int t = tid; t < 20 * nK; t += SWEEP_BLOCK) {
            const int c = t / nK, i = t - c * nK;
            s_prior[c * KK + i] = (c < 4) ? w.prior_eta[(size_t)c * w.V + v0 + i] : w.prior_lam[(size_t)(c - 4) * w.V + v0 + i];
        }
        for (int t = tid; t < 16 * nK; t += SWEEP_BLOCK) s_cov[(t / nK) * KK + (t % nK)] = w.bel_cov[(size_t)(t / nK) * w.V + v0 + (t % nK)];
        for (int t = tid; t < 4 * nK; t += SWEEP_BLOCK) s_mu[(t / nK) * KK + (t % nK)] = w.bel_mu[(size_t)(t / nK) * w.V + v0 + (t % nK)];
        for (int t = tid; t < nK; t += SWEEP_BLOCK) {
            s_epoch[t] = w.snap_epoch[w.cur][v0 + t];
            s_valid[t] = w.bel_valid[v0 + t];
        }
    }

    // ---- this lane's robot in each of its roles ---------------------------------------------------
    // DYN wave: lane -> (robot sd, message l); UV wave: unary lane -> (su, factor l), variable lane -> (sv, i)
    const int sd = lane / n_dyn, ld = lane - sd * n_dyn;
    const int su = lane / n_una, lu = lane - su * n_una;
    const int sv = lane / K, iv = lane - sv * K;
    const bool is_dyn = role == ROLE_DYN && lane < n_sub * n_dyn;
    const bool is_una = role == ROLE_UV && lane < n_sub * n_una;
    const bool is_obs = is_una && lu < K - 2;
    const bool is_trk = is_una && lu >= K - 2;
    const bool is_var = role == ROLE_UV && lane < nK;
    // the robot whose flags / counters this lane follows (edge lanes look theirs up per edge)
    const int my_sub = role == ROLE_DYN ? (is_dyn ? sd : 0) : (is_una ? su : (is_var ? sv : 0));
    const bool idle_u = w.idle[r0 + (is_una ? su : 0)] != 0;      // unary-factor lane's robot
    const bool idle_v = w.idle[r0 + (is_var ? sv : 0)] != 0;      // variable lane's robot
    const bool idle_d = w.idle[r0 + (is_dyn ? sd : 0)] != 0;      // dynamic-message lane's robot
    const bool radio_v = (w.antenna[r0 + (is_var ? sv : 0)] != 0) && !idle_v;
    // iteration_count.factor of the robot this lane follows (every lane applies the same increments
    // as its robot: one per internal factor sweep if not idle, one per external factor sweep if on air)
    int itf = w.iter_factor[r0 + my_sub];
    const bool my_idle = w.idle[r0 + my_sub] != 0;
    const bool my_radio = (w.antenna[r0 + my_sub] != 0) && !my_idle;

    int ir_e0 = 0, ir_mid = 0, ir_e1 = 0;
    if (is_var) {
        ir_e0 = w.ir_var_ptr[v0 + lane];
        ir_mid = w.ir_var_mid[v0 + lane];
        ir_e1 = w.ir_var_ptr[v0 + lane + 1];
    }
    // which variable sweep of this launch is the last one for this lane's robot (its belief goes out)
    const bool has_int_var = (int_mask & PH_INT_VARIABLE) && n_int > 0 && !idle_v;

    // ---- DYN wave: constant potential blocks of this lane's message -------------------------------
    double maa[4], mab[4], mba[4], mbb[4];
    int dyn_other_var = 0, dyn_other_edge = 0, dyn_edge = 0;
    if (is_dyn) {
        const int f = ld % (K - 1), slot = ld / (K - 1);
        const int a2 = 2 * slot, b2 = 2 * (1 - slot);
        const int it = (r0 + sd) * (K - 1) + f;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                maa[i * 2 + j] = w.dyn_m[(size_t)((a2 + i) * 4 + (a2 + j)) * w.ND + it];
                mab[i * 2 + j] = w.dyn_m[(size_t)((a2 + i) * 4 + (b2 + j)) * w.ND + it];
                mba[i * 2 + j] = w.dyn_m[(size_t)((b2 + i) * 4 + (a2 + j)) * w.ND + it];
                mbb[i * 2 + j] = w.dyn_m[(size_t)((b2 + i) * 4 + (b2 + j)) * w.ND + it];
            }
        dyn_other_var = sd * K + f + 1 - slot;
        dyn_other_edge = sd * E + (1 - slot) * (K - 1) + f;
        dyn_edge = sd * E + ld;
    }

    // ---- UV wave, factor phase: per robot obstacle lanes [0, K-2), tracking lanes [K-2, 2(K-2)) ----
    const int uvar = su * K + (is_trk ? lu - (K - 2) : lu) + 1;     // LDS column of the factor's variable
    const int uedge = su * E + n_dyn + lu;                          // its internal-edge column
    int trk_rec = 0;
    float trk_lp[2] = {0.f, 0.f};
    double trk_lv = 0.0;
    const int trk_item = (r0 + su) * (K - 2) + (lu - (K - 2));
    if (is_trk) {
        trk_rec = w.trk_record[trk_item];
        trk_lp[0] = w.trk_last_pos[trk_item];
        trk_lp[1] = w.trk_last_pos[(size_t)w.NT + trk_item];
        trk_lv = w.trk_last_val[trk_item];
    }
    __syncthreads();
    STAMP(t_staged);

    // ======================= external factor sweep (pull form) ================================
    // factorgraph.rs:745-754 keeps only the message to the other graph's variable, so F_AB is
    // evaluated here, at B, from A's snapshot record and B's last response mean.
    if (ext_mask & PH_EXT_FACTOR) {
        if (ir_on) {
            for (int j = tid; j < ne; j += SWEEP_BLOCK) {
                const int e = ie0 + j;
                const int B = w.ir_dst_var[e] / K;  // target robot (device index)
                if (!(w.antenna[B] != 0 && w.idle[B] == 0)) continue;  // B cannot receive
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
        if (my_radio) itf += 1;  // iteration_count.factor of the robot's own external sweep (factorgraph.rs:757)
        __syncthreads();
    }

    // internal-edge columns of this lane's variable
    const int eoff = sv * E;
    // (absent edges read the all-zero column: x + 0.0 == x exactly, and a running sum that starts from
    // the prior is never -0.0, so this equals skipping the entry as the reference's inbox does)
    const int e_left = (iv >= 1) ? eoff + (K - 1) + (iv - 1) : ZCOL;   // dynamic factor i-1 -> slot 1
    const int e_right = (iv <= K - 2) ? eoff + iv : ZCOL;               // dynamic factor i   -> slot 0
    const int e_obs = (iv >= 1 && iv <= K - 2) ? eoff + n_dyn + (iv - 1) : ZCOL;
    const int e_trk = (iv >= 1 && iv <= K - 2) ? eoff + n_dyn + (K - 2) + (iv - 1) : ZCOL;

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

    // VariableNode::update_belief_and_create_factor_responses (variable.rs:251-342) in two halves:
    //  variable_sum:    eta / lam = prior + inbox (:254-271) and, for an internal sweep, the (eta, lam)
    //                   part of the responses to own-graph factors (:301-330, factorgraph.rs:771-786);
    //  variable_finish: covariance, validity and mean from (eta, lam) (:273-297) and the mean part of
    //                   the responses.
    // Dynamic factors never read a mean, so in the internal loop the finish of sweep t runs in the UV
    // wave NEXT TO the dynamic messages of sweep t+1 in the DYN wave instead of in front of them.
    auto variable_sum = [&](bool deliver_internal, bool last, double (&b_eta)[4], double (&b_lam)[16]) {
#pragma unroll
        for (int c = 0; c < 4; c++) b_eta[c] = s_prior[c * KK + lane];
#pragma unroll
        for (int c = 0; c < 16; c++) b_lam[c] = s_prior[(4 + c) * KK + lane];
        // inbox order of the reference (BTreeMap<FactorId, _>, id.rs:19-54): factors of graphs with a
        // lower key, own factors by node index (dynamic i-1, dynamic i, obstacle, tracking; own
        // inter-robot factors are forever empty), then factors of graphs with a higher key
        ir_accumulate(ir_e0, ir_mid, b_eta, b_lam);
        const int es[4] = {e_left, e_right, e_obs, e_trk};
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int c = 0; c < 4; c++) b_eta[c] += s_fv[c * EE + es[q]];
#pragma unroll
            for (int c = 0; c < 16; c++) b_lam[c] += s_fv[(4 + c) * EE + es[q]];
        }
        ir_accumulate(ir_mid, ir_e1, b_eta, b_lam);
        if (deliver_internal) {
#pragma unroll
            for (int c = 0; c < 4; c++) s_snap[c * KK + lane] = b_eta[c];
#pragma unroll
            for (int c = 0; c < 16; c++) s_snap[(4 + c) * KK + lane] = b_lam[c];
            s_epoch[lane] += 1;
        }
        if (last) {  // the prior is not needed again in this launch: its LDS column carries the
                     // belief (eta, lam) of the last sweep to the coalesced write-back
#pragma unroll
            for (int c = 0; c < 4; c++) s_prior[c * KK + lane] = b_eta[c];
#pragma unroll
            for (int c = 0; c < 16; c++) s_prior[(4 + c) * KK + lane] = b_lam[c];
        }
    };
    // Internal-sweep sums, one lane per (variable, row): lane (col, r) accumulates eta[r] and lam[r][0..3]
    // in inbox order — each element sees exactly the additions of the per-variable form.  All 128
    // threads take part (the DYN wave has nothing else to do in the variable phase).
    auto variable_sum_rows = [&](bool last) {
        for (int t = tid; t < 4 * nK; t += SWEEP_BLOCK) {
            const int col = t >> 2, rr = t & 3;
            const int sub = col / K, i = col - sub * K;
            if (w.idle[r0 + sub] != 0) continue;
            const int eo = sub * E;
            const int es[4] = {(i >= 1) ? eo + (K - 1) + (i - 1) : ZCOL, (i <= K - 2) ? eo + i : ZCOL,
                               (i >= 1 && i <= K - 2) ? eo + n_dyn + (i - 1) : ZCOL,
                               (i >= 1 && i <= K - 2) ? eo + n_dyn + (K - 2) + (i - 1) : ZCOL};
            double acc[5];  // eta[rr], lam[rr][0..3]
            acc[0] = s_prior[rr * KK + col];
#pragma unroll
            for (int c = 0; c < 4; c++) acc[1 + c] = s_prior[(4 + rr * 4 + c) * KK + col];
            const int x0 = w.ir_var_ptr[v0 + col], xm = w.ir_var_mid[v0 + col], x1 = w.ir_var_ptr[v0 + col + 1];
            auto ir_rows = [&](int e_from, int e_to) {
                for (int e = e_from; e < e_to; e++) {
                    if (STAGE_IR) {
                        const double *m = s_ir + (e - ie0) * IR_STRIDE;
                        acc[0] += m[rr];
#pragma unroll
                        for (int c = 0; c < 4; c++) acc[1 + c] += m[4 + rr * 4 + c];
                    } else {
                        acc[0] += w.ir_fv_eta[(size_t)rr * w.NI + e];
#pragma unroll
                        for (int c = 0; c < 4; c++) acc[1 + c] += w.ir_fv_lam[(size_t)(rr * 4 + c) * w.NI + e];
                    }
                }
            };
            ir_rows(x0, xm);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                acc[0] += s_fv[rr * EE + es[q]];
#pragma unroll
                for (int c = 0; c < 4; c++) acc[1 + c] += s_fv[(4 + rr * 4 + c) * EE + es[q]];
            }
            ir_rows(xm, x1);
            s_snap[rr * KK + col] = acc[0];
#pragma unroll
            for (int c = 0; c < 4; c++) s_snap[(4 + rr * 4 + c) * KK + col] = acc[1 + c];
            if (rr == 0) s_epoch[col] += 1;
            if (last) {
                s_prior[rr * KK + col] = acc[0];
#pragma unroll
                for (int c = 0; c < 4; c++) s_prior[(4 + rr * 4 + c) * KK + col] = acc[1 + c];
            }
        }
    };
    auto variable_finish = [&](bool deliver_internal, const double (&b_eta)[4], const double (&b_lam)[16]) {
        double mu[4], cov[16];
#pragma unroll
        for (int c = 0; c < 4; c++) mu[c] = s_mu[c * KK + lane];
        int valid = s_valid[lane];
        if (belief_update(b_eta, b_lam, mu, cov, valid)) {  // covariance (and maybe mean) changed
#pragma unroll
            for (int c = 0; c < 16; c++) s_cov[c * KK + lane] = cov[c];
#pragma unroll
            for (int c = 0; c < 4; c++) s_mu[c * KK + lane] = mu[c];
            s_valid[lane] = valid;
        }
        if (deliver_internal) {
#pragma unroll
            for (int c = 0; c < 4; c++) s_snap[(20 + c) * KK + lane] = mu[c];
        }
    };
    // finish of an internal sweep whose sums are in the snapshot columns (they ARE the responses)
    auto variable_finish_from_snapshot = [&]() {
        double b_eta[4], b_lam[16];
#pragma unroll
        for (int c = 0; c < 4; c++) b_eta[c] = s_snap[c * KK + lane];
#pragma unroll
        for (int c = 0; c < 16; c++) b_lam[c] = s_snap[(4 + c) * KK + lane];
        variable_finish(true, b_eta, b_lam);
    };

    // ======================= external variable sweep ==========================================
    if (ext_mask & PH_EXT_VARIABLE) {
        if (is_var && radio_v) {
            double b_eta[4], b_lam[16];
            variable_sum(false, !has_int_var, b_eta, b_lam);
            variable_finish(false, b_eta, b_lam);
        }
        __syncthreads();
        if (ir_on) {
            // responses to the foreign factors attached to our variables, routed to their inbox
            // (robot.rs:1842-1858): only the mean of that inbox entry is ever used (it sets the
            // linearisation point; eta / lam of the target side never reach the kept message)
            for (int j = tid; j < ne; j += SWEEP_BLOCK) {
                const int e = ie0 + j;
                const int col = w.ir_dst_var[e] - v0;
                const int B = r0 + col / K;
                if (!(w.antenna[B] != 0 && w.idle[B] == 0)) continue;  // B did not run its sweep
                const int A = w.ir_src_robot[e];
                if (!(w.antenna[A] != 0 && w.idle[A] == 0)) continue;  // A cannot receive
#pragma unroll
                for (int c = 0; c < 4; c++) w.ir_bmu[(size_t)c * w.NI + e] = s_mu[c * KK + col] - 0.0;
            }
        }
        __syncthreads();
    }

    // ======================= internal iterations ==============================================
    {
        const SdfView sdf = make_sdf_view(w.sdf, w.sdf_w, w.sdf_h, w.world_w, w.world_h);
#ifdef MGX_STAMPS
        unsigned long long c_f = 0, c_fb = 0, c_v = 0, c_vb = 0;
#endif
        STAMP(t_loop0);
#ifdef MGX_STAMPS
        const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
        bool pending = false;  // this variable lane's last internal sums still await their finish
        for (int it = 0; it < n_int; it++) {
            STAMP(t0);
            if (int_mask & PH_INT_FACTOR) {
                // The two messages of one dynamic factor read each other's previous value; both lanes
                // sit in the SAME wave, whose LDS reads all issue before its LDS writes, so no barrier
                // is needed between reading the old and writing the new messages.
                if (is_dyn && !idle_d && (w.enable & 1u)) {
                    double me[4], ml[16], oe[4], ol[16];
                    const int o = dyn_other_var, oe_ix = dyn_other_edge;
                    if (s_epoch[o] > 0) {  // other variable has answered: belief - our last message
#pragma unroll
                        for (int c = 0; c < 4; c++) me[c] = s_snap[c * KK + o] - s_fv[c * EE + oe_ix];
#pragma unroll
                        for (int c = 0; c < 16; c++) ml[c] = s_snap[(4 + c) * KK + o] - s_fv[(4 + c) * EE + oe_ix];
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
                    for (int c = 0; c < 4; c++) s_fv[c * EE + dyn_edge] = oe[c];
#pragma unroll
                    for (int c = 0; c < 16; c++) s_fv[(4 + c) * EE + dyn_edge] = ol[c];
                }
                // UV wave: first the belief of the previous sweep (mean, covariance) that the unary
                // factors linearise at — same wave, so its LDS writes precede their LDS reads
                if (pending) {
                    variable_finish_from_snapshot();
                    pending = false;
                }
                if (is_obs && !idle_u && (w.enable & 4u)) {
                    double x0[4], oe[4], ol[16];
                    const bool pres = s_epoch[uvar] > 0;
#pragma unroll
                    for (int c = 0; c < 4; c++) x0[c] = pres ? s_snap[(20 + c) * KK + uvar] : 0.0;
                    long long idx[4];
                    obstacle_taps(sdf, x0[0], x0[1], w.obs_delta, idx);
                    double h[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) h[q] = (idx[q] >= 0) ? sdf_value(w.sdf[idx[q]]) : 0.0;
                    obstacle_message(h, w.obs_delta, w.inv_s2_obs, x0, oe, ol);
#pragma unroll
                    for (int c = 0; c < 4; c++) s_fv[c * EE + uedge] = oe[c];
#pragma unroll
                    for (int c = 0; c < 16; c++) s_fv[(4 + c) * EE + uedge] = ol[c];
                }
                if (is_trk && !idle_u && (w.enable & 8u) && itf >= 10) {  // factorgraph.rs:701
                    double x0[4], oe[4], ol[16];
#pragma unroll
                    for (int c = 0; c < 4; c++) x0[c] = s_snap[(20 + c) * KK + uvar];
                    const int p0 = w.path_ptr[r0 + su], np = w.path_ptr[r0 + su + 1] - p0;
                    if (!tracking_message(w.path_xy + 2 * (size_t)p0, np, w.trk_pad, w.trk_attr, w.inv_s2_trk, x0, trk_rec,
                                          trk_lp, trk_lv, oe, ol)) {
#pragma unroll
                        for (int c = 0; c < 4; c++) oe[c] = 0.0;
#pragma unroll
                        for (int c = 0; c < 16; c++) ol[c] = 0.0;
                    }
#pragma unroll
                    for (int c = 0; c < 4; c++) s_fv[c * EE + uedge] = oe[c];
#pragma unroll
                    for (int c = 0; c < 16; c++) s_fv[(4 + c) * EE + uedge] = ol[c];
                }
                if (!my_idle) itf += 1;
                STAMP(t1);
                __syncthreads();
                STAMP(t2);
                STAMP_ADD(c_f, t0, t1);
                STAMP_ADD(c_fb, t1, t2);
            }
            STAMP(t3);
            if (int_mask & PH_INT_VARIABLE) {
                variable_sum_rows(it == n_int - 1);
                if (is_var && !idle_v) pending = true;
                STAMP(t4);
                __syncthreads();
                STAMP(t5);
                STAMP_ADD(c_v, t3, t4);
                STAMP_ADD(c_vb, t4, t5);
            }
        }
        if (pending) variable_finish_from_snapshot();
        __syncthreads();
#ifdef MGX_STAMPS
        if (w.dbg && lane == 0) {  // per wave: cycles in factor phase, its barrier, variable phase, its barrier
            unsigned long long *d = w.dbg + ((size_t)blockIdx.x * 2 + role) * 8;
            d[0] = c_f; d[1] = c_fb; d[2] = c_v; d[3] = c_vb; d[4] = __builtin_readcyclecounter() - t_loop0;
            d[5] = __builtin_amdgcn_s_memrealtime() - rt0;  // 100 MHz ticks over the same span
            d[6] = t_staged - t_k0;
            d[7] = t_loop0 - t_staged;
        }
#endif
    }

    // ---- write back (coalesced) ------------------------------------------------------------------
    for (int t = tid; t < 20 * nE; t += SWEEP_BLOCK) {
        const int c = t / nE, e = t - c * nE;
        if (c < 4)
            w.fv_eta[(size_t)c * w.EI + eb + e] = s_fv[c * EE + e];
        else
            w.fv_lam[(size_t)(c - 4) * w.EI + eb + e] = s_fv[c * EE + e];
    }
    if (snap_out >= 0) {
        double *dst = w.snap[snap_out] + (size_t)v0 * SNAP_W;
        for (int t = tid; t < SNAP_W * nK; t += SWEEP_BLOCK) dst[t] = s_snap[(t % SNAP_W) * KK + (t / SNAP_W)];
        for (int t = tid; t < nK; t += SWEEP_BLOCK) w.snap_epoch[snap_out][v0 + t] = s_epoch[t];
    }
    // beliefs: the s_prior columns of robots that ran a variable sweep now hold (eta, lam) of their last sweep
    const bool any_int_var = (int_mask & PH_INT_VARIABLE) && n_int > 0;
    const bool any_ext_var = (ext_mask & PH_EXT_VARIABLE) != 0;
    auto swept = [&](int col) {
        const int rr = r0 + col / K;
        const bool idl = w.idle[rr] != 0;
        return (any_int_var && !idl) || (any_ext_var && w.antenna[rr] != 0 && !idl);
    };
    for (int t = tid; t < 20 * nK; t += SWEEP_BLOCK) {
        const int c = t / nK, i = t - c * nK;
        if (!swept(i)) continue;
        if (c < 4)
            w.bel_eta[(size_t)c * w.V + v0 + i] = s_prior[c * KK + i];
        else
            w.bel_lam[(size_t)(c - 4) * w.V + v0 + i] = s_prior[c * KK + i];
    }
    for (int t = tid; t < 16 * nK; t += SWEEP_BLOCK)
        if (swept(t % nK)) w.bel_cov[(size_t)(t / nK) * w.V + v0 + (t % nK)] = s_cov[(t / nK) * KK + (t % nK)];
    for (int t = tid; t < 4 * nK; t += SWEEP_BLOCK)
        if (swept(t % nK)) w.bel_mu[(size_t)(t / nK) * w.V + v0 + (t % nK)] = s_mu[(t / nK) * KK + (t % nK)];
    for (int t = tid; t < nK; t += SWEEP_BLOCK)
        if (swept(t)) w.bel_valid[v0 + t] = s_valid[t];
    if (is_trk) {
        w.trk_record[trk_item] = trk_rec;
        w.trk_last_pos[trk_item] = trk_lp[0];
        w.trk_last_pos[(size_t)w.NT + trk_item] = trk_lp[1];
        w.trk_last_val[trk_item] = trk_lv;
    }
    if (is_dyn && ld == 0) w.iter_factor[r0 + sd] = itf;  // one lane per robot carries its counter out
#ifdef MGX_STAMPS
    if (w.dbg && lane == 0) w.dbg[((size_t)blockIdx.x * 2 + role) * 8 + 7] = __builtin_readcyclecounter() - t_k0;  // whole kernel
#endif
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
size_t sweep_lds_bytes(int K, int rpb, int ir_edges) {
    const int E = 4 * K - 6;
    return sizeof(double) * (size_t)(rpb * ((SNAP_W + 20 + 16 + 4) * K + 20 * E) + 20 + IR_STRIDE * ir_edges) + 8 * (size_t)(rpb * K);
}
bool sweep_supports(int K) { return K >= 3 && 2 * (K - 1) <= 64; }
int sweep_rpb(int K) { return rpb_for(K); }

template <int KT, int RPB>
static void launch_k(const DevWorld &w, int robot0, int n_robots, uint32_t ext_mask, uint32_t int_mask, int n_int, int snap_out,
                     hipStream_t stream) {
    const int blocks = (n_robots + RPB - 1) / RPB;
    // LDS: staging the inter-robot messages needs IR_STRIDE f64 per edge of the workgroup's robots;
    // beyond 64 KB fall back to reading them from L2 in every variable sweep
    const size_t staged = sweep_lds_bytes(w.K, RPB, RPB * w.ir_max_edges);
    if (staged <= 64 * 1024)
        hipLaunchKernelGGL((k_robot_sweep<KT, RPB, true>), dim3(blocks), dim3(SWEEP_BLOCK), staged, stream, w, robot0,
                           robot0 + n_robots, ext_mask, int_mask, n_int, snap_out);
    else
        hipLaunchKernelGGL((k_robot_sweep<KT, RPB, false>), dim3(blocks), dim3(SWEEP_BLOCK), sweep_lds_bytes(w.K, RPB, 0), stream, w,
                           robot0, robot0 + n_robots, ext_mask, int_mask, n_int, snap_out);
}

hipError_t launch_robot_sweep(const DevWorld &w, int robot0, int n_robots, uint32_t ext_mask, uint32_t int_mask, int n_int,
                              int snap_out, hipStream_t stream) {
    if (n_robots <= 0) return hipSuccess;
    switch (w.K) {  // horizon lengths of BASELINE.json / the reference scenarios get constant-K code
    case 10: launch_k<10, rpb_for(10)>(w, robot0, n_robots, ext_mask, int_mask, n_int, snap_out, stream); break;
    case 12: launch_k<12, rpb_for(12)>(w, robot0, n_robots, ext_mask, int_mask, n_int, snap_out, stream); break;
    case 16: launch_k<16, rpb_for(16)>(w, robot0, n_robots, ext_mask, int_mask, n_int, snap_out, stream); break;
    case 21: launch_k<21, rpb_for(21)>(w, robot0, n_robots, ext_mask, int_mask, n_int, snap_out, stream); break;
    case 32: launch_k<32, rpb_for(32)>(w, robot0, n_robots, ext_mask, int_mask, n_int, snap_out, stream); break;
    default: launch_k<0, 1>(w, robot0, n_robots, ext_mask, int_mask, n_int, snap_out, stream); break;
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
