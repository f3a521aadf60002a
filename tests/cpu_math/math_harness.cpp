// Test-only host build of the engine's per-lane arithmetic (magics_amd/csrc/gbp_math.h) so that
// the exact functions the gfx950 kernels call can be checked on a CPU-only machine
// (tests/test_lane_math.py).  Not part of the product library.
#include "../../magics_amd/csrc/gbp_math.h"
#include "../../experiments/four_lane_forms.h"  // record of the cooperative-lane experiment, not product code

extern "C" {
int h_inv4(const double *m, double *o) {
    double a[16], b[16];
    for (int i = 0; i < 16; i++) a[i] = m[i];
    bool ok = mgx::inv4(a, b);
    for (int i = 0; i < 16; i++) o[i] = b[i];
    return ok;
}
int h_dynamic_message(const double *M, int slot, const double *eo, const double *lo, double *oe, double *ol) {
    double maa[4], mab[4], mba[4], mbb[4], e[4], l[16], re[4], rl[16];
    const int a2 = 2 * slot, b2 = 2 * (1 - slot);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) {
            maa[i * 2 + j] = M[(a2 + i) * 4 + a2 + j];
            mab[i * 2 + j] = M[(a2 + i) * 4 + b2 + j];
            mba[i * 2 + j] = M[(b2 + i) * 4 + a2 + j];
            mbb[i * 2 + j] = M[(b2 + i) * 4 + b2 + j];
        }
    for (int i = 0; i < 4; i++) e[i] = eo[i];
    for (int i = 0; i < 16; i++) l[i] = lo[i];
    bool ok = mgx::dynamic_message(maa, mab, mba, mbb, e, l, re, rl);
    for (int i = 0; i < 4; i++) oe[i] = re[i];
    for (int i = 0; i < 16; i++) ol[i] = rl[i];
    return ok;
}
int h_interrobot_message(const double *xlo, const double *xhi, double dsafe, double off, double inv_s2, int dst_slot,
                         const double *eo, const double *lo, double *oe, double *ol) {
    double a[4], b[4], e[4], l[16], re[4], rl[16];
    for (int i = 0; i < 4; i++) { a[i] = xlo[i]; b[i] = xhi[i]; e[i] = eo[i]; }
    for (int i = 0; i < 16; i++) l[i] = lo[i];
    bool ok = mgx::interrobot_message(a, b, dsafe, off, inv_s2, dst_slot, e, l, re, rl);
    for (int i = 0; i < 4; i++) oe[i] = re[i];
    for (int i = 0; i < 16; i++) ol[i] = rl[i];
    return ok;
}
// four-lane dynamic message emulated lane by lane, stage by stage (what the DYN waves do in LDS)
int h_dynamic_message_4lane(const double *M, int slot, const double *eo, const double *lo, double *oe, double *ol) {
    double maa[4], mab[4], mba[4], mbb[4];
    const int a2 = 2 * slot, b2 = 2 * (1 - slot);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) {
            maa[i * 2 + j] = M[(a2 + i) * 4 + a2 + j];
            mab[i * 2 + j] = M[(a2 + i) * 4 + b2 + j];
            mba[i * 2 + j] = M[(b2 + i) * 4 + a2 + j];
            mbb[i * 2 + j] = M[(b2 + i) * 4 + b2 + j];
        }
    double scr[16], row0[4] = {0, 0, 0, 0};
    auto S = [&](int r, int c) { return scr[r * 4 + c]; };
    for (int q = 0; q < 4; q++) {  // stage 1
        double ml_q[4] = {lo[q * 4], lo[q * 4 + 1], lo[q * 4 + 2], lo[q * 4 + 3]}, row[4];
        mgx::dyn4_row_of_lbb(q, mbb, ml_q, row);
        for (int c = 0; c < 4; c++) scr[q * 4 + c] = row[c];
        if (q == 0)
            for (int c = 0; c < 4; c++) row0[c] = row[c];
    }
    double cf[4][4], det = 0.0;
    for (int q = 0; q < 4; q++) mgx::cofactor_row_from_scratch(q, S, cf[q]);
    det = mgx::det_from_row0(row0, cf[0]);  // lane 0
    if (det == 0.0) return 0;
    const double id = 1.0 / det;
    double tc[4][4], pr[4][4];
    for (int q = 0; q < 4; q++) mgx::dyn4_columns(cf[q], id, mab, eo[q], tc[q], pr[q]);
    bool inf = false;
    for (int q = 0; q < 4; q++) {
        double p_row[4] = {pr[0][q], pr[1][q], pr[2][q], pr[3][q]};  // products of row q come from the four lanes
        oe[q] = mgx::dyn4_eta(p_row);
        double col[4];
        mgx::dyn4_lam_column(q, maa, mba, tc[q], tc[q ^ 2], col);
        for (int r = 0; r < 4; r++) {
            ol[r * 4 + q] = col[r];
            inf = inf || std::isinf(col[r]);
        }
    }
    return inf ? 0 : 1;
}
void h_obstacle(const unsigned char *red, unsigned w, unsigned h, double ww, double wh, double delta, double inv_s2,
                const double *x0, double *oe, double *ol) {
    mgx::SdfView s = mgx::make_sdf_view(red, w, h, ww, wh);
    long long idx[4];
    double x[4], hv[4], re[4], rl[16];
    for (int i = 0; i < 4; i++) x[i] = x0[i];
    mgx::obstacle_taps(s, x[0], x[1], delta, idx);
    for (int q = 0; q < 4; q++) hv[q] = idx[q] >= 0 ? mgx::sdf_value(red[idx[q]]) : 0.0;
    mgx::obstacle_message(hv, delta, inv_s2, x, re, rl);
    for (int i = 0; i < 4; i++) oe[i] = re[i];
    for (int i = 0; i < 16; i++) ol[i] = rl[i];
}
void h_belief(const double *eta, const double *lam, double *mu, double *cov, int *valid) {
    double e[4], l[16], m[4], c[16];
    for (int i = 0; i < 4; i++) { e[i] = eta[i]; m[i] = mu[i]; }
    for (int i = 0; i < 16; i++) { l[i] = lam[i]; c[i] = cov[i]; }
    int v = *valid;
    mgx::belief_from_information(e, l, m, c, v);
    for (int i = 0; i < 4; i++) mu[i] = m[i];
    for (int i = 0; i < 16; i++) cov[i] = c[i];
    *valid = v;
}
int h_tracking(const float *path, int n_path, double pad, double attr, double inv_s2, const double *x0, int *record,
               float *last_pos, double *last_val, double *oe, double *ol) {
    double x[4], re[4], rl[16];
    float lp[2] = {last_pos[0], last_pos[1]};
    for (int i = 0; i < 4; i++) x[i] = x0[i];
    bool ok = mgx::tracking_message(path, n_path, pad, attr, inv_s2, x, *record, lp, *last_val, re, rl);
    last_pos[0] = lp[0];
    last_pos[1] = lp[1];
    for (int i = 0; i < 4; i++) oe[i] = re[i];
    for (int i = 0; i < 16; i++) ol[i] = rl[i];
    return ok;
}
}
