// four_lane_forms.h — cooperative (four lanes per 4x4 problem) forms of the dynamic-factor message and
// the belief finish, bit-identical to the single-lane forms of magics_amd/csrc/gbp_math.h.  The
// product does not use them (the cooperative kernels were slower, see README.md here); they are
// kept with the experiment, and tests/test_lane_math.py still checks them against the single-lane
// forms through tests/cpu_math/math_harness.cpp.
#pragma once
#include "../magics_amd/csrc/gbp_math.h"

namespace mgx {

// ---------------------------------------------------------------------------------------
// FOUR-LANE FORMS.  Four consecutive lanes q = 0..3 of one wave cooperate on one 4x4 problem; they
// exchange data through a small scratch area (LDS on the device: a wave's LDS operations execute in
// program order, so a value written by one lane is visible to a later read of another lane of the
// same wave without a barrier).  scr(r, c) addresses a 4x4 scratch matrix, aux(k) four extra words.
// Each element is produced by exactly the operations of the single-lane forms above.
// ---------------------------------------------------------------------------------------

// cofactors C(q, 0..3) of row q from the three other rows of the matrix held in scratch
template <class Scr>
MGX_HD void cofactor_row_from_scratch(int q, const Scr &scr, double (&cf)[4]) {
    double r[3][4];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int row = k + (k >= q ? 1 : 0);
#pragma unroll
        for (int c = 0; c < 4; c++) r[k][c] = scr(row, c);
    }
    double mn[4];
    minors_of_removed_row(r[0], r[1], r[2], mn);
    cofactors_from_minors(q, mn, cf);
}

// Dynamic-factor message, lane q of 4 (see dynamic_message).  Stage 1: row q of lam_bb and element q
// of the other variable's eta.  ml_q / me_q: row q / element q of the other variable's message.
MGX_HD void dyn4_row_of_lbb(int q, const double (&mbb)[4], const double (&ml_q)[4], double (&row)[4]) {
    const int b = q >> 1, p = q & 1;
#pragma unroll
    for (int c = 0; c < 4; c++) row[c] = ((c & 1) == p) ? mbb[b * 2 + (c >> 1)] + ml_q[c] : ml_q[c];
}
// Stage 2 (after the rows are in scratch and det is known): column q of W = lam_bb^-1, column q of
// T = (M_ab (x) I2) W, and the products T[r][q] * eta_b[q] that make up row r of T eta_b.
MGX_HD void dyn4_columns(const double (&cf)[4], double id, const double (&mab)[4], double me_q, double (&tc)[4],
                         double (&pr)[4]) {
    double wc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) wc[j] = cf[j] * id;  // inverse[j][q]
    tc[0] = mab[0] * wc[0] + mab[1] * wc[2];
    tc[1] = mab[0] * wc[1] + mab[1] * wc[3];
    tc[2] = mab[2] * wc[0] + mab[3] * wc[2];
    tc[3] = mab[2] * wc[1] + mab[3] * wc[3];
#pragma unroll
    for (int r = 0; r < 4; r++) pr[r] = tc[r] * me_q;
}
// Stage 3: element q of the message's eta from the four products of row q, and column q of its lam from
// this lane's and the partner lane's (q ^ 2) columns of T.
MGX_HD double dyn4_eta(const double (&p_row)[4]) { return 0.0 - (((p_row[0] + p_row[1]) + p_row[2]) + p_row[3]); }
MGX_HD void dyn4_lam_column(int q, const double (&maa)[4], const double (&mba)[4], const double (&tc_own)[4],
                            const double (&tc_partner)[4], double (&col)[4]) {
    const int d = q >> 1, qq = q & 1;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const double t_lo = d ? tc_partner[r] : tc_own[r];  // T[r][qq]
        const double t_hi = d ? tc_own[r] : tc_partner[r];  // T[r][2 + qq]
        const double tm = t_lo * mba[0 * 2 + d] + t_hi * mba[1 * 2 + d];
        const double base = ((r & 1) == qq) ? maa[(r >> 1) * 2 + d] : 0.0;
        col[r] = base - tm;
    }
}

// Belief finish, lane q of 4 (see belief_update): column q of the covariance and the products
// cov[r][q] * eta[q] that make up row r of cov eta.
MGX_HD void fin4_column(const double (&cf)[4], double id, double eta_q, double (&covc)[4], double (&pr)[4]) {
#pragma unroll
    for (int j = 0; j < 4; j++) covc[j] = cf[j] * id;  // cov[j][q]
#pragma unroll
    for (int r = 0; r < 4; r++) pr[r] = covc[r] * eta_q;
}
MGX_HD double fin4_mean(const double (&p_row)[4]) { return ((p_row[0] + p_row[1]) + p_row[2]) + p_row[3]; }

}  // namespace mgx
