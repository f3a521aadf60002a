// gbp_math.h — per-lane GBP arithmetic of the MI355X engine (f64, DOFS = 4).
//
// Pure inline functions on register-resident fixed-size arrays; used by the gfx950 kernels
// in mgx_kernels.hip (one call per lane).  Every function names the reference code whose
// result it reproduces (paths relative to /root/reference/crates/magics/src/factorgraph).
// Unlike the reference these exploit the known sparsity of each factor's Jacobian
// (zeros are never multiplied), which changes nothing for finite operands.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define MGX_HD __host__ __device__ __forceinline__
#else
#define MGX_HD inline
#endif

namespace mgx {

// 4x4 inverse by cofactor expansion — the `Option` contract of ndarray-inverse 0.1.9 `inv()` used at
// variable.rs:153,278 and factor/marginalise_factor_distance.rs:79: `None` iff det == 0 exactly.
//   cofactor(i, j) = (-1)^(i+j) * det3(rows != i, columns != j),
//   det3([[a b c],[d e f],[g h k]]) = (a (e k - f h) - b (d k - f g)) + c (d h - e g),
//   det = ((m00 C00 + m01 C01) + m02 C02) + m03 C03,   inverse[j][i] = C(i, j) * (1 / det).
// Every cofactor is the SAME expression of the three remaining rows, so a row of cofactors can be
// computed by one lane from the other three rows of the matrix (experiments/four_lane_forms.h) with
// results identical to this single-lane form.

// unsigned minors of the row that was removed: r0, r1, r2 are the remaining rows in ascending order
MGX_HD void minors_of_removed_row(const double (&r0)[4], const double (&r1)[4], const double (&r2)[4], double (&mn)[4]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int c0 = (j == 0) ? 1 : 0, c1 = (j <= 1) ? 2 : 1, c2 = (j <= 2) ? 3 : 2;
        const double a = r0[c0], b = r0[c1], c = r0[c2];
        const double d = r1[c0], e = r1[c1], f = r1[c2];
        const double g = r2[c0], h = r2[c1], k = r2[c2];
        mn[j] = (a * (e * k - f * h) - b * (d * k - f * g)) + c * (d * h - e * g);
    }
}
// the first two of them (columns 0, 1)
MGX_HD void minors_of_removed_row_first2(const double (&r0)[4], const double (&r1)[4], const double (&r2)[4], double (&mn)[2]) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int c0 = (j == 0) ? 1 : 0, c1 = 2, c2 = 3;
        const double a = r0[c0], b = r0[c1], c = r0[c2];
        const double d = r1[c0], e = r1[c1], f = r1[c2];
        const double g = r2[c0], h = r2[c1], k = r2[c2];
        mn[j] = (a * (e * k - f * h) - b * (d * k - f * g)) + c * (d * h - e * g);
    }
}
// cofactors C(i, 0..3) of row i from the minors of that row
MGX_HD void cofactors_from_minors(int i, const double (&mn)[4], double (&cf)[4]) {
#pragma unroll
    for (int j = 0; j < 4; j++) cf[j] = ((i + j) & 1) ? -mn[j] : mn[j];
}
MGX_HD double det_from_row0(const double (&row0)[4], const double (&cf0)[4]) {
    return ((row0[0] * cf0[0] + row0[1] * cf0[1]) + row0[2] * cf0[2]) + row0[3] * cf0[3];
}

MGX_HD bool inv4(const double (&m)[16], double (&o)[16]) {
    double cf[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double rows[3][4];
        int n = 0;
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (r != i) {
#pragma unroll
                for (int c = 0; c < 4; c++) rows[n][c] = m[r * 4 + c];
                n++;
            }
        double mn[4];
        minors_of_removed_row(rows[0], rows[1], rows[2], mn);
        cofactors_from_minors(i, mn, cf[i]);
    }
    const double row0[4] = {m[0], m[1], m[2], m[3]};
    const double det = det_from_row0(row0, cf[0]);
    if (det == 0.0) return false;
    const double id = 1.0 / det;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) o[j * 4 + i] = cf[i][j] * id;
    return true;
}

MGX_HD bool any_inf16(const double (&a)[16]) {
    bool r = false;
#pragma unroll
    for (int i = 0; i < 16; i++) r = r || std::isinf(a[i]);
    return r;
}

// Belief update of VariableNode::update_belief_and_create_factor_responses
// (variable.rs:273-297) given the already summed (eta, lam).  Returns false — mu / cov / valid keep
// their previous values — when the precision is "zero" (no element > 1e-6) or singular; otherwise
// cov is replaced, valid recomputed, and mu replaced iff the covariance is finite.
MGX_HD bool belief_update(const double (&eta)[4], const double (&lam)[16], double (&mu)[4], double (&cov)[16],
                          int &valid) {
    // variable.rs:276 tests `x - 1e-6 > 0.0`; for every double x that is the same predicate as
    // `x > 1e-6` (the subtraction is exact within a factor 2 of 1e-6 — Sterbenz — and cannot change
    // sign outside it; NaN fails both), so the subtraction is not performed
    bool not_zero = false;
#pragma unroll
    for (int i = 0; i < 16; i++) not_zero = not_zero || (lam[i] > 1e-6);
    if (!not_zero) return false;
    if (!inv4(lam, cov)) return false;
    bool fin = true;
#pragma unroll
    for (int i = 0; i < 16; i++) fin = fin && std::isfinite(cov[i]);
    valid = fin ? 1 : 0;
    if (fin) {
#pragma unroll
        for (int r = 0; r < 4; r++)
            mu[r] = ((cov[r * 4 + 0] * eta[0] + cov[r * 4 + 1] * eta[1]) + cov[r * 4 + 2] * eta[2]) + cov[r * 4 + 3] * eta[3];
    }
    return true;
}

MGX_HD void belief_from_information(const double (&eta)[4], const double (&lam)[16], double (&mu)[4],
                                    double (&cov)[16], int &valid) {
    double c[16];
    if (belief_update(eta, lam, mu, c, valid)) {
#pragma unroll
        for (int i = 0; i < 16; i++) cov[i] = c[i];
    }
}

// Schur-complement marginalisation of an 8-dim potential onto block `a`
// (factor/marginalise_factor_distance.rs:74-127):
//   eta = ea - Lab Lbb^-1 eb ;  lam = Laa - Lab Lbb^-1 Lba
// Returns false for the reference's `Message::empty()` cases: Lbb singular (det == 0) or an
// infinite element in the result (NaN does not trigger, :117).
MGX_HD bool schur4(const double (&laa)[16], const double (&lab)[16], const double (&lba)[16],
                   const double (&lbb)[16], const double (&ea)[4], const double (&eb)[4], double (&eo)[4],
                   double (&lo)[16]) {
    double w[16];
    if (!inv4(lbb, w)) return false;
    double t[16];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++)
            t[r * 4 + c] = ((lab[r * 4 + 0] * w[0 * 4 + c] + lab[r * 4 + 1] * w[1 * 4 + c]) + lab[r * 4 + 2] * w[2 * 4 + c]) +
                           lab[r * 4 + 3] * w[3 * 4 + c];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const double tv = ((t[r * 4 + 0] * eb[0] + t[r * 4 + 1] * eb[1]) + t[r * 4 + 2] * eb[2]) + t[r * 4 + 3] * eb[3];
        eo[r] = ea[r] - tv;
    }
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const double tm = ((t[r * 4 + 0] * lba[0 * 4 + c] + t[r * 4 + 1] * lba[1 * 4 + c]) + t[r * 4 + 2] * lba[2 * 4 + c]) +
                              t[r * 4 + 3] * lba[3 * 4 + c];
            lo[r * 4 + c] = laa[r * 4 + c] - tm;
        }
    return !any_inf16(lo);
}

// ---------------------------------------------------------------------------------------
// Dynamic factor (factor/dynamic.rs:22-76, factor/mod.rs:334-454).
// Its potential is constant: lam_p = J^T Q J = M (x) I2 with M the 4x4 over the blocks
// (pos_i, vel_i, pos_i+1, vel_i+1), and eta_p = J^T Q (J x0 + (0 - J x0)) == 0 exactly.
// The lane that sends to slot `a` holds maa, mab, mba, mbb = the 2x2 blocks of M seen from a.
// (eo, lo) is the OTHER variable's variable->factor message (zeros when empty).
// ---------------------------------------------------------------------------------------
MGX_HD bool dynamic_message(const double (&maa)[4], const double (&mab)[4], const double (&mba)[4],
                            const double (&mbb)[4], const double (&eo)[4], const double (&lo)[16],
                            double (&out_eta)[4], double (&out_lam)[16]) {
    double lbb[16];
#pragma unroll
    for (int i = 0; i < 16; i++) lbb[i] = lo[i];
    // (M_bb (x) I2)[2b+p][2c+p] = mbb[b][c]
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int p = 0; p < 2; p++) lbb[(2 * b + p) * 4 + (2 * c + p)] = mbb[b * 2 + c] + lo[(2 * b + p) * 4 + (2 * c + p)];
    double w[16];
    if (!inv4(lbb, w)) return false;
    // T = (M_ab (x) I2) W : T[2a+p][col] = sum_b mab[a][b] W[2b+p][col]
    double t[16];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int c = 0; c < 4; c++)
                t[(2 * a + p) * 4 + c] = mab[a * 2 + 0] * w[(0 + p) * 4 + c] + mab[a * 2 + 1] * w[(2 + p) * 4 + c];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const double tv = ((t[r * 4 + 0] * eo[0] + t[r * 4 + 1] * eo[1]) + t[r * 4 + 2] * eo[2]) + t[r * 4 + 3] * eo[3];
        out_eta[r] = 0.0 - tv;
    }
    // lam = (M_aa (x) I2) - T (M_ba (x) I2): (T Mba)[r][2d+q] = sum_a' T[r][2a'+q] mba[a'][d]
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int d = 0; d < 2; d++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const double tm = t[r * 4 + (0 + q)] * mba[0 * 2 + d] + t[r * 4 + (2 + q)] * mba[1 * 2 + d];
                const int a = r >> 1, p = r & 1;
                const double base = (p == q) ? maa[a * 2 + d] : 0.0;
                out_lam[r * 4 + (2 * d + q)] = base - tm;
            }
    return !any_inf16(out_lam);
}

// ---------------------------------------------------------------------------------------
// Obstacle factor (factor/obstacle.rs:141-188 + Factor::first_order_jacobian,
// factor/mod.rs:102-128).
// ---------------------------------------------------------------------------------------
// Rust `f64 as u32`: saturating, NaN -> 0.
MGX_HD uint32_t sat_u32(double v) {
    if (!(v > 0.0)) return 0u;
    if (v >= 4294967295.0) return 4294967295u;
    return (uint32_t)v;
}

struct SdfView {
    const uint8_t *red;  // single channel (red of the reference's Rgb<u8> image), row-major
    uint32_t w, h;
    // ObstacleFactor::measure recomputes these four from (world size, image size) on every call
    // (obstacle.rs:147-150); they are loop invariants, evaluated once with the same f64 operations
    double x_off, y_off, x_scale, y_scale;
};

MGX_HD SdfView make_sdf_view(const uint8_t *red, uint32_t w, uint32_t h, double world_w, double world_h) {
    SdfView s;
    s.red = red;
    s.w = w;
    s.h = h;
    s.x_off = world_w / 2.0;
    s.y_off = world_h / 2.0;
    s.x_scale = (double)w / world_w;
    s.y_scale = (double)h / world_h;
    return s;
}

// pixel index of ObstacleFactor::measure, or -1 when outside the image (=> h = 0)
MGX_HD long long sdf_index(const SdfView &s, double x, double y) {
    const uint32_t xp = sat_u32((x + s.x_off) * s.x_scale);
    const uint32_t yp = sat_u32((-y + s.y_off) * s.y_scale);
    if (!(xp < s.w && yp < s.h)) return -1;
    return (long long)yp * s.w + xp;
}
MGX_HD double sdf_value(uint8_t red) { return 1.0 - (double)red / 255.0; }

// The four sample positions of the forward-difference Jacobian.  The reference perturbs
// x[i] += delta ... x[i] -= delta in place, so later columns see (x+d)-d, not x.
MGX_HD void obstacle_taps(const SdfView &s, double x, double y, double delta, long long (&idx)[4]) {
    idx[0] = sdf_index(s, x, y);
    idx[1] = sdf_index(s, x + delta, y);
    const double xr = (x + delta) - delta;
    idx[2] = sdf_index(s, xr, y + delta);
    const double yr = (y + delta) - delta;
    idx[3] = sdf_index(s, xr, yr);  // columns 2 and 3 (velocity perturbations) sample here
}

// Message of the obstacle factor = its 4-dim potential (marginalise passthrough,
// marginalise_factor_distance.rs:62-72): lam = J^T (1/sigma^2) J, eta = J^T (1/s^2)(J x0 + (0 - h0)).
MGX_HD void obstacle_message(const double (&h)[4], double delta, double inv_sigma2, const double (&x0)[4],
                             double (&eta)[4], double (&lam)[16]) {
    double J[4];
    J[0] = (h[1] - h[0]) / delta;
    J[1] = (h[2] - h[0]) / delta;
    J[2] = (h[3] - h[0]) / delta;
    J[3] = (h[3] - h[0]) / delta;
    double jl[4];
#pragma unroll
    for (int i = 0; i < 4; i++) jl[i] = J[i] * inv_sigma2;
    const double jx = ((J[0] * x0[0] + J[1] * x0[1]) + J[2] * x0[2]) + J[3] * x0[3];
    const double rhs = jx + (0.0 - h[0]);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        eta[i] = jl[i] * rhs;
#pragma unroll
        for (int j = 0; j < 4; j++) lam[i * 4 + j] = jl[i] * J[j];
    }
}

// Row-split form of the two functions above for four cooperating lanes (q = 0..3): lane q samples
// tap q, and, given all four samples, produces row q of the message.  Every element is computed by
// the same operations as in obstacle_message.
MGX_HD long long obstacle_tap(const SdfView &s, double x, double y, double delta, int q) {
    const double xr = (x + delta) - delta, yr = (y + delta) - delta;
    const double tx = (q == 0) ? x : (q == 1 ? x + delta : xr);
    const double ty = (q <= 1) ? y : (q == 2 ? y + delta : yr);
    return sdf_index(s, tx, ty);
}
MGX_HD void obstacle_message_row(const double (&h)[4], double delta, double inv_sigma2, const double (&x0)[4], int q,
                                 double &eta_q, double (&lam_q)[4]) {
    double J[4];
    J[0] = (h[1] - h[0]) / delta;
    J[1] = (h[2] - h[0]) / delta;
    J[2] = (h[3] - h[0]) / delta;
    J[3] = J[2];
    const double jq = (q == 0) ? J[0] : (q == 1 ? J[1] : J[2]);
    const double jl = jq * inv_sigma2;
    const double jx = ((J[0] * x0[0] + J[1] * x0[1]) + J[2] * x0[2]) + J[3] * x0[3];
    const double rhs = jx + (0.0 - h[0]);
    eta_q = jl * rhs;
#pragma unroll
    for (int j = 0; j < 4; j++) lam_q[j] = jl * J[j];
}

// ---------------------------------------------------------------------------------------
// Inter-robot factor (factor/interrobot.rs:91-226, factor/mod.rs:334-454).
// x_lo / x_hi: linearisation-point halves of slot 0 / slot 1 (the graph with the lower /
// higher order key; zeros when that inbox entry is empty).  Computes the message to slot
// `dst_slot` given the other slot's variable->factor message (eo, lo) (zeros when empty).
// Returns false for an empty message (skip, singular, inf).
// ---------------------------------------------------------------------------------------
MGX_HD bool interrobot_message(const double (&x_lo)[4], const double (&x_hi)[4], double d_safe, double tiny_offset,
                               double inv_sigma2, int dst_slot, const double (&eo)[4], const double (&lo)[16],
                               double (&out_eta)[4], double (&out_lam)[16]) {
    const double dx = x_lo[0] - x_hi[0], dy = x_lo[1] - x_hi[1];
    if (dx * dx + dy * dy >= d_safe * d_safe) return false;  // skip(), :213-226 (no offset)
    const double d0 = dx + tiny_offset, d1 = dy + tiny_offset;  // :91-106
    const double r = std::sqrt(d0 * d0 + d1 * d1);
    double h0 = 0.0, jl0 = 0.0, jl1 = 0.0, jh0 = 0.0, jh1 = 0.0;
    if (r <= d_safe) {  // :148-160, :188-200
        h0 = 1.0 * (1.0 - r / d_safe);
        const double cl = -1.0 / d_safe / r, ch = 1.0 / d_safe / r;
        jl0 = cl * d0; jl1 = cl * d1; jh0 = ch * d0; jh1 = ch * d1;
    }
    // row 0 of J = [jl0 jl1 0 0 | jh0 jh1 0 0]; lam_p = (J0^T / s^2) J0 ; eta_p = (J0^T / s^2) rhs
    // J x0 (factor/mod.rs:399) is an ndarray mat-vec: each row . x0 goes through numeric_util::unrolled_dot
    // (ndarray 0.15.6, no BLAS), which for the 8-long rows of a two-variable factor keeps eight partial sums and
    // adds them pairwise, (p0 + p4) + (p1 + p5) + (p2 + p6) + (p3 + p7): lane 0 of slot 0 with lane 0 of slot 1
    const double jx = (jl0 * x_lo[0] + jh0 * x_hi[0]) + (jl1 * x_lo[1] + jh1 * x_hi[1]);
    const double rhs = jx + (0.0 - h0);
    const double ja0 = dst_slot ? jh0 : jl0, ja1 = dst_slot ? jh1 : jl1;  // target block
    const double jb0 = dst_slot ? jl0 : jh0, jb1 = dst_slot ? jl1 : jh1;  // marginalised block
    const double wa0 = ja0 * inv_sigma2, wa1 = ja1 * inv_sigma2;
    const double wb0 = jb0 * inv_sigma2, wb1 = jb1 * inv_sigma2;
    double laa[16], lab[16], lba[16], lbb[16], ea[4], eb[4];
#pragma unroll
    for (int i = 0; i < 16; i++) { laa[i] = 0.0; lab[i] = 0.0; lba[i] = 0.0; lbb[i] = lo[i]; }
    laa[0] = wa0 * ja0; laa[1] = wa0 * ja1; laa[4] = wa1 * ja0; laa[5] = wa1 * ja1;
    lab[0] = wa0 * jb0; lab[1] = wa0 * jb1; lab[4] = wa1 * jb0; lab[5] = wa1 * jb1;
    lba[0] = wb0 * ja0; lba[1] = wb0 * ja1; lba[4] = wb1 * ja0; lba[5] = wb1 * ja1;
    lbb[0] = wb0 * jb0 + lo[0]; lbb[1] = wb0 * jb1 + lo[1]; lbb[4] = wb1 * jb0 + lo[4]; lbb[5] = wb1 * jb1 + lo[5];
    ea[0] = wa0 * rhs; ea[1] = wa1 * rhs; ea[2] = 0.0; ea[3] = 0.0;
    eb[0] = wb0 * rhs + eo[0]; eb[1] = wb1 * rhs + eo[1]; eb[2] = eo[2]; eb[3] = eo[3];
    return schur4(laa, lab, lba, lbb, ea, eb, out_eta, out_lam);
}

// The same message in the form the engine keeps it (six numbers — eta[0..1] and the position block of lam; the
// velocity rows and columns of an inter-robot message are structural zeros): only the terms that are not products with a
// structural zero of the Jacobian blocks are evaluated, each kept output by the same operations in the same order as above.
//   Lab, Lba, Laa have entries in their top-left 2x2 only, so T = Lab W needs rows 0, 1 of W = Lbb^-1 (ten of the sixteen
//   cofactors: columns 0, 1 of the cofactor matrix and row 0 for the determinant), eta[r] = ea[r] - T[r][0..3] . eb, and
//   lam[r][c] = Laa[r][c] - (T[r][0] Lba[0][c] + T[r][1] Lba[1][c]) for r, c < 2; everything else of the dense result is
//   0 - (sums of x * 0), which is never infinite, so the `Message::empty()` test reads the four kept entries.
// x + 0.0 * y == x for finite y: the kept numbers equal interrobot_message's unless the dense form meets inf / NaN there
// (structural zeros).  out: eta0, eta1, lam00, lam01, lam10, lam11.
MGX_HD bool interrobot_message_compact(const double (&x_lo)[4], const double (&x_hi)[4], double d_safe, double tiny_offset,
                                       double inv_sigma2, int dst_slot, const double (&eo)[4], const double (&lo)[16],
                                       double (&out)[6]) {
    const double dx = x_lo[0] - x_hi[0], dy = x_lo[1] - x_hi[1];
    if (dx * dx + dy * dy >= d_safe * d_safe) return false;
    const double d0 = dx + tiny_offset, d1 = dy + tiny_offset;
    const double r = std::sqrt(d0 * d0 + d1 * d1);
    double h0 = 0.0, jl0 = 0.0, jl1 = 0.0, jh0 = 0.0, jh1 = 0.0;
    if (r <= d_safe) {
        h0 = 1.0 * (1.0 - r / d_safe);
        const double cl = -1.0 / d_safe / r, ch = 1.0 / d_safe / r;
        jl0 = cl * d0; jl1 = cl * d1; jh0 = ch * d0; jh1 = ch * d1;
    }
    const double jx = (jl0 * x_lo[0] + jh0 * x_hi[0]) + (jl1 * x_lo[1] + jh1 * x_hi[1]);
    const double rhs = jx + (0.0 - h0);
    const double ja0 = dst_slot ? jh0 : jl0, ja1 = dst_slot ? jh1 : jl1;
    const double jb0 = dst_slot ? jl0 : jh0, jb1 = dst_slot ? jl1 : jh1;
    const double wa0 = ja0 * inv_sigma2, wa1 = ja1 * inv_sigma2;
    const double wb0 = jb0 * inv_sigma2, wb1 = jb1 * inv_sigma2;
    double lbb[16];
#pragma unroll
    for (int i = 0; i < 16; i++) lbb[i] = lo[i];
    lbb[0] = wb0 * jb0 + lo[0]; lbb[1] = wb0 * jb1 + lo[1]; lbb[4] = wb1 * jb0 + lo[4]; lbb[5] = wb1 * jb1 + lo[5];
    // rows 0, 1 of W: W[j][i] = C(i, j) / det
    double cf0[4], c0[4], c1[4];  // cofactors of row 0; columns 0, 1 of the cofactor matrix
    {
        const double r0[4] = {lbb[0], lbb[1], lbb[2], lbb[3]}, r1[4] = {lbb[4], lbb[5], lbb[6], lbb[7]};
        const double r2[4] = {lbb[8], lbb[9], lbb[10], lbb[11]}, r3[4] = {lbb[12], lbb[13], lbb[14], lbb[15]};
        double mn[4];
        minors_of_removed_row(r1, r2, r3, mn);
        cofactors_from_minors(0, mn, cf0);
        c0[0] = cf0[0]; c1[0] = cf0[1];
        double m2[2];
        minors_of_removed_row_first2(r0, r2, r3, m2);
        c0[1] = -m2[0]; c1[1] = m2[1];
        minors_of_removed_row_first2(r0, r1, r3, m2);
        c0[2] = m2[0]; c1[2] = -m2[1];
        minors_of_removed_row_first2(r0, r1, r2, m2);
        c0[3] = -m2[0]; c1[3] = m2[1];
        const double det = det_from_row0(r0, cf0);
        if (det == 0.0) return false;
        const double id = 1.0 / det;
#pragma unroll
        for (int i = 0; i < 4; i++) { c0[i] = c0[i] * id; c1[i] = c1[i] * id; }  // W[0][i], W[1][i]
    }
    const double lab00 = wa0 * jb0, lab01 = wa0 * jb1, lab10 = wa1 * jb0, lab11 = wa1 * jb1;
    double t0[4], t1[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        t0[c] = lab00 * c0[c] + lab01 * c1[c];
        t1[c] = lab10 * c0[c] + lab11 * c1[c];
    }
    const double eb0 = wb0 * rhs + eo[0], eb1 = wb1 * rhs + eo[1];
    out[0] = wa0 * rhs - (((t0[0] * eb0 + t0[1] * eb1) + t0[2] * eo[2]) + t0[3] * eo[3]);
    out[1] = wa1 * rhs - (((t1[0] * eb0 + t1[1] * eb1) + t1[2] * eo[2]) + t1[3] * eo[3]);
    const double lba00 = wb0 * ja0, lba01 = wb0 * ja1, lba10 = wb1 * ja0, lba11 = wb1 * ja1;
    out[2] = wa0 * ja0 - (t0[0] * lba00 + t0[1] * lba10);
    out[3] = wa0 * ja1 - (t0[0] * lba01 + t0[1] * lba11);
    out[4] = wa1 * ja0 - (t1[0] * lba00 + t1[1] * lba10);
    out[5] = wa1 * ja1 - (t1[0] * lba01 + t1[1] * lba11);
    return !(std::isinf(out[2]) || std::isinf(out[3]) || std::isinf(out[4]) || std::isinf(out[5]));
}

// ---------------------------------------------------------------------------------------
// Tracking factor (factor/tracking.rs:171-346,362-381).  `record`, `last_pos`, `last_val`
// are the factor's mutable state (Mutex<Cell<..>> in the reference).  Returns false when the
// factor is skipped (empty message).  path: f32 (x,y) points, n_path >= 2.
// ---------------------------------------------------------------------------------------
MGX_HD double norm2(double a, double b) { return std::sqrt(a * a + b * b); }

MGX_HD bool tracking_message(const float *path, int n_path, double pad, double attraction, double inv_sigma2,
                             const double (&x0)[4], int &record, float (&last_pos)[2], double &last_val,
                             double (&eta)[4], double (&lam)[16]) {
    if (n_path < 2 || record >= n_path - 1) return false;  // skip(), :373-379
    const int rec = record;
    const double px = x0[0], py = x0[1];
    const double csx = (double)path[2 * rec], csy = (double)path[2 * rec + 1];
    const double cex = (double)path[2 * rec + 2], cey = (double)path[2 * rec + 3];
    const double lx = cex - csx, ly = cey - csy;
    const double tt = ((px - csx) * lx + (py - csy) * ly) / (lx * lx + ly * ly);
    const double cx = csx + tt * lx, cy = csy + tt * ly;  // current projection, :222-223
    const double d0 = pad, d1 = pad * 0.01;                 // :231-244
    const double dist_end = norm2(cex - cx, cey - cy);
    bool use_prev = false;
    double ppx = 0.0, ppy = 0.0;
    if (rec > 0) {  // :255-286
        const double psx = (double)path[2 * (rec - 1)], psy = (double)path[2 * (rec - 1) + 1];
        const double pex = csx, pey = csy;
        const double plx = pex - psx, ply = pey - psy;
        const double t2 = ((px - psx) * plx + (py - psy) * ply) / (plx * plx + ply * ply);
        ppx = psx + t2 * plx;
        ppy = psy + t2 * ply;
        const double a = norm2(pex - cx, pey - cy);
        const double b = norm2(csx - ppx, csy - ppy);
        use_prev = (a < d0) && (a > d1) && (b < d0);
    }
    if (dist_end < d0) {  // :294-296, increment_record :54-64
        int nr = rec + 1;
        if (nr > n_path - 2) nr = n_path - 2;
        record = nr;
    }
    double mx, my;
    if (use_prev) {  // :300-311
        mx = px + ((cx - px) + (ppx - px));
        my = py + ((cy - py) + (ppy - py));
    } else {  // :312-316; normalized() leaves a zero / infinite vector untouched
        double nx = lx, ny = ly;
        const double mag = norm2(lx, ly);
        if (!(mag == 0.0 || std::isinf(mag))) { nx = lx / mag; ny = ly / mag; }
        const double vn = norm2(x0[2], x0[3]);
        mx = cx + nx * vn / 5.0;
        my = cy + ny * vn / 5.0;
    }
    const double dist = norm2(mx - px, my - py);
    const double meas = (dist < attraction) ? dist / attraction : 1.0;  // :322-333
    last_pos[0] = (float)mx;  // :336-339 (f32 Vec2)
    last_pos[1] = (float)my;
    last_val = meas;
    // jacobian(), :171-194: uses the state just stored
    const double inv_h = 1.0 / meas;
    double J[4] = {inv_h * (px - (double)last_pos[0]), inv_h * (py - (double)last_pos[1]), 0.0, 0.0};
    double jl[4];
#pragma unroll
    for (int i = 0; i < 4; i++) jl[i] = J[i] * inv_sigma2;
    const double jx = J[0] * x0[0] + J[1] * x0[1];
    const double rhs = jx + (0.0 - meas);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        eta[i] = jl[i] * rhs;
#pragma unroll
        for (int j = 0; j < 4; j++) lam[i * 4 + j] = jl[i] * J[j];
    }
    return true;
}

// TrackingFactor::skip's timeout (tracking.rs:153-155,362-371: Option<usize>, set by FactorGraph::reset_tracking_factors)
// rides in the upper half of the factor's `record` word: 0 = None, n + 1 = Some(n).  Returns true when the factor
// skips this update because of it (Some(n > 0) -> Some(n - 1)); Some(0) becomes None and the update goes ahead.
MGX_HD bool tracking_timeout_skips(int &record_packed) {
    const int code = record_packed >> 16;
    if (code == 0) return false;
    const int rec = record_packed & 0xffff;
    if (code == 1) { record_packed = rec; return false; }
    record_packed = rec | ((code - 1) << 16);
    return true;
}
// tracking_message on the packed record word (the timeout survives the update)
MGX_HD bool tracking_update(const float *path, int n_path, double pad, double attraction, double inv_sigma2, const double (&x0)[4],
                            int &record_packed, float (&last_pos)[2], double &last_val, double (&eta)[4], double (&lam)[16]) {
    if (tracking_timeout_skips(record_packed)) return false;
    int rec = record_packed & 0xffff;
    const int code = record_packed >> 16;
    const bool ok = tracking_message(path, n_path, pad, attraction, inv_sigma2, x0, rec, last_pos, last_val, eta, lam);
    record_packed = rec | (code << 16);
    return ok;
}

}  // namespace mgx
