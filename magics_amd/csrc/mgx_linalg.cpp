// mgx_linalg.cpp — host-side value types of the C ABI (no device): the two small crates the
// factor-graph code is written against,
//   crates/gbp_linalg/src/lib.rs:47-128          vector norms / normalize
//   crates/gbp_multivariate_normal/src/lib.rs    MultivariateNormal in information form
// `magics` itself does not use MultivariateNormal at run time; it is here so that
// code written against that crate finds the same operations, error cases and (sic) cached
// "mean = precision . information" (lib.rs:85,292).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/mgx.h"

extern "C" int mgx_set_error_(int code, const char *text);  // mgx_world.hip: thread-local last error

namespace {

int fail(int code, const char *fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return mgx_set_error_(code, buf);
}

// ndarray-inverse 0.1.9 (Cargo.lock:4870) is not vendored: its contract is restated — det() by
// cofactor expansion along the first row for n <= 4 (the order the engine's 4x4 inverse uses), by
// LU with partial pivoting beyond; inv() = None exactly when det() == 0, else cofactors / det for
// n <= 4 and Gauss-Jordan beyond.
double det_small(const double *a, int n, int stride) {
    if (n == 1) return a[0];
    if (n == 2) return a[0] * a[stride + 1] - a[1] * a[stride];
    double minor[9], acc = 0.0;
    for (int j = 0; j < n; j++) {
        for (int r = 1; r < n; r++) {
            int cc = 0;
            for (int c = 0; c < n; c++)
                if (c != j) minor[(r - 1) * (n - 1) + cc++] = a[r * stride + c];
        }
        const double m = det_small(minor, n - 1, n - 1);
        const double term = a[j] * ((j & 1) ? -m : m);
        acc = (j == 0) ? term : acc + term;
    }
    return acc;
}
double det_lu(std::vector<double> a, int n) {
    double det = 1.0;
    for (int k = 0; k < n; k++) {
        int p = k;
        for (int r = k + 1; r < n; r++)
            if (std::fabs(a[(size_t)r * n + k]) > std::fabs(a[(size_t)p * n + k])) p = r;
        if (a[(size_t)p * n + k] == 0.0) return 0.0;
        if (p != k) {
            for (int c = 0; c < n; c++) std::swap(a[(size_t)p * n + c], a[(size_t)k * n + c]);
            det = -det;
        }
        det *= a[(size_t)k * n + k];
        for (int r = k + 1; r < n; r++) {
            const double f = a[(size_t)r * n + k] / a[(size_t)k * n + k];
            for (int c = k; c < n; c++) a[(size_t)r * n + c] -= f * a[(size_t)k * n + c];
        }
    }
    return det;
}
double det_n(const double *a, int n) {
    if (n == 0) return 1.0;
    if (n <= 4) return det_small(a, n, n);
    return det_lu(std::vector<double>(a, a + (size_t)n * n), n);
}
bool inverse_n(const double *a, int n, double *out) {
    const double det = det_n(a, n);
    if (det == 0.0) return false;
    if (n == 1) { out[0] = 1.0 / a[0]; return true; }
    if (n <= 4) {  // adjugate / determinant
        const double inv_det = 1.0 / det;
        double minor[9];
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                int rr = 0;
                for (int r = 0; r < n; r++) {
                    if (r == i) continue;
                    int cc = 0;
                    for (int c = 0; c < n; c++)
                        if (c != j) minor[rr * (n - 1) + cc++] = a[r * n + c];
                    rr++;
                }
                const double m = det_small(minor, n - 1, n - 1);
                out[j * n + i] = (((i + j) & 1) ? -m : m) * inv_det;
            }
        return true;
    }
    std::vector<double> m(a, a + (size_t)n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) out[(size_t)i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int k = 0; k < n; k++) {
        int p = k;
        for (int r = k + 1; r < n; r++)
            if (std::fabs(m[(size_t)r * n + k]) > std::fabs(m[(size_t)p * n + k])) p = r;
        if (m[(size_t)p * n + k] == 0.0) return false;
        if (p != k)
            for (int c = 0; c < n; c++) {
                std::swap(m[(size_t)p * n + c], m[(size_t)k * n + c]);
                std::swap(out[(size_t)p * n + c], out[(size_t)k * n + c]);
            }
        const double piv = m[(size_t)k * n + k];
        for (int c = 0; c < n; c++) { m[(size_t)k * n + c] /= piv; out[(size_t)k * n + c] /= piv; }
        for (int r = 0; r < n; r++) {
            if (r == k) continue;
            const double f = m[(size_t)r * n + k];
            if (f == 0.0) continue;
            for (int c = 0; c < n; c++) { m[(size_t)r * n + c] -= f * m[(size_t)k * n + c]; out[(size_t)r * n + c] -= f * out[(size_t)k * n + c]; }
        }
    }
    return true;
}
void matvec(const std::vector<double> &a, const std::vector<double> &x, std::vector<double> &y) {  // ndarray `dot`, k ascending
    const size_t n = x.size();
    y.assign(n, 0.0);
    for (size_t i = 0; i < n; i++) {
        double s = 0.0;
        for (size_t k = 0; k < n; k++) s += a[i * n + k] * x[k];
        y[i] = s;
    }
}

}  // namespace

struct mgx_mvn {
    std::vector<double> information, precision, mean;
    bool dirty = false;
};

extern "C" {

// ---- gbp_linalg -------------------------------------------------------------------------------------
double mgx_euclidean_norm(const double *x, uint32_t n) {  // lib.rs:66-68: sqrt(fold(0, acc + x*x))
    double acc = 0.0;
    for (uint32_t i = 0; i < n; i++) acc = acc + x[i] * x[i];
    return std::sqrt(acc);
}
double mgx_l1_norm(const double *x, uint32_t n) {  // lib.rs:71-73
    double acc = 0.0;
    for (uint32_t i = 0; i < n; i++) acc = acc + std::fabs(x[i]);
    return acc;
}
void mgx_normalize(double *x, uint32_t n) {  // lib.rs:113-124: untouched when the norm is 0 or infinite
    const double mag = mgx_euclidean_norm(x, n);
    if (mag == 0.0 || std::isinf(mag)) return;
    for (uint32_t i = 0; i < n; i++) x[i] /= mag;
}
double mgx_det(const double *a, uint32_t n) { return det_n(a, (int)n); }
int mgx_inverse(const double *a, uint32_t n, double *out) {
    if (!a || !out) return fail(MGX_ERR_INVALID, "null argument");
    return inverse_n(a, (int)n, out) ? 1 : 0;
}

// ---- gbp_multivariate_normal ------------------------------------------------------------------------
static int check_shape(uint32_t len, uint32_t rows, uint32_t cols) {
    if (rows != cols)  // lib.rs:67-71,118-122 (both constructors report NonSquarePrecisionMatrix)
        return fail(MGX_MVN_ERR_NON_SQUARE, "NonSquarePrecisionMatrix(%u, %u)", rows, cols);
    if (len != rows || len != cols)  // lib.rs:72-79,123-128
        return fail(MGX_MVN_ERR_LENGTH, "VectorLengthNotEqualMatrixShape(%u, %u, %u)", len, rows, cols);
    return MGX_OK;
}

int mgx_mvn_from_information_and_precision(const double *information, uint32_t len, const double *precision, uint32_t rows,
                                           uint32_t cols, mgx_mvn **out) {
    if (!information || !precision || !out) return fail(MGX_ERR_INVALID, "null argument");
    int rc = check_shape(len, rows, cols);
    if (rc != MGX_OK) return rc;
    if (det_n(precision, (int)len) == 0.0) return fail(MGX_MVN_ERR_SINGULAR_PRECISION, "NonInvertiblePrecisionMatrix");  // :82-84
    mgx_mvn *m = new (std::nothrow) mgx_mvn();
    if (!m) return fail(MGX_ERR_NOMEM, "out of memory");
    m->information.assign(information, information + len);
    m->precision.assign(precision, precision + (size_t)len * len);
    matvec(m->precision, m->information, m->mean);  // :85 (precision . information, as the reference has it)
    *out = m;
    return MGX_OK;
}

int mgx_mvn_from_mean_and_covariance(const double *mean, uint32_t len, const double *covariance, uint32_t rows, uint32_t cols,
                                     mgx_mvn **out) {
    if (!mean || !covariance || !out) return fail(MGX_ERR_INVALID, "null argument");
    int rc = check_shape(len, rows, cols);
    if (rc != MGX_OK) return rc;
    mgx_mvn *m = new (std::nothrow) mgx_mvn();
    if (!m) return fail(MGX_ERR_NOMEM, "out of memory");
    m->precision.resize((size_t)len * len);
    if (!inverse_n(covariance, (int)len, m->precision.data())) {  // :130-132
        delete m;
        return fail(MGX_MVN_ERR_SINGULAR_COVARIANCE, "NonInvertibleCovarianceMatrix");
    }
    m->mean.assign(mean, mean + len);
    matvec(m->precision, m->mean, m->information);  // :133
    *out = m;
    return MGX_OK;
}

void mgx_mvn_destroy(mgx_mvn *m) { delete m; }
uint32_t mgx_mvn_len(const mgx_mvn *m) { return m ? (uint32_t)m->information.size() : 0; }

int mgx_mvn_get(const mgx_mvn *m, double *information, double *precision, double *mean) {
    if (!m) return fail(MGX_ERR_INVALID, "null argument");
    if (information) memcpy(information, m->information.data(), sizeof(double) * m->information.size());
    if (precision) memcpy(precision, m->precision.data(), sizeof(double) * m->precision.size());
    if (mean) memcpy(mean, m->mean.data(), sizeof(double) * m->mean.size());
    return MGX_OK;
}
int mgx_mvn_covariance(const mgx_mvn *m, double *covariance) {  // :195-199 (`expect`: the invariant may have been broken through set_*)
    if (!m || !covariance) return fail(MGX_ERR_INVALID, "null argument");
    if (!inverse_n(m->precision.data(), (int)m->information.size(), covariance))
        return fail(MGX_MVN_ERR_SINGULAR_PRECISION, "the precision matrix is not invertible");
    return MGX_OK;
}

// :271-279: recompute the cached mean if something was set without updating; 1 = recomputed
int mgx_mvn_update(mgx_mvn *m) {
    if (!m) return fail(MGX_ERR_INVALID, "null argument");
    if (!m->dirty) return 0;
    matvec(m->precision, m->information, m->mean);
    m->dirty = false;
    return 1;
}
// Note: update_information_vector / update_precision_matrix call update() without marking the
// value dirty (lib.rs:158-161,169-178), so the cached mean is NOT refreshed by them.
int mgx_mvn_update_information_vector(mgx_mvn *m, const double *value) {
    if (!m || !value) return fail(MGX_ERR_INVALID, "null argument");
    m->information.assign(value, value + m->information.size());
    mgx_mvn_update(m);
    return MGX_OK;
}
int mgx_mvn_update_precision_matrix(mgx_mvn *m, const double *value) {
    if (!m || !value) return fail(MGX_ERR_INVALID, "null argument");
    if (det_n(value, (int)m->information.size()) == 0.0) return fail(MGX_MVN_ERR_SINGULAR_PRECISION, "NonInvertiblePrecisionMatrix");
    m->precision.assign(value, value + m->precision.size());
    mgx_mvn_update(m);
    return MGX_OK;
}
// the `unsafe` setters (:212-263): no checks, the mean goes stale until mgx_mvn_update
int mgx_mvn_set_information_vector(mgx_mvn *m, const double *value) {
    if (!m || !value) return fail(MGX_ERR_INVALID, "null argument");
    m->information.assign(value, value + m->information.size());
    m->dirty = true;
    return MGX_OK;
}
int mgx_mvn_set_precision_matrix(mgx_mvn *m, const double *value) {
    if (!m || !value) return fail(MGX_ERR_INVALID, "null argument");
    m->precision.assign(value, value + m->precision.size());
    m->dirty = true;
    return MGX_OK;
}
int mgx_mvn_add_assign_information_vector(mgx_mvn *m, const double *value) {
    if (!m || !value) return fail(MGX_ERR_INVALID, "null argument");
    for (size_t i = 0; i < m->information.size(); i++) m->information[i] += value[i];
    m->dirty = true;
    return MGX_OK;
}
int mgx_mvn_add_assign_precision_matrix(mgx_mvn *m, const double *value) {
    if (!m || !value) return fail(MGX_ERR_INVALID, "null argument");
    for (size_t i = 0; i < m->precision.size(); i++) m->precision[i] += value[i];
    m->dirty = true;
    return MGX_OK;
}

// Add / Sub / Mul (:300-410): information and precision are added (Mul = Add in information form)
// or subtracted; the mean of the result is precision . information; no invertibility check.
static int combine_into(mgx_mvn *dst, const mgx_mvn *a, const mgx_mvn *b, int op) {
    if (op != MGX_MVN_ADD && op != MGX_MVN_SUB && op != MGX_MVN_MUL) return fail(MGX_ERR_INVALID, "bad operator");
    if (a->information.size() != b->information.size()) return fail(MGX_MVN_ERR_LENGTH, "operands of different dimension");
    const double sgn = (op == MGX_MVN_SUB) ? -1.0 : 1.0;
    std::vector<double> info(a->information.size()), prec(a->precision.size());
    for (size_t i = 0; i < info.size(); i++) info[i] = (sgn > 0) ? a->information[i] + b->information[i] : a->information[i] - b->information[i];
    for (size_t i = 0; i < prec.size(); i++) prec[i] = (sgn > 0) ? a->precision[i] + b->precision[i] : a->precision[i] - b->precision[i];
    dst->information.swap(info);
    dst->precision.swap(prec);
    matvec(dst->precision, dst->information, dst->mean);
    dst->dirty = false;
    return MGX_OK;
}
int mgx_mvn_combine(const mgx_mvn *a, const mgx_mvn *b, int32_t op, mgx_mvn **out) {
    if (!a || !b || !out) return fail(MGX_ERR_INVALID, "null argument");
    mgx_mvn *m = new (std::nothrow) mgx_mvn();
    if (!m) return fail(MGX_ERR_NOMEM, "out of memory");
    int rc = combine_into(m, a, b, op);
    if (rc != MGX_OK) { delete m; return rc; }
    *out = m;
    return MGX_OK;
}
int mgx_mvn_combine_assign(mgx_mvn *a, const mgx_mvn *b, int32_t op) {  // AddAssign / SubAssign / MulAssign
    if (!a || !b) return fail(MGX_ERR_INVALID, "null argument");
    return combine_into(a, a, b, op);
}

}  // extern "C"
