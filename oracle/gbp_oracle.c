/* gbp_oracle.c — CPU restatement of the GBP hot path of AU-Master-Thesis/magics.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (magics_amd/, include/) may
 * link, import or call this file; only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py use it, as the checker / timed CPU baseline.
 *
 * It follows the reference object model line by line (per-robot graphs, ordered
 * inboxes of optional messages, generic-dimension factor linearisation), NOT the
 * flattened formulation of the HIP engine, so that agreement between the two checks
 * the engine's reformulation as well as its arithmetic.  All paths are relative to
 * /root/reference/; FG = crates/magics/src/factorgraph, ROBOT = crates/magics/src/planner/robot.rs.
 *
 * Parity status: the reference is Rust and cannot be built here (no cargo/rustc; 803
 * un-vendored crates).  Its own tests pin only: schedules, get_variable_timesteps,
 * the 4-dim marginalise passthrough, norms — all checked in tests/.  The 4x4 inverse is
 * third-party (ndarray-inverse 0.1.9, Cargo.lock:4870; determinant/adjugate form,
 * `None` when det == 0) and the small GEMMs are ndarray 0.15.6 / matrixmultiply 0.3.8:
 * for those and for factor/variable updates the reference holds no numeric vectors, so
 * general-matrix parity is UNPINNED by the reference ("parity unpinned"); it is pinned
 * instead by analytic known-answer tests (tests/test_oracle_*.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define DOFS 4 /* FG/mod.rs:21 */

enum { K_DYNAMIC = 0, K_INTERROBOT = 1, K_OBSTACLE = 2, K_TRACKING = 3 };
enum { EN_DYN = 1u, EN_IR = 2u, EN_OBS = 4u, EN_TRK = 8u };

typedef struct {
    double sigma_dynamics, sigma_interrobot, sigma_obstacle, sigma_tracking;
    double safety_multiplier, tracking_switch_padding, tracking_attraction_distance;
    uint32_t enable_mask, reserved;
} orc_params;

typedef struct {
    uint32_t K, n_path;
    const double *mean0, *prior_diag, *dt;
    const float *path_xy;
    double radius;
    uint64_t order_key;
    uint32_t ghost, reserved;
} orc_robot_desc;

/* FG/message.rs:19-29,49-53: Option<Box<Payload{eta, lam, mu}>> */
typedef struct {
    int present;
    double eta[DOFS], lam[DOFS * DOFS], mu[DOFS];
} Msg;

/* FG/id.rs: (factorgraph_id, node index), ordered lexicographically (id.rs:19-54,85-117) */
typedef struct {
    int64_t graph; /* order key of the owning graph (Bevy Entity order) */
    int robot;     /* world index of the owning graph */
    int index;     /* node index inside that graph */
} NodeId;

static int id_cmp(NodeId a, NodeId b) {
    if (a.graph != b.graph) return a.graph < b.graph ? -1 : 1;
    if (a.index != b.index) return a.index < b.index ? -1 : 1;
    return 0;
}

typedef struct {
    NodeId key;
    Msg msg;
} InboxEntry;

typedef struct {
    InboxEntry *e;
    int n, cap;
} Inbox; /* BTreeMap<Id, Message> (FG/message.rs:210,217) */

static Msg *inbox_insert(Inbox *ib, NodeId key) {
    int lo = 0;
    while (lo < ib->n && id_cmp(ib->e[lo].key, key) < 0) lo++;
    if (lo < ib->n && id_cmp(ib->e[lo].key, key) == 0) return &ib->e[lo].msg;
    if (ib->n == ib->cap) {
        ib->cap = ib->cap ? 2 * ib->cap : 4;
        ib->e = (InboxEntry *)realloc(ib->e, sizeof(InboxEntry) * (size_t)ib->cap);
    }
    memmove(&ib->e[lo + 1], &ib->e[lo], sizeof(InboxEntry) * (size_t)(ib->n - lo));
    ib->n++;
    ib->e[lo].key = key;
    memset(&ib->e[lo].msg, 0, sizeof(Msg));
    return &ib->e[lo].msg;
}

static void inbox_remove_if_graph(Inbox *ib, int robot) {
    int j = 0;
    for (int i = 0; i < ib->n; i++)
        if (ib->e[i].key.robot != robot) ib->e[j++] = ib->e[i];
    ib->n = j;
}
static void inbox_remove_key(Inbox *ib, NodeId key) {
    int j = 0;
    for (int i = 0; i < ib->n; i++)
        if (id_cmp(ib->e[i].key, key) != 0) ib->e[j++] = ib->e[i];
    ib->n = j;
}

/* FG/variable.rs:86-104 */
typedef struct {
    double prior_eta[DOFS], prior_lam[16];
    double eta[DOFS], lam[16], mu[DOFS], cov[16];
    int valid;
    Inbox inbox;
    int robot;       /* owning graph (world index) */
    uint64_t cnt[4]; /* MessageCount (FG/mod.rs:29-137): sent internal / external, received internal / external */
} Variable;

/* FG/factor/mod.rs:133-148,597-623 + the per-kind structs */
typedef struct {
    int kind, enabled;
    int nvars;           /* neighbours(): 1 or 2 */
    int zdim;            /* len(initial_measurement) */
    double z[4];         /* initial_measurement */
    double lam_meas[16]; /* measurement_precision, zdim x zdim */
    double x0[8];        /* linearisation_point */
    Inbox inbox;
    int robot;           /* owning graph (world index) */
    uint64_t cnt[4];     /* MessageCount, as in Variable */
    /* dynamic (factor/dynamic.rs:14-52) */
    double J_dyn[4 * 8];
    /* interrobot (factor/interrobot.rs:40-77) */
    double safety_distance, tiny_offset;
    int ext_robot, ext_var;
    /* obstacle (factor/obstacle.rs:12-22,97-110) */
    double jac_delta;
    /* tracking (factor/tracking.rs:15-28,72-85) */
    int record;
    float last_pos[2];
    double last_value;
    int timeout; /* Option<usize> (tracking.rs:80,153-155): -1 = None */
} Factor;

typedef struct {
    int is_factor, alive;
    Variable v;
    Factor f;
} Node;

typedef struct {
    int64_t order_key;
    int K, ghost;
    double radius;
    Node *nodes;
    int n_nodes, cap_nodes;
    int *free_list; /* StableGraph vacant-slot reuse, LIFO */
    int n_free, cap_free;
    int *factor_indices; /* FG/factorgraph.rs: factor_indices (creation order, retained on delete) */
    int n_factors, cap_factors;
    int *ir_indices; /* interrobot_factor_indices: never pruned (factorgraph.rs:729-733) */
    int n_ir, cap_ir;
    int *var_indices;
    int iter_factor, iter_variable; /* iteration_count */
    int antenna, idle;
    int removed; /* entity despawned (ROBOT:2172): out of every query from now on */
    float *path;
    int n_path;
    int *connected; /* RobotConnections::robots_connected_with (ROBOT:515-531), sorted by order key */
    int n_connected, cap_connected;
} Graph;

typedef struct {
    orc_params p;
    Graph *g;
    int n, cap;
    uint8_t *sdf; /* RGB interleaved */
    uint32_t sdf_w, sdf_h;
    double world_w, world_h;
    int n_threads;
} World;

#define ORC_OK 0
#define ORC_ERR_INVALID (-1)

/* ---------------------------------------------------------------------------------------
 * small dense helpers, as ndarray 0.15.6 (Cargo.lock:4857, `blas` feature off, Cargo.toml:46-47)
 * evaluates them — third-party code absent from /root/reference, restated from its published
 * source; built -ffp-contract=off so that nothing contracts unless a flavour asks for it.
 *
 *  matvec  Array2.dot(&Array1) = general_mat_vec_mul: one `row.dot(x)` per row, and a dot of two
 *          contiguous slices is numeric_util::unrolled_dot: eight partial sums p0..p7 over chunks
 *          of 8, sum = 0 + (p0+p4) + (p1+p5) + (p2+p6) + (p3+p7), then the < 8 leftover products
 *          added one by one.  Rows of 4 (J^T L rhs, lam_ab lam_bb^-1 eta_b, cov eta, prior lam mu) are
 *          therefore plain left-to-right sums; the 8-long rows of J x0 (FG/factor/mod.rs:399, dynamic
 *          and inter-robot factors) are paired (x0 y0 + x4 y4) + (x1 y1 + x5 y5) + ...
 *  matmul  Array2.dot(&Array2) = matrixmultiply 0.3.8 dgemm (Cargo.lock:4686): every C[i][j] is a
 *          k-ascending accumulation from 0.  Its x86-64 kernels are chosen at run time; the FMA one
 *          fuses each step (flavour ORC_GEMM_FMA), the fallback does not (default).
 *
 * FLAVOURS (compile-time; tests/test_oracle_variants.py): the reference binary's rounding is
 * not reproducible here (no Rust toolchain, third-party sources absent), so the tolerance claim of
 * BASELINE.json is checked against every plausible arithmetic instead of one guess:
 *   ORC_GEMM_FMA   matmul steps fused (matrixmultiply's FMA kernel on a machine with FMA)
 *   ORC_INV_LU     4x4 inverse by LU with partial pivoting instead of cofactors (None iff a pivot is 0)
 *   ORC_DOT_SEQ    every dot product a plain left-to-right sum (no unrolled_dot pairing)
 * --------------------------------------------------------------------------------------- */
static void matmul(const double *A, const double *B, double *C, int m, int k, int n) {
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = 0.0;
#ifdef ORC_GEMM_FMA
            for (int l = 0; l < k; l++) s = fma(A[i * k + l], B[l * n + j], s);
#else
            for (int l = 0; l < k; l++) s += A[i * k + l] * B[l * n + j];
#endif
            C[i * n + j] = s;
        }
}
/* ndarray::numeric_util::unrolled_dot */
static double unrolled_dot(const double *x, const double *y, int n) {
#ifdef ORC_DOT_SEQ
    double s = 0.0;
    for (int l = 0; l < n; l++) s += x[l] * y[l];
    return s;
#else
    double p[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, sum = 0.0;
    while (n >= 8) {
        for (int l = 0; l < 8; l++) p[l] = p[l] + x[l] * y[l];
        x += 8;
        y += 8;
        n -= 8;
    }
    sum = sum + (p[0] + p[4]);
    sum = sum + (p[1] + p[5]);
    sum = sum + (p[2] + p[6]);
    sum = sum + (p[3] + p[7]);
    for (int l = 0; l < n; l++) sum = sum + x[l] * y[l];
    return sum;
#endif
}
static void matvec(const double *A, const double *x, double *y, int m, int n) {
    for (int i = 0; i < m; i++) y[i] = unrolled_dot(A + i * n, x, n);
}

/* ndarray-inverse 0.1.9 `Inverse::inv` for a 4x4 (Cargo.lock:4870).  Third-party, source absent
 * from /root/reference: restated from its published contract (`None` when the determinant is
 * exactly 0, cofactor / determinant inverse) as the plain cofactor expansion:
 *   C(i,j) = (-1)^(i+j) det3(rows != i, cols != j),
 *   det3([[a b c],[d e f],[g h k]]) = (a (e k - f h) - b (d k - f g)) + c (d h - e g),
 *   det = ((m00 C00 + m01 C01) + m02 C02) + m03 C03,  inverse[j][i] = C(i,j) * (1/det). */
static double det3(const double *r0, const double *r1, const double *r2, int j) {
    int c0 = (j == 0) ? 1 : 0, c1 = (j <= 1) ? 2 : 1, c2 = (j <= 2) ? 3 : 2;
    double a = r0[c0], b = r0[c1], c = r0[c2];
    double d = r1[c0], e = r1[c1], f = r1[c2];
    double g = r2[c0], h = r2[c1], k = r2[c2];
    return (a * (e * k - f * h) - b * (d * k - f * g)) + c * (d * h - e * g);
}
#ifdef ORC_INV_LU
/* flavour: Gauss-Jordan on [m | I] with partial pivoting; "None" iff a pivot column is all zeros */
static int inv4(const double *m, double *out) {
    double a[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            a[i][j] = m[i * 4 + j];
            a[i][4 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; c++) {
        int piv = c;
        for (int r = c + 1; r < 4; r++)
            if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        if (a[piv][c] == 0.0) return 0;
        if (piv != c)
            for (int j = 0; j < 8; j++) {
                double t = a[c][j];
                a[c][j] = a[piv][j];
                a[piv][j] = t;
            }
        double ip = 1.0 / a[c][c];
        for (int j = 0; j < 8; j++) a[c][j] *= ip;
        for (int r = 0; r < 4; r++) {
            if (r == c) continue;
            double f = a[r][c];
            if (f == 0.0) continue;
            for (int j = 0; j < 8; j++) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[i * 4 + j] = a[i][4 + j];
    return 1;
}
#else
static int inv4(const double *m, double *out) {
    double cf[4][4];
    for (int i = 0; i < 4; i++) {
        const double *rows[3];
        int n = 0;
        for (int r = 0; r < 4; r++)
            if (r != i) rows[n++] = m + 4 * r;
        for (int j = 0; j < 4; j++) {
            double mn = det3(rows[0], rows[1], rows[2], j);
            cf[i][j] = ((i + j) & 1) ? -mn : mn;
        }
    }
    double det = ((m[0] * cf[0][0] + m[1] * cf[0][1]) + m[2] * cf[0][2]) + m[3] * cf[0][3];
    if (det == 0.0) return 0;
    double id = 1.0 / det;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[j * 4 + i] = cf[i][j] * id;
    return 1;
}
#endif
int orc_inv4(const double *m, double *out) { return inv4(m, out); }

/* crates/gbp_linalg/src/lib.rs:47-93 */
static double euclidean_norm(const double *x, int n) {
    double acc = 0.0;
    for (int i = 0; i < n; i++) acc = acc + x[i] * x[i];
    return sqrt(acc);
}
double orc_euclidean_norm(const double *x, int n) { return euclidean_norm(x, n); }
double orc_l1_norm(const double *x, int n) {
    double acc = 0.0;
    for (int i = 0; i < n; i++) acc = acc + fabs(x[i]);
    return acc;
}
/* crates/gbp_linalg/src/lib.rs:113-128: unchanged when |x| is 0 or inf */
static void normalize(double *x, int n) {
    double mag = euclidean_norm(x, n);
    if (mag == 0.0 || isinf(mag)) return;
    for (int i = 0; i < n; i++) x[i] /= mag;
}
void orc_normalize(double *x, int n) { normalize(x, n); }

/* ---------------------------------------------------------------------------------------
 * FG/factor/marginalise_factor_distance.rs:55-127
 * --------------------------------------------------------------------------------------- */
static void marginalise_factor_distance(const double *eta, const double *lam, int dim, int marg_idx,
                                        Msg *out) {
    memset(out, 0, sizeof(Msg));
    if (dim == DOFS) { /* :62-72 */
        out->present = 1;
        memcpy(out->eta, eta, sizeof(double) * DOFS);
        memcpy(out->lam, lam, sizeof(double) * 16);
        return;
    }
    /* dim == 8: a = block at marg_idx, b = the other block (:74-108) */
    int a = marg_idx, b = (marg_idx == 0) ? DOFS : 0;
    double laa[16], lab[16], lba[16], lbb[16], ea[4], eb[4];
    for (int i = 0; i < 4; i++) {
        ea[i] = eta[a + i];
        eb[i] = eta[b + i];
        for (int j = 0; j < 4; j++) {
            laa[i * 4 + j] = lam[(a + i) * dim + (a + j)];
            lab[i * 4 + j] = lam[(a + i) * dim + (b + j)];
            lba[i * 4 + j] = lam[(b + i) * dim + (a + j)];
            lbb[i * 4 + j] = lam[(b + i) * dim + (b + j)];
        }
    }
    double lbb_inv[16];
    if (!inv4(lbb, lbb_inv)) return; /* :79-81 => Message::empty() */
    double t[16], tv[4], tm[16];
    matmul(lab, lbb_inv, t, 4, 4, 4);
    matvec(t, eb, tv, 4, 4);
    matmul(t, lba, tm, 4, 4, 4);
    for (int i = 0; i < 4; i++) out->eta[i] = ea[i] - tv[i];      /* :114 */
    for (int i = 0; i < 16; i++) out->lam[i] = laa[i] - tm[i];    /* :115 */
    for (int i = 0; i < 16; i++)
        if (isinf(out->lam[i])) { /* :117-118 (NaN does not trigger) */
            memset(out, 0, sizeof(Msg));
            return;
        }
    out->present = 1; /* mean = zeros (:120) */
}
/* exported for known-answer tests */
int orc_marginalise(const double *eta, const double *lam, int dim, int marg_idx, double *o_eta,
                    double *o_lam, double *o_mu) {
    Msg m;
    marginalise_factor_distance(eta, lam, dim, marg_idx, &m);
    if (!m.present) return 0;
    memcpy(o_eta, m.eta, sizeof m.eta);
    memcpy(o_lam, m.lam, sizeof m.lam);
    memcpy(o_mu, m.mu, sizeof m.mu);
    return 1;
}

/* ---------------------------------------------------------------------------------------
 * measurement functions
 * --------------------------------------------------------------------------------------- */
/* Rust `as u32` on f64: saturating, NaN -> 0 */
static uint32_t sat_u32(double v) {
    if (!(v > 0.0)) return 0u; /* negatives, -0, NaN */
    if (v >= 4294967295.0) return 4294967295u;
    return (uint32_t)v;
}

/* FG/factor/obstacle.rs:141-188 */
static double obstacle_measure(const World *w, const double *x) {
    double x_offset = w->world_w / 2.0, y_offset = w->world_h / 2.0;
    double x_scale = (double)w->sdf_w / w->world_w;
    double y_scale = (double)w->sdf_h / w->world_h;
    uint32_t xp = sat_u32((x[0] + x_offset) * x_scale);
    uint32_t yp = sat_u32((-x[1] + y_offset) * y_scale);
    if (!(xp < w->sdf_w && yp < w->sdf_h)) return 0.0; /* get_pixel_checked -> None (:169-176) */
    uint8_t red = w->sdf[((size_t)yp * w->sdf_w + xp) * 3];
    return 1.0 - (double)red / 255.0;
}
double orc_obstacle_measure(World *w, const double *x) { return obstacle_measure(w, x); }

/* FG/factor/interrobot.rs:91-106 */
static void ir_diff(const Factor *f, const double *x, double *d) {
    for (int i = 0; i < 2; i++) d[i] = (x[i] - x[DOFS + i]) + f->tiny_offset;
}

/* FG/factor/tracking.rs:197-346. Mutates record / last_measurement like the Mutex<Cell>s. */
static double tracking_measure(const World *w, Factor *f, const Graph *g, const double *x) {
    int rec = f->record;
    const float *p = g->path;
    double xp[2] = {x[0], x[1]}, xv[2] = {x[2], x[3]};
    double cs[2] = {(double)p[2 * rec], (double)p[2 * rec + 1]};
    double ce[2] = {(double)p[2 * rec + 2], (double)p[2 * rec + 3]};
    double line[2] = {ce[0] - cs[0], ce[1] - cs[1]};
    double d0[2] = {xp[0] - cs[0], xp[1] - cs[1]};
    double tt = (d0[0] * line[0] + d0[1] * line[1]) / (line[0] * line[0] + line[1] * line[1]);
    double cur[2] = {cs[0] + tt * line[0], cs[1] + tt * line[1]};
    double d = w->p.tracking_switch_padding;
    double cd0 = d, cd1 = d * 0.01; /* :231-244 */
    double e2c[2] = {ce[0] - cur[0], ce[1] - cur[1]};
    double dist_end = euclidean_norm(e2c, 2);
    int use_prev = 0;
    double prevp[2] = {0, 0};
    if (rec > 0) { /* :255-286 */
        double ps[2] = {(double)p[2 * (rec - 1)], (double)p[2 * (rec - 1) + 1]};
        double pe[2] = {(double)p[2 * rec], (double)p[2 * rec + 1]};
        double pl[2] = {pe[0] - ps[0], pe[1] - ps[1]};
        double q0[2] = {xp[0] - ps[0], xp[1] - ps[1]};
        double t2 = (q0[0] * pl[0] + q0[1] * pl[1]) / (pl[0] * pl[0] + pl[1] * pl[1]);
        prevp[0] = ps[0] + t2 * pl[0];
        prevp[1] = ps[1] + t2 * pl[1];
        double a_[2] = {pe[0] - cur[0], pe[1] - cur[1]};
        double b_[2] = {cs[0] - prevp[0], cs[1] - prevp[1]};
        double a = euclidean_norm(a_, 2), b = euclidean_norm(b_, 2);
        use_prev = (a < cd0 && a > cd1 && b < cd0);
    }
    if (dist_end < cd0) { /* :294-296, increment_record :54-64 */
        int nr = rec + 1;
        if (nr > g->n_path - 2) nr = g->n_path - 2;
        f->record = nr;
    }
    double mp[2];
    if (use_prev) { /* :300-311 */
        double x2c[2] = {cur[0] - xp[0], cur[1] - xp[1]};
        double x2p[2] = {prevp[0] - xp[0], prevp[1] - xp[1]};
        mp[0] = xp[0] + (x2c[0] + x2p[0]);
        mp[1] = xp[1] + (x2c[1] + x2p[1]);
    } else { /* :312-316 */
        double ln[2] = {line[0], line[1]};
        normalize(ln, 2);
        double vn = euclidean_norm(xv, 2);
        mp[0] = cur[0] + ln[0] * vn / 5.0;
        mp[1] = cur[1] + ln[1] * vn / 5.0;
    }
    double x2m[2] = {mp[0] - xp[0], mp[1] - xp[1]};
    double dist = euclidean_norm(x2m, 2);
    double ad = w->p.tracking_attraction_distance;
    double meas = (dist < ad) ? dist / ad : 1.0; /* :322-333 */
    f->last_pos[0] = (float)mp[0];                /* :336-339 (f32 Vec2) */
    f->last_pos[1] = (float)mp[1];
    f->last_value = meas;
    return meas;
}

/* `Factor::skip` per kind */
static int factor_skip(const Graph *g, Factor *f) {
    if (f->kind == K_INTERROBOT) { /* interrobot.rs:213-226 (no tiny offset) */
        double dx = f->x0[0] - f->x0[DOFS], dy = f->x0[1] - f->x0[DOFS + 1];
        double sq = dx * dx + dy * dy; /* mapv(powi(2)).sum() */
        return sq >= f->safety_distance * f->safety_distance;
    }
    if (f->kind == K_TRACKING) { /* tracking.rs:362-381 */
        if (f->timeout >= 0) { /* :363-371: Some(0) -> None and go on; Some(n) -> Some(n - 1) and skip */
            if (f->timeout == 0) f->timeout = -1;
            else { f->timeout -= 1; return 1; }
        }
        int len = g->n_path;
        if (len < 2) return 1;
        return f->record >= len - 1;
    }
    return 0;
}

/* measure(): h has zdim entries */
static void factor_measure(const World *w, Graph *g, Factor *f, const double *x, double *h) {
    switch (f->kind) {
    case K_DYNAMIC: matvec(f->J_dyn, x, h, 4, 8); break; /* dynamic.rs:72-75 */
    case K_INTERROBOT: {                                 /* interrobot.rs:165-204 */
        for (int i = 0; i < f->zdim; i++) h[i] = 0.0;
        double d[2];
        ir_diff(f, x, d);
        double r = euclidean_norm(d, 2);
        if (r <= f->safety_distance) h[0] = 1.0 * (1.0 - r / f->safety_distance);
        break;
    }
    case K_OBSTACLE: h[0] = obstacle_measure(w, x); break;
    case K_TRACKING: h[0] = tracking_measure(w, f, g, x); break;
    }
}

/* jacobian(): J is zdim x (4*nvars), row-major */
static void factor_jacobian(const World *w, Graph *g, Factor *f, const double *x, double *J) {
    int n = DOFS * f->nvars;
    switch (f->kind) {
    case K_DYNAMIC: memcpy(J, f->J_dyn, sizeof(double) * 32); break; /* dynamic.rs:68-70 */
    case K_INTERROBOT: {                                             /* interrobot.rs:121-161 */
        for (int i = 0; i < f->zdim * n; i++) J[i] = 0.0;
        double d[2];
        ir_diff(f, x, d);
        double r = euclidean_norm(d, 2);
        if (r <= f->safety_distance) {
            double c0 = -1.0 / f->safety_distance / r;
            double c1 = 1.0 / f->safety_distance / r;
            for (int i = 0; i < 2; i++) {
                J[i] = c0 * d[i];
                J[DOFS + i] = c1 * d[i];
            }
        }
        break;
    }
    case K_OBSTACLE: { /* Factor::first_order_jacobian, FG/factor/mod.rs:102-128 */
        double xx[4] = {x[0], x[1], x[2], x[3]};
        double h0 = obstacle_measure(w, xx);
        double delta = f->jac_delta;
        for (int i = 0; i < n; i++) {
            xx[i] += delta;
            double h1 = obstacle_measure(w, xx);
            J[i] = (h1 - h0) / delta;
            xx[i] -= delta;
        }
        break;
    }
    case K_TRACKING: { /* tracking.rs:171-194: uses the state measure() just stored */
        double inv_h0 = 1.0 / f->last_value;
        J[0] = inv_h0 * (x[0] - (double)f->last_pos[0]);
        J[1] = inv_h0 * (x[1] - (double)f->last_pos[1]);
        J[2] = 0.0;
        J[3] = 0.0;
        break;
    }
    }
}

/* ---------------------------------------------------------------------------------------
 * FactorNode::update — FG/factor/mod.rs:334-454.  out[j] = message for inbox entry j.
 * --------------------------------------------------------------------------------------- */
static void factor_update(const World *w, Graph *g, Factor *f, Msg *out) {
    int nv = f->inbox.n, n = DOFS * f->nvars, m = f->zdim;
    for (int j = 0; j < nv; j++) f->cnt[f->inbox.e[j].key.robot != f->robot] += 1; /* :353-367,410-452: one message per key, skipped or not */
    /* :336-349 linearisation point from inbox means (empty => zeros) */
    for (int j = 0; j < nv; j++)
        for (int i = 0; i < DOFS; i++)
            f->x0[j * DOFS + i] = f->inbox.e[j].msg.present ? f->inbox.e[j].msg.mu[i] : 0.0;
    if (factor_skip(g, f)) { /* :352-369 */
        for (int j = 0; j < nv; j++) memset(&out[j], 0, sizeof(Msg));
        return;
    }
    double h[4], J[4 * 8];
    factor_measure(w, g, f, f->x0, h);  /* :374-377 measure BEFORE jacobian */
    factor_jacobian(w, g, f, f->x0, J); /* :388 */
    /* :391-401 */
    double Jt[8 * 4], JtL[8 * 4], lam_p[64], Jx[4], rhs[4], eta_p[8];
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) Jt[j * m + i] = J[i * n + j];
    matmul(Jt, f->lam_meas, JtL, n, m, m);
    matmul(JtL, J, lam_p, n, m, n);
    matvec(J, f->x0, Jx, m, n);
    for (int i = 0; i < m; i++) rhs[i] = Jx[i] + (f->z[i] - h[i]);
    matvec(JtL, rhs, eta_p, n, m);
    /* :406-450 */
    int marg_idx = 0;
    for (int k = 0; k < nv; k++) {
        double eta[8], lam[64];
        memcpy(eta, eta_p, sizeof(double) * (size_t)n);
        memcpy(lam, lam_p, sizeof(double) * (size_t)(n * n));
        for (int j = 0; j < nv; j++) {
            if (j == k) continue;
            const Msg *om = &f->inbox.e[j].msg;
            if (!om->present) continue;
            for (int i = 0; i < DOFS; i++) eta[j * DOFS + i] += om->eta[i];
            for (int r = 0; r < DOFS; r++)
                for (int c = 0; c < DOFS; c++) lam[(j * DOFS + r) * n + (j * DOFS + c)] += om->lam[r * 4 + c];
        }
        marginalise_factor_distance(eta, lam, n, marg_idx, &out[k]);
        marg_idx += DOFS;
    }
}

/* FactorNode::receive_message_from — FG/factor/mod.rs:307-317 */
static void factor_receive(Factor *f, NodeId from, const Msg *m) {
    if (!f->enabled) return;
    *inbox_insert(&f->inbox, from) = *m;
    f->cnt[2 + (from.robot != f->robot)] += 1; /* :312-316 */
}
/* VariableNode::receive_message_from — FG/variable.rs:179-191 */
static void variable_receive(Variable *v, NodeId from, const Msg *m) {
    *inbox_insert(&v->inbox, from) = *m;
    v->cnt[2 + (from.robot != v->robot)] += 1; /* :185-189 */
}

/* VariableNode::prepare_message — FG/variable.rs:234-240 */
static void variable_prepare_message(const Variable *v, Msg *m) {
    m->present = 1;
    memcpy(m->eta, v->eta, sizeof m->eta);
    memcpy(m->lam, v->lam, sizeof m->lam);
    memcpy(m->mu, v->mu, sizeof m->mu);
}

/* VariableNode::new — FG/variable.rs:140-166 */
static void variable_new(Variable *v, const double *mean, double prior_diag) {
    memset(v, 0, sizeof *v);
    double lam[16] = {0};
    for (int i = 0; i < 4; i++) lam[i * 5] = prior_diag;
    int finite = 1;
    for (int i = 0; i < 16; i++)
        if (!isfinite(lam[i])) finite = 0;
    if (!finite) memset(lam, 0, sizeof lam); /* :146-148 */
    matvec(lam, mean, v->prior_eta, 4, 4);   /* :150 */
    memcpy(v->prior_lam, lam, sizeof lam);
    if (!inv4(lam, v->cov)) memset(v->cov, 0, sizeof v->cov); /* :152-154 */
    memcpy(v->eta, v->prior_eta, sizeof v->eta);
    memcpy(v->lam, lam, sizeof lam);
    memcpy(v->mu, mean, sizeof v->mu);
    v->valid = 1;
    for (int i = 0; i < 16; i++)
        if (!isfinite(v->cov[i])) v->valid = 0;
}

/* VariableNode::update_belief_and_create_factor_responses — FG/variable.rs:251-342.
 * out[k] = response for inbox entry k. */
static void variable_update(Variable *v, Msg *out) {
    for (int k = 0; k < v->inbox.n; k++) v->cnt[v->inbox.e[k].key.robot != v->robot] += 1; /* :299-332, one response per key */
    memcpy(v->eta, v->prior_eta, sizeof v->eta);
    memcpy(v->lam, v->prior_lam, sizeof v->lam);
    for (int k = 0; k < v->inbox.n; k++) { /* :263-271 */
        const Msg *m = &v->inbox.e[k].msg;
        if (!m->present) continue;
        for (int i = 0; i < 4; i++) v->eta[i] = v->eta[i] + m->eta[i];
        for (int i = 0; i < 16; i++) v->lam[i] = v->lam[i] + m->lam[i];
    }
    int not_zero = 0; /* :276 */
    for (int i = 0; i < 16; i++)
        if (v->lam[i] - 1e-6 > 0.0) not_zero = 1;
    if (not_zero) {
        double cov[16];
        if (inv4(v->lam, cov)) { /* :278 */
            memcpy(v->cov, cov, sizeof cov);
            v->valid = 1;
            for (int i = 0; i < 16; i++)
                if (!isfinite(cov[i])) v->valid = 0;
            if (v->valid) matvec(v->cov, v->eta, v->mu, 4, 4); /* :281-285 */
        }
    }
    for (int k = 0; k < v->inbox.n; k++) { /* :301-330 */
        const Msg *m = &v->inbox.e[k].msg;
        if (!m->present) {
            variable_prepare_message(v, &out[k]);
        } else {
            out[k].present = 1;
            for (int i = 0; i < 4; i++) out[k].eta[i] = v->eta[i] - m->eta[i];
            for (int i = 0; i < 16; i++) out[k].lam[i] = v->lam[i] - m->lam[i];
            for (int i = 0; i < 4; i++) out[k].mu[i] = v->mu[i] - m->mu[i];
        }
    }
}

/* ---------------------------------------------------------------------------------------
 * graph container — FG/factorgraph.rs
 * --------------------------------------------------------------------------------------- */
static int graph_add_node(Graph *g) { /* petgraph StableGraph::add_node */
    int ix;
    if (g->n_free > 0) {
        ix = g->free_list[--g->n_free];
    } else {
        if (g->n_nodes == g->cap_nodes) {
            g->cap_nodes = g->cap_nodes ? 2 * g->cap_nodes : 64;
            g->nodes = (Node *)realloc(g->nodes, sizeof(Node) * (size_t)g->cap_nodes);
        }
        ix = g->n_nodes++;
    }
    memset(&g->nodes[ix], 0, sizeof(Node));
    g->nodes[ix].alive = 1;
    return ix;
}
static void push_int(int **a, int *n, int *cap, int v) {
    if (*n == *cap) {
        *cap = *cap ? 2 * *cap : 16;
        *a = (int *)realloc(*a, sizeof(int) * (size_t)*cap);
    }
    (*a)[(*n)++] = v;
}
static NodeId mkid(const World *w, int robot, int index) {
    NodeId id = {(int64_t)w->g[robot].order_key, robot, index};
    return id;
}

/* FactorState::new — FG/factor/mod.rs:627-643 */
static void factor_state_new(Factor *f, int kind, int zdim, double strength, int nvars, int enabled) {
    memset(f, 0, sizeof *f);
    f->kind = kind;
    f->enabled = enabled;
    f->zdim = zdim;
    f->nvars = nvars;
    f->timeout = -1;
    double s2 = strength * strength; /* powi(strength, 2) */
    for (int i = 0; i < zdim; i++) f->lam_meas[i * zdim + i] = 1.0 / s2;
}

/* add_internal_edge — FG/factorgraph.rs:304-330 */
static void add_internal_edge(World *w, int robot, int var_ix, int fac_ix) {
    Graph *g = &w->g[robot];
    Variable *v = &g->nodes[var_ix].v;
    Factor *f = &g->nodes[fac_ix].f;
    Msg empty;
    memset(&empty, 0, sizeof empty);
    variable_receive(v, mkid(w, robot, fac_ix), &empty);
    if (f->kind == K_TRACKING) {
        Msg m;
        variable_prepare_message(v, &m);
        factor_receive(f, mkid(w, robot, var_ix), &m);
    } else {
        factor_receive(f, mkid(w, robot, var_ix), &empty);
    }
}

World *orc_world_create(const orc_params *p) {
    World *w = (World *)calloc(1, sizeof(World));
    w->p = *p;
    w->n_threads = 1;
    return w;
}
void orc_set_threads(World *w, int n) { w->n_threads = n < 1 ? 1 : n; }

static void graph_free(Graph *g) {
    for (int i = 0; i < g->n_nodes; i++) {
        free(g->nodes[i].v.inbox.e);
        free(g->nodes[i].f.inbox.e);
    }
    free(g->nodes);
    free(g->free_list);
    free(g->factor_indices);
    free(g->ir_indices);
    free(g->var_indices);
    free(g->path);
    free(g->connected);
}
void orc_world_destroy(World *w) {
    if (!w) return;
    for (int i = 0; i < w->n; i++) graph_free(&w->g[i]);
    free(w->g);
    free(w->sdf);
    free(w);
}

int orc_world_set_sdf(World *w, const uint8_t *rgb, uint32_t width, uint32_t height, double world_w,
                      double world_h) {
    if (!w || !rgb || !width || !height) return ORC_ERR_INVALID;
    free(w->sdf);
    w->sdf = (uint8_t *)malloc((size_t)width * height * 3);
    memcpy(w->sdf, rgb, (size_t)width * height * 3);
    w->sdf_w = width;
    w->sdf_h = height;
    w->world_w = world_w;
    w->world_h = world_h;
    return ORC_OK;
}

/* RobotBundle::new — ROBOT:1134-1356 */
int orc_robot_add(World *w, const orc_robot_desc *d, int32_t *robot_id) {
    if (!w || !d || d->K < 2 || !d->mean0 || !d->prior_diag || !d->dt) return ORC_ERR_INVALID;
    if (w->n == w->cap) {
        w->cap = w->cap ? 2 * w->cap : 16;
        w->g = (Graph *)realloc(w->g, sizeof(Graph) * (size_t)w->cap);
    }
    int r = w->n++;
    Graph *g = &w->g[r];
    memset(g, 0, sizeof *g);
    g->order_key = (int64_t)d->order_key;
    g->K = (int)d->K;
    g->ghost = (int)d->ghost;
    g->radius = d->radius;
    g->antenna = 1;
    if (d->n_path && d->path_xy) {
        g->n_path = (int)d->n_path;
        g->path = (float *)malloc(sizeof(float) * 2 * d->n_path);
        memcpy(g->path, d->path_xy, sizeof(float) * 2 * d->n_path);
    }
    int K = g->K;
    g->var_indices = (int *)malloc(sizeof(int) * (size_t)K);
    for (int i = 0; i < K; i++) { /* :1179-1223 */
        int ix = graph_add_node(g);
        g->nodes[ix].is_factor = 0;
        variable_new(&g->nodes[ix].v, d->mean0 + 4 * i, d->prior_diag[i]);
        g->nodes[ix].v.robot = r;
        g->var_indices[i] = ix;
    }
    for (int i = 0; i < K - 1; i++) { /* dynamic factors :1225-1255, dynamic.rs:22-52 */
        int ix = graph_add_node(g);
        g->nodes[ix].is_factor = 1;
        Factor *f = &g->nodes[ix].f;
        factor_state_new(f, K_DYNAMIC, 4, w->p.sigma_dynamics, 2, (w->p.enable_mask & EN_DYN) != 0);
        f->robot = r;
        double dt = d->dt[i];
        double qc = 1.0 / (w->p.sigma_dynamics * w->p.sigma_dynamics); /* powi(strength,-2) */
        double q11 = 12.0 * (1.0 / (dt * dt * dt)) * qc;                /* powi(dt,-3) */
        double q12 = -6.0 * (1.0 / (dt * dt)) * qc;
        double q22 = (4.0 / dt) * qc;
        memset(f->lam_meas, 0, sizeof f->lam_meas);
        for (int a = 0; a < 2; a++) {
            f->lam_meas[a * 4 + a] = q11;
            f->lam_meas[a * 4 + (a + 2)] = q12;
            f->lam_meas[(a + 2) * 4 + a] = q12;
            f->lam_meas[(a + 2) * 4 + (a + 2)] = q22;
        }
        memset(f->J_dyn, 0, sizeof f->J_dyn);
        for (int a = 0; a < 2; a++) {
            f->J_dyn[a * 8 + a] = 1.0;
            f->J_dyn[a * 8 + (a + 2)] = dt;
            f->J_dyn[a * 8 + (a + 4)] = -1.0;
            f->J_dyn[(a + 2) * 8 + (a + 2)] = 1.0;
            f->J_dyn[(a + 2) * 8 + (a + 6)] = -1.0;
        }
        push_int(&g->factor_indices, &g->n_factors, &g->cap_factors, ix);
        add_internal_edge(w, r, g->var_indices[i + 1], ix); /* :1245-1253 */
        add_internal_edge(w, r, g->var_indices[i], ix);
    }
    for (int i = 1; i < K - 1; i++) { /* obstacle factors :1269-1285 */
        int ix = graph_add_node(g);
        g->nodes[ix].is_factor = 1;
        Factor *f = &g->nodes[ix].f;
        factor_state_new(f, K_OBSTACLE, 1, w->p.sigma_obstacle, 1, (w->p.enable_mask & EN_OBS) != 0);
        f->robot = r;
        push_int(&g->factor_indices, &g->n_factors, &g->cap_factors, ix);
        add_internal_edge(w, r, g->var_indices[i], ix);
    }
    for (int i = 1; i < K - 1; i++) { /* tracking factors :1305-1334, FG/factor/mod.rs:256-277 */
        int ix = graph_add_node(g);
        g->nodes[ix].is_factor = 1;
        Factor *f = &g->nodes[ix].f;
        factor_state_new(f, K_TRACKING, 1, w->p.sigma_tracking, 1, (w->p.enable_mask & EN_TRK) != 0);
        f->robot = r;
        f->x0[0] = d->mean0[4 * i];
        f->x0[1] = d->mean0[4 * i + 1];
        f->last_pos[0] = (float)d->mean0[4 * i];
        f->last_pos[1] = (float)d->mean0[4 * i + 1];
        f->last_value = 0.0;
        push_int(&g->factor_indices, &g->n_factors, &g->cap_factors, ix);
        add_internal_edge(w, r, g->var_indices[i], ix);
    }
    if (robot_id) *robot_id = r;
    return ORC_OK;
}

/* create_interrobot_factors, one direction — ROBOT:1500-1585 */
static int ir_connect(World *w, int32_t owner, int32_t other, uint64_t first_robot_number) {
    if (!w || owner < 0 || other < 0 || owner >= w->n || other >= w->n || owner == other)
        return ORC_ERR_INVALID;
    Graph *g = &w->g[owner];
    Graph *o = &w->g[other];
    if (g->K != o->K) return ORC_ERR_INVALID;
    /* obstacle jacobian delta is set lazily (needs the image); see set below */
    for (int i = 1; i < g->K; i++) {
        int ix = graph_add_node(g);
        g = &w->g[owner];
        g->nodes[ix].is_factor = 1;
        Factor *f = &g->nodes[ix].f;
        factor_state_new(f, K_INTERROBOT, 4, w->p.sigma_interrobot, 2, (w->p.enable_mask & EN_IR) != 0);
        f->robot = owner;
        f->safety_distance = w->p.safety_multiplier * g->radius; /* interrobot.rs:64 */
        f->tiny_offset = (double)1e-6f * (double)(first_robot_number + (uint64_t)(i - 1)); /* :75 */
        f->ext_robot = other;
        f->ext_var = o->var_indices[i];
        push_int(&g->factor_indices, &g->n_factors, &g->cap_factors, ix);
        push_int(&g->ir_indices, &g->n_ir, &g->cap_ir, ix);
        add_internal_edge(w, owner, g->var_indices[i], ix); /* :1538 */
        /* add_external_edge on the other graph — FG/factorgraph.rs:340-353 */
        Variable *ov = &o->nodes[o->var_indices[i]].v;
        Msg empty;
        memset(&empty, 0, sizeof empty);
        variable_receive(ov, mkid(w, owner, ix), &empty);
        /* ROBOT:1557-1578: the other variable's current belief goes into the new factor */
        Msg m;
        variable_prepare_message(ov, &m);
        factor_receive(f, mkid(w, other, o->var_indices[i]), &m);
    }
    return ORC_OK;
}

/* delete_interrobot_factors_connected_to — FG/factorgraph.rs:380-436 */
static void delete_ir_connected_to(World *w, int self, int other) {
    Graph *g = &w->g[self];
    int *removed = NULL, n_removed = 0, cap_removed = 0;
    for (int ix = 0; ix < g->n_nodes; ix++) {
        Node *nd = &g->nodes[ix];
        if (!nd->alive) continue;
        if (!nd->is_factor) {
            inbox_remove_if_graph(&nd->v.inbox, other);
            continue;
        }
        if (nd->f.kind != K_INTERROBOT || nd->f.ext_robot != other) continue;
        free(nd->f.inbox.e);
        memset(&nd->f, 0, sizeof nd->f);
        nd->alive = 0;
        push_int(&g->free_list, &g->n_free, &g->cap_free, ix);
        int j = 0;
        for (int i = 0; i < g->n_factors; i++)
            if (g->factor_indices[i] != ix) g->factor_indices[j++] = g->factor_indices[i];
        g->n_factors = j;
        push_int(&removed, &n_removed, &cap_removed, ix);
    }
    for (int k = 0; k < g->K; k++)
        for (int i = 0; i < n_removed; i++)
            inbox_remove_key(&g->nodes[g->var_indices[k]].v.inbox, mkid(w, self, removed[i]));
    free(removed);
}
/* delete_interrobot_factors — ROBOT:1386-1439 */
static int ir_disconnect(World *w, int32_t a, int32_t b) {
    if (!w || a < 0 || b < 0 || a >= w->n || b >= w->n || a == b) return ORC_ERR_INVALID;
    delete_ir_connected_to(w, a, b);
    delete_ir_connected_to(w, b, a);
    return ORC_OK;
}

/* ---------------------------------------------------------------------------------------
 * dynamic inter-robot topology — the three FixedUpdate systems of ROBOT:1362-1586
 * --------------------------------------------------------------------------------------- */
/* update_robot_neighbours — ROBOT:1362-1384.  `pos` = Transform::translation of every robot
 * (f32 xyz).  j is within range of i unless `radius < distance(i, j)` (so a NaN distance IS in
 * range); glam Vec3::distance = sqrt((dx*dx + dy*dy) + dz*dz) in f32.  Result = BTreeSet per
 * robot, i.e. ascending Entity order (= order_key), as CSR. */
static int within_range(const float *a, const float *b, float radius) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    float d = sqrtf((dx * dx + dy * dy) + dz * dz);
    return !(radius < d);
}
static void sort_by_key(const World *w, int *a, int n) { /* BTreeSet<Entity> order */
    for (int i = 1; i < n; i++) {
        int v = a[i], j = i;
        while (j > 0 && w->g[a[j - 1]].order_key > w->g[v].order_key) { a[j] = a[j - 1]; j--; }
        a[j] = v;
    }
}
int orc_neighbours(World *w, const float *pos, float radius, int32_t *ptr, int32_t *idx, uint64_t capacity,
                   uint64_t *needed) {
    if (!w || !pos || !ptr) return ORC_ERR_INVALID;
    for (int r = 0; r < w->n; r++)
        if (w->g[r].ghost) return ORC_ERR_INVALID;
    uint64_t cnt = 0;
    int *row = (int *)malloc(sizeof(int) * (size_t)(w->n > 0 ? w->n : 1));
    for (int i = 0; i < w->n; i++) {
        ptr[i] = (int32_t)cnt;
        int m = 0;
        for (int j = 0; j < w->n && !w->g[i].removed; j++)
            if (j != i && !w->g[j].removed && within_range(pos + 3 * i, pos + 3 * j, radius)) row[m++] = j;
        sort_by_key(w, row, m);
        for (int k = 0; k < m; k++, cnt++)
            if (idx && cnt < capacity) idx[cnt] = row[k];
    }
    ptr[w->n] = (int32_t)cnt;
    free(row);
    if (needed) *needed = cnt;
    return (idx && cnt <= capacity) || !idx ? ORC_OK : ORC_ERR_INVALID;
}

static int set_contains(const int *a, int n, int v) {
    for (int i = 0; i < n; i++)
        if (a[i] == v) return 1;
    return 0;
}
int orc_connections(World *w, int32_t r, int32_t *out, uint32_t capacity, uint32_t *n) {
    if (!w || r < 0 || r >= w->n || !n) return ORC_ERR_INVALID;
    Graph *g = &w->g[r];
    *n = (uint32_t)g->n_connected;
    if ((uint32_t)g->n_connected > capacity) return out ? ORC_ERR_INVALID : ORC_OK;
    for (int i = 0; i < g->n_connected; i++) out[i] = g->connected[i];
    return ORC_OK;
}
/* public forms keep robots_connected_with in step (ROBOT:1406-1408,1546) */
int orc_ir_connect(World *w, int32_t owner, int32_t other, uint64_t first_robot_number) {
    int rc = ir_connect(w, owner, other, first_robot_number);
    if (rc != ORC_OK) return rc;
    Graph *g = &w->g[owner];
    if (!set_contains(g->connected, g->n_connected, other)) {
        push_int(&g->connected, &g->n_connected, &g->cap_connected, other);
        sort_by_key(w, g->connected, g->n_connected);
    }
    return ORC_OK;
}
static void set_remove(int *a, int *n, int v) {
    int k = 0;
    for (int i = 0; i < *n; i++)
        if (a[i] != v) a[k++] = a[i];
    *n = k;
}
int orc_ir_disconnect(World *w, int32_t a, int32_t b) {
    int rc = ir_disconnect(w, a, b);
    if (rc != ORC_OK) return rc;
    set_remove(w->g[a].connected, &w->g[a].n_connected, b);
    set_remove(w->g[b].connected, &w->g[b].n_connected, a);
    return ORC_OK;
}

/* update_robot_neighbours + delete_interrobot_factors + create_interrobot_factors in the order
 * the schedule runs them (ROBOT:86-99).  Query iteration = robot id order (spawn order).
 * `next_number` is RobotNumberGenerator (ROBOT:122-140): one number per created factor.
 * stats[0] = connections (one direction, K-1 factors each) created, stats[1] = robot pairs whose
 * factors were deleted. */
int orc_update_topology(World *w, const float *pos, float radius, uint64_t *next_number, uint32_t *stats) {
    if (!w || !pos || !next_number || *next_number == 0) return ORC_ERR_INVALID;
    const int n = w->n;
    int32_t *ptr = (int32_t *)malloc(sizeof(int32_t) * ((size_t)(n > 0 ? n : 0) + 1));
    uint64_t need = 0;
    int rc = orc_neighbours(w, pos, radius, ptr, NULL, 0, &need);
    if (rc != ORC_OK) { free(ptr); return rc; }
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(need ? need : 1));
    orc_neighbours(w, pos, radius, ptr, idx, need, &need);
    uint32_t created = 0, deleted = 0;

    /* delete_interrobot_factors — ROBOT:1386-1439.  The (robot -> other) pairs go through a
     * HashMap<RobotId, RobotId> (:1391,1400-1404): `extend` keeps only the LAST pair per robot,
     * i.e. the largest out-of-range id; the others are dropped from robots_connected_with
     * (:1406-1408) but their factors stay unless the other side's entry names this robot.
     * HashMap iteration order is unspecified in the reference; ascending robot id here. */
    int *victim = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int r = 0; r < n; r++) {
        Graph *g = &w->g[r];
        victim[r] = -1;
        if (g->removed) continue; /* not in the query any more */
        int keep = 0;
        for (int k = 0; k < g->n_connected; k++) {
            int o = g->connected[k];
            if (set_contains(idx + ptr[r], ptr[r + 1] - ptr[r], o)) g->connected[keep++] = o;
            else victim[r] = o; /* ascending key order: the last one wins */
        }
        g->n_connected = keep;
    }
    for (int r = 0; r < n; r++)
        if (victim[r] >= 0) {
            ir_disconnect(w, r, victim[r]);
            deleted++;
        }
    free(victim);

    /* create_interrobot_factors — ROBOT:1441-1586: new = within_range \ connected_with,
     * ascending; robots in query order; K-1 numbers drawn per connection (:1527) */
    int **fresh = (int **)calloc((size_t)(n > 0 ? n : 1), sizeof(int *));
    int *n_fresh = (int *)calloc((size_t)(n > 0 ? n : 1), sizeof(int));
    for (int r = 0; r < n; r++) { /* :1449-1461 snapshot before any insertion */
        Graph *g = &w->g[r];
        int m = ptr[r + 1] - ptr[r];
        fresh[r] = (int *)malloc(sizeof(int) * (size_t)(m ? m : 1));
        for (int k = 0; k < m; k++) {
            int o = idx[ptr[r] + k];
            if (!set_contains(g->connected, g->n_connected, o)) fresh[r][n_fresh[r]++] = o;
        }
    }
    for (int r = 0; r < n && rc == ORC_OK; r++) {
        for (int k = 0; k < n_fresh[r]; k++) {
            int o = fresh[r][k];
            rc = ir_connect(w, r, o, *next_number);
            if (rc != ORC_OK) break;
            *next_number += (uint64_t)(w->g[r].K - 1);
            Graph *g = &w->g[r];
            push_int(&g->connected, &g->n_connected, &g->cap_connected, o); /* :1546 */
            sort_by_key(w, g->connected, g->n_connected);
            created++;
        }
    }
    for (int r = 0; r < n; r++) free(fresh[r]);
    free(fresh);
    free(n_fresh);
    free(ptr);
    free(idx);
    if (stats) { stats[0] = created; stats[1] = deleted; }
    return rc;
}

/* Entity despawn (ROBOT:2172, despawn_entity_after): the graph leaves every query — it is never
 * iterated again and messages addressed to it are dropped (`query.get_mut` fails, ROBOT:1815,1844);
 * the other robots drop their factors towards it through the next topology passes. */
int orc_robot_remove(World *w, int32_t r) {
    if (!w || r < 0 || r >= w->n || w->g[r].removed) return ORC_ERR_INVALID;
    w->g[r].removed = 1;
    w->g[r].idle = 1;
    w->g[r].antenna = 0;
    w->g[r].n_connected = 0;
    return ORC_OK;
}

int orc_set_antenna(World *w, int32_t r, int32_t on) {
    if (!w || r < 0 || r >= w->n || w->g[r].removed) return ORC_ERR_INVALID;
    w->g[r].antenna = on != 0;
    return ORC_OK;
}
int orc_set_idle(World *w, int32_t r, int32_t idle) {
    if (!w || r < 0 || r >= w->n || w->g[r].removed) return ORC_ERR_INVALID;
    w->g[r].idle = idle != 0;
    return ORC_OK;
}

/* FactorGraph::change_factor_enabled (FG/factorgraph.rs:1529-1539) for every graph, as the settings
 * panel applies it (ui/settings.rs:491-496) together with the config entry new factors read. */
int orc_set_enabled(World *w, uint32_t mask) {
    if (!w || (mask & ~15u)) return ORC_ERR_INVALID;
    w->p.enable_mask = mask;
    for (int r = 0; r < w->n; r++) {
        Graph *g = &w->g[r];
        if (g->removed) continue;
        for (int i = 0; i < g->n_factors; i++) {
            Node *nd = &g->nodes[g->factor_indices[i]];
            if (!nd->alive || !nd->is_factor) continue;
            Factor *f = &nd->f;
            f->enabled = f->kind == K_DYNAMIC ? (mask & EN_DYN) != 0 : f->kind == K_INTERROBOT ? (mask & EN_IR) != 0
                       : f->kind == K_OBSTACLE ? (mask & EN_OBS) != 0 : (mask & EN_TRK) != 0;
        }
    }
    return ORC_OK;
}

static void prepare_factor(const World *w, Factor *f) {
    if (f->kind == K_OBSTACLE) /* obstacle.rs:98-102 */
        f->jac_delta = (w->world_w / (double)w->sdf_w + w->world_h / (double)w->sdf_h) / 2.0;
}

/* internal_factor_iteration — FG/factorgraph.rs:688-714 */
static void internal_factor_iteration(World *w, int r) {
    Graph *g = &w->g[r];
    for (int i = 0; i < g->n_factors; i++) {
        int ix = g->factor_indices[i];
        Factor *f = &g->nodes[ix].f;
        if (!f->enabled) continue;
        if (f->kind == K_INTERROBOT) continue;
        if (f->kind == K_TRACKING && g->iter_factor < 10) continue;
        prepare_factor(w, f);
        Msg out[2];
        factor_update(w, g, f, out);
        for (int k = 0; k < f->inbox.n; k++) {
            NodeId to = f->inbox.e[k].key;
            variable_receive(&g->nodes[to.index].v, mkid(w, r, ix), &out[k]);
        }
    }
    g->iter_factor += 1;
}

/* internal_variable_iteration — FG/factorgraph.rs:762-790 */
static void internal_variable_iteration(World *w, int r) {
    Graph *g = &w->g[r];
    Msg *out = NULL;
    int cap = 0;
    for (int k = 0; k < g->K; k++) {
        int vix = g->var_indices[k];
        Variable *v = &g->nodes[vix].v;
        if (v->inbox.n > cap) {
            cap = v->inbox.n;
            out = (Msg *)realloc(out, sizeof(Msg) * (size_t)cap);
        }
        variable_update(v, out);
        for (int j = 0; j < v->inbox.n; j++) {
            NodeId to = v->inbox.e[j].key;
            if (to.robot != r) continue; /* :772-776 */
            Factor *f = &g->nodes[to.index].f;
            if (!f->enabled) continue; /* :781-783 */
            factor_receive(f, mkid(w, r, vix), &out[j]);
        }
    }
    free(out);
    g->iter_variable += 1;
}

typedef struct {
    NodeId from, to;
    Msg msg;
} Routed;
typedef struct {
    Routed *e;
    int n, cap;
} RoutedVec;
static void routed_push(RoutedVec *v, NodeId from, NodeId to, const Msg *m) {
    if (v->n == v->cap) {
        v->cap = v->cap ? 2 * v->cap : 256;
        v->e = (Routed *)realloc(v->e, sizeof(Routed) * (size_t)v->cap);
    }
    v->e[v->n].from = from;
    v->e[v->n].to = to;
    v->e[v->n].msg = *m;
    v->n++;
}

/* external_factor_iteration — FG/factorgraph.rs:719-760 */
static void external_factor_iteration(World *w, int r, RoutedVec *outv) {
    Graph *g = &w->g[r];
    for (int i = 0; i < g->n_ir; i++) {
        int ix = g->ir_indices[i];
        if (ix >= g->n_nodes || !g->nodes[ix].alive || !g->nodes[ix].is_factor) continue; /* contains_node */
        Factor *f = &g->nodes[ix].f;
        if (f->kind != K_INTERROBOT) continue;
        if (!f->enabled) continue;
        Msg out[2];
        factor_update(w, g, f, out);
        for (int k = 0; k < f->inbox.n; k++) {
            NodeId to = f->inbox.e[k].key;
            if (to.robot != r) routed_push(outv, mkid(w, r, ix), to, &out[k]); /* :745-754 */
        }
    }
    g->iter_factor += 1;
}

/* external_variable_iteration — FG/factorgraph.rs:794-826 */
static void external_variable_iteration(World *w, int r, RoutedVec *outv) {
    Graph *g = &w->g[r];
    Msg *out = NULL;
    int cap = 0;
    for (int k = 0; k < g->K; k++) {
        int vix = g->var_indices[k];
        Variable *v = &g->nodes[vix].v;
        if (v->inbox.n > cap) {
            cap = v->inbox.n;
            out = (Msg *)realloc(out, sizeof(Msg) * (size_t)cap);
        }
        variable_update(v, out);
        for (int j = 0; j < v->inbox.n; j++) {
            NodeId to = v->inbox.e[j].key;
            if (to.robot != r) routed_push(outv, mkid(w, r, vix), to, &out[j]);
        }
    }
    free(out);
    g->iter_variable += 1;
}

static int factor_node_exists(const World *w, NodeId id) {
    const Graph *g = &w->g[id.robot];
    return id.index < g->n_nodes && g->nodes[id.index].alive && g->nodes[id.index].is_factor;
}

/* one schedule step — iterate_gbp_v2 body, ROBOT:1787-1860 */
static void step(World *w, int internal, int external) {
    if (internal) { /* :1788-1801 par_iter_mut over robots */
#pragma omp parallel for schedule(static) num_threads(w->n_threads) if (w->n_threads > 1)
        for (int r = 0; r < w->n; r++) {
            Graph *g = &w->g[r];
            if (g->ghost || g->idle) continue;
            internal_factor_iteration(w, r);
            internal_variable_iteration(w, r);
        }
    }
    if (external) { /* :1803-1859, serial */
        RoutedVec tv = {0, 0, 0};
        for (int r = 0; r < w->n; r++) {
            Graph *g = &w->g[r];
            if (g->ghost || !g->antenna || g->idle) continue;
            external_factor_iteration(w, r, &tv);
        }
        for (int i = 0; i < tv.n; i++) { /* :1813-1831 */
            Graph *og = &w->g[tv.e[i].to.robot];
            if (!og->antenna || og->idle) continue;
            variable_receive(&og->nodes[tv.e[i].to.index].v, tv.e[i].from, &tv.e[i].msg);
        }
        free(tv.e);
        RoutedVec fv = {0, 0, 0};
        for (int r = 0; r < w->n; r++) {
            Graph *g = &w->g[r];
            if (g->ghost || !g->antenna || g->idle) continue;
            external_variable_iteration(w, r, &fv);
        }
        for (int i = 0; i < fv.n; i++) { /* :1842-1858 */
            Graph *og = &w->g[fv.e[i].to.robot];
            if (!og->antenna || og->idle) continue;
            if (!factor_node_exists(w, fv.e[i].to)) continue;
            factor_receive(&og->nodes[fv.e[i].to.index].f, fv.e[i].from, &fv.e[i].msg);
        }
        free(fv.e);
    }
}

int orc_iterate(World *w, const uint8_t *steps, uint32_t n) {
    if (!w || (!steps && n)) return ORC_ERR_INVALID;
    for (uint32_t i = 0; i < n; i++) step(w, steps[i] & 1u, (steps[i] & 2u) != 0);
    return ORC_OK;
}

/* fine-grained sweeps (robot = -1: all robots, with the caller's gating and routing) */
int orc_internal_factor_iteration(World *w, int32_t robot) {
    for (int r = 0; r < w->n; r++)
        if ((robot < 0 && !w->g[r].idle) || r == robot) internal_factor_iteration(w, r);
    return ORC_OK;
}
int orc_internal_variable_iteration(World *w, int32_t robot) {
    for (int r = 0; r < w->n; r++)
        if ((robot < 0 && !w->g[r].idle) || r == robot) internal_variable_iteration(w, r);
    return ORC_OK;
}
int orc_external_factor_iteration(World *w, int32_t robot) {
    if (robot >= 0) return ORC_ERR_INVALID;
    RoutedVec tv = {0, 0, 0};
    for (int r = 0; r < w->n; r++) {
        Graph *g = &w->g[r];
        if (!g->antenna || g->idle) continue;
        external_factor_iteration(w, r, &tv);
    }
    for (int i = 0; i < tv.n; i++) {
        Graph *og = &w->g[tv.e[i].to.robot];
        if (!og->antenna || og->idle) continue;
        variable_receive(&og->nodes[tv.e[i].to.index].v, tv.e[i].from, &tv.e[i].msg);
    }
    free(tv.e);
    return ORC_OK;
}
int orc_external_variable_iteration(World *w, int32_t robot) {
    if (robot >= 0) return ORC_ERR_INVALID;
    RoutedVec fv = {0, 0, 0};
    for (int r = 0; r < w->n; r++) {
        Graph *g = &w->g[r];
        if (!g->antenna || g->idle) continue;
        external_variable_iteration(w, r, &fv);
    }
    for (int i = 0; i < fv.n; i++) {
        Graph *og = &w->g[fv.e[i].to.robot];
        if (!og->antenna || og->idle) continue;
        if (!factor_node_exists(w, fv.e[i].to)) continue;
        factor_receive(&og->nodes[fv.e[i].to.index].f, fv.e[i].from, &fv.e[i].msg);
    }
    free(fv.e);
    return ORC_OK;
}

/* VariableNode::change_prior + FactorGraph::change_prior_of_variable + caller routing
 * — FG/variable.rs:203-230, FG/factorgraph.rs:494-528, ROBOT:2262-2282 */
int orc_change_prior(World *w, int32_t r, uint32_t var_ix, const double *mean) {
    if (!w || r < 0 || r >= w->n || (int)var_ix >= w->g[r].K || !mean || w->g[r].removed) return ORC_ERR_INVALID;
    Graph *g = &w->g[r];
    int vix = g->var_indices[var_ix];
    Variable *v = &g->nodes[vix].v;
    matvec(v->prior_lam, mean, v->prior_eta, 4, 4); /* :204 */
    memcpy(v->mu, mean, sizeof v->mu);              /* :206 */
    Msg m;
    variable_prepare_message(v, &m);
    for (int j = 0; j < v->inbox.n; j++) {
        NodeId to = v->inbox.e[j].key;
        if (factor_node_exists(w, to)) /* FG/factorgraph.rs:511-513, ROBOT:2277-2281 */
            factor_receive(&w->g[to.robot].nodes[to.index].f, mkid(w, r, vix), &m);
        memset(&v->inbox.e[j].msg, 0, sizeof(Msg)); /* :224-227 */
    }
    return ORC_OK;
}

/* FactorGraph::reset_variables — FG/factorgraph.rs:1541-1564 with VariableNode::reset (FG/variable.rs:350-360) and
 * FactorNode::empty_inbox (FG/factor/mod.rs:480-483): belief mean and precision replaced (the `sigma` arguments are used
 * AS the diagonal of the precision, infinite included; information vector, covariance, prior untouched), every message
 * in the variables' inboxes and in the inboxes of the graph's own factors (inter-robot ones included) becomes empty.
 * Called by the path-finding completion handler with (means, 1e30, +inf) (ROBOT:768). */
int orc_reset_variables(World *w, int32_t r, const double *means, double first_last_sigma, double inbetween_sigma) {
    if (!w || r < 0 || r >= w->n || !means || w->g[r].removed || w->g[r].ghost) return ORC_ERR_INVALID;
    Graph *g = &w->g[r];
    for (int i = 0; i < g->K; i++) {
        Variable *v = &g->nodes[g->var_indices[i]].v;
        double sigma = (i == 0 || i == g->K - 1) ? first_last_sigma : inbetween_sigma;
        memcpy(v->mu, means + 4 * i, sizeof v->mu);
        for (int q = 0; q < 16; q++) v->lam[q] = (q % 5 == 0) ? sigma : 0.0; /* Matrix::from_diag_elem */
        for (int j = 0; j < v->inbox.n; j++) memset(&v->inbox.e[j].msg, 0, sizeof(Msg));
    }
    for (int q = 0; q < g->n_factors; q++) {
        Node *nd = &g->nodes[g->factor_indices[q]];
        if (!nd->alive || !nd->is_factor) continue;
        for (int j = 0; j < nd->f.inbox.n; j++) memset(&nd->f.inbox.e[j].msg, 0, sizeof(Msg));
    }
    return ORC_OK;
}
/* FactorGraph::reset_tracking_factors — FG/factorgraph.rs:1566-1590: every tracking factor attached to a variable
 * 1 .. K-2 (all of them, ROBOT:1290-1334) gets set_timeout(10): its next ten updates are skipped (tracking.rs:362-371) */
int orc_reset_tracking_factors(World *w, int32_t r) {
    if (!w || r < 0 || r >= w->n || w->g[r].removed || w->g[r].ghost) return ORC_ERR_INVALID;
    Graph *g = &w->g[r];
    for (int q = 0; q < g->n_factors; q++) {
        Node *nd = &g->nodes[g->factor_indices[q]];
        if (nd->alive && nd->is_factor && nd->f.kind == K_TRACKING) nd->f.timeout = 10;
    }
    return ORC_OK;
}

/* update_prior_of_horizon_state (ROBOT:2182-2283) and update_prior_of_current_state_v3
 * (ROBOT:2286-2338) for the listed robots; the caller has already applied the systems' skip rules
 * (finished / idle / no next waypoint).  The reference runs the first system over all robots, then
 * the second. */
int orc_update_priors(World *w, uint32_t n, const int32_t *robots, const double *waypoints_xy, const double *time_scale,
                      const uint8_t *what, double max_speed, double delta_t) {
    for (uint32_t t = 0; t < n; t++)
        if (robots[t] < 0 || robots[t] >= w->n || w->g[robots[t]].removed) return ORC_ERR_INVALID;
    for (uint32_t t = 0; t < n; t++) {
        if (!(what[t] & 1u)) continue;
        Graph *g = &w->g[robots[t]];
        Variable *hv = &g->nodes[g->var_indices[g->K - 1]].v; /* last_variable_mut :2239 */
        double est[2] = {hv->mu[0], hv->mu[1]};                /* :2242 */
        double h2w[2] = {waypoints_xy[2 * t] - est[0], waypoints_xy[2 * t + 1] - est[1]}; /* :2251 */
        double dist = euclidean_norm(h2w, 2);                  /* :2252 */
        double nrm[2] = {h2w[0], h2w[1]};
        normalize(nrm, 2);
        double sp = fmin(max_speed, dist);                      /* Float::min :2254 */
        double vel[2] = {sp * nrm[0], sp * nrm[1]};
        double mean[4] = {est[0] + vel[0] * delta_t, est[1] + vel[1] * delta_t, vel[0], vel[1]}; /* :2255-2258 */
        memcpy(hv->mu, mean, sizeof mean);                      /* :2263 */
        orc_change_prior(w, robots[t], (uint32_t)(g->K - 1), mean); /* :2266-2282 */
    }
    for (uint32_t t = 0; t < n; t++) {
        if (!(what[t] & 2u)) continue;
        Graph *g = &w->g[robots[t]];
        const Variable *v0 = &g->nodes[g->var_indices[0]].v, *v1 = &g->nodes[g->var_indices[1]].v;
        double mean[4];
        for (int c = 0; c < 4; c++) { /* :2309-2316 */
            double change = time_scale[t] * (v1->mu[c] - v0->mu[c]);
            mean[c] = v0->mu[c] + change;
        }
        orc_change_prior(w, robots[t], 0u, mean); /* :2318-2322 */
    }
    return ORC_OK;
}

int orc_get_belief(World *w, int32_t r, uint32_t var_ix, double *eta, double *lam, double *mean,
                   double *cov, int32_t *valid) {
    if (!w || r < 0 || r >= w->n || (int)var_ix >= w->g[r].K) return ORC_ERR_INVALID;
    const Variable *v = &w->g[r].nodes[w->g[r].var_indices[var_ix]].v;
    if (eta) memcpy(eta, v->eta, sizeof v->eta);
    if (lam) memcpy(lam, v->lam, sizeof v->lam);
    if (mean) memcpy(mean, v->mu, sizeof v->mu);
    if (cov) memcpy(cov, v->cov, sizeof v->cov);
    if (valid) *valid = v->valid;
    return ORC_OK;
}

int orc_read_beliefs(World *w, double *eta, double *lam, double *means) {
    size_t o = 0;
    for (int r = 0; r < w->n; r++) {
        const Graph *g = &w->g[r];
        if (g->ghost) continue;
        for (int k = 0; k < g->K; k++, o++) {
            const Variable *v = &g->nodes[g->var_indices[k]].v;
            if (eta) memcpy(eta + 4 * o, v->eta, sizeof v->eta);
            if (lam) memcpy(lam + 16 * o, v->lam, sizeof v->lam);
            if (means) memcpy(means + 4 * o, v->mu, sizeof v->mu);
        }
    }
    return ORC_OK;
}

int orc_num_robots(World *w, uint32_t *n_robots, uint32_t *n_variables) {
    uint32_t nr = 0, nv = 0;
    for (int r = 0; r < w->n; r++)
        if (!w->g[r].ghost) {
            nr++;
            nv += (uint32_t)w->g[r].K;
        }
    if (n_robots) *n_robots = nr;
    if (n_variables) *n_variables = nv;
    return ORC_OK;
}

/* FactorGraph::messages_sent / messages_received (FG/factorgraph.rs:876-890): sums over the nodes
 * the graph holds NOW (a deleted factor takes its counts with it).  out = sent internal, sent
 * external, received internal, received external. */
int orc_message_counts(World *w, int32_t r, uint64_t out[4]) {
    if (!w || r < 0 || r >= w->n || !out) return ORC_ERR_INVALID;
    const Graph *g = &w->g[r];
    for (int c = 0; c < 4; c++) out[c] = 0;
    for (int ix = 0; ix < g->n_nodes; ix++) {
        const Node *nd = &g->nodes[ix];
        if (!nd->alive) continue;
        for (int c = 0; c < 4; c++) out[c] += nd->is_factor ? nd->f.cnt[c] : nd->v.cnt[c];
    }
    return ORC_OK;
}

/* debugging / white-box access for tests: message factor->variable currently in a
 * variable's inbox from the j-th inbox entry; returns present flag, -1 if out of range */
int orc_variable_inbox(World *w, int32_t r, uint32_t var_ix, int32_t j, int32_t *from_robot,
                       int32_t *from_index, double *eta, double *lam) {
    const Variable *v = &w->g[r].nodes[w->g[r].var_indices[var_ix]].v;
    if (j < 0 || j >= v->inbox.n) return -1;
    if (from_robot) *from_robot = v->inbox.e[j].key.robot;
    if (from_index) *from_index = v->inbox.e[j].key.index;
    if (eta) memcpy(eta, v->inbox.e[j].msg.eta, sizeof(double) * 4);
    if (lam) memcpy(lam, v->inbox.e[j].msg.lam, sizeof(double) * 16);
    return v->inbox.e[j].msg.present;
}

/* ---------------------------------------------------------------------------------------
 * crates/gbp_schedule/src/schedules/ *.rs
 * --------------------------------------------------------------------------------------- */
static void sched_interleave_recurse(uint8_t *s, int len, int n) { /* interleave_evenly.rs:41-104 */
    int max = len, half = max / 2;
    if (n == max) {
        memset(s, 1, (size_t)len);
    } else if (n == 0) {
        memset(s, 0, (size_t)len);
    } else if ((n % 2 == 1) && (max % 2 == 1)) {
        if (max % n == 0) {
            int td = max / n;
            for (int i = 0; i < len; i++) s[i] = (i % td) == 0;
        } else {
            int h = n / 2;
            sched_interleave_recurse(s, half, h);
            s[half] = 1;
            sched_interleave_recurse(s + half + 1, len - half - 1, h);
            for (int i = half + 1, j = len - 1; i < j; i++, j--) {
                uint8_t t = s[i];
                s[i] = s[j];
                s[j] = t;
            }
        }
    } else if ((n % 2 == 0) && (max % 2 == 1)) {
        int h = n / 2;
        sched_interleave_recurse(s, half, h);
        for (int i = 0, j = half - 1; i < j; i++, j--) {
            uint8_t t = s[i];
            s[i] = s[j];
            s[j] = t;
        }
        s[half] = 0;
        sched_interleave_recurse(s + half + 1, len - half - 1, h);
    } else if ((n % 2 == 0) && (max % 2 == 0)) {
        if (max % n == 0) {
            int td = max / n;
            for (int i = 0; i < len; i++) s[i] = (i % td) == 0;
        } else {
            int h = n / 2;
            sched_interleave_recurse(s, half, h);
            sched_interleave_recurse(s + half, len - half, h);
        }
    } else { /* odd n, even max */
        int h = n / 2;
        sched_interleave_recurse(s, half, h + 1);
        for (int i = 0, j = half - 1; i < j; i++, j--) {
            uint8_t t = s[i];
            s[i] = s[j];
            s[j] = t;
        }
        sched_interleave_recurse(s + half, len - half, h);
    }
}

static void sched_stream(int kind, int n, int max, uint8_t *s) {
    switch (kind) {
    case 0: /* centered.rs:19-48 */
        for (int i = 0; i < max; i++) {
            if (n == 0 && max == 1) {
                s[i] = 0;
                continue;
            }
            int mid = max / 2, hn = n / 2;
            int start = mid >= hn ? mid - hn : 0;
            /* int arithmetic: n == 0 gives end = start - 1 => all false (the reference's u8
             * `start + n - 1` can only underflow for max == 1, handled above) */
            int end = (start + n <= max) ? start + n - 1 : max - 1;
            s[i] = (i >= start && i <= end);
        }
        break;
    case 1: /* soon_as_possible.rs:27-51 */
        for (int i = 0; i < max; i++) s[i] = i < n;
        break;
    case 2: /* late_as_possible.rs:30-48 */
        for (int i = 0; i < max; i++) s[i] = (n == max) ? 1 : (n == 0 ? 0 : i >= max - n);
        break;
    case 3: sched_interleave_recurse(s, max, n); break;
    case 4: { /* half_beginning_half_end.rs:19-43 */
        int hn = n / 2, rem = n % 2;
        for (int i = 0; i < max; i++) s[i] = (i < hn || i >= max - hn - rem);
        break;
    }
    }
}

int orc_schedule(int32_t kind, uint8_t n_int, uint8_t n_ext, uint8_t *steps, uint32_t capacity) {
    int max = n_int > n_ext ? n_int : n_ext;
    if (kind < 0 || kind > 4 || !steps || (int)capacity < max) return ORC_ERR_INVALID;
    uint8_t a[256], b[256];
    sched_stream(kind, n_int, max, a);
    sched_stream(kind, n_ext, max, b);
    for (int i = 0; i < max; i++) steps[i] = (uint8_t)((a[i] ? 1 : 0) | (b[i] ? 2 : 0));
    return max;
}

/* crates/magics/src/utils.rs:35-75 (f32 arithmetic, mul_add = fmaf) */
int orc_variable_timesteps(uint32_t h, uint32_t m, uint32_t *ts, uint32_t capacity) {
    if (!ts || m == 0) return ORC_ERR_INVALID;
    uint32_t n = 1u + (uint32_t)(0.5f * (-1.0f + sqrtf(1.0f + 8.0f * (float)h / (float)m)));
    uint32_t cnt = 0;
    for (uint32_t i = 0; i < m * (n + 1u); i++) {
        uint32_t section = i / m;
        float f = fmaf((float)m / 2.0f, (float)section, fmaf((float)section, -(float)m, (float)i)) *
                  ((float)section + 1.0f);
        if (f >= (float)h) {
            if (cnt >= capacity) return ORC_ERR_INVALID;
            ts[cnt++] = h;
            break;
        }
        if (cnt >= capacity) return ORC_ERR_INVALID;
        ts[cnt++] = (uint32_t)f;
    }
    return (int)cnt;
}
