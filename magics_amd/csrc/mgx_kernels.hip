// mgx_kernels.hip — gfx950 kernels of the GBP engine.
//
// k_robot_sweep: ONE 128-THREAD WORKGROUP PER ROBOT, TWO ROLE-SPECIALISED WAVES.
// The robot's whole factor graph (its private state blob: priors, beliefs, factor->variable
// messages; its snapshot records; the inter-robot messages attached to its variables) is staged in
// LDS once per launch by straight 16-byte copies and stays there for every phase the launch runs: an optional external phase
// (external_factor_iteration + routing + external_variable_iteration,
// factorgraph.rs:719-760,794-826, robot.rs:1803-1859) followed by `n_int` internal iterations
// (internal_factor_iteration + internal_variable_iteration, factorgraph.rs:688-714,762-790).
//
//   wave 0 (DYN)  factor phase: one lane per dynamic-factor MESSAGE (2(K-1) lanes), one 4x4 Schur
//                 complement each
//   wave 1 (UV)   factor phase: mean / covariance of the previous sweep (one lane per variable: 4x4
//                 inverse), then one lane per obstacle / tracking factor
//   both waves    variable phase: inbox sums, one lane per (variable, row);
//                 external factor sweep: one lane per incoming inter-robot edge ("pull" form: every
//                 factor F_AB is evaluated by the workgroup of its only consumer B)
//
// A dynamic factor never reads a mean, so the expensive half of a variable update (inverse, mean) of
// sweep t runs in the UV wave NEXT TO the dynamic messages of sweep t+1 in the DYN wave.  One wave
// alone issues an f64 VALU instruction only every ~8 cycles, so what bounds an iteration is the
// longest dependent instruction stream per robot, not lane count: the design shortens that stream.
// All per-variable state lives in LDS, not registers, so each wave stays within 256 VGPRs.  Each
// phase is a Jacobi sweep separated by workgroup barriers only.  Robots couple only through the
// inter-robot edges, which gather the OTHER robot's 192-byte snapshot records from buffer `cur` in
// HBM while this launch writes buffer `1 - cur`: no inter-workgroup synchronisation inside a launch.
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
// Compact inter-robot messages.  The factor's Jacobian touches the two positions only
// (interrobot.rs:149-159), so J^T L J and J^T L (..) vanish outside the position rows / columns,
// and so does the Schur complement towards either variable: Lab has two non-zero rows and columns,
// hence Lab W Lba and Lab W eb live in the top-left 2x2 block / first two entries — as exact zeros
// as long as the arithmetic stays finite (0 * x is 0; tests/test_oracle_known_answers.py checks it
// on the oracle's full 20-entry messages).  Such a message is six numbers: eta0, eta1, lam00,
// lam01, lam10, lam11; only those are computed, stored and summed (x + 0 == x).  With NaN / inf in
// play the reference would smear NaNs over the other entries (0 * NaN); that is one of the places
// where a non-finite world is not reproduced (known deviations).
constexpr int IR_STRIDE = 7;  // one staged inter-robot message: 6 f64 + 1 pad (bank spread)
// IRM, how a launch treats inter-robot messages: the world has no inter-robot edges at all (every
// trace of them is compiled out: configs[1] runs this), they are read from HBM / L2 in every
// variable sweep (a robot with too many edges for LDS), or they are staged in LDS.
enum { IR_NONE = 0, IR_GLOBAL = 1, IR_STAGED = 2 };

// Workgroups are handed to the eight XCDs round-robin by workgroup id, and each XCD has its own L2.
// Robots are numbered along the grid, so inter-robot neighbours have nearby ids: giving XCD k the
// k-th CONTIGUOUS eighth of the robots (instead of every eighth robot) lets the snapshot records that
// several neighbours gather be fetched into that XCD's L2 once.  Bijection of [0, n) for any n.
constexpr int N_XCD = 8;
__device__ __forceinline__ int xcd_local_index(int block, int n) {
    const int xcd = block % N_XCD, idx = block / N_XCD;
    // workgroups with id = k (mod 8): ceil((n - k) / 8) of them; robots of XCD k start after those of 0..k-1
    int start = 0;
    for (int k = 0; k < xcd; k++) start += (n - k + N_XCD - 1) / N_XCD;
    return start + idx;
}

// straight copies between a robot's blob in HBM and its LDS image, 16 bytes per lane
__device__ __forceinline__ void copy_words(double *dst, const double *src, int n, int tid) {
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
    double2 *d2 = reinterpret_cast<double2 *>(dst);
    for (int t = tid; t < (n >> 1); t += SWEEP_BLOCK) d2[t] = s2[t];
    if ((n & 1) && tid == 0) dst[n - 1] = src[n - 1];
}

// HBM -> LDS with every load of the thread in flight before its first LDS store: a copy loop that
// waits for each 16 bytes before asking for the next costs one memory round trip per iteration, and
// staging is a handful of such loops.  N (f64 words) is a compile-time constant: the loops unroll into
// independent loads held in registers.
template <int N>
struct StageRegs {
    static constexpr int N2 = N / 2, ITERS = (N2 + SWEEP_BLOCK - 1) / SWEEP_BLOCK;
    double2 v[ITERS];
    double last;
    __device__ __forceinline__ void load(const double *src, int tid) {
        const double2 *s2 = reinterpret_cast<const double2 *>(src);
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const int t = tid + it * SWEEP_BLOCK;
            v[it] = (t < N2) ? s2[t] : make_double2(0.0, 0.0);
        }
        last = ((N & 1) && tid == 0) ? src[N - 1] : 0.0;
    }
    __device__ __forceinline__ void store(double *dst, int tid) const {
        double2 *d2 = reinterpret_cast<double2 *>(dst);
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const int t = tid + it * SWEEP_BLOCK;
            if (t < N2) d2[t] = v[it];
        }
        if ((N & 1) && tid == 0) dst[N - 1] = last;
    }
};

// the same, issued by one wave (64 lanes)
__device__ __forceinline__ void copy_words_wave(double *dst, const double *src, int n, int lane) {
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
    double2 *d2 = reinterpret_cast<double2 *>(dst);
    for (int t = lane; t < (n >> 1); t += 64) d2[t] = s2[t];
    if ((n & 1) && lane == 0) dst[n - 1] = src[n - 1];
}

// Accesses to what ANOTHER workgroup of the same launch writes or reads (resident schedule launches): relaxed
// agent-scope atomics = global_load / global_store ... sc1 — they bypass this CU's L1 and write through the XCD's L2,
// which is what makes a record published by one workgroup readable by another without cache maintenance
// (MI355X_MICROARCH.md, inter-workgroup visibility: all-sc1 stores and loads, every storing wave drains vmcnt,
// one lane signals behind a workgroup barrier, the consumer polls that word and loads behind a barrier).
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The same for 16 bytes: raw buffer accesses with aux = 16 (sc1); the descriptor is built from wave-uniform values.
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sc1_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void ld16_agent(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, double &a, double &b) {
    const v4u32 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 16);
    a = __hiloint2double((int)v.y, (int)v.x);
    b = __hiloint2double((int)v.w, (int)v.z);
}
__device__ __forceinline__ v4u32 ld16_agent_raw(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 16);
}
__device__ __forceinline__ void st16_agent(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, double a, double b) {
    v4u32 v;
    v.x = (unsigned)__double2loint(a); v.y = (unsigned)__double2hiint(a);
    v.z = (unsigned)__double2loint(b); v.w = (unsigned)__double2hiint(b);
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)byte_off, 0, 16);
}

// 4 x 4 transposition inside every quad of lanes: on entry lane v holds x[s] = item (s, v), on exit lane s holds x[v] = item (s, v)
// — two butterfly stages of DPP quad permutes, no LDS.
__device__ __forceinline__ void quad_transpose4(unsigned (&x)[4], int lane) {
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
#pragma unroll
    for (int p = 0; p < 4; p += 2) {  // lanes v, v ^ 1 exchange x[p + 1] of the even lane with x[p] of the odd one
        const unsigned send = b0 ? x[p] : x[p + 1];
        const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xf, 0xf, true);  // quad_perm [1, 0, 3, 2]
        x[p] = b0 ? recv : x[p];
        x[p + 1] = b0 ? x[p + 1] : recv;
    }
#pragma unroll
    for (int p = 0; p < 2; p++) {  // lanes v, v ^ 2 exchange x[p + 2] of the lower lane with x[p] of the upper one
        const unsigned send = b1 ? x[p] : x[p + 2];
        const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xf, 0xf, true);  // quad_perm [2, 3, 0, 1]
        x[p] = b1 ? recv : x[p];
        x[p + 2] = b1 ? x[p + 2] : recv;
    }
}

// PERSIST: the launch runs a whole schedule (SegPlan, mgx_dev.h) instead of one segment.
template <int KT, int IRM, bool PERSIST>
__global__ void __launch_bounds__(SWEEP_BLOCK, 2) k_robot_sweep(DevWorld w, int robot0, uint32_t ext_mask, uint32_t int_mask,
                                                             int n_int, int snap_out, uint32_t hints, const SegPlan plan) {
    constexpr bool HAS_IR = IRM != IR_NONE, STAGE_IR = IRM == IR_STAGED;
    static_assert(!PERSIST || IRM == IR_STAGED, "resident schedule launches exist for worlds with staged inter-robot messages");
    const int nseg = PERSIST ? plan.n : 1;
    // KT > 0: horizon length fixed at compile time; 0: read from the world (K <= 33); -1: read from the world, any K.
    // BIG: more than 64 dynamic-factor messages / tracking factors per robot (K > 33): some lanes carry two.
    constexpr bool BIG = KT < 0 || KT > 33;
    STAMP(t_k0);
    const int r = robot0 + xcd_local_index(blockIdx.x, gridDim.x);
    int tid = threadIdx.x;  // not const: resident launches make them opaque once per segment, see the segment loop
    const int role = tid >> 6;
    int lane = tid & 63;
    const int K = KT > 0 ? KT : w.K, E = 4 * K - 6, E1 = E + 1;
    const BlobLayout L(K);
    const int ZCOL = E;  // all-zero message column (absent edges)
    double *s_snap = lds;                                 // [24][K] variable -> own-factor snapshots
    double *s_prior = s_snap + SNAP_W * K;                // [20][K] prior eta, lam (belief after the last sweep)
    double *s_tmp = s_prior + 20 * K;                     // [20][K] scratch sums (external sweep)
    double *s_io = s_tmp + 20 * K;                        // image of the blob's in/out region:
    double *s_cov = s_io;                                 //   [16][K] belief covariance
    double *s_mu = s_io + 16 * K;                         //   [4][K]  belief mean
    double *s_fv = s_io + 20 * K;                         //   [20][E1] factor -> variable messages
    int32_t *s_valid = (int32_t *)(s_io + 20 * K + 20 * E1);  // [K]
    uint32_t *s_epoch = (uint32_t *)(s_io + L.inout_words() + (L.inout_words() & 1));  // [K] deliveries
    int32_t *s_irp = (int32_t *)(s_epoch + ((K + 1) & ~1));  // [3][K+1] inbox ranges of foreign factors
    int32_t *s_covset = s_irp + ((3 * (K + 1) + 1) & ~1);     // [K] this launch recomputed the variable's covariance
    // resident launches: shadow of s_fv that takes the factor sweep computed AHEAD of the external iteration it follows
    int32_t *s_xok = s_covset + ((K + 1) & ~1);               // [K] (PERSIST only) outcome of a concurrent external belief update
    double *s_sh = (double *)(s_xok + (PERSIST ? ((K + 1) & ~1) : 0));  // [20][E1] (PERSIST only)
    double *s_ir = s_sh + (PERSIST ? 20 * E1 : 0);            // [ne][IR_STRIDE] inter-robot messages (STAGE_IR)

    double *blob = w.blob + (size_t)r * w.BS;
    const int v0 = r * K;
    const int ie0 = HAS_IR ? w.ir_var_ptr[v0] : 0, ne = HAS_IR ? w.ir_var_ptr[v0 + K] - ie0 : 0;
    const bool ir_on = HAS_IR && (w.enable & 2u) != 0;
    const int n_dyn = 2 * (K - 1);
    // kinds whose first internal factor sweep of this launch has already been computed from the inbox they
    // froze with (k_thaw, mgx_set_enabled); the pointer is null unless some robot is thawing
    const uint32_t skip0 = w.skip0 ? (uint32_t)w.skip0[r] : 0u;
    const bool idle = w.idle[r] != 0;
    const bool radio = (w.antenna[r] != 0) && !idle;

    int itf = w.iter_factor[r];  // iteration_count.factor (every lane applies the same increments)

    // ---- roles ------------------------------------------------------------------------------------
    const bool is_dyn = role == ROLE_DYN && lane < n_dyn;
    const bool is_obs = role == ROLE_UV && lane < K - 2;
    const bool is_trk = role == ROLE_UV && lane >= K - 2 && lane < 2 * (K - 2);
    const bool is_var = role == ROLE_UV && lane < K;
    // Horizons of at most 16 variables: the belief finish runs on FOUR lanes per variable (lane (q, i) computes cofactor row q of
    // variable i, variable_finish_quad), and in resident launches the whole variable sweep — inbox sums and finish — stays on the
    // UV wave with no workgroup barrier in between (FUSED).
    constexpr bool QUADFIN = KT > 0 && 4 * KT <= 64;
    constexpr bool FUSED = PERSIST && QUADFIN;
    const int sum_t = FUSED ? (role == ROLE_UV ? lane : 4 * K) : tid;        // (variable, row) this thread sums: t = rr * K + i
    const int sum_step = FUSED ? 4 * K : SWEEP_BLOCK;
    // which variable sweep of this launch is the robot's last one (its belief goes out)
    bool plan_int = false, plan_ext = false;  // PERSIST: some segment has internal iterations / an external iteration
    if (PERSIST)
        for (int k = 0; k < nseg; k++) { plan_int = plan_int || plan.n_int[k] > 0; plan_ext = plan_ext || plan.ext[k] != 0; }
    const bool has_int_var = PERSIST ? (plan_int && !idle) : ((int_mask & PH_INT_VARIABLE) && n_int > 0 && !idle);
    const bool any_sweep = has_int_var || ((PERSIST ? plan_ext : (ext_mask & PH_EXT_VARIABLE) != 0) && radio);
    // a later launch of the same call rewrites this robot's belief image: this one's copy is never read
    const bool bel_dead = ((hints & HINT_LATER_EXT_VARIABLE) && radio) || ((hints & HINT_LATER_INT_VARIABLE) && !idle);

    // DYN wave: constant potential blocks of this lane's message
    // (resident launches fetch them from L2 in every sweep instead: the DYN wave has the time — it computes its messages ahead,
    // under the hand-off — and the 32 registers are what the segment loop would otherwise spill)
    double maa[4], mab[4], mba[4], mbb[4];
    int dyn_other_var = 0, dyn_other_edge = 0;
    auto load_dyn_potential = [&](double (&paa)[4], double (&pab)[4], double (&pba)[4], double (&pbb)[4]) __attribute__((always_inline)) {
        const int f = lane % (K - 1), slot = lane / (K - 1);
        const int a2 = 2 * slot, b2 = 2 * (1 - slot);
        const int it = r * (K - 1) + f;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                paa[i * 2 + j] = w.dyn_m[(size_t)((a2 + i) * 4 + (a2 + j)) * w.ND + it];
                pab[i * 2 + j] = w.dyn_m[(size_t)((a2 + i) * 4 + (b2 + j)) * w.ND + it];
                pba[i * 2 + j] = w.dyn_m[(size_t)((b2 + i) * 4 + (a2 + j)) * w.ND + it];
                pbb[i * 2 + j] = w.dyn_m[(size_t)((b2 + i) * 4 + (b2 + j)) * w.ND + it];
            }
    };
    if (is_dyn) {
        const int f = lane % (K - 1), slot = lane / (K - 1);
        if (!PERSIST) load_dyn_potential(maa, mab, mba, mbb);
        dyn_other_var = f + 1 - slot;
        dyn_other_edge = (1 - slot) * (K - 1) + f;
    }
    // obstacle factors: four lanes per factor when they fit one wave and no tracking lanes are needed
    const bool obs_rows = (4 * (K - 2) <= 64) && !(w.enable & 8u);
    // UV wave, factor phase (otherwise): obstacle lanes [0, K-2), tracking lanes [K-2, 2(K-2))
    const int uvar = (is_trk ? lane - (K - 2) : lane) + 1;  // variable of the unary factor
    const int uedge = n_dyn + lane;                         // its internal-edge column
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

    // ---- operands of the external factor sweep, requested AROUND the staging copies ------------------
    // An edge lane needs a chain of dependent loads (gate / constants -> the owner's delivery count and
    // snapshot record); started here they travel while the blob is staged, instead of after the barrier.
    // Each thread prefetches its first edge (robots have at most a few more edges than threads).
    const bool do_extf = !PERSIST && HAS_IR && (ext_mask & PH_EXT_FACTOR) && radio && ir_on;
    bool pf_on = false, pf_present = false;
    IrEdgeRec pf_er{};
    double pf_bmu[4] = {0.0, 0.0, 0.0, 0.0}, pf_rec[SNAP_W];
#pragma unroll
    for (int c = 0; c < SNAP_W; c++) pf_rec[c] = 0.0;
    uint8_t pf_gate = 0;
    int pf_dst = 0;
    if (HAS_IR && tid < ne) pf_gate = w.ir_gate[ie0 + tid];
    if (do_extf && tid < ne) {
        pf_er = w.ir_rec[ie0 + tid];
        pf_dst = pf_er.dst;
        ld_soa4(w.ir_bmu, w.NI, ie0 + tid, pf_bmu);
    }
    if (PERSIST && tid < ne) {  // resident launches: the edge's constants stay in registers for every external iteration
        pf_er = w.ir_rec[ie0 + tid];
        pf_dst = pf_er.dst;
    }
    // The owners' records of the threads' edges are fetched by QUADS of lanes: for each of its four lanes' records in turn, lane v
    // of a quad asks for bytes [64 t + 16 v, + 16), t = 0..2 — the four requests of a quad are one contiguous 64 bytes — and a
    // 4 x 4 transposition inside the quad hands every lane its own record.  One lane fetching its own 192 bytes makes sixty-four
    // scattered 16-byte requests per load instruction, and the gather was bound by their number: 0.45 us per 16 bytes per lane
    // at 1000 robots, 5 us of an iteration (experiments/README.md); the same bytes by quads take a quarter of that.  Every lane of
    // the wave takes part (a lane without an edge passes offset 0: record 0 is fetched and dropped).
    auto quad_gather = [&](auto fetch, unsigned ro_mine, double (&out)[SNAP_W]) __attribute__((always_inline)) {
        v4u32 R[4][3];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            unsigned rb;  // the record of lane s4 of the quad
            if (s4 == 0) rb = (unsigned)__builtin_amdgcn_mov_dpp((int)ro_mine, 0x00, 0xf, 0xf, true);
            else if (s4 == 1) rb = (unsigned)__builtin_amdgcn_mov_dpp((int)ro_mine, 0x55, 0xf, 0xf, true);
            else if (s4 == 2) rb = (unsigned)__builtin_amdgcn_mov_dpp((int)ro_mine, 0xAA, 0xf, 0xf, true);
            else rb = (unsigned)__builtin_amdgcn_mov_dpp((int)ro_mine, 0xFF, 0xf, 0xf, true);
#pragma unroll
            for (int t3 = 0; t3 < 3; t3++) R[s4][t3] = fetch(rb + 64u * t3 + 16u * (unsigned)(lane & 3));
        }
#pragma unroll
        for (int t3 = 0; t3 < 3; t3++) {
            unsigned dw[4][4];  // [dword of the 16 bytes][piece v of the lane's own record]
#pragma unroll
            for (int wd = 0; wd < 4; wd++) {
                unsigned x[4] = {R[0][t3][wd], R[1][t3][wd], R[2][t3][wd], R[3][t3][wd]};
                quad_transpose4(x, lane);
#pragma unroll
                for (int v = 0; v < 4; v++) dw[wd][v] = x[v];
            }
#pragma unroll
            for (int v = 0; v < 4; v++) {
                out[2 * (4 * t3 + v)] = __hiloint2double((int)dw[1][v], (int)dw[0][v]);
                out[2 * (4 * t3 + v) + 1] = __hiloint2double((int)dw[3][v], (int)dw[2][v]);
            }
        }
    };
    auto fetch_plain = [&](unsigned off) __attribute__((always_inline)) {  // records written by an earlier launch
        return *reinterpret_cast<const v4u32 *>(reinterpret_cast<const char *>(w.snap[w.cur]) + off);
    };
    // per-variable words (K <= 64 < threads: one pass)
    uint32_t r_epoch = 0;
    int r_irp[3] = {0, 0, 0};
    if (tid < K) {
        r_epoch = w.snap_epoch[w.cur][v0 + tid];
        if (HAS_IR) {
            r_irp[0] = w.ir_var_ptr[v0 + tid];
            r_irp[1] = w.ir_var_mid[v0 + tid];
            r_irp[2] = w.ir_var_ptr[v0 + tid + 1];
        }
    }
    // mgx_tick: this robot's prior updates (waypoint, time scale, what) and, lane c < 20 of each wave, entry c of
    // the belief (eta, lam) the variable it updates holds in HBM — wave 0: the horizon variable, wave 1: variable 0
    double u_rec[4] = {0.0, 0.0, 0.0, 0.0}, u_bel = 0.0;
    if (w.upd) {
#pragma unroll
        for (int c = 0; c < 4; c++) u_rec[c] = w.upd[(size_t)r * 4 + c];
        if (lane < 20) u_bel = blob[L.bel() + lane * K + (role == 0 ? K - 1 : 0)];
    }
    // messages that this launch's external factor sweep recomputes before anyone reads them are not fetched
    const bool recompute = (PERSIST ? plan.ext[0] != 0 : (ext_mask & PH_EXT_FACTOR) != 0) && radio && ir_on;
    double r_ir[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    bool r_ir_on = false;

    // ---- stage the robot in LDS (all 128 threads) -----------------------------------------------
    // Constant-K instantiations: TWO memory round trips for the whole stage.  First every load that
    // needs no other load's result (prior, mean | messages | valid, own snapshot records, and above the
    // first link of the edge chain), then the chain's second link, then the LDS stores.  The covariance
    // is output only (a variable whose covariance this launch does not recompute keeps the HBM copy).
    auto chain_second_link = [&]() {
        if (STAGE_IR && tid < ne && !(recompute && pf_gate == 1)) {  // the thread's first staged inter-robot message
            const size_t e = (size_t)(ie0 + tid);
            r_ir_on = true;
            r_ir[0] = w.ir_fv_eta[0 * (size_t)w.NI + e];
            r_ir[1] = w.ir_fv_eta[1 * (size_t)w.NI + e];
            r_ir[2] = w.ir_fv_lam[0 * (size_t)w.NI + e];
            r_ir[3] = w.ir_fv_lam[1 * (size_t)w.NI + e];
            r_ir[4] = w.ir_fv_lam[4 * (size_t)w.NI + e];
            r_ir[5] = w.ir_fv_lam[5 * (size_t)w.NI + e];
        }
        if (do_extf) {  // the owners' records (other robots, HBM / L2); absent (not yet delivered) ones are zeroed when used
            pf_on = tid < ne && pf_gate == 1;
            if (pf_on) pf_present = w.snap_epoch[w.cur][pf_er.src_var] > pf_er.created;
            quad_gather(fetch_plain, pf_on ? (unsigned)pf_er.src_var * (unsigned)(SNAP_W * sizeof(double)) : 0u, pf_rec);
        }
    };
    {
        const double *src = w.snap[w.cur] + (size_t)v0 * SNAP_W;
        if constexpr (KT > 0) {
            StageRegs<20 * KT> r_prior;
            StageRegs<BlobLayout(KT).inout_words() - 16 * KT> r_io;
            constexpr int IT = (SNAP_W * KT + SWEEP_BLOCK - 1) / SWEEP_BLOCK;
            double r_snap[IT];
            r_prior.load(blob + L.prior(), tid);
            r_io.load(blob + L.mu(), tid);
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int t = tid + it * SWEEP_BLOCK;
                r_snap[it] = (t < SNAP_W * K) ? src[t] : 0.0;
            }
            chain_second_link();
            r_prior.store(s_prior, tid);
            r_io.store(s_mu, tid);
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int t = tid + it * SWEEP_BLOCK;
                if (t < SNAP_W * K) s_snap[(t % SNAP_W) * K + (t / SNAP_W)] = r_snap[it];
            }
        } else {
            copy_words(s_prior, blob + L.prior(), 20 * K, tid);
            chain_second_link();
            copy_words(s_mu, blob + L.mu(), L.inout_words() - 16 * K, tid);
            for (int t = tid; t < SNAP_W * K; t += SWEEP_BLOCK) s_snap[(t % SNAP_W) * K + (t / SNAP_W)] = src[t];
        }
        for (int t = tid; t < K; t += SWEEP_BLOCK) s_covset[t] = 0;
        if (STAGE_IR && tid < IR_STRIDE) s_ir[ne * IR_STRIDE + tid] = 0.0;  // the all-zero message behind the last edge
        if (tid < K) {
            s_epoch[tid] = r_epoch;
            if (HAS_IR) {
                s_irp[tid] = r_irp[0];
                s_irp[(K + 1) + tid] = r_irp[1];
                s_irp[2 * (K + 1) + tid] = r_irp[2];
            }
        }
        if (STAGE_IR) {
            if (r_ir_on) {
#pragma unroll
                for (int c = 0; c < 6; c++) s_ir[tid * IR_STRIDE + c] = r_ir[c];
            }
            for (int j = tid + SWEEP_BLOCK; j < ne; j += SWEEP_BLOCK) {  // robots with more edges than threads
                if (recompute && w.ir_gate[ie0 + j] == 1) continue;
                const size_t e = (size_t)(ie0 + j);
                double m[6];
                m[0] = w.ir_fv_eta[0 * (size_t)w.NI + e];
                m[1] = w.ir_fv_eta[1 * (size_t)w.NI + e];
                m[2] = w.ir_fv_lam[0 * (size_t)w.NI + e];
                m[3] = w.ir_fv_lam[1 * (size_t)w.NI + e];
                m[4] = w.ir_fv_lam[4 * (size_t)w.NI + e];
                m[5] = w.ir_fv_lam[5 * (size_t)w.NI + e];
#pragma unroll
                for (int c = 0; c < 6; c++) s_ir[j * IR_STRIDE + c] = m[c];
            }
        }
    }
    __syncthreads();
    // ---- mgx_tick: update_prior_of_horizon_state (wave 0, variable K-1) and update_prior_of_current_state_v3
    // (wave 1, variable 0) on the staged image, each ending in change_prior of that variable
    // (robot.rs:2182-2338, variable.rs:203-230; same arithmetic as k_update_priors / apply_change_prior).  For
    // K >= 3 the two touch disjoint state, and nobody else reads this robot's snapshot in a launch without an
    // external factor sweep, so the change needs no other synchronisation than the barrier below.
    if (w.upd) {
        const uint32_t what = (uint32_t)u_rec[3];
        const int i = role == 0 ? K - 1 : 0;
        if (role == 0 ? (what & 1u) : (what & 2u)) {
            double m[4];
            if (role == 0) {
                const double ex = s_mu[0 * K + i], ey = s_mu[1 * K + i];         // estimated position (:2242)
                double hx = u_rec[0] - ex, hy = u_rec[1] - ey;                    // horizon2waypoint
                const double dist = std::sqrt(hx * hx + hy * hy);                 // euclidean_norm
                double nx = hx, ny = hy;                                           // .normalized(): unchanged if |.| is 0 / inf
                if (!(dist == 0.0 || std::isinf(dist))) { nx = hx / dist; ny = hy / dist; }
                const double sp = (w.upd_max_speed < dist || dist != dist) ? w.upd_max_speed : dist;  // Float::min(max_speed, dist)
                const double vx = sp * nx, vy = sp * ny;                           // new_velocity
                m[0] = ex + vx * w.upd_delta_t; m[1] = ey + vy * w.upd_delta_t; m[2] = vx; m[3] = vy;  // (:2253-2256)
            } else {
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const double m0 = s_mu[c * K + 0], m1 = s_mu[c * K + 1];
                    m[c] = m0 + u_rec[2] * (m1 - m0);                              // (:2309-2316)
                }
            }
            __builtin_amdgcn_wave_barrier();  // every lane has read the means before lanes 4..7 overwrite them
            if (lane < 4) {  // prior eta = prior lam . mean (:204), in LDS and in the blob (the sweep never writes priors back)
                double pl[4];
#pragma unroll
                for (int c = 0; c < 4; c++) pl[c] = s_prior[(4 + lane * 4 + c) * K + i];
                const double pe = ((pl[0] * m[0] + pl[1] * m[1]) + pl[2] * m[2]) + pl[3] * m[3];
                s_prior[lane * K + i] = pe;
                blob[L.prior() + lane * K + i] = pe;
            } else if (lane < 8) {  // belief mean (:206) and the mean of the message the variable sends (:210-221)
                const double mc = lane == 4 ? m[0] : (lane == 5 ? m[1] : (lane == 6 ? m[2] : m[3]));
                s_mu[(lane - 4) * K + i] = mc;
                s_snap[(20 + lane - 4) * K + i] = mc;
            } else if (lane == 8) {
                s_epoch[i] += 1;
            }
            if (lane < 20) s_snap[lane * K + i] = u_bel;  // (stale eta, stale lam) of that message
            // every inbox message of the variable becomes empty (:224-227)
            const int es[4] = {(i >= 1) ? (K - 1) + (i - 1) : -1, (i <= K - 2) ? i : -1,
                               (i >= 1 && i <= K - 2) ? n_dyn + (i - 1) : -1, (i >= 1 && i <= K - 2) ? n_dyn + (K - 2) + (i - 1) : -1};
            for (int t = lane; t < 80; t += 64) {
                const int col = es[t & 3];
                if (col >= 0) s_fv[(t >> 2) * E1 + col] = 0.0;
            }
            if (HAS_IR) {  // foreign inter-robot factors attached to the variable: their message goes, they get the new mean
                const int x0 = w.ir_var_ptr[v0 + i], x1 = w.ir_var_ptr[v0 + i + 1];
                for (int e = x0 + lane; e < x1; e += 64) {
                    if (w.enable & 2u) {
#pragma unroll
                        for (int c = 0; c < 4; c++) w.ir_bmu[(size_t)c * w.NI + e] = m[c];
                    }
                    w.ir_fv_eta[0 * (size_t)w.NI + e] = 0.0;
                    w.ir_fv_eta[1 * (size_t)w.NI + e] = 0.0;
                    w.ir_fv_lam[0 * (size_t)w.NI + e] = 0.0;
                    w.ir_fv_lam[1 * (size_t)w.NI + e] = 0.0;
                    w.ir_fv_lam[4 * (size_t)w.NI + e] = 0.0;
                    w.ir_fv_lam[5 * (size_t)w.NI + e] = 0.0;
                    if (STAGE_IR) {
#pragma unroll
                        for (int c = 0; c < 6; c++) s_ir[(e - ie0) * IR_STRIDE + c] = 0.0;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (PERSIST) {  // columns no sweep recomputes (disabled kinds, tracking in front of its gate) must be equal in both
        for (int t = tid; t < 20 * E1; t += SWEEP_BLOCK) s_sh[t] = s_fv[t];
        __syncthreads();
    }
    uint32_t my_epoch = (sum_t < 4 * K) ? s_epoch[sum_t % K] : 0u;  // deliveries of the variable this thread sums
    STAMP(t_staged);

    // ======================= external factor sweep (pull form) ================================
    // factorgraph.rs:745-754 keeps only the message to the other graph's variable, so F_AB is
    // evaluated here, at B, from A's snapshot record and B's last response mean.
    // k: segment of a resident schedule launch (0 otherwise); store_fv: the HBM copy of the messages is needed
    // descriptors of the two snapshot buffers for the 16-byte agent-scope accesses of resident launches
    const unsigned snap_bytes = (unsigned)w.V * (unsigned)(SNAP_W * sizeof(double));
    const __amdgpu_buffer_rsrc_t rs_snap[2] = {sc1_rsrc(w.snap[0], PERSIST ? snap_bytes : 0u), sc1_rsrc(w.snap[1], PERSIST ? snap_bytes : 0u)};
    // resident launches: the response means (ir_bmu) of every edge of variable i that is on air are the variable's mean after the
    // external variable sweep — kept in LDS (the scratch sums' block, idle between that sweep's finish and the next one's sums)
    // from one segment to the next; HBM gets them once, after the launch's last external iteration
    double *s_xmu = s_tmp;
    bool have_xmu = false;
#ifdef MGX_STAMPS
    unsigned long long q_arrive = 0ull, t_edges0 = 0ull;  // cycles from the start of the factor sweep until the thread's record is there
#endif
    auto external_factor_sweep = [&](int k, bool store_fv) __attribute__((always_inline)) {
        const int buf = PERSIST ? ((w.cur + k) & 1) : w.cur;  // snapshot buffer the owners' records are read from
#ifdef MGX_STAMPS
        t_edges0 = __builtin_readcyclecounter();
#endif
        if (radio && ir_on) {
            auto fetch_sc1 = [&](unsigned off) __attribute__((always_inline)) { return ld16_agent_raw(rs_snap[buf], off); };
            for (int j0 = 0; j0 < ne; j0 += SWEEP_BLOCK) {  // rounds of the whole workgroup: every lane takes part in the gather
                const int j = j0 + tid;
                const int e = ie0 + j;
                IrEdgeRec er{};
                double ao_eta[4], ao_lam[16], a_mu[4], b_mu[4];
                bool a_present;
                // the round's record, unless it was prefetched while staging (launch-per-segment path, first round)
                double grec[SNAP_W];
                bool mine = false;
                if (PERSIST || j0 > 0) {
                    if (j0 == 0) {
                        mine = tid < ne && pf_gate == 1;
                        if (mine) er = pf_er;
                    } else if (j < ne && w.ir_gate[e] == 1) {  // robots with more edges than threads
                        mine = true;
                        er = w.ir_rec[e];
                    }
                    const unsigned ro_mine = mine ? (unsigned)er.src_var * (unsigned)(SNAP_W * sizeof(double)) : 0u;
                    if (PERSIST) quad_gather(fetch_sc1, ro_mine, grec);  // published by other workgroups of THIS launch: agent scope
                    else quad_gather(fetch_plain, ro_mine, grec);
                    if (!mine) continue;
                }

                if (PERSIST) {
                    if (have_xmu) {  // the means this robot's external variable sweep of the previous segment answered with
                        const int i = er.dst & 0xffff;
#pragma unroll
                        for (int c = 0; c < 4; c++) b_mu[c] = s_xmu[c * K + i];
                    } else {
                        ld_soa4(w.ir_bmu, w.NI, e, b_mu);  // written by an earlier launch
                    }
                    // has the owner's variable answered this factor yet?  Once it has it stays so: the thread's first edge
                    // remembers (one scattered 4-byte request per lane and iteration less)
                    if (j0 == 0 && pf_present) {
                        a_present = true;
                    } else {
                        a_present = ld_agent(&w.snap_epoch[buf][er.src_var]) > er.created;
                        if (j0 == 0) pf_present = a_present;
                    }
#ifdef MGX_STAMPS
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (j0 == 0) { const unsigned long long _n = __builtin_readcyclecounter(); q_arrive += _n - t_edges0; }
#endif
#pragma unroll
                    for (int c = 0; c < 4; c++) ao_eta[c] = grec[c];
#pragma unroll
                    for (int c = 0; c < 16; c++) ao_lam[c] = grec[4 + c];
#pragma unroll
                    for (int c = 0; c < 4; c++) a_mu[c] = grec[20 + c];
                } else if (j0 == 0) {  // operands prefetched during staging
                    if (j >= ne) continue;
                    if (!pf_on) continue;  // the owner did not run its external factor sweep
                    er = pf_er;
                    a_present = pf_present;
#pragma unroll
                    for (int c = 0; c < 4; c++) b_mu[c] = pf_bmu[c];
#pragma unroll
                    for (int c = 0; c < 4; c++) ao_eta[c] = pf_rec[c];
#pragma unroll
                    for (int c = 0; c < 16; c++) ao_lam[c] = pf_rec[4 + c];
#pragma unroll
                    for (int c = 0; c < 4; c++) a_mu[c] = pf_rec[20 + c];
                } else {
                    ld_soa4(w.ir_bmu, w.NI, e, b_mu);
                    a_present = w.snap_epoch[w.cur][er.src_var] > er.created;
#pragma unroll
                    for (int c = 0; c < 4; c++) ao_eta[c] = grec[c];
#pragma unroll
                    for (int c = 0; c < 16; c++) ao_lam[c] = grec[4 + c];
#pragma unroll
                    for (int c = 0; c < 4; c++) a_mu[c] = grec[20 + c];
                }
                if (!a_present) {  // the owner's variable has not answered this factor yet: empty inbox entry
#pragma unroll
                    for (int c = 0; c < 4; c++) { ao_eta[c] = 0.0; a_mu[c] = 0.0; }
#pragma unroll
                    for (int c = 0; c < 16; c++) ao_lam[c] = 0.0;
                }
                // ONE evaluation for both slot orders (selects on the linearisation point): the lanes of a wave hold edges of
                // both orders, and a branch around two inlined copies would run both for every wave
                const int dslot = er.dst >> 16;
                double x_lo[4], x_hi[4], o6[6];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    x_lo[c] = dslot ? a_mu[c] : b_mu[c];
                    x_hi[c] = dslot ? b_mu[c] : a_mu[c];
                }
                if (!interrobot_message_compact(x_lo, x_hi, er.d_safe, er.offset, w.inv_s2_ir, dslot, ao_eta, ao_lam, o6)) {
#pragma unroll
                    for (int c = 0; c < 6; c++) o6[c] = 0.0;
                }
                // HINT_IR_DEAD: the caller's next sweep recomputes these messages before reading them (it
                // starts with an external factor sweep under the same flags), and this launch reads
                // them from LDS — then the HBM copy is dead and not stored; likewise every external
                // iteration of a resident schedule launch but its last one
                if (store_fv && !(STAGE_IR && (hints & HINT_IR_DEAD))) {
                    w.ir_fv_eta[0 * (size_t)w.NI + e] = o6[0];
                    w.ir_fv_eta[1 * (size_t)w.NI + e] = o6[1];
                    w.ir_fv_lam[0 * (size_t)w.NI + e] = o6[2];
                    w.ir_fv_lam[1 * (size_t)w.NI + e] = o6[3];
                    w.ir_fv_lam[4 * (size_t)w.NI + e] = o6[4];
                    w.ir_fv_lam[5 * (size_t)w.NI + e] = o6[5];
                }
                if (STAGE_IR) {
                    double *p = s_ir + j * IR_STRIDE;
#pragma unroll
                    for (int c = 0; c < 6; c++) p[c] = o6[c];
                }
            }
        }
        if (radio) itf += 1;  // iteration_count.factor of the robot's own external sweep (factorgraph.rs:757)
    };

    // Inbox sums of a variable sweep, one lane per (variable, row): lane (i, rr) accumulates eta[rr] and
    // lam[rr][0..3] in the reference's inbox order (BTreeMap<FactorId, _>, id.rs:19-54): factors of
    // graphs with a lower key, own factors by node index (dynamic i-1, dynamic i, obstacle, tracking;
    // own inter-robot factors are forever empty), then factors of graphs with a higher key — each
    // element sees exactly the additions of VariableNode::update_belief... (variable.rs:254-271).
    // Absent edges read the all-zero column: x + 0.0 == x exactly and a running sum that starts from
    // the prior is never -0.0, so this equals skipping the entry.  For an internal sweep the sums are
    // also the (eta, lam) of the responses to own-graph factors (:301-330, factorgraph.rs:771-786).
    // s_out receives the sums ([20][K] image: the snapshot for internal sweeps).
    auto variable_sums_from = [&](int t_first, int t_step, double *s_out, bool internal, bool last) __attribute__((always_inline)) {
        for (int t = t_first; t < 4 * K; t += t_step) {
            const int rr = t / K, i = t - rr * K;  // consecutive lanes -> consecutive variables: conflict-free LDS rows
            uint32_t epoch_reg = (4 * K <= SWEEP_BLOCK) ? my_epoch : s_epoch[i];
            const int es[4] = {(i >= 1) ? (K - 1) + (i - 1) : ZCOL, (i <= K - 2) ? i : ZCOL,
                               (i >= 1 && i <= K - 2) ? n_dyn + (i - 1) : ZCOL,
                               (i >= 1 && i <= K - 2) ? n_dyn + (K - 2) + (i - 1) : ZCOL};
            // all 25 LDS operands of the own-graph part are fetched before the first add: one LDS
            // round trip instead of one per message
            double pr[5], ms[4][5];
            pr[0] = s_prior[rr * K + i];
#pragma unroll
            for (int c = 0; c < 4; c++) pr[1 + c] = s_prior[(4 + rr * 4 + c) * K + i];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                ms[q][0] = s_fv[rr * E1 + es[q]];
#pragma unroll
                for (int c = 0; c < 4; c++) ms[q][1 + c] = s_fv[(4 + rr * 4 + c) * E1 + es[q]];
            }
            double acc[5];  // eta[rr], lam[rr][0..3]
#pragma unroll
            for (int c = 0; c < 5; c++) acc[c] = pr[c];
            const int x0 = HAS_IR ? s_irp[i] : 0, xm = HAS_IR ? s_irp[(K + 1) + i] : 0, x1 = HAS_IR ? s_irp[2 * (K + 1) + i] : 0;
            auto ir_rows = [&](int e_from, int e_to) {
                // compact messages: rows 0, 1 add eta[rr], lam[rr][0], lam[rr][1]; rows 2, 3 only zeros
                if (!HAS_IR || rr >= 2) return;
                if (STAGE_IR) {
                    // staged messages: a batch that runs past the end reads the all-zero slot behind the robot's last edge
                    // (x + 0.0 == x, as for the absent own edges above) — no clamps, no conditional adds, 32-bit LDS offsets
                    for (int e = e_from; e < e_to; e += 4) {  // four messages are fetched before the adds
                        double m[4][3];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int b = ((e + u < e_to) ? e + u - ie0 : ne) * IR_STRIDE;
                            m[u][0] = s_ir[b + rr];
                            m[u][1] = s_ir[b + 2 + 2 * rr];
                            m[u][2] = s_ir[b + 3 + 2 * rr];
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            acc[0] += m[u][0];
                            acc[1] += m[u][1];
                            acc[2] += m[u][2];
                        }
                    }
                    return;
                }
                for (int e = e_from; e < e_to; e += 4) {  // four messages are fetched before the adds
                    double m[4][3];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int ee = (e + u < e_to) ? e + u : e;  // clamp: the value is not added
                        if (STAGE_IR) {
                            const double *p = s_ir + (ee - ie0) * IR_STRIDE;
                            m[u][0] = p[rr];
                            m[u][1] = p[2 + 2 * rr];
                            m[u][2] = p[3 + 2 * rr];
                        } else {
                            m[u][0] = w.ir_fv_eta[(size_t)rr * w.NI + ee];
                            m[u][1] = w.ir_fv_lam[(size_t)(rr * 4) * w.NI + ee];
                            m[u][2] = w.ir_fv_lam[(size_t)(rr * 4 + 1) * w.NI + ee];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (e + u < e_to) {
                            acc[0] += m[u][0];
                            acc[1] += m[u][1];
                            acc[2] += m[u][2];
                        }
                }
            };
            ir_rows(x0, xm);
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int c = 0; c < 5; c++) acc[c] += ms[q][c];
            ir_rows(xm, x1);
            s_out[rr * K + i] = acc[0];
#pragma unroll
            for (int c = 0; c < 4; c++) s_out[(4 + rr * 4 + c) * K + i] = acc[1 + c];
            if (internal && rr == 0) {
                s_epoch[i] = ++epoch_reg;  // register copy when each thread owns one (variable, row)
                if (4 * K <= SWEEP_BLOCK) my_epoch = epoch_reg;
            }
            if (last && s_out != s_prior) {  // the prior is not needed again in this launch: its LDS image
                                             // carries the belief (eta, lam) to the write-back
                s_prior[rr * K + i] = acc[0];
#pragma unroll
                for (int c = 0; c < 4; c++) s_prior[(4 + rr * 4 + c) * K + i] = acc[1 + c];
            }
        }
    };
    auto variable_sums = [&](double *s_out, bool internal, bool last) __attribute__((always_inline)) {
        variable_sums_from(sum_t, sum_step, s_out, internal, last);
    };
    // Second half of the variable update: covariance, validity and mean from (eta, lam)
    // (variable.rs:273-297), one lane per variable; for an internal sweep the mean also completes the
    // snapshot (the responses' mean, :317).
    auto variable_finish = [&](const double *s_in, bool internal) {
        double b_eta[4], b_lam[16], mu[4], cov[16];
#pragma unroll
        for (int c = 0; c < 4; c++) b_eta[c] = s_in[c * K + lane];
#pragma unroll
        for (int c = 0; c < 16; c++) b_lam[c] = s_in[(4 + c) * K + lane];
#pragma unroll
        for (int c = 0; c < 4; c++) mu[c] = s_mu[c * K + lane];
        int valid = s_valid[lane];
        if (belief_update(b_eta, b_lam, mu, cov, valid)) {  // covariance (and maybe mean) changed
#pragma unroll
            for (int c = 0; c < 16; c++) s_cov[c * K + lane] = cov[c];
#pragma unroll
            for (int c = 0; c < 4; c++) s_mu[c * K + lane] = mu[c];
            s_valid[lane] = valid;
            s_covset[lane] = 1;
        }
        if (internal) {
#pragma unroll
            for (int c = 0; c < 4; c++) s_snap[(20 + c) * K + lane] = mu[c];
        }
    };

    // The same on four lanes per variable (UV wave, lane = q * K + i): lane (q, i) computes the cofactors of row q from the three
    // other rows — the expression inv4 evaluates for that row — i.e. column q of the covariance; the determinant comes from the
    // q == 0 lane, the four columns meet in the covariance image in LDS, and lane (q, i) reads row q back for component q of
    // the mean.  Every number is produced by the operations of belief_update in the same order.
    // core: (eta, lam) of variable i from s_in; when the precision is neither "zero" nor singular (ok) the covariance goes to
    // cov_img ([16][K]) and, if it is finite (fin), mu_q becomes component q of the new mean
    auto quad_core = [&](const double *s_in, double *cov_img, bool &ok, bool &fin, double &mu_q) __attribute__((always_inline)) {
        const int q = lane / K, i = lane - q * K;
        double eta[4], lam[16];
#pragma unroll
        for (int c = 0; c < 4; c++) eta[c] = s_in[c * K + i];
#pragma unroll
        for (int c = 0; c < 16; c++) lam[c] = s_in[(4 + c) * K + i];
        bool not_zero = false;
#pragma unroll
        for (int c = 0; c < 16; c++) not_zero = not_zero || (lam[c] > 1e-6);
        double r0[4], r1[4], r2[4], mn[4], cf[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            r0[c] = (q == 0) ? lam[4 + c] : lam[c];
            r1[c] = (q <= 1) ? lam[8 + c] : lam[4 + c];
            r2[c] = (q <= 2) ? lam[12 + c] : lam[8 + c];
        }
        minors_of_removed_row(r0, r1, r2, mn);
#pragma unroll
        for (int j = 0; j < 4; j++) cf[j] = ((q + j) & 1) ? -mn[j] : mn[j];
        const double row0[4] = {lam[0], lam[1], lam[2], lam[3]};
        const double det = __shfl(det_from_row0(row0, cf), i, 64);  // lane i is (q == 0, i)
        ok = not_zero && det != 0.0;
        const double id = 1.0 / det;
        double col[4];  // cov[j][q]
        bool fin_own = true;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            col[j] = cf[j] * id;
            fin_own = fin_own && std::isfinite(col[j]);
        }
        unsigned long long bad = __ballot(!fin_own);
        constexpr int KS = QUADFIN ? KT : 1;  // (instantiated, never run, for longer horizons: keep the shift counts in range)
        bad |= (bad >> KS) | (bad >> (2 * KS)) | (bad >> (3 * KS));
        fin = ((bad >> i) & 1ull) == 0ull;
        if (ok) {
#pragma unroll
            for (int j = 0; j < 4; j++) cov_img[(j * 4 + q) * K + i] = col[j];
            __builtin_amdgcn_wave_barrier();  // one wave: its LDS reads below follow its LDS writes above
            if (fin) {
                double cq[4];
#pragma unroll
                for (int c = 0; c < 4; c++) cq[c] = cov_img[(q * 4 + c) * K + i];
                mu_q = ((cq[0] * eta[0] + cq[1] * eta[1]) + cq[2] * eta[2]) + cq[3] * eta[3];
            }
        }
    };
    auto variable_finish_quad = [&](const double *s_in, bool internal) {
        const int q = lane / K, i = lane - q * K;
        double mu_q = s_mu[q * K + i];
        bool ok, fin;
        quad_core(s_in, s_cov, ok, fin, mu_q);
        if (ok) {
            if (fin) s_mu[q * K + i] = mu_q;
            if (q == 0) {
                s_valid[i] = fin ? 1 : 0;
                s_covset[i] = 1;
            }
        }
        if (internal) s_snap[(20 + q) * K + i] = mu_q;
    };
    auto finish = [&](const double *s_in, bool internal) __attribute__((always_inline)) {
        if constexpr (QUADFIN) {
            if (role == ROLE_UV && lane < 4 * K) variable_finish_quad(s_in, internal);
        } else {
            if (is_var) variable_finish(s_in, internal);
        }
    };

    // Internal factor sweep, DYN wave: one lane per dynamic-factor message.  The two messages of one
    // factor read each other's previous value; both lanes sit in the SAME wave, whose LDS reads all
    // issue before its LDS writes, so no barrier is needed between reading the old and writing the new
    // messages.  Reads s_snap, s_epoch and the dynamic columns of s_fv only.
    auto dynamic_messages = [&](double *s_out) {
        double me[4], ml[16], oe[4], ol[16];
        const int o = dyn_other_var, oe_ix = dyn_other_edge;
        double paa[4], pab[4], pba[4], pbb[4];
        if (PERSIST) {
            load_dyn_potential(paa, pab, pba, pbb);  // requested first: they travel under the LDS reads below
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) { paa[c] = maa[c]; pab[c] = mab[c]; pba[c] = mba[c]; pbb[c] = mbb[c]; }
        }
        if (s_epoch[o] > 0) {  // other variable has answered: belief - our last message
#pragma unroll
            for (int c = 0; c < 4; c++) me[c] = s_snap[c * K + o] - s_fv[c * E1 + oe_ix];
#pragma unroll
            for (int c = 0; c < 16; c++) ml[c] = s_snap[(4 + c) * K + o] - s_fv[(4 + c) * E1 + oe_ix];
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) me[c] = 0.0;
#pragma unroll
            for (int c = 0; c < 16; c++) ml[c] = 0.0;
        }
        // Horizons beyond 33 variables (BIG instantiations only) have more messages than the wave has lanes:
        // lane l also computes message l + 64.  Its operands are read here, before ANY message of this sweep
        // is written — the partner of a second-pass message may be a first-pass message.
        double me2[4], ml2[16];
        const int lane2 = lane + 64;
        const bool has2 = BIG && lane2 < n_dyn;
        if (has2) {
            const int f2 = lane2 % (K - 1), slot2 = lane2 / (K - 1), o2 = f2 + 1 - slot2, oe2 = (1 - slot2) * (K - 1) + f2;
            const bool pres = s_epoch[o2] > 0;
#pragma unroll
            for (int c = 0; c < 4; c++) me2[c] = pres ? s_snap[c * K + o2] - s_fv[c * E1 + oe2] : 0.0;
#pragma unroll
            for (int c = 0; c < 16; c++) ml2[c] = pres ? s_snap[(4 + c) * K + o2] - s_fv[(4 + c) * E1 + oe2] : 0.0;
        }
        if (!dynamic_message(paa, pab, pba, pbb, me, ml, oe, ol)) {
#pragma unroll
            for (int c = 0; c < 4; c++) oe[c] = 0.0;
#pragma unroll
            for (int c = 0; c < 16; c++) ol[c] = 0.0;
        }
#pragma unroll
        for (int c = 0; c < 4; c++) s_out[c * E1 + lane] = oe[c];
#pragma unroll
        for (int c = 0; c < 16; c++) s_out[(4 + c) * E1 + lane] = ol[c];
        if (has2) {  // potential blocks of the second message straight from HBM / L2: a rare shape, not worth registers
            const int f2 = lane2 % (K - 1), slot2 = lane2 / (K - 1), a2 = 2 * slot2, b2 = 2 * (1 - slot2), it2 = r * (K - 1) + f2;
            double naa[4], nab[4], nba[4], nbb[4];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    naa[i * 2 + j] = w.dyn_m[(size_t)((a2 + i) * 4 + (a2 + j)) * w.ND + it2];
                    nab[i * 2 + j] = w.dyn_m[(size_t)((a2 + i) * 4 + (b2 + j)) * w.ND + it2];
                    nba[i * 2 + j] = w.dyn_m[(size_t)((b2 + i) * 4 + (a2 + j)) * w.ND + it2];
                    nbb[i * 2 + j] = w.dyn_m[(size_t)((b2 + i) * 4 + (b2 + j)) * w.ND + it2];
                }
            if (!dynamic_message(naa, nab, nba, nbb, me2, ml2, oe, ol)) {
#pragma unroll
                for (int c = 0; c < 4; c++) oe[c] = 0.0;
#pragma unroll
                for (int c = 0; c < 16; c++) ol[c] = 0.0;
            }
#pragma unroll
            for (int c = 0; c < 4; c++) s_out[c * E1 + lane2] = oe[c];
#pragma unroll
            for (int c = 0; c < 16; c++) s_out[(4 + c) * E1 + lane2] = ol[c];
        }
    };

    // Unary factors of an internal factor sweep, UV wave: obstacle factors (four lanes each when they fit) and
    // tracking factors.  Reads the snapshot means and delivery counts, writes its own message columns.
    // skip: kinds whose sweep k_thaw has already computed (first sweep after mgx_set_enabled only).
    const SdfView sdf = make_sdf_view(w.sdf, w.sdf_w, w.sdf_h, w.world_w, w.world_h);
    auto unary_messages = [&](uint32_t skip, double *s_out, int itf_gate) __attribute__((always_inline)) {
        if (obs_rows) {
            // four lanes per obstacle factor: lane q samples tap q and writes row q of the message
            if (role == ROLE_UV && lane < 4 * (K - 2) && (w.enable & 4u) && !(skip & 4u)) {
                const int j = lane >> 2, q = lane & 3, var = j + 1, col = n_dyn + j;
                double x0[4];
                const bool pres = s_epoch[var] > 0;
#pragma unroll
                for (int c = 0; c < 4; c++) x0[c] = pres ? s_snap[(20 + c) * K + var] : 0.0;
                const long long idx = obstacle_tap(sdf, x0[0], x0[1], w.obs_delta, q);
                const double hq = (idx >= 0) ? sdf_value(w.sdf[idx]) : 0.0;
                double h[4];
#pragma unroll
                for (int t = 0; t < 4; t++) h[t] = __shfl(hq, (lane & ~3) + t, 64);
                double eta_q, lam_q[4];
                obstacle_message_row(h, w.obs_delta, w.inv_s2_obs, x0, q, eta_q, lam_q);
                s_out[q * E1 + col] = eta_q;
#pragma unroll
                for (int c = 0; c < 4; c++) s_out[(4 + q * 4 + c) * E1 + col] = lam_q[c];
            }
        } else if (is_obs && (w.enable & 4u) && !(skip & 4u)) {
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
            for (int c = 0; c < 4; c++) s_out[c * E1 + uedge] = oe[c];
#pragma unroll
            for (int c = 0; c < 16; c++) s_out[(4 + c) * E1 + uedge] = ol[c];
        }
        if (is_trk && (w.enable & 8u) && itf_gate >= 10 && !(skip & 8u)) {  // factorgraph.rs:701
            double x0[4], oe[4], ol[16];
#pragma unroll
            for (int c = 0; c < 4; c++) x0[c] = s_snap[(20 + c) * K + uvar];
            const int p0 = w.path_ptr[r], np = w.path_ptr[r + 1] - p0;
            if (!tracking_update(w.path_xy + 2 * (size_t)p0, np, w.trk_pad, w.trk_attr, w.inv_s2_trk, x0, trk_rec,
                                  trk_lp, trk_lv, oe, ol)) {
#pragma unroll
                for (int c = 0; c < 4; c++) oe[c] = 0.0;
#pragma unroll
                for (int c = 0; c < 16; c++) ol[c] = 0.0;
            }
#pragma unroll
            for (int c = 0; c < 4; c++) s_out[c * E1 + uedge] = oe[c];
#pragma unroll
            for (int c = 0; c < 16; c++) s_out[(4 + c) * E1 + uedge] = ol[c];
        }
        // horizons beyond 33 variables: tracking factors K-2+64 .. 2(K-2)-1 have no lane of their own; lanes
        // 0 .. of the UV wave take them on, with their state in HBM (BIG instantiations only)
        if (BIG && role == ROLE_UV && lane + 64 >= K - 2 && lane + 64 < 2 * (K - 2) && (w.enable & 8u) && itf_gate >= 10 &&
            !(skip & 8u)) {
            const int j2 = lane + 64 - (K - 2), var2 = j2 + 1, col2 = n_dyn + (K - 2) + j2, item2 = r * (K - 2) + j2;
            double x0[4], oe[4], ol[16];
#pragma unroll
            for (int c = 0; c < 4; c++) x0[c] = s_snap[(20 + c) * K + var2];
            int rec2 = w.trk_record[item2];
            float lp2[2] = {w.trk_last_pos[item2], w.trk_last_pos[(size_t)w.NT + item2]};
            double lv2 = w.trk_last_val[item2];
            const int p0 = w.path_ptr[r], np = w.path_ptr[r + 1] - p0;
            if (!tracking_update(w.path_xy + 2 * (size_t)p0, np, w.trk_pad, w.trk_attr, w.inv_s2_trk, x0, rec2, lp2, lv2, oe, ol)) {
#pragma unroll
                for (int c = 0; c < 4; c++) oe[c] = 0.0;
#pragma unroll
                for (int c = 0; c < 16; c++) ol[c] = 0.0;
            }
            w.trk_record[item2] = rec2;
            w.trk_last_pos[item2] = lp2[0];
            w.trk_last_pos[(size_t)w.NT + item2] = lp2[1];
            w.trk_last_val[item2] = lv2;
#pragma unroll
            for (int c = 0; c < 4; c++) s_out[c * E1 + col2] = oe[c];
#pragma unroll
            for (int c = 0; c < 16; c++) s_out[(4 + c) * E1 + col2] = ol[c];
        }
    };

    // Resident schedule launch: in front of the external iteration of segment k, wait until every robot this one
    // exchanges snapshot records with (and that is on air) has completed segment k - 1 — its records for this
    // iteration are then published, and it has finished reading ours of the iteration before, whose buffer the
    // end of this segment overwrites.  One lane per peer polls that robot's progress word (relaxed agent-scope
    // loads, s_sleep in between); a wait that outlasts the wall-clock bound raises the world's abort word, which
    // releases every waiter: the launch then ends with wrong beliefs and the host reports it (never a hung GPU).
    // The peer list does not change during the launch: lane l of the polling wave keeps peer l (the one it polls in every
    // segment) in a register — looked up per segment, the three dependent loads in front of the first look at a progress
    // word (list range, peer, its antenna / idle flags) were a microsecond of pure latency on the hand-off.
    int my_peer = -1, peer_q0 = 0, peer_q1 = 0;
    if (PERSIST && radio && ir_on && role == ROLE_UV) {
        peer_q0 = w.peer_ptr[r];
        peer_q1 = w.peer_ptr[r + 1];
        if (peer_q0 + lane < peer_q1) {
            const int pr = w.peer_idx[peer_q0 + lane];
            if (w.antenna[pr] && !w.idle[pr]) my_peer = pr;  // not on air: neither reads our records nor has its own read
        }
    }
    auto wait_for_peers = [&](int k) __attribute__((always_inline)) {
        if (PERSIST && radio && ir_on && role == ROLE_UV) {
            const unsigned long long want = plan.flag_base + (unsigned long long)k;
            for (int q = peer_q0 + lane; q < peer_q1; q += 64) {
                int pr = my_peer;
                if (q != peer_q0 + lane) {  // more than 64 peers: the rest is looked up
                    pr = w.peer_idx[q];
                    if (!w.antenna[pr] || w.idle[pr]) pr = -1;
                }
                if (pr < 0) continue;
                const long long t0 = wall_clock64();
                unsigned spins = 0;
                while (__hip_atomic_load(&w.sweep_flag[pr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(4);
                    if ((++spins & 31u) == 0u) {
                        if (__hip_atomic_load(w.sweep_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) break;
                        if (wall_clock64() - t0 > plan.timeout_ticks) {
                            __hip_atomic_store(w.sweep_abort, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(w.sweep_err, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            break;
                        }
                    }
                }
            }
        }
    };

    bool prefired = false;  // both waves already ran the factor sweep of internal iteration 0
    bool par_done = false;  // resident, K <= 16: this segment's two variable sweeps ran side by side
    bool pending = false;   // the last internal sums still await their finish (mean, covariance)
    // resident launches: the factor sweep of the coming segment's first internal iteration has been computed into
    // s_sh at the end of the previous segment, under the publish / wait latency of the hand-off (it reads nothing an
    // external iteration produces: snapshots of the last INTERNAL variable sweep and the factors' own last messages)
    bool early = false;
    auto adopt_early = [&](int t0, int step) __attribute__((always_inline)) {  // s_sh -> s_fv, columns 0 .. E-1
        for (int t = t0; t < 20 * E; t += step) {
            const int c = t / E, e = t - c * E;
            s_fv[c * E1 + e] = s_sh[c * E1 + e];
        }
    };
    int last_int_seg = -1, last_ext_seg = -1;  // PERSIST: last segment with internal iterations / an external iteration
    if (PERSIST)
        for (int k = 0; k < nseg; k++) {
            if (plan.n_int[k] > 0) last_int_seg = k;
            if (plan.ext[k]) last_ext_seg = k;
        }
#ifdef MGX_STAMPS
    unsigned long long c_f = 0, c_fb = 0, c_v = 0, c_vb = 0, t_extf = t_staged, t_extv = t_staged, t_loop0 = t_staged;
    unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long p_wait = 0, p_extf = 0, p_extv = 0, p_int = 0, p_pub = 0;  // resident launches: cycles per stage, all segments
    unsigned long long qs[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // ... and inside the stages
#define PSTAMP(v) const unsigned long long v = __builtin_readcyclecounter()
#define QSTAMP(i, since) do { const unsigned long long _n = __builtin_readcyclecounter(); qs[i] += _n - (since); (since) = _n; } while (0)
#define QBEGIN(v) unsigned long long v = __builtin_readcyclecounter()
#else
#define PSTAMP(v)
#define QSTAMP(i, since)
#define QBEGIN(v)
#endif
    for (int k = 0; k < nseg; k++) {
        // Nothing derived from the thread index stays live across segments: left alone, the compiler hoists every per-thread
        // address and predicate of the loop body in front of the loop and then spills them around the f64 blocks (64 spilled
        // VGPRs, 244 B of scratch per lane at K = 16); recomputing them per segment is a handful of integer instructions.
        if (PERSIST) asm volatile("" : "+v"(tid), "+v"(lane));
        const uint32_t ext_k = PERSIST ? (plan.ext[k] ? (PH_EXT_FACTOR | PH_EXT_VARIABLE) : 0u) : ext_mask;
        const uint32_t int_k = PERSIST ? (PH_INT_FACTOR | PH_INT_VARIABLE) : int_mask;
        const int n_int_k = PERSIST ? (int)plan.n_int[k] : n_int;
        const bool last_seg = k == nseg - 1;
        // ======================= external factor sweep ============================================
        PSTAMP(ps0);
        QBEGIN(qt);
        if (PERSIST && ext_k && k > 0) {  // k == 0: the launch boundary has published everything
            wait_for_peers(k);
            QSTAMP(0, qt);
            __syncthreads();
            QSTAMP(1, qt);
        }
        PSTAMP(ps1);
        if (ext_k & PH_EXT_FACTOR) {
            external_factor_sweep(k, !PERSIST || k == last_ext_seg);
            QSTAMP(2, qt);
            __syncthreads();
            QSTAMP(3, qt);
        }
        PSTAMP(ps2);
#ifdef MGX_STAMPS
        if (k == 0) t_extf = __builtin_readcyclecounter();
#endif
        // ======================= external variable sweep ==========================================
        prefired = false;
        if constexpr (FUSED) {
            // Resident launches, K <= 16.  The UV wave runs the whole external variable sweep — inbox sums, then the finish on
            // four lanes per variable — with no workgroup barrier inside; a factor sweep of this segment's first internal
            // iteration that was not computed ahead goes into the shadow block next to it (DYN wave: dynamic messages, UV
            // wave after its finish: unary factors), and the shadow is adopted by exchanging the two blocks' roles.
            // The steady state of an alternating schedule (external iteration, ONE internal iteration whose factor sweep was
            // computed ahead): the two variable sweeps read the same inboxes but for the internal factors' messages — the
            // external one the old block, the internal one the shadow — and neither reads what the other writes, EXCEPT that
            // a belief update which finds its precision "zero", singular or its covariance non-finite keeps the state the
            // previous update left.  So they run side by side: the DYN wave the external sweep, into shadow images (means
            // = the response means of the next factor sweep, covariance, outcome), the UV wave the internal one, keeping its
            // means and flags back until the barrier; only a variable whose internal update did not go through looks at the
            // external sweep's outcome afterwards.  Takes a whole variable sweep off the chain from record to publication.
            par_done = (ext_k & PH_EXT_VARIABLE) && radio && early && n_int_k == 1;
            if (par_done) {
                const bool is_last = k == last_int_seg && last_seg;
                const int q = lane / K, i = lane - q * K;  // lanes < 4 K of either wave
                bool ok_i = false, fin_i = false;
                double mu_i = 0.0;
                if (role == ROLE_UV) {  // the messages computed ahead are this sweep's
                    double *t_ = s_fv;
                    s_fv = s_sh;
                    s_sh = t_;
                    // (the robot's last sweep leaves its sums in the belief image as well — after the barrier: the other wave is
                    // still reading that image as the prior)
                    variable_sums(s_snap, true, false);
                    QSTAMP(4, qt);
                    if (lane < 4 * K) quad_core(s_snap, s_cov, ok_i, fin_i, mu_i);
                } else {
                    if (lane < 4 * K) {
                        variable_sums_from(lane, 4 * K, s_tmp, false, false);
                        QSTAMP(4, qt);
                        double mu_x = s_mu[q * K + i];  // the state before this segment: the UV wave stores after the barrier
                        bool ok_x, fin_x;
                        quad_core(s_tmp, s_tmp + 4 * K, ok_x, fin_x, mu_x);  // covariance over the lam sums it has read
                        s_xmu[q * K + i] = mu_x - 0.0;                        // likewise over the eta sums
                        if (q == 0) s_xok[i] = (ok_x ? 1 : 0) | (fin_x ? 2 : 0);
                    }
                    double *t_ = s_fv;
                    s_fv = s_sh;
                    s_sh = t_;
                }
                QSTAMP(6, qt);
                __syncthreads();
                QSTAMP(7, qt);
                if (role == ROLE_UV && lane < 4 * K) {
                    double mu_fin = mu_i;
                    const int xs = s_xok[i];
                    if (!(ok_i && fin_i)) {  // variable.rs:273-297 kept what the external update left
                        mu_fin = s_xmu[q * K + i];
                        if (!ok_i && (xs & 1)) {
#pragma unroll
                            for (int j = 0; j < 4; j++) s_cov[(j * 4 + q) * K + i] = s_tmp[(4 + j * 4 + q) * K + i];
                        }
                    }
                    s_mu[q * K + i] = mu_fin;
                    s_snap[(20 + q) * K + i] = mu_fin;
                    if (is_last) {  // row q of (eta, lam): what this lane summed
                        s_prior[q * K + i] = s_snap[q * K + i];
#pragma unroll
                        for (int c = 0; c < 4; c++) s_prior[(4 + q * 4 + c) * K + i] = s_snap[(4 + q * 4 + c) * K + i];
                    }
                    if (q == 0) {
                        if (ok_i) {
                            s_valid[i] = fin_i ? 1 : 0;
                            s_covset[i] = 1;
                        } else if (xs & 1) {
                            s_valid[i] = (xs & 2) ? 1 : 0;
                            s_covset[i] = 1;
                        }
                    }
                }
                if (ir_on && k != last_ext_seg) {
                    have_xmu = true;
                } else if (ir_on) {  // the launch's last external iteration: the response means go to HBM (robot.rs:1842-1858)
                    for (int j = tid; j < ne; j += SWEEP_BLOCK) {
                        const int e = ie0 + j;
                        int dst;
                        if (j == tid) {
                            if (!pf_gate) continue;
                            dst = pf_dst;
                        } else {
                            if (!w.ir_gate[e]) continue;
                            dst = w.ir_rec[e].dst;
                        }
                        const int iv = dst & 0xffff;
#pragma unroll
                        for (int c = 0; c < 4; c++) w.ir_bmu[(size_t)c * w.NI + e] = s_xmu[c * K + iv];
                    }
                }
                if (is_last) __syncthreads();  // the tail's write-back (other wave) reads the belief image
                itf += 1;  // the factor sweep that was computed ahead
                early = false;
            } else if ((ext_k & PH_EXT_VARIABLE) && radio) {
                const bool ext_is_last = last_seg && n_int_k == 0;
                double *s_sum = ext_is_last ? s_prior : s_tmp;
                prefired = !early && n_int_k > 0 && !idle && (int_k & PH_INT_FACTOR) && skip0 == 0u;
                const bool keep_means = ir_on && k != last_ext_seg;
                if (role == ROLE_UV) {
                    variable_sums(s_sum, false, false);  // reads the messages of the last internal factor sweep (s_fv)
                    QSTAMP(4, qt);
                    finish(s_sum, false);
                    // the response means stay in LDS for the next segment's factor sweep (same wave: these writes follow the
                    // finish's reads of the eta sums they overwrite)
                    if (keep_means && lane < 4 * K) s_xmu[lane] = s_mu[lane] - 0.0;
                    if (prefired) unary_messages(0u, s_sh, itf);
                } else if (prefired && is_dyn && (w.enable & 1u)) {
                    dynamic_messages(s_sh);
                }
                QSTAMP(6, qt);
                if (keep_means) {
                    have_xmu = true;
                } else if (ir_on) {  // the launch's last external iteration: the means go to HBM (robot.rs:1842-1858)
                    __syncthreads();
                    for (int j = tid; j < ne; j += SWEEP_BLOCK) {
                        const int e = ie0 + j;
                        int dst;
                        if (j == tid) {
                            if (!pf_gate) continue;
                            dst = pf_dst;
                        } else {
                            if (!w.ir_gate[e]) continue;
                            dst = w.ir_rec[e].dst;
                        }
                        const int i = dst & 0xffff;
#pragma unroll
                        for (int c = 0; c < 4; c++) w.ir_bmu[(size_t)c * w.NI + e] = s_mu[c * K + i] - 0.0;
                    }
                }
                if (prefired) __syncthreads();  // both waves' columns of the shadow are complete
            }
            if (early || prefired) {  // adopt the factor sweep that was computed into the shadow
                double *t_ = s_fv;
                s_fv = s_sh;
                s_sh = t_;
            }
        } else if (ext_k & PH_EXT_VARIABLE) {
            // beliefs are recomputed, nothing is delivered to own factors (factorgraph.rs:794-826): the
            // sums go to the belief image if this is the robot's last sweep of the launch, else to scratch
            // (the image doubles as the prior, which every later sweep of the launch still needs)
            const bool ext_is_last = PERSIST ? (last_seg && n_int_k == 0) : !has_int_var;
            double *s_sum = ext_is_last ? s_prior : s_tmp;
            if (radio) variable_sums(s_sum, false, false);
            QSTAMP(4, qt);
            __syncthreads();
            QSTAMP(5, qt);
            // The first internal factor sweep of this segment does not depend on anything the external
            // sweeps produce (a dynamic factor reads the snapshot of the last INTERNAL variable sweep and
            // its own previous messages): the DYN wave computes its messages now, next to the UV wave's
            // mean / covariance of the external variable sweep (one 4x4 inverse per variable either way).
            // The unary factors do not either (they linearise at the means of the last INTERNAL sweep): the UV wave
            // runs them right after its finish instead of idling until the DYN wave is done.
            prefired = !early && radio && n_int_k > 0 && !idle && (int_k & PH_INT_FACTOR) && skip0 == 0u;  // same for the whole workgroup
            if (prefired && is_dyn && (w.enable & 1u)) dynamic_messages(s_fv);
            if (PERSIST && early && radio && role == ROLE_DYN) adopt_early(lane, 64);  // the sums above were the last readers of the old messages
            if (radio && is_var) variable_finish(s_sum, false);
            if (prefired) unary_messages(0u, s_fv, itf);
            QSTAMP(6, qt);
            __syncthreads();
            QSTAMP(7, qt);
            if (PERSIST && radio && ir_on && k != last_ext_seg) {
                if (tid < 4 * K) s_xmu[tid] = s_mu[tid] - 0.0;  // read after the barrier that opens the next segment's factor sweep
                have_xmu = true;
            } else if (radio && ir_on) {
                // responses to the foreign factors attached to our variables, routed to their inbox
                // (robot.rs:1842-1858): only the mean of that inbox entry is ever used (it sets the
                // linearisation point; eta / lam of the target side never reach the kept message).
                // Plain stores of LDS values: nothing in this launch but the storing thread itself reads them
                // (the means are next written after the barrier that ends the coming factor sweep / by nobody).
                for (int j = tid; j < ne; j += SWEEP_BLOCK) {
                    const int e = ie0 + j;
                    int dst;
                    if (j == tid && (do_extf || PERSIST)) {  // gate and constants of the thread's first edge are in registers
                        if (!pf_gate) continue;
                        dst = pf_dst;
                    } else {
                        if (!w.ir_gate[e]) continue;  // the owner cannot receive
                        dst = w.ir_rec[e].dst;
                    }
                    const int i = dst & 0xffff;
#pragma unroll
                    for (int c = 0; c < 4; c++) w.ir_bmu[(size_t)c * w.NI + e] = s_mu[c * K + i] - 0.0;
                }
            }
        }
        QSTAMP(8, qt);
        PSTAMP(ps3);
#ifdef MGX_STAMPS
        if (k == 0) { t_extv = __builtin_readcyclecounter(); t_loop0 = t_extv; rt0 = __builtin_amdgcn_s_memrealtime(); }
#endif
        // ======================= internal iterations ==============================================
        if (PERSIST && !FUSED && early && !((ext_k & PH_EXT_VARIABLE) && radio)) {  // no external variable sweep ran: adopt here
            adopt_early(tid, SWEEP_BLOCK);
            __syncthreads();
        }
        if (PERSIST && early) prefired = true;
        early = false;
        for (int it = 0; it < n_int_k && !idle && !par_done; it++) {
            asm volatile("" : "+v"(tid), "+v"(lane));  // as at the top of a segment: nothing per-thread hoisted out of this loop either
            STAMP(t0);
            if ((int_k & PH_INT_FACTOR) && it == 0 && prefired) {
                itf += 1;  // this sweep ran next to the external variable sweep, in front of that block's last barrier
            } else if (int_k & PH_INT_FACTOR) {
                if (is_dyn && (w.enable & 1u) && !(it == 0 && (skip0 & 1u))) dynamic_messages(s_fv);
                // UV wave: first the belief of the previous sweep (mean, covariance) that the unary
                // factors linearise at — same wave, so its LDS writes precede their LDS reads
                if (pending) finish(s_snap, true);
                pending = false;
                unary_messages(it == 0 ? skip0 : 0u, s_fv, itf);
                itf += 1;
                STAMP(t1);
                __syncthreads();
                STAMP(t2);
                STAMP_ADD(c_f, t0, t1);
                STAMP_ADD(c_fb, t1, t2);
            }
            STAMP(t3);
            if (int_k & PH_INT_VARIABLE) {
                // the robot's last sweep of the launch leaves its sums in the belief image as well (a robot that is
                // off the air runs no external sweep: its last one is the last internal iteration of the schedule)
                const bool is_last = it == n_int_k - 1 && (!PERSIST || (k == last_int_seg && (last_seg || !radio)));
                variable_sums(s_snap, true, is_last);
                pending = true;
                STAMP(t4);
                __syncthreads();
                STAMP(t5);
                STAMP_ADD(c_v, t3, t4);
                STAMP_ADD(c_vb, t4, t5);
            }
        }
        // ======================= end of a segment of a resident schedule launch ====================
        // The snapshot records of this robot (what its variables last sent to their own factors: all that another
        // robot's inter-robot factors read) go out for the external iteration that opens the next segment: into the
        // buffer nobody reads during this segment, write-through, every wave drained, then the progress word.
        PSTAMP(ps4);
        // ONE wave does all of it (the one that completes the means), so it may signal for itself after its own drain.
        if (PERSIST && !last_seg) {
            QSTAMP(9, qt);
            if (role == ROLE_UV) {
                if (pending) finish(s_snap, true);
                QSTAMP(10, qt);
                __builtin_amdgcn_wave_barrier();  // the wave's LDS writes (means) precede its LDS reads below
                const int ob = (w.cur + k + 1) & 1;
                const unsigned base = (unsigned)v0 * (unsigned)(SNAP_W * sizeof(double));
                for (int t = lane; t < (SNAP_W / 2) * K; t += 64) {  // 16 bytes = components 2c, 2c + 1 of variable i
                    const int i = t / (SNAP_W / 2), c = t - i * (SNAP_W / 2);
                    st16_agent(rs_snap[ob], base + (unsigned)(i * SNAP_W + 2 * c) * 8u, s_snap[(2 * c) * K + i], s_snap[(2 * c + 1) * K + i]);
                }
                for (int t = lane; t < K; t += 64) st_agent(&w.snap_epoch[ob][v0 + t], s_epoch[t]);
                QSTAMP(11, qt);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                QSTAMP(12, qt);
                if (lane == 0)
                    __hip_atomic_store(&w.sweep_flag[r], plan.flag_base + (unsigned long long)k + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            pending = false;
            // the factor sweep that opens the next segment's internal iterations, while the records travel: the DYN wave
            // starts at once (a dynamic factor reads no mean), the UV wave after its publish.  The tracking factors' gate
            // (factorgraph.rs:701) counts the external factor sweep that the reference runs in between.
            early = plan.n_int[k + 1] > 0 && !idle && skip0 == 0u;
            if (early) {
                if (is_dyn && (w.enable & 1u)) dynamic_messages(s_sh);
                unary_messages(0u, s_sh, itf + ((plan.ext[k + 1] && radio) ? 1 : 0));
            }
            QSTAMP(13, qt);
        }
#ifdef MGX_STAMPS
        {
            PSTAMP(ps5);
            p_wait += ps1 - ps0; p_extf += ps2 - ps1; p_extv += ps3 - ps2; p_int += ps4 - ps3; p_pub += ps5 - ps4;
        }
#endif
    }
    {
        // Tail: the UV wave completes the last variable sweep (mean, covariance) while the DYN wave
        // already writes back what that does not touch — the factor -> variable messages and the belief
        // (eta, lam) image, three quarters of the robot's output.
        if (role == ROLE_UV) {
            if (pending) finish(s_snap, true);
        } else {
            copy_words_wave(blob + L.fv(), s_fv, 20 * E1, lane);
            if (any_sweep && !bel_dead) copy_words_wave(blob + L.bel(), s_prior, 20 * K, lane);
        }
        __syncthreads();
#ifdef MGX_STAMPS
        if (w.dbg && lane == 0) {  // per wave: cycles in factor phase, its barrier, variable phase, its barrier
            unsigned long long *d = w.dbg + ((size_t)blockIdx.x * 2 + role) * 8;
            d[0] = c_f; d[1] = c_fb; d[2] = c_v; d[3] = c_vb; d[4] = __builtin_readcyclecounter() - t_loop0;
            d[5] = __builtin_amdgcn_s_memrealtime() - rt0;  // 100 MHz ticks over the same span
            d[6] = t_staged - t_k0;
            d[0] = (d[0] & 0xffffffffull) | ((t_extf - t_staged) << 32);  // external factor sweep (high word)
            d[1] = (d[1] & 0xffffffffull) | ((t_extv - t_extf) << 32);    // external variable sweep (high word)
            if (PERSIST) {
                d[0] = p_wait; d[1] = p_extf; d[2] = p_extv; d[3] = p_int; d[4] = p_pub;
                unsigned long long *d2 = w.dbg + (size_t)(gridDim.x + 4) * 16 + ((size_t)blockIdx.x * 2 + role) * 16;
                for (int i = 0; i < 16; i++) d2[i] = qs[i];
                d2[14] = q_arrive;
            }
        }
#endif
    }
    if (PERSIST) snap_out = (w.cur + nseg) & 1;  // where the records of the schedule's last sweep go (the host follows)

    // ---- write back: straight copies of the LDS images ----------------------------------------------
    for (int t = tid; t < 16 * K; t += SWEEP_BLOCK)  // covariance of the variables that recomputed it
        if (s_covset[t % K]) blob[L.cov() + t] = s_cov[t];
    copy_words(blob + L.mu(), s_mu, 4 * K, tid);
    copy_words(blob + L.valid(), (const double *)s_valid, K, tid);
    if (snap_out >= 0) {
        double *dst = w.snap[snap_out] + (size_t)v0 * SNAP_W;
        for (int t = tid; t < SNAP_W * K; t += SWEEP_BLOCK) dst[t] = s_snap[(t % SNAP_W) * K + (t / SNAP_W)];
        for (int t = tid; t < K; t += SWEEP_BLOCK) w.snap_epoch[snap_out][v0 + t] = s_epoch[t];
    }
    if (is_trk) {
        w.trk_record[trk_item] = trk_rec;
        w.trk_last_pos[trk_item] = trk_lp[0];
        w.trk_last_pos[(size_t)w.NT + trk_item] = trk_lp[1];
        w.trk_last_val[trk_item] = trk_lv;
    }
    if (tid == 0) w.iter_factor[r] = itf;
#ifdef MGX_STAMPS
    if (w.dbg && lane == 0) w.dbg[((size_t)blockIdx.x * 2 + role) * 8 + 7] = __builtin_readcyclecounter() - t_k0;  // whole kernel
#endif
}

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
__global__ void __launch_bounds__(256) k_halo_wait_unpack(DevWorld w, int n, const int32_t *ghosts, const double *recv, int n_sources,
                                                          const unsigned long long *flags, unsigned long long seq,
                                                          unsigned long long *err, long long timeout_ticks, unsigned long long *ready,
                                                          unsigned long long *host_err) {
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
    const double val = __hip_atomic_load(&recv[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
// k_edge_rebuild lays out the new edge arrays from the new slot list (one thread per factor):
// a surviving connection (old_slot >= 0) carries its message (the six live numbers), response mean
// and creation epoch over from the arrays being replaced; a new one starts empty, created at the
// owner variable's current delivery count, with the target variable's current belief mean as the
// response it has seen (robot.rs:1549-1585).  The constant record of every edge is derived here too.
__global__ void k_edge_rebuild(DevWorld w, int n_slots, const IrSlotRec *__restrict__ slots, const int32_t *__restrict__ in_new,
                               const int32_t *__restrict__ in_old, int stride_new, IrEdgeRec *__restrict__ recs,
                               double *__restrict__ fv_eta, double *__restrict__ fv_lam, double *__restrict__ bmu) {
    const int K1 = w.K - 1;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_slots * K1) return;
    const int g = t / K1, j = t - g * K1;
    const IrSlotRec sl = slots[g];
    const int r = sl.tgt_robot;
    const int n_in = in_new[r + 1] - in_new[r];
    const int e = K1 * in_new[r] + j * n_in + (g - in_new[r]);
    const size_t sn = (size_t)stride_new, so = (size_t)w.NI;
    IrEdgeRec rec;
    rec.src_var = sl.src_robot * w.K + j + 1;
    rec.src_robot = sl.src_robot;
    rec.dst = (int32_t)(j + 1) | ((sl.flags & 1) ? (1 << 16) : 0);
    rec.d_safe = sl.d_safe;
    rec.offset = (double)1e-6f * (double)(sl.first_number + (unsigned long long)j);  // interrobot.rs:52,75
    if (sl.old_slot >= 0) {
        const int n_old = in_old[r + 1] - in_old[r];
        const int o = K1 * in_old[r] + j * n_old + (sl.old_slot - in_old[r]);
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
        const BlobLayout L(w.K);
#pragma unroll
        for (int c = 0; c < 4; c++)  // the target's belief goes into the new factor — which drops it while its kind is off
            bmu[c * sn + e] = (w.enable & 2u) ? w.blob[(size_t)r * w.BS + L.mu() + c * w.K + (j + 1)] : 0.0;
        rec.created = w.snap_epoch[w.cur][rec.src_var];
    }
    recs[e] = rec;
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
size_t sweep_lds_bytes(int K, int ir_edges, bool resident) {  // resident: + the shadow of the factor -> variable messages
    const BlobLayout L(K);
    const int io = L.inout_words() + (L.inout_words() & 1);
    return sizeof(double) * (size_t)((SNAP_W + 20 + 20) * K + io + IR_STRIDE * (ir_edges + 1) + (resident ? 20 * L.E1 : 0)) +
           4 * (size_t)((resident ? 3 : 2) * ((K + 1) & ~1) + ((3 * (K + 1) + 1) & ~1));
}
size_t sweep_lds_bytes(int K, int ir_edges) { return sweep_lds_bytes(K, ir_edges, false); }
bool sweep_supports(int K) { return K >= 3 && 2 * (K - 1) <= 128; }  // beyond 33 variables: two dynamic messages per lane
int blob_words(int K) { const BlobLayout L(K); return (L.words() + 1) & ~1; }

template <int KT>
static void launch_k(const DevWorld &w, int robot0, int n_robots, uint32_t ext_mask, uint32_t int_mask, int n_int, int snap_out,
                     uint32_t hints, hipStream_t stream) {
    // staging the inter-robot messages needs IR_STRIDE f64 per edge; beyond 64 KB of LDS fall back to
    // reading them from L2 in every variable sweep
    const size_t staged = sweep_lds_bytes(w.K, w.ir_max_edges);
    const SegPlan none{};
    if (w.ir_max_edges == 0)
        hipLaunchKernelGGL((k_robot_sweep<KT, IR_NONE, false>), dim3(n_robots), dim3(SWEEP_BLOCK), sweep_lds_bytes(w.K, 0), stream, w,
                           robot0, ext_mask, int_mask, n_int, snap_out, hints, none);
    else if (staged <= 64 * 1024)
        hipLaunchKernelGGL((k_robot_sweep<KT, IR_STAGED, false>), dim3(n_robots), dim3(SWEEP_BLOCK), staged, stream, w, robot0, ext_mask,
                           int_mask, n_int, snap_out, hints, none);
    else
        hipLaunchKernelGGL((k_robot_sweep<KT, IR_GLOBAL, false>), dim3(n_robots), dim3(SWEEP_BLOCK), sweep_lds_bytes(w.K, 0), stream, w,
                           robot0, ext_mask, int_mask, n_int, snap_out, hints, none);
}

// horizon lengths of BASELINE.json / the reference scenarios get constant-K code
#define MGX_FOR_K(K_, DO)                                                     \
    switch (K_) {                                                             \
    case 10: DO(10); break;                                                   \
    case 11: DO(11); break; /* Tracking Factor Showcase */                    \
    case 12: DO(12); break; /* Junction Twoway */                             \
    case 13: DO(13); break; /* Junction Experiment */                         \
    case 16: DO(16); break;                                                   \
    case 17: DO(17); break; /* Merge, Iteration Amount */                     \
    case 20: DO(20); break; /* Schedules Experiment */                        \
    case 21: DO(21); break; /* Circle Experiment */                           \
    case 32: DO(32); break;                                                   \
    case 35: DO(35); break; /* Communications Failure */                      \
    default:                                                                  \
        if (2 * ((K_) - 1) > 64) { DO(-1); } else { DO(0); }                  \
        break;                                                                \
    }

hipError_t launch_robot_sweep(const DevWorld &w, int robot0, int n_robots, uint32_t ext_mask, uint32_t int_mask, int n_int,
                              int snap_out, uint32_t hints, hipStream_t stream) {
    if (n_robots <= 0) return hipSuccess;
#define MGX_DO(KT) launch_k<KT>(w, robot0, n_robots, ext_mask, int_mask, n_int, snap_out, hints, stream)
    MGX_FOR_K(w.K, MGX_DO)
#undef MGX_DO
    return hipGetLastError();
}

// ---- resident schedule launches ------------------------------------------------------------------------
// How many workgroups of the resident kernel the device holds at once (0: this world's shape has no resident
// form: no inter-robot edges, or too many per robot to stage in LDS).  Every workgroup of such a launch waits for
// its neighbours INSIDE the launch, so all of them have to be resident together.
// A workgroup may take up to the CU's whole 160 KB of LDS (MI355X_MICROARCH.md); beyond 64 KB the kernel has to be told.
constexpr size_t RESIDENT_LDS_MAX = 160 * 1024;
size_t sweep_resident_lds_max() { return RESIDENT_LDS_MAX; }
template <int KT>
static bool resident_allow_lds(size_t staged) {
    if (staged <= 64 * 1024) return true;
    static size_t allowed = 0;  // per instantiation
    if (staged <= allowed) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_robot_sweep<KT, IR_STAGED, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)staged) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    allowed = staged;
    return true;
}
template <int KT>
static int resident_capacity_k(const DevWorld &w) {
    const size_t staged = sweep_lds_bytes(w.K, w.ir_max_edges, true);
    if (w.ir_max_edges == 0 || staged > RESIDENT_LDS_MAX || !resident_allow_lds<KT>(staged)) return 0;
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_robot_sweep<KT, IR_STAGED, true>, SWEEP_BLOCK, staged) != hipSuccess) return 0;
    return per_cu * cus;
}
int sweep_resident_capacity(const DevWorld &w) {
    int cap = 0;
#define MGX_DO(KT) cap = resident_capacity_k<KT>(w)
    MGX_FOR_K(w.K, MGX_DO)
#undef MGX_DO
    return cap;
}
hipError_t launch_robot_schedule(const DevWorld &w, int n_robots, const SegPlan &plan, hipStream_t stream) {
    if (n_robots <= 0 || plan.n <= 0) return hipSuccess;
    const size_t staged = sweep_lds_bytes(w.K, w.ir_max_edges, true);
#define MGX_DO(KT)                                                                                                                 \
    if (!resident_allow_lds<KT>(staged)) return hipErrorInvalidValue;                                                              \
    hipLaunchKernelGGL((k_robot_sweep<KT, IR_STAGED, true>), dim3(n_robots), dim3(SWEEP_BLOCK), staged, stream, w, 0, 0u, 0u, 0, -1, \
                       0u, plan)
    MGX_FOR_K(w.K, MGX_DO)
#undef MGX_DO
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
                            const unsigned long long *peer_flags, unsigned long long seq, unsigned int *done, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const int total = n * (SNAP_W + 1) * w.K;
    hipLaunchKernelGGL(k_halo_push, dim3((total + 255) / 256), dim3(256), 0, stream, w, n, robots, dst, n_peers, peer_flags, seq, done);
    return hipGetLastError();
}
hipError_t launch_halo_wait_unpack(const DevWorld &w, int n, const int32_t *ghosts, const double *recv, int n_sources,
                                   const unsigned long long *flags, unsigned long long seq, unsigned long long *err,
                                   long long timeout_ticks, unsigned long long *ready, unsigned long long *host_err, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const int total = n * (SNAP_W + 1) * w.K;
    hipLaunchKernelGGL(k_halo_wait_unpack, dim3((total + 255) / 256), dim3(256), 0, stream, w, n, ghosts, recv, n_sources, flags, seq,
                       err, timeout_ticks, ready, host_err);
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
hipError_t launch_edge_rebuild(const DevWorld &w, int n_slots, const IrSlotRec *slots, const int32_t *in_new, const int32_t *in_old,
                               int stride_new, IrEdgeRec *recs, double *fv_eta, double *fv_lam, double *bmu, hipStream_t stream) {
    const int total = n_slots * (w.K - 1);
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_edge_rebuild, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, n_slots, slots, in_new, in_old,
                       stride_new, recs, fv_eta, fv_lam, bmu);
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
