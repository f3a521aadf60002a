// mgx_sweep.h — the sweep kernel of the GBP engine (k_robot_sweep) and the device helpers every kernel file shares.
// Included by mgx_kernels.hip (the small kernels) and by mgx_sweep_inst.hip (the instantiations, several translation units).
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
#pragma once
#include <hip/hip_runtime.h>

#include "gbp_math.h"
#include "mgx_dev.h"

namespace mgx {

constexpr int SWEEP_BLOCK = 128;
enum { ROLE_DYN = 0, ROLE_UV = 1 };
// FOUR WAVES PER ROBOT (resident launches, horizons of at most 16 variables; -DMGX_WIDE=1).  What bounds a resident launch is the
// dependent chain of ONE robot's iteration — gather, inter-robot factors, inbox sums, beliefs, publication — with the factor sweep
// of the coming internal iteration on the same two waves (measured by delaying single points of the kernel: the whole loop of the
// UV wave is critical).  With four waves the two chain waves keep the chain (DYN: external variable sweep, UV: internal one + the
// publication), wave 2 computes the dynamic factors' messages and wave 3 the unary factors' — beside the hand-off, the gather and the
// inter-robot factors of the chain waves instead of in front of them.
#ifndef MGX_WIDE
#define MGX_WIDE 0
#endif
template <int KT, bool PERSIST>
constexpr int sweep_waves() { return (MGX_WIDE && PERSIST && KT > 0 && 4 * KT <= 64) ? 4 : 2; }
template <int KT, bool PERSIST>
constexpr int sweep_threads() { return 64 * sweep_waves<KT, PERSIST>(); }

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

// An edge's constants in registers: loaded as two 16-byte pieces and kept as five scalars, never as an aggregate — a copy of the
// 32-byte record that is chosen between a register copy and memory ends up as a pointer choice, and the register copy in scratch.
#define MGX_EDGE_REGS(p) int p##_src_var = 0, p##_dst = 0; uint32_t p##_created = 0u; double p##_d_safe = 0.0, p##_offset = 0.0
#define MGX_EDGE_LOAD(p, ptr)                                                                                     \
    do {                                                                                                          \
        const int4 a_ = *reinterpret_cast<const int4 *>(ptr);                                                     \
        const double2 b_ = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(ptr) + 16);          \
        p##_src_var = a_.x; p##_created = (uint32_t)a_.z; p##_dst = a_.w; p##_d_safe = b_.x; p##_offset = b_.y;   \
    } while (0)
#define MGX_EDGE_COPY(p, q) do { p##_src_var = q##_src_var; p##_created = q##_created; p##_dst = q##_dst; p##_d_safe = q##_d_safe; p##_offset = q##_offset; } while (0)

// In-kernel cycle stamps exist only in the diagnostic build (never in libmgx.so): they go to a
// buffer of their own and no output value depends on them.
// Causal profiling (diagnostic builds only: -DMGX_DELAY_POINT=n [-DMGX_DELAY=16]): a sleep of MGX_DELAY x 64 clocks at ONE point of the
// resident launch's iteration; how much of it shows up in the iteration's period says how much of the point lies on the critical path.
#ifdef MGX_DELAY_POINT
#ifndef MGX_DELAY
#define MGX_DELAY 16
#endif
#define DELAY_AT(n, cond) do { if (MGX_DELAY_POINT == (n) && (cond)) __builtin_amdgcn_s_sleep(MGX_DELAY); } while (0)
#else
#define DELAY_AT(n, cond)
#endif
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
__device__ __forceinline__ void copy_words(double *dst, const double *src, int n, int tid, int nt = SWEEP_BLOCK) {
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
    double2 *d2 = reinterpret_cast<double2 *>(dst);
    for (int t = tid; t < (n >> 1); t += nt) d2[t] = s2[t];
    if ((n & 1) && tid == 0) dst[n - 1] = src[n - 1];
}

// HBM -> LDS with every load of the thread in flight before its first LDS store: a copy loop that
// waits for each 16 bytes before asking for the next costs one memory round trip per iteration, and
// staging is a handful of such loops.  N (f64 words) is a compile-time constant: the loops unroll into
// independent loads held in registers.
template <int N, int NT = SWEEP_BLOCK>
struct StageRegs {
    static constexpr int N2 = N / 2, ITERS = (N2 + NT - 1) / NT;
    double2 v[ITERS];
    double last;
    __device__ __forceinline__ void load(const double *src, int tid) {
        const double2 *s2 = reinterpret_cast<const double2 *>(src);
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const int t = tid + it * NT;
            v[it] = (t < N2) ? s2[t] : make_double2(0.0, 0.0);
        }
        last = ((N & 1) && tid == 0) ? src[N - 1] : 0.0;
    }
    __device__ __forceinline__ void store(double *dst, int tid) const {
        double2 *d2 = reinterpret_cast<double2 *>(dst);
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const int t = tid + it * NT;
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

// The same at SYSTEM scope (aux = sc0 | sc1): fine-grained memory that another GPU stores into or reads from over xGMI (the
// ghost areas of sharded resident launches).
__device__ __forceinline__ v4u32 ld16_system_raw(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 17);
}
__device__ __forceinline__ void st16_system(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, double a, double b) {
    v4u32 v;
    v.x = (unsigned)__double2loint(a); v.y = (unsigned)__double2hiint(a);
    v.z = (unsigned)__double2loint(b); v.w = (unsigned)__double2hiint(b);
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)byte_off, 0, 17);
}
// four dwords as they are (exchange records: three payload dwords and the sequence word, mgx_dev.h); `sbase` is a wave-uniform
// byte offset that rides in the instruction's scalar operand (the parity of the records: ONE descriptor serves both)
__device__ __forceinline__ v4u32 ld16_agent_raw(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, unsigned sbase) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, (int)sbase, 16);
}
__device__ __forceinline__ v4u32 ld16_system_raw(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, unsigned sbase) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, (int)sbase, 17);
}
// (stores: the wave-uniform part is ADDED to the per-lane offset, never passed in the scalar operand.  A buffer store of more than
// 8 bytes reads its data registers some cycles after issue; the compiler keeps the next VALU write of those registers away from
// it — except behind a store with an SGPR soffset, which older parts issued a cycle later (GCNHazardRecognizer: "this hazard only
// exists if the instruction is not using a register in the soffset field").  On gfx950 that exemption does not hold: with the
// parity in soffset, lanes 32..47 of one publication in ~10^4 went out with the NEXT chunk's payload under a valid sequence word
// — found with a per-record checksum, experiments/README.md.)
__device__ __forceinline__ void st16_agent_raw(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, unsigned sbase, v4u32 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(byte_off + sbase), 0, 16);
}
__device__ __forceinline__ void st16_system_raw(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, v4u32 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)byte_off, 0, 17);
}
// a buffer descriptor over [base, base + bytes) from a pointer every lane of the wave holds (made scalar here)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(unsigned long long base, unsigned bytes) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(base & 0xffffffffull));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0, (int)bytes, 0x00020000);
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
// SHARD (resident launches of a sharded world): the snapshot records of ghost robots arrive INSIDE the launch — their owners'
// ranks store them into this rank's peer-mapped ghost area and this rank's boundary robots store theirs into the peers'.
template <int KT, int IRM, bool PERSIST, bool SHARD = false>
__global__ void __launch_bounds__((sweep_threads<KT, PERSIST>()), 2) k_robot_sweep(DevWorld w, int robot0, uint32_t ext_mask, uint32_t int_mask,
                                                             int n_int, int snap_out, uint32_t hints, const SegPlan plan) {
    constexpr bool HAS_IR = IRM != IR_NONE, STAGE_IR = IRM == IR_STAGED;
    constexpr int NW = sweep_waves<KT, PERSIST>(), NT = 64 * NW;  // waves / threads of a workgroup
    constexpr int ROLE_DF = NW == 4 ? 2 : ROLE_DYN, ROLE_UF = NW == 4 ? 3 : ROLE_UV;  // who computes the dynamic / the unary factors' messages
    static_assert(!PERSIST || IRM == IR_STAGED, "resident schedule launches exist for worlds with staged inter-robot messages");
    static_assert(!SHARD || PERSIST, "ghost records arrive in-launch only in resident schedule launches");
    // LINGER: the launch may go on with plans the host posts while it runs (mgx_dev.h, "lingering resident launches"): the plan in
    // force lives in LDS (s_plan: the launch's own from the kernel's arguments, later ones from the host-mapped box), and the
    // segment counter that parities and sequence numbers derive from runs on over the plans (kbase)
    constexpr bool LINGER = PERSIST && !SHARD;
    constexpr bool PLDS = PERSIST;  // the plan in force and what is derived from it live in LDS (s_plan, s_ps, s_seg) in every resident kernel
    int nseg = PERSIST ? plan.n : 1;
    // Fields of the world that only COLD paths read (the residency census and its decider, give-up paths of the waits, the ranks'
    // agreement, the push records of boundary robots): fetched from the kernel's argument block where they are used, through a
    // pointer the compiler cannot see through — read as `w.field` they are loaded once at the top and held in scalar registers for
    // the whole kernel, and the sharded instantiation had more of those than spill lanes (the rest went to scratch).
    struct ArgBlock {  // the kernel's arguments as they lie in its argument block
        DevWorld w; int robot0; uint32_t ext_mask, int_mask; int n_int, snap_out; uint32_t hints; SegPlan plan;
    };
    // (where the block is: the pointer the kernel is entered with, parked in the first 16 bytes of LDS by thread 0 in front of the
    // staging barrier — held in registers for the whole kernel it was one more pair to spill)
    auto cold_args = [&]() __attribute__((always_inline)) -> const ArgBlock & {
        unsigned long long a = PERSIST ? *reinterpret_cast<const unsigned long long *>(lds) : (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
        if (PERSIST) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a & 0xffffffffull));
            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
            a = ((unsigned long long)hi << 32) | lo;
        }
        const ArgBlock *p = reinterpret_cast<const ArgBlock *>(a);
        asm volatile("" : "+s"(p));
        return *p;
    };
    auto cold_args_at_entry = [&]() __attribute__((always_inline)) -> const ArgBlock & {  // (in front of the staging barrier)
        const ArgBlock *p = reinterpret_cast<const ArgBlock *>((unsigned long long)__builtin_amdgcn_kernarg_segment_ptr());
        asm volatile("" : "+s"(p));
        return *p;
    };
    if (PERSIST && threadIdx.x == 0) *reinterpret_cast<unsigned long long *>(lds) = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    auto cold = [&]() __attribute__((always_inline)) -> const DevWorld & { return cold_args().w; };
    auto cold0 = [&]() __attribute__((always_inline)) -> const DevWorld & { return cold_args_at_entry().w; };
    auto cold_plan0 = [&]() __attribute__((always_inline)) -> const SegPlan & { return cold_args_at_entry().plan; };
    auto cold_plan = [&]() __attribute__((always_inline)) -> const SegPlan & { return cold_args().plan; };
    // The plan's per-segment bytes, read as DWORDS of the kernel's arguments: a byte indexed by the segment counter is fetched by a
    // VECTOR load, and the wait for a vector load waits for every write-through store the wave has in flight as well (one counter) —
    // `plan.n_int[k + 1]` right behind the publication made the publishing wave sit out the drain of its own stores.
    static_assert(MAX_SEGS % 4 == 0 && offsetof(SegPlan, ext) % 4 == 0 && offsetof(SegPlan, n_int) % 4 == 0, "SegPlan bytes are read as dwords");
    // (LINGER: the bytes of the plan in force, from its LDS copy — a wave-uniform value, and told so)
    uint32_t *s_plan = reinterpret_cast<uint32_t *>(lds + 2);  // [LINGER_PLAN_DWORDS] (PERSIST only), behind the parked argument pointer
    double *s_urec = lds + 2 + LINGER_PLAN_DWORDS / 2;          // [4] a posted plan's prior-update record of this robot
    // What is derived from the plan in force, kept HERE and fetched at the top of every segment (wave-uniform values that would
    // otherwise sit in scalar registers — and their spill lanes — for the whole launch).  Every wave works them out for itself where
    // a plan begins and keeps a copy of its own (no barrier between writing and reading them):
    //   s_ps  [4 waves][4]   [0] plans run before the one in force; [1] launch-wide index BEHIND the plan's last segment;
    //                        [2] flags (1: some internal variable sweep runs on this robot, 2: some variable sweep)
    //   s_seg [4 waves][32]  per segment k, ONE 8-byte read at its top: { SEG_* bits | internal iterations << 8 | external? of
    //                        segment k + 1 << 16 | its internal iterations << 24, the segment's launch-wide index }
    constexpr int PS_WORDS = 4, PREFIX_F64 = 2 + LINGER_PLAN_DWORDS / 2 + 4 + (4 * PS_WORDS + 4 * 2 * MAX_SEGS) / 2;
    static_assert(LINGER_PLAN_DWORDS % 2 == 0 && PREFIX_F64 == 154, "the resident kernels' LDS prefix (mgx_kernels.hip: sweep_lds_bytes)");
    constexpr uint32_t SEG_EXT = 1u, SEG_LAST = 2u, SEG_LAST_INT = 4u, SEG_LAST_EXT = 8u, SEG_FIRST_OF_LAUNCH = 16u, SEG_LINGERS = 32u;
    uint32_t *s_ps = reinterpret_cast<uint32_t *>(lds + 2 + LINGER_PLAN_DWORDS / 2 + 4) + (threadIdx.x >> 6) * PS_WORDS;
    uint2 *s_seg = reinterpret_cast<uint2 *>(lds + 2 + LINGER_PLAN_DWORDS / 2 + 4 + (4 * PS_WORDS) / 2) + (threadIdx.x >> 6) * MAX_SEGS;
    auto ps_word = [&](int i) __attribute__((always_inline)) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ps[i]); };
    auto plan_dword = [&](int i) __attribute__((always_inline)) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)s_plan[i]); };
    auto plan_ext = [&](int k) __attribute__((always_inline)) {
        if constexpr (PLDS) return (plan_dword(2 + (k >> 2)) >> (8 * (k & 3))) & 0xffu;
        return (reinterpret_cast<const uint32_t *>(plan.ext)[k >> 2] >> (8 * (k & 3))) & 0xffu;
    };
    auto plan_n_int = [&](int k) __attribute__((always_inline)) {
        if constexpr (PLDS) return (int)((plan_dword(2 + MAX_SEGS / 4 + (k >> 2)) >> (8 * (k & 3))) & 0xffu);
        return (int)((reinterpret_cast<const uint32_t *>(plan.n_int)[k >> 2] >> (8 * (k & 3))) & 0xffu);
    };
    if (PLDS && threadIdx.x == 0) {  // the launch's own plan, as a posted one would arrive (read behind the staging barrier)
        s_plan[0] = (uint32_t)plan.n;
        s_plan[1] = w.upd ? 1u : 0u;
#pragma unroll
        for (int i = 0; i < MAX_SEGS / 4; i++) {
            s_plan[2 + i] = reinterpret_cast<const uint32_t *>(plan.ext)[i];
            s_plan[2 + MAX_SEGS / 4 + i] = reinterpret_cast<const uint32_t *>(plan.n_int)[i];
        }
        double *pd = reinterpret_cast<double *>(s_plan + 2 + MAX_SEGS / 2);
        pd[0] = w.upd_max_speed;
        pd[1] = w.upd_delta_t;
    }
    // KT > 0: horizon length fixed at compile time; 0: read from the world (K <= 33); -1: read from the world, any K.
    // BIG: more than 64 dynamic-factor messages / tracking factors per robot (K > 33): some lanes carry two.
    constexpr bool BIG = KT < 0 || KT > 33;
    STAMP(t_k0);
    // resident launches: "this workgroup has started" (SegPlan: residency census) — a word of its own, a plain write-through
    // store: nothing contended is outstanding when the staging loads are waited for
    const bool census = PERSIST && plan.launch_seq != 0ull;
    if (census && threadIdx.x == 0) __hip_atomic_store(&cold0().census[blockIdx.x], plan.launch_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ... and the launch's LAST workgroup owns no robot: it is the decider.  Dispatched behind every robot's workgroup, it
    // finds every word signed within a microsecond of the launch's start when the whole grid is on the device — while the
    // robots' workgroups are still staging — and says go; if they are not all signed within the bound it says abort.  Either
    // way by compare-and-swap on a word that is monotonic in the launch number (no launch resets it).
    if (census && blockIdx.x == gridDim.x - 1) {
        const unsigned long long seq = plan.launch_seq;
        const long long t0 = wall_clock64();
        // sharded worlds: this rank's answer is the RANKS' answer (SegPlan::agree_seq) — complete here means signed in there
        const bool ranks = SHARD && cold_plan0().agree_seq != 0ull;
        bool signed_in = false;
        for (;;) {
            int missing = 0;
            for (unsigned b2 = threadIdx.x; b2 < gridDim.x; b2 += NT)
                missing |= __hip_atomic_load(&cold0().census[b2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seq ? 1 : 0;
            const bool all = __syncthreads_or(missing) == 0;
            int done = 0;
            if (threadIdx.x == 0) {
                unsigned long long v = __hip_atomic_load(cold0().decision, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool late = wall_clock64() - t0 > cold_plan0().census_ticks;
                unsigned want;
                if (!ranks) {
                    want = all ? RESIDENT_GO : (late ? RESIDENT_ABORT : 0u);
                } else if (all && !signed_in) {
                    want = agree_on_launch(cold0().agree, cold_plan0().agree_seq, (unsigned)cold0().n_ranks, AGREE_SIGN_IN);
                    signed_in = true;
                } else {
                    want = agree_on_launch(cold0().agree, cold_plan0().agree_seq, (unsigned)cold0().n_ranks, late ? AGREE_ABORT : AGREE_LOOK);
                }
                if (want == AGREE_LOST) {  // the other ranks are more schedules ahead than the word remembers: this world has parted
                    __hip_atomic_store(cold0().sweep_err, cold_plan0().agree_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // from theirs (reported)
                    want = RESIDENT_ABORT;
                }
                if ((v >> 2) == seq) {
                    done = 1;  // a workgroup that gave up on this one has decided
                } else if (want) {
                    if (__hip_atomic_compare_exchange_strong(cold0().decision, &v, seq * 4ull + want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                        __hip_atomic_store(cold0().decision_host, seq * 4ull + want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    done = 1;  // (lost the exchange: somebody else's decision stands)
                }
            }
            if (__syncthreads_or(done)) break;
            __builtin_amdgcn_s_sleep(1);
            if (ranks) __builtin_amdgcn_s_sleep(8);  // (the word is another rank's memory)
        }
        if constexpr (LINGER) {
            // A LINGERING launch (mgx_dev.h): this workgroup stays as the launch's postman — the only poller of the host-mapped box.
            // Every round one look at the box and at most one move on the go word: a new post is COPIED into the launch's device-side
            // slot (a thousand workgroups reading host memory over PCIe cost more than the tick they saved) and then raises the word
            // (2 S: plan S may be run); the host's request to end turns it odd behind everything posted (2 S + 1); a robot's
            // workgroup that waited out its bound has turned it odd itself.  Every move is a compare-and-swap from the value just
            // read, so the word is decided once for everybody.  Ends when the word is odd: no bound of its own is needed — the
            // robots' waits are all bounded, and the last of them ends the launch.
            int lingers = 0;
            if (threadIdx.x == 0) {
                const unsigned long long v = __hip_atomic_load(cold0().decision, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lingers = ((v >> 2) == seq && (v & 3ull) == RESIDENT_GO && cold_plan0().linger_ticks > 0) ? 1 : 0;
                if (lingers) __hip_atomic_fetch_max(cold_plan0().linger_go, 2ull * seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (__syncthreads_or(lingers)) {
                unsigned long long consumed = seq;  // every robot's workgroup has picked up the plan of this number (every thread keeps the same count)
                for (;;) {
                    // one look at the box and the word (thread 0), handed to everybody through LDS
                    unsigned long long *sh = reinterpret_cast<unsigned long long *>(lds + 32);  // [0] the go word, [1] the newest post
                    if (threadIdx.x == 0) {
                        const LingerBox *box = cold_plan0().linger_box;
                        unsigned long long g = __hip_atomic_load(cold_plan0().linger_go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (!(g & 1ull)) {
                            const unsigned long long posted = __hip_atomic_load(&box->posted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            const unsigned long long creq = __hip_atomic_load(&box->close_req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            sh[1] = posted;
                            if (posted <= (g >> 1) && creq >= seq) {  // asked to end — behind everything posted
                                if (__hip_atomic_compare_exchange_strong(cold_plan0().linger_go, &g, g + 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) g += 1ull;
                            }
                        }
                        if (g & 1ull) __hip_atomic_store(const_cast<unsigned long long *>(&box->closed), g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        sh[0] = g;
                    }
                    __syncthreads();
                    const unsigned long long g = sh[0], next = (g >> 1) + 1ull;
                    const bool pending_post = !(g & 1ull) && sh[1] >= next;
                    __syncthreads();
                    if (g & 1ull) break;
                    if (pending_post) {
                        // Post `next` goes from the box (host memory: ONE reader, wide loads) into the launch's device-side slot
                        // next & 1 — which held post next - 2: not before every robot's workgroup has picked that one up (their
                        // census words hold the number of the plan they picked up last).
                        bool free_slot = consumed + 2ull >= next;
                        while (!free_slot) {
                            int lag = 0;
                            for (unsigned b2 = threadIdx.x; b2 + 1 < gridDim.x; b2 += NT)
                                lag |= __hip_atomic_load(&cold0().census[b2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < consumed + 1ull ? 1 : 0;
                            if (__syncthreads_or(lag)) break;  // (not yet: look at the box and the word again first)
                            consumed += 1ull;
                            free_slot = consumed + 2ull >= next;
                        }
                        if (free_slot) {
                            // The word first: 2 next says plan `next` WILL run — from here on no robot's workgroup ends the launch
                            // behind the plan before (its give-up is an atomic max that finds the word moved on) — then the slot, as
                            // 16-byte chunks that carry the post's sequence word (mgx_dev.h: a chunk that shows it is complete): the
                            // robots' workgroups poll the chunks themselves, one round trip, and need no second look at anything.
                            int won = 0;
                            if (threadIdx.x == 0) {
                                unsigned long long e = g;
                                won = __hip_atomic_compare_exchange_strong(cold_plan0().linger_go, &e, 2ull * next, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
                                // (lost: a robot's workgroup has ended the launch — the next look finds the word odd)
                            }
                            if (__syncthreads_or(won)) {
                                const unsigned slot = (unsigned)(next & 1ull);
                                const LingerBox *box = cold_plan0().linger_box;
                                const unsigned n_rob = (unsigned)(gridDim.x - 1);
                                unsigned char *dst = cold_plan0().linger_dev + (size_t)slot * cold_plan0().linger_dev_stride;
                                const __amdgpu_buffer_rsrc_t rs_d = sc1_rsrc(dst, LINGER_SLOT_HEAD + LINGER_UPD_BYTES * n_rob);
                                const uint32_t seq = xrec_seq(next);
                                uint32_t has_upd = 0u;
                                if (threadIdx.x < LINGER_PLAN_DWORDS / 3) {
                                    const uint32_t *ps = reinterpret_cast<const uint32_t *>(&box->plan[slot]) + 3 * threadIdx.x;
                                    v4u32 c;
                                    c.x = __hip_atomic_load(ps + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                    c.y = __hip_atomic_load(ps + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                    c.z = __hip_atomic_load(ps + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                    c.w = seq;
                                    if (threadIdx.x == 0) has_upd = c.y;
                                    __builtin_amdgcn_raw_buffer_store_b128(c, rs_d, (int)(16u * threadIdx.x), 0, 16);
                                }
                                if (__syncthreads_or((int)has_upd)) {  // the tick's prior-update records: 32 bytes per robot in the box, three chunks here
                                    const v4u32 *src = reinterpret_cast<const v4u32 *>(cold_plan0().linger_upd + (size_t)slot * cold_plan0().linger_upd_stride);
                                    for (unsigned i0 = 0; i0 < n_rob; i0 += 2u * NT) {
                                        v4u32 a[2], b[2];
#pragma unroll
                                        for (int u = 0; u < 2; u++) {
                                            const unsigned i = i0 + (unsigned)u * NT + threadIdx.x;
                                            a[u] = i < n_rob ? __builtin_nontemporal_load(src + 2 * i) : v4u32{0u, 0u, 0u, 0u};
                                            b[u] = i < n_rob ? __builtin_nontemporal_load(src + 2 * i + 1) : v4u32{0u, 0u, 0u, 0u};
                                        }
#pragma unroll
                                        for (int u = 0; u < 2; u++) {
                                            const unsigned i = i0 + (unsigned)u * NT + threadIdx.x;
                                            if (i < n_rob) {
                                                const int o = (int)(LINGER_SLOT_HEAD + LINGER_UPD_BYTES * i);
                                                __builtin_amdgcn_raw_buffer_store_b128(v4u32{a[u].x, a[u].y, a[u].z, seq}, rs_d, o, 0, 16);
                                                __builtin_amdgcn_raw_buffer_store_b128(v4u32{a[u].w, b[u].x, b[u].y, seq}, rs_d, o + 16, 0, 16);
                                                __builtin_amdgcn_raw_buffer_store_b128(v4u32{b[u].z, b[u].w, 0u, seq}, rs_d, o + 32, 0, 16);
                                            }
                                        }
                                    }
                                }
                                __syncthreads();  // (every thread's loads from the box are back: the host may write that slot again)
                                if (threadIdx.x == 0) __hip_atomic_store(const_cast<unsigned long long *>(&box->taken), next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            }
                            continue;
                        }
                    }
                    __builtin_amdgcn_s_sleep(12);
                }
            }
        }
        return;
    }
    // (the workgroup's robot: the same in every lane, and said so — what is derived from it stays out of the vector registers)
    const int r = __builtin_amdgcn_readfirstlane(robot0 + xcd_local_index(blockIdx.x, census ? gridDim.x - 1 : gridDim.x));
    int tid = threadIdx.x;  // not const: resident launches make them opaque once per segment, see the segment loop
    const int role = tid >> 6;
    int lane = tid & 63;
    const int K = KT > 0 ? KT : w.K, E = 4 * K - 6, E1 = E + 1;
    const BlobLayout L(K);
    const int ZCOL = E;  // all-zero message column (absent edges)
    double *s_snap = lds + (PERSIST ? 154 : 0);            // [24][K] variable -> own-factor snapshots (resident: behind the parked argument pointer and the plan)
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
    const bool is_dyn = role == ROLE_DF && lane < n_dyn;
    const bool is_obs = role == ROLE_UF && lane < K - 2;
    const bool is_trk = role == ROLE_UF && lane >= K - 2 && lane < 2 * (K - 2);
    const bool is_var = role == ROLE_UV && lane < K;
    // Horizons of at most 16 variables: the belief finish runs on FOUR lanes per variable (lane (q, i) computes cofactor row q of
    // variable i, variable_finish_quad), and in resident launches the whole variable sweep — inbox sums and finish — stays on the
    // UV wave with no workgroup barrier in between (FUSED).
    constexpr bool QUADFIN = KT > 0 && 4 * KT <= 64;
    constexpr bool FUSED = PERSIST && QUADFIN;
    const int sum_t = FUSED ? (role == ROLE_UV ? lane : 4 * K) : tid;        // (variable, row) this thread sums: t = rr * K + i
    const int sum_step = FUSED ? 4 * K : NT;
    // which variable sweep of this launch is the robot's last one (its belief goes out); resident launches: per plan, set where a
    // plan begins (below, behind the staging barrier)
    bool has_int_var = !PERSIST && (int_mask & PH_INT_VARIABLE) && n_int > 0 && !idle;
    bool any_sweep = !PERSIST && (has_int_var || ((ext_mask & PH_EXT_VARIABLE) != 0 && radio));
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
    MGX_EDGE_REGS(pf_er);
    double pf_bmu[4] = {0.0, 0.0, 0.0, 0.0}, pf_rec[SNAP_W];
#pragma unroll
    for (int c = 0; c < SNAP_W; c++) pf_rec[c] = 0.0;
    uint8_t pf_gate = 0;
    int pf_dst = 0;
    // Which edge an edge lane takes.  The robot's incoming edges sit in the order of its variables' inboxes: variable 1's n_in
    // edges (one per incoming connection, by owner key), variable 2's, ... (a connection hangs one factor on each of the variables
    // 1 .. K-1 of its target: ne = n_in (K - 1), edge (variable v, connection c) = (v - 1) n_in + c).  Launch-per-segment kernels
    // take them as they come, lane q edge q.  Resident launches take them OWNER-MAJOR — lane q = connection q / (K - 1), variable
    // 1 + q % (K - 1) — so that consecutive lanes read consecutive variables' exchange records of one owner (mgx_dev.h: chunk-major
    // records, each lane fetches its own, contiguous requests) and both waves hold the same mix of near- and far-horizon factors.
    const int n_in = (PERSIST && K > 1) ? ne / (K - 1) : 0;
    auto edge_of_lane = [&](int q) __attribute__((always_inline)) {  // local index of lane q's edge (q < ne)
        if (!PERSIST) return q;
        const int c = q / (K - 1);
        return (q - c * (K - 1)) * n_in + c;
    };
    int my_j = (HAS_IR && tid < ne) ? edge_of_lane(tid) : 0;  // the thread's first edge (not const: opaque per segment, see the segment loop)
    if (HAS_IR && tid < ne) pf_gate = w.ir_gate[ie0 + my_j];
    if (do_extf && tid < ne) {
        MGX_EDGE_LOAD(pf_er, &w.ir_rec[ie0 + my_j]);
        pf_dst = pf_er_dst;
        ld_soa4(w.ir_bmu, w.NI, ie0 + my_j, pf_bmu);
    }
    if (PERSIST && tid < ne) {  // resident launches: the edge's constants stay in registers for every external iteration
        MGX_EDGE_LOAD(pf_er, &w.ir_rec[ie0 + my_j]);
        pf_dst = pf_er_dst;
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
        if (lane < 20 && role < 2) u_bel = blob[L.bel() + lane * K + (role == 0 ? K - 1 : 0)];
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
            const size_t e = (size_t)(ie0 + my_j);
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
            if (pf_on) pf_present = w.snap_epoch[w.cur][pf_er_src_var] > pf_er_created;
            quad_gather(fetch_plain, pf_on ? (unsigned)pf_er_src_var * (unsigned)(SNAP_W * sizeof(double)) : 0u, pf_rec);
        }
    };
    unsigned long long early_decision = 0ull;  // resident launches: the go / abort word as it stands when the staging loads are back
    {
        const double *src = w.snap[w.cur] + (size_t)v0 * SNAP_W;
        if constexpr (KT > 0) {
            StageRegs<20 * KT, NT> r_prior;
            StageRegs<BlobLayout(KT).inout_words() - 16 * KT, NT> r_io;
            constexpr int IT = (SNAP_W * KT + NT - 1) / NT;
            double r_snap[IT];
            r_prior.load(blob + L.prior(), tid);
            r_io.load(blob + L.mu(), tid);
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int t = tid + it * NT;
                r_snap[it] = (t < SNAP_W * K) ? src[t] : 0.0;
            }
            chain_second_link();
            if (census && BIG && tid == 0) early_decision = __hip_atomic_load(cold().decision, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r_prior.store(s_prior, tid);
            r_io.store(s_mu, tid);
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int t = tid + it * NT;
                if (t < SNAP_W * K) s_snap[(t % SNAP_W) * K + (t / SNAP_W)] = r_snap[it];
            }
        } else {
            copy_words(s_prior, blob + L.prior(), 20 * K, tid, NT);
            chain_second_link();
            if (census && BIG && tid == 0) early_decision = __hip_atomic_load(cold().decision, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            copy_words(s_mu, blob + L.mu(), L.inout_words() - 16 * K, tid, NT);
            for (int t = tid; t < SNAP_W * K; t += NT) s_snap[(t % SNAP_W) * K + (t / SNAP_W)] = src[t];
        }
        for (int t = tid; t < K; t += NT) s_covset[t] = 0;
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
                for (int c = 0; c < 6; c++) s_ir[my_j * IR_STRIDE + c] = r_ir[c];
            }
            for (int q = tid + NT; q < ne; q += NT) {  // robots with more edges than threads
                const int j = edge_of_lane(q);
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
    // ---- residency census (SegPlan): go or abort, before anything that cannot be taken back has been written ------------------
    // The decision is made by the launch's extra workgroup while this one stages (see the top of the kernel).  It is consumed
    // at the END of segment 0, in front of the first publication — by then it is microseconds old, the look costs nothing —
    // because nothing a workgroup does before that point survives an abort or hurts: LDS and registers are simply dropped, and
    // the one thing written to HBM, mgx_tick's prior updates right below, is a function of state the launch has not changed
    // (means, beliefs), so the launch-by-launch re-run writes the same values again.  Horizons beyond 33 variables keep part of
    // their tracking factors' state in HBM and update it inside the sweeps: they look right here, after staging (the look was
    // issued in front of the staging stores).  Still undecided when looked at: poll; a decider that never started (four times
    // the bound) is an abort too, by compare-and-swap so that nobody can decide differently.
    constexpr bool CENSUS_EARLY = BIG;
    auto census_says_abort = [&](unsigned long long v) __attribute__((always_inline)) {
        int abort_launch = 0;
        if (tid == 0) {
            const unsigned long long seq = cold_plan().launch_seq;
            const long long t0 = wall_clock64();
            while ((v >> 2) != seq) {
                __builtin_amdgcn_s_sleep(1);
                v = __hip_atomic_load(cold().decision, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((v >> 2) != seq && wall_clock64() - t0 > 4 * cold_plan().census_ticks) {
                    // (sharded worlds: only if the ranks' word says abort too — with this vote it does unless every rank,
                    // hence this one's decider, has signed in: then the decider is alive and about to say so here)
                    if (SHARD && cold_plan().agree_seq != 0ull) {
                        const unsigned a = agree_on_launch(cold().agree, cold_plan().agree_seq, (unsigned)cold().n_ranks, AGREE_ABORT);
                        if (a == RESIDENT_GO) continue;
                        if (a == AGREE_LOST) __hip_atomic_store(cold().sweep_err, cold_plan().agree_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    if (__hip_atomic_compare_exchange_strong(cold().decision, &v, seq * 4ull + RESIDENT_ABORT, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT)) {
                        __hip_atomic_store(cold().decision_host, seq * 4ull + RESIDENT_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        v = seq * 4ull + RESIDENT_ABORT;
                    }
                }
            }
            abort_launch = (v & 3ull) == RESIDENT_ABORT ? 1 : 0;
        }
        return __syncthreads_or(abort_launch) != 0;
    };
    if (census && CENSUS_EARLY && census_says_abort(early_decision)) return;  // nothing has been written: the world is as it was
    if (PLDS && w.upd) {  // mgx_tick's prior updates of the launch's own plan: parked where a posted plan's would be (see where a plan begins)
        if (tid < 4) s_urec[tid] = tid == 0 ? u_rec[0] : (tid == 1 ? u_rec[1] : (tid == 2 ? u_rec[2] : u_rec[3]));
        if (lane < 20 && role < 2) s_tmp[4 * K + role * 20 + lane] = u_bel;
        __syncthreads();
    }
    uint32_t my_epoch = 0u;  // deliveries of the variable this thread sums (set where a plan begins)
    STAMP(t_staged);

    // ======================= external factor sweep (pull form) ================================
    // factorgraph.rs:745-754 keeps only the message to the other graph's variable, so F_AB is
    // evaluated here, at B, from A's snapshot record and B's last response mean.
    // k: segment of a resident schedule launch (0 otherwise); store_fv: the HBM copy of the messages is needed
    // ONE descriptor over the two parities of exchange records (mgx_dev.h: contiguous), for the 16-byte agent-scope accesses of
    // resident launches; the parity is a scalar offset of the access
    const int VL = w.R_local * K;  // first ghost variable
    const unsigned xrec_bytes = PERSIST ? (unsigned)VL * (unsigned)XREC_BYTES : 0u;
    const __amdgpu_buffer_rsrc_t rs_x = sc1_rsrc(PERSIST ? w.xrec[0] : nullptr, 2u * xrec_bytes);
    // sharded worlds: the ghosts' records of segments k > 0 come from this rank's ghost area (their owners' ranks store them there
    // from inside their own launches), marked by the top bit of the byte offset a lane asks for
    constexpr unsigned GHOST_BIT = 0x80000000u;
    const unsigned gx_bytes = SHARD ? (unsigned)(w.V - VL) * (unsigned)XREC_BYTES : 0u;
    const __amdgpu_buffer_rsrc_t rs_gx = sc1_rsrc(SHARD ? w.gxrec[0] : nullptr, 2u * gx_bytes);
    // The gather of a resident launch's segments k > 0: the owners' EXCHANGE RECORDS (mgx_dev.h), polled.  Every edge lane asks for
    // the XREC_CHUNKS chunks of ITS owner variable's record — chunk c of variable i of robot A sits at (A K XREC_CHUNKS + c K + i) 16,
    // and consecutive lanes hold consecutive variables of one owner, so each load instruction asks for whole runs of (K - 1) x 16
    // contiguous bytes; a chunk is there when its sequence word is the expected one, and what is not there yet is asked for again —
    // only that.  The 45 payload dwords need no exchange between lanes: (eta, lam) and the two position means go to `out`, the
    // delivery count to `deliveries`.  off_mine: byte offset of chunk 0 of the lane's record (top bit: in the ghost area); lanes
    // without an edge ask for nothing.  A wait that outlasts the bound raises the world's abort word like any other wait of the
    // launch (reported, never a hang).
    auto gather_records = [&](auto fetch, bool mine, unsigned off_mine, uint32_t seq, double (&out)[SNAP_W], uint32_t &deliveries) __attribute__((always_inline)) {
        v4u32 R[XREC_CHUNKS];
        const unsigned cstride = 16u * (unsigned)K;
#pragma unroll
        for (int ch = 0; ch < XREC_CHUNKS; ch++) R[ch] = v4u32{0u, 0u, 0u, 0u};
        // (ghost records crossed the fabric: their sequence word carries a mix of the payload it arrived with, mgx_dev.h — undone
        // ONCE, where a chunk is fetched, and only in waves that gather a ghost record at all: fifteen mixes per look of every lane
        // of every robot were 0.7 us of a sharded world's iteration)
        const bool remote = SHARD && (off_mine & GHOST_BIT) != 0u;
        const bool any_remote = SHARD && __ballot(remote) != 0ull;
        if (mine) {
#pragma unroll
            for (int ch = 0; ch < XREC_CHUNKS; ch++) R[ch] = fetch(off_mine + (unsigned)ch * cstride);
        }
        if constexpr (SHARD) {  // (behind ALL the fetches: a mix between two of them would wait for the first before asking for the second)
            if (any_remote) {
#pragma unroll
                for (int ch = 0; ch < XREC_CHUNKS; ch++) R[ch].w ^= remote ? xrec_mix(R[ch].x, R[ch].y, R[ch].z) : 0u;
            }
        }
        auto word_of = [&](const v4u32 &c) __attribute__((always_inline)) { return c.w; };
        long long t0 = 0;
        for (unsigned spins = 0;; spins++) {
            bool missing = false;
#pragma unroll
            for (int ch = 0; ch < XREC_CHUNKS; ch++) missing = missing || word_of(R[ch]) != seq;
            missing = missing && mine;
            if (__ballot(missing) == 0ull) break;
            // (a waiting workgroup's re-requests load the memory pipeline of its CU, which the workgroups beside it — the ones it
            // may be waiting FOR — gather through: 2048 clocks between looks measured 2.5 % faster than 128, 8192 slower)
#ifdef MGX_POLL_SLEEP
            __builtin_amdgcn_s_sleep(MGX_POLL_SLEEP);
#else
            __builtin_amdgcn_s_sleep(32);
#endif
            if (spins == 0u) t0 = wall_clock64();
            if ((spins & 31u) == 31u) {
                bool stop = __hip_atomic_load(cold().sweep_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;
                if (!stop && wall_clock64() - t0 > cold_plan().timeout_ticks) {
                    __hip_atomic_store(cold().sweep_abort, (unsigned long long)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(cold().sweep_err, (unsigned long long)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    stop = true;
                }
                if (__ballot(stop) != 0ull) break;  // (the launch ends with wrong beliefs; the host reports it)
            }
            if (missing) {
                unsigned again = 0u;  // the chunks asked for again
#pragma unroll
                for (int ch = 0; ch < XREC_CHUNKS; ch++)
                    if (word_of(R[ch]) != seq) { again |= 1u << ch; R[ch] = fetch(off_mine + (unsigned)ch * cstride); }
                if constexpr (SHARD) {
                    if (remote) {
#pragma unroll
                        for (int ch = 0; ch < XREC_CHUNKS; ch++)
                            if ((again >> ch) & 1u) R[ch].w ^= xrec_mix(R[ch].x, R[ch].y, R[ch].z);
                    }
                }
            }
        }
        auto D = [&](int n) __attribute__((always_inline)) { return R[n / 3][n % 3]; };  // payload dword n
#pragma unroll
        for (int n = 0; n < 22; n++) out[n] = __hiloint2double((int)D(2 * n + 1), (int)D(2 * n));
        out[22] = 0.0;
        out[23] = 0.0;
        deliveries = D(XREC_EPOCH_DWORD);
#ifdef MGX_XREC_CHECKSUM
        if (mine) {
            unsigned x = 0u;
#pragma unroll
            for (int n = 0; n < XREC_PAYLOAD_DWORDS; n++) x ^= D(n);
            const unsigned o = off_mine & 0x7fffffffu, rob = o / ((unsigned)K * (unsigned)XREC_BYTES);
            const unsigned want_id = rob * 64u + (o - rob * (unsigned)K * (unsigned)XREC_BYTES) / 16u;  // robot * 64 + variable
            if (x != R[15].x || R[15].y != want_id)
                __hip_atomic_store(cold().sweep_err, 0xBAD0000000000000ull | ((unsigned long long)(R[15].y & 0xfffffu) << 28) | ((unsigned long long)(want_id & 0xfffffu) << 8) | (seq & 0xffu),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
#endif
    };
    // resident launches: the response means (ir_bmu) of every edge of variable i that is on air are the variable's mean after the
    // external variable sweep — kept in LDS (the scratch sums' block, idle between that sweep's finish and the next one's sums)
    // from one segment to the next; HBM gets them once, after the launch's last external iteration
    double *s_xmu = s_tmp;
    bool have_xmu = false;
#ifdef MGX_STAMPS
    unsigned long long q_arrive = 0ull, t_edges0 = 0ull;  // cycles from the start of the factor sweep until the thread's record is there
    // hand-off timeline (100 MHz wall clock, the same on every CU), per robot and segment k: [0] publication of segment k begins,
    // [1] its stores are issued, [2] the gather of segment k + 1 begins, [3] every record of that gather is there (UV wave)
    unsigned long long *tl = (PERSIST && w.dbg) ? w.dbg + (size_t)(w.R_local + 4) * 48 + (size_t)r * 64 : nullptr;
#define TLSTAMP(kk, what) do { if (tl && role == ROLE_UV && lane == 0 && (kk) >= 0 && (kk) < 16) tl[(kk) * 4 + (what)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TLSTAMP(kk, what)
#endif
    auto external_factor_sweep = [&](int k, int kg, bool store_fv) __attribute__((always_inline)) {  // k: segment of the plan in force, kg: of the launch
        const int buf = PERSIST ? ((w.cur + kg) & 1) : w.cur;  // snapshot buffer the owners' records are read from
#ifdef MGX_STAMPS
        t_edges0 = __builtin_readcyclecounter();
#endif
        if (radio && ir_on) {
            auto fetch_xrec = [&](unsigned off) __attribute__((always_inline)) {
                if constexpr (SHARD) {
                    if (off & GHOST_BIT) return ld16_system_raw(rs_gx, off & ~GHOST_BIT, buf ? gx_bytes : 0u);
                }
                return ld16_agent_raw(rs_x, off, buf ? xrec_bytes : 0u);
            };
            const uint32_t want_seq = xrec_seq(plan.flag_base + (unsigned long long)kg);
            for (int j0 = 0; j0 < ne; j0 += NT) {  // rounds of the whole workgroup: every lane takes part in the gather
                const int q = j0 + tid;                                        // edge lane
                const int j = j0 == 0 ? my_j : (q < ne ? edge_of_lane(q) : 0);  // its edge
                const int e = ie0 + j;
                MGX_EDGE_REGS(er);
                double ao_eta[4], ao_lam[16], a_mu[4], b_mu[4];
                bool a_present;
                // the round's record, unless it was prefetched while staging (launch-per-segment path, first round)
                double grec[SNAP_W];
                uint32_t grec_deliveries = 0u;
                bool mine = false;
                if (PERSIST || j0 > 0) {
                    if (j0 == 0) {
                        mine = tid < ne && pf_gate == 1;
                        if (mine) MGX_EDGE_COPY(er, pf_er);
                    } else if (j0 > 0 && q < ne && w.ir_gate[e] == 1) {  // robots with more edges than threads
                        mine = true;
                        MGX_EDGE_LOAD(er, &w.ir_rec[e]);
                    }
                    if (PERSIST && k > 0) {
                        // published by other workgroups of THIS launch (agent scope) or by other ranks' launches into this rank's
                        // ghost area (system scope): exchange records, polled
                        const bool ghost_src = SHARD && mine && er_src_var >= VL;
                        const int sv = mine ? (ghost_src ? er_src_var - VL : er_src_var) : 0, si = sv % K;  // variable si of robot sv / K
                        const unsigned off_mine = ((unsigned)(sv - si) * (unsigned)XREC_BYTES + 16u * (unsigned)si) | (ghost_src ? GHOST_BIT : 0u);
                        if (j0 == 0) TLSTAMP(k - 1, 2);
                        gather_records(fetch_xrec, mine, off_mine, want_seq, grec, grec_deliveries);
                        if (j0 == 0) TLSTAMP(k - 1, 3);
                        DELAY_AT(4, j0 == 0);
                        DELAY_AT(6, j0 == 0 && role == ROLE_UV);
                        DELAY_AT(7, j0 == 0 && role == ROLE_DYN);
                    } else {
                        // written by an earlier launch (segment 0 of a sharded launch reads the ghosts' plain copies, filled by the
                        // exchange in front of the launch)
                        quad_gather(fetch_plain, mine ? (unsigned)er_src_var * (unsigned)(SNAP_W * sizeof(double)) : 0u, grec);
                    }
                    if (!mine) continue;
                }

                if (PERSIST) {
                    if (have_xmu) {  // the means this robot's external variable sweep of the previous segment answered with
                        const int i = er_dst & 0xffff;
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
                        if (k > 0) a_present = grec_deliveries > er_created;  // the delivery count travels in the record
                        else a_present = w.snap_epoch[buf][er_src_var] > er_created;
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
                    if (q >= ne) continue;
                    if (!pf_on) continue;  // the owner did not run its external factor sweep
                    MGX_EDGE_COPY(er, pf_er);
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
                    a_present = w.snap_epoch[w.cur][er_src_var] > er_created;
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
                const int dslot = er_dst >> 16;
                double x_lo[4], x_hi[4], o6[6];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    x_lo[c] = dslot ? a_mu[c] : b_mu[c];
                    x_hi[c] = dslot ? b_mu[c] : a_mu[c];
                }
                const bool live_msg = interrobot_message_compact(x_lo, x_hi, er_d_safe, er_offset, w.inv_s2_ir, dslot, ao_eta, ao_lam, o6);
                if (!live_msg) {
#pragma unroll
                    for (int c = 0; c < 6; c++) o6[c] = 0.0;
                }
#ifdef MGX_STAMPS
                if (tl && k == 5 && j0 == 0 && role < 2) {  // who runs where, and how many of the wave's factors are inside their safety distance
                    const unsigned long long lv = __ballot(live_msg);
                    if (lane == __ffsll((unsigned long long)__ballot(true)) - 1) {
                        tl[60 + role] = (unsigned long long)__popcll(lv);
                        if (role == ROLE_UV) {
                            unsigned hw, xcc;
                            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                            tl[62] = ((unsigned long long)xcc << 32) | hw;
                        }
                    }
                }
#endif
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
            uint32_t epoch_reg = (4 * K <= NT) ? my_epoch : s_epoch[i];
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
                if (4 * K <= NT) my_epoch = epoch_reg;
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
    // (the tracking factors' switch padding, opaque where it is used: `pad * 0.01` hoisted in front of the segment loop was a pair of
    // registers held — and in the sharded instantiation spilled to scratch — for a path most worlds never take)
    auto trk_pad_here = [&]() __attribute__((always_inline)) {
        double p_ = w.trk_pad;
        asm volatile("" : "+v"(p_));
        return p_;
    };
    auto unary_messages = [&](uint32_t skip, double *s_out, int itf_gate) __attribute__((always_inline)) {
        if (obs_rows) {
            // four lanes per obstacle factor: lane q samples tap q and writes row q of the message
            if (role == ROLE_UF && lane < 4 * (K - 2) && (w.enable & 4u) && !(skip & 4u)) {
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
            if (!tracking_update(w.path_xy + 2 * (size_t)p0, np, trk_pad_here(), w.trk_attr, w.inv_s2_trk, x0, trk_rec,
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
        if (BIG && role == ROLE_UF && lane + 64 >= K - 2 && lane + 64 < 2 * (K - 2) && (w.enable & 8u) && itf_gate >= 10 &&
            !(skip & 8u)) {
            const int j2 = lane + 64 - (K - 2), var2 = j2 + 1, col2 = n_dyn + (K - 2) + j2, item2 = r * (K - 2) + j2;
            double x0[4], oe[4], ol[16];
#pragma unroll
            for (int c = 0; c < 4; c++) x0[c] = s_snap[(20 + c) * K + var2];
            int rec2 = w.trk_record[item2];
            float lp2[2] = {w.trk_last_pos[item2], w.trk_last_pos[(size_t)w.NT + item2]};
            double lv2 = w.trk_last_val[item2];
            const int p0 = w.path_ptr[r], np = w.path_ptr[r + 1] - p0;
            if (!tracking_update(w.path_xy + 2 * (size_t)p0, np, trk_pad_here(), w.trk_attr, w.inv_s2_trk, x0, rec2, lp2, lv2, oe, ol)) {
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

    // Resident schedule launch, flow control.  A robot's records of segment k + 1 overwrite those of segment k - 1 (two parities),
    // so every robot that reads them must be through with segment k - 1's gather first.  A reader whose OWN records this robot
    // gathers — the rule: inter-robot factors come in pairs — has said so already: its record of segment k, validated by this
    // robot's edge lanes, was published behind that gather.  Only a reader this robot does not read (the reference's bookkeeping
    // can leave a connection one-sided) is asked through its progress word: one lane per such peer polls it in front of segment
    // k's external iteration (relaxed agent-scope loads, s_sleep in between); a wait that outlasts the wall-clock bound raises
    // the world's abort word, which releases every waiter: the launch then ends with wrong beliefs and the host reports it
    // (never a hung GPU).  The peer list (owners of incoming and targets of outgoing connections) does not change during the
    // launch: lane l keeps peer l in a register, and whether that peer owns one of this robot's incoming connections (the
    // edges of variable 1 are one per connection; a peer off the air is not polled at all).
    int my_peer = -1, peer_q0 = 0, peer_q1 = 0;
    bool my_peer_read = false;  // this robot gathers that peer's records in every external iteration
    if (PERSIST && radio && ir_on && role == ROLE_UV) {
        peer_q0 = w.peer_ptr[r];
        peer_q1 = w.peer_ptr[r + 1];
        if (peer_q0 + lane < peer_q1) {
            const int pr = w.peer_idx[peer_q0 + lane];
            if (w.antenna[pr] && !w.idle[pr]) my_peer = pr;  // not on air: neither reads our records nor has its own read
        }
        for (int e = w.ir_var_ptr[v0 + 1]; e < w.ir_var_ptr[v0 + 2]; e++) my_peer_read = my_peer_read || w.ir_rec[e].src_robot == my_peer;
    }
    // ghosts_only (sharded worlds, see the end of a segment): look at the peers on other ranks only
    auto progress_of = [&](int pr) __attribute__((always_inline)) {
        if (SHARD && pr >= w.R_local)  // a ghost: its owner's rank stores the word into this rank's ghost area
            return __hip_atomic_load(&cold().gflag[pr - w.R_local], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return __hip_atomic_load(&w.sweep_flag[pr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto wait_for_progress = [&](unsigned long long want, bool ghosts_only) __attribute__((always_inline)) {
        if (PERSIST && radio && ir_on && role == ROLE_UV) {
            for (int q = peer_q0 + lane; q < peer_q1; q += 64) {
                int pr = my_peer;
                if (q != peer_q0 + lane) {  // more than 64 peers: the rest is looked up
                    pr = w.peer_idx[q];
                    if (!w.antenna[pr] || w.idle[pr]) pr = -1;
                }
                if (pr < 0 || (ghosts_only && pr < w.R_local)) continue;
                if (!ghosts_only && q == peer_q0 + lane && my_peer_read) continue;  // its records speak for it (see above)
                const long long t0 = wall_clock64();
                unsigned spins = 0;
                while (progress_of(pr) < want) {
                    __builtin_amdgcn_s_sleep(4);
                    if ((++spins & 31u) == 0u) {
                        if (__hip_atomic_load(cold().sweep_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) break;
                        if (wall_clock64() - t0 > cold_plan().timeout_ticks) {
                            // (what is reported: the launch's first segment count — any non-zero word stops the waiters; `want`
                            // itself kept alive into this path was one more register pair to spill)
                            const unsigned long long word = cold_plan().flag_base + 1ull;
                            __hip_atomic_store(cold().sweep_abort, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(cold().sweep_err, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            break;
                        }
                    }
                }
            }
        }
    };
    auto wait_for_peers = [&](int k) __attribute__((always_inline)) { wait_for_progress(plan.flag_base + (unsigned long long)k, false); };
    // sharded worlds: the ranks that hold this robot as a ghost
    int xp0 = 0, xp1 = 0;
    if (SHARD && role == ROLE_UV) {  // (wave-uniform, and told so: the push records are then fetched by scalar loads, not into registers)
        xp0 = __builtin_amdgcn_readfirstlane(cold().xp_ptr[r]);
        xp1 = __builtin_amdgcn_readfirstlane(cold().xp_ptr[r + 1]);
    }

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
    int last_int_seg = -1, last_ext_seg = -1;  // PERSIST: last segment with internal iterations / an external iteration (of the plan in force)
    // LINGER: plans run so far, and the launch-wide index of the current plan's segment 0 — buffer parities, sequence numbers and
    // progress counts derive from kbase + k, so that they run on over the plans exactly as over the segments of one plan (the
    // first segment of a posted plan has no external iteration: it CONTINUES the last segment of the plan before — same index)
    // (LINGER: none of it lives in registers across the segments — see s_ps, s_seg)
    int kbase = 0;
    if (PLDS && lane == 0) s_ps[0] = 0u;
    if (PLDS) __builtin_amdgcn_wave_barrier();
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
    for (int k = 0; PLDS || k < nseg; k++) {
        if (k == 0) {  // ======================= a plan begins: the launch's own, or (LINGER) one the host posted into the running launch
            const int plans_done = PLDS ? (int)ps_word(0) : 0;
            const bool upd_now = PLDS ? plan_dword(1) != 0u : w.upd != nullptr;  // the plan carries mgx_tick's prior updates (u_rec, u_bel)
            // ---- mgx_tick: update_prior_of_horizon_state (wave 0, variable K-1) and update_prior_of_current_state_v3
            // (wave 1, variable 0) on the staged image, each ending in change_prior of that variable
            // (robot.rs:2182-2338, variable.rs:203-230; same arithmetic as k_update_priors / apply_change_prior).  For
            // K >= 3 the two touch disjoint state, and nobody else reads this robot's snapshot in a launch without an
            // external factor sweep, so the change needs no other synchronisation than the barrier below.
            if (upd_now) {
                // (LINGER: the record and the stale (eta, lam) entries wait in LDS — the launch's own parked there behind the staging
                // barrier, a posted plan's at the end of the plan before: nothing of them stays in registers across the segments)
                if constexpr (PLDS) {
#pragma unroll
                    for (int c = 0; c < 4; c++) u_rec[c] = s_urec[c];
                    u_bel = (lane < 20 && role < 2) ? s_tmp[4 * K + role * 20 + lane] : 0.0;
                }
                const double upd_max_speed = PLDS ? reinterpret_cast<const double *>(s_plan + 2 + MAX_SEGS / 2)[0] : w.upd_max_speed;
                const double upd_delta_t = PLDS ? reinterpret_cast<const double *>(s_plan + 2 + MAX_SEGS / 2)[1] : w.upd_delta_t;
                const uint32_t what = (uint32_t)u_rec[3];
                const int i = role == 0 ? K - 1 : 0;
                if (role < 2 && (role == 0 ? (what & 1u) : (what & 2u))) {
                    double m[4];
                    if (role == 0) {
                        const double ex = s_mu[0 * K + i], ey = s_mu[1 * K + i];         // estimated position (:2242)
                        double hx = u_rec[0] - ex, hy = u_rec[1] - ey;                    // horizon2waypoint
                        const double dist = std::sqrt(hx * hx + hy * hy);                 // euclidean_norm
                        double nx = hx, ny = hy;                                           // .normalized(): unchanged if |.| is 0 / inf
                        if (!(dist == 0.0 || std::isinf(dist))) { nx = hx / dist; ny = hy / dist; }
                        const double sp = (upd_max_speed < dist || dist != dist) ? upd_max_speed : dist;  // Float::min(max_speed, dist)
                        const double vx = sp * nx, vy = sp * ny;                           // new_velocity
                        m[0] = ex + vx * upd_delta_t; m[1] = ey + vy * upd_delta_t; m[2] = vx; m[3] = vy;  // (:2253-2256)
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
                        // (a plan posted into a lingering launch: the response means of the variable's foreign factors live in LDS)
                        if (PERSIST && have_xmu) s_xmu[(lane - 4) * K + i] = mc;
                    } else if (lane == 8) {
                        s_epoch[i] += 1;
                    }
                    if (lane < 20) s_snap[lane * K + i] = u_bel;  // (stale eta, stale lam) of that message
                    // every inbox message of the variable becomes empty (:224-227)
                    const int es[4] = {(i >= 1) ? (K - 1) + (i - 1) : -1, (i <= K - 2) ? i : -1,
                                       (i >= 1 && i <= K - 2) ? n_dyn + (i - 1) : -1, (i >= 1 && i <= K - 2) ? n_dyn + (K - 2) + (i - 1) : -1};
                    for (int t = lane; t < 80; t += 64) {
                        const int col = es[t & 3];
                        if (col >= 0) {
                            s_fv[(t >> 2) * E1 + col] = 0.0;
                            if (PERSIST) s_sh[(t >> 2) * E1 + col] = 0.0;  // (the shadow block too: the launch's first plan copies it below, a posted one finds it in use)
                        }
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
            if (PERSIST && plans_done == 0) {  // columns no sweep recomputes (disabled kinds, tracking in front of its gate) must be equal in both
                for (int t = tid; t < 20 * E1; t += NT) s_sh[t] = s_fv[t];
                __syncthreads();
            }
            my_epoch = (sum_t < 4 * K) ? s_epoch[sum_t % K] : 0u;
            if constexpr (PLDS) {
                // the plan's bytes, one segment per lane: two ballots say which segments have internal iterations / an external one
                nseg = (int)plan_dword(0);
                const int q = lane & (MAX_SEGS - 1), q1 = (q + 1) & (MAX_SEGS - 1);
                auto byte_of = [&](int base, int i) __attribute__((always_inline)) { return (s_plan[base + (i >> 2)] >> (8 * (i & 3))) & 0xffu; };
                const uint32_t e_q = byte_of(2, q), n_q = byte_of(2 + MAX_SEGS / 4, q);
                const uint32_t e_1 = q + 1 < MAX_SEGS ? byte_of(2, q1) : 0u, n_1 = q + 1 < MAX_SEGS ? byte_of(2 + MAX_SEGS / 4, q1) : 0u;
                const bool in = lane < nseg;
                const unsigned long long m_int = __ballot(in && n_q > 0u), m_ext = __ballot(in && e_q != 0u);
                last_int_seg = m_int ? 63 - __clzll((long long)m_int) : -1;
                last_ext_seg = m_ext ? 63 - __clzll((long long)m_ext) : -1;
                has_int_var = m_int != 0ull && !idle;
                any_sweep = has_int_var || (m_ext != 0ull && radio);
                // (segment 0 of a posted plan continues the last segment of the plan before: the launch-wide index does not move)
                kbase = plans_done ? (int)ps_word(1) - 1 : 0;
                const bool lingers = LINGER && census && plan.linger_ticks > 0;
                if (lane < MAX_SEGS) {
                    const uint32_t bits = (e_q ? SEG_EXT : 0u) | (lane == nseg - 1 ? SEG_LAST : 0u) | (lane == last_int_seg ? SEG_LAST_INT : 0u) |
                                          (lane == last_ext_seg ? SEG_LAST_EXT : 0u) | ((lane == 0 && plans_done == 0) ? SEG_FIRST_OF_LAUNCH : 0u) |
                                          (lingers ? SEG_LINGERS : 0u);
                    s_seg[lane] = uint2{bits | (n_q << 8) | ((e_1 ? 1u : 0u) << 16) | (n_1 << 24), (uint32_t)(kbase + lane)};
                }
                __builtin_amdgcn_wave_barrier();  // (ps_word(1) above is read before it is written below)
                if (lane == 0) {
                    s_ps[1] = (uint32_t)(kbase + nseg);
                    s_ps[2] = (has_int_var ? 1u : 0u) | (any_sweep ? 2u : 0u);
                }
                __builtin_amdgcn_wave_barrier();  // (the wave's own LDS reads below follow its LDS writes above)
            } else if (PERSIST) {
                bool plan_int = false, plan_has_ext = false;  // some segment has internal iterations / an external iteration
                last_int_seg = last_ext_seg = -1;
                for (int q = 0; q < nseg; q++) {
                    if (plan_n_int(q) > 0) { plan_int = true; last_int_seg = q; }
                    if (plan_ext(q)) { plan_has_ext = true; last_ext_seg = q; }
                }
                has_int_var = plan_int && !idle;
                any_sweep = has_int_var || (plan_has_ext && radio);
            }
        }
        // this segment's (and the next one's) bytes of the plan, and what is derived from the plan as a whole (see s_ps, s_seg)
        uint32_t seg_b = 0u;
        int kg = k;  // the segment's launch-wide index
        if constexpr (PLDS) {
            // (asked for here and waited for at once: the word of the NEXT segment fetched a segment ahead and turned scalar at its end
            // measured no better, 7.67 - 7.77 us per iteration against 7.66 - 7.68 — experiments/README.md)
            const uint2 e = s_seg[k];
            seg_b = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x);
            kg = __builtin_amdgcn_readfirstlane((int)e.y);
        } else if (PERSIST) {
            seg_b = (plan_ext(k) ? SEG_EXT : 0u) | ((uint32_t)plan_n_int(k) << 8) | (k == nseg - 1 ? SEG_LAST : 0u) | (k == last_int_seg ? SEG_LAST_INT : 0u) |
                    (k == last_ext_seg ? SEG_LAST_EXT : 0u) | (k == 0 ? SEG_FIRST_OF_LAUNCH : 0u);
            if (k + 1 < nseg) seg_b |= ((plan_ext(k + 1) ? 1u : 0u) << 16) | ((uint32_t)plan_n_int(k + 1) << 24);
        }
        const bool is_last_int_seg = (seg_b & SEG_LAST_INT) != 0u, is_last_ext_seg = (seg_b & SEG_LAST_EXT) != 0u;
        const bool lingers = LINGER && (seg_b & SEG_LINGERS) != 0u;
        // Nothing derived from the thread index stays live across segments: left alone, the compiler hoists every per-thread
        // address and predicate of the loop body in front of the loop and then spills them around the f64 blocks (64 spilled
        // VGPRs, 244 B of scratch per lane at K = 16); recomputing them per segment is a handful of integer instructions.
        // (... nor from the per-thread indices made of it in front of the loop: a kept copy of `&w.ir_bmu[c * NI + ie0 + my_j]` is two
        // registers per row, and the sharded instantiation spilled four of those to scratch)
        if (PERSIST) asm volatile("" : "+v"(tid), "+v"(lane), "+v"(my_j), "+v"(pf_dst));
        const uint32_t ext_k = PERSIST ? ((seg_b & SEG_EXT) ? (PH_EXT_FACTOR | PH_EXT_VARIABLE) : 0u) : ext_mask;
        const uint32_t int_k = PERSIST ? (PH_INT_FACTOR | PH_INT_VARIABLE) : int_mask;
        const int n_int_k = PERSIST ? (int)((seg_b >> 8) & 0xffu) : n_int;
        const bool last_seg = PERSIST ? (seg_b & SEG_LAST) != 0u : true;
        // ======================= external factor sweep ============================================
        PSTAMP(ps0);
        QBEGIN(qt);
        if (PERSIST && ext_k && k > 0) {  // k == 0: the launch boundary has published everything
            wait_for_peers(kg);  // (one-sided readers only: nobody, as a rule)
            QSTAMP(0, qt);
            // The edge lanes read the response means the previous external variable sweep left in LDS: written in front of a
            // barrier when the two variable sweeps ran side by side (the steady state of an alternating schedule: each wave goes
            // straight on to its gather), behind the last one otherwise.
            if (!(FUSED && par_done)) __syncthreads();
            QSTAMP(1, qt);
        }
        PSTAMP(ps1);
        if (ext_k & PH_EXT_FACTOR) {
            external_factor_sweep(k, kg, !PERSIST || is_last_ext_seg);
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
                const bool is_last = is_last_int_seg && last_seg;
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
                    if (role == ROLE_DYN && lane < 4 * K) {
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
                DELAY_AT(5, true);
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
                // (a lingering launch keeps them in LDS behind its plan's last external iteration as well: a posted plan goes on from there)
                if (ir_on && (!is_last_ext_seg || lingers)) have_xmu = true;
                if (ir_on && is_last_ext_seg) {  // the plan's last external iteration: the response means go to HBM (robot.rs:1842-1858)
                    for (int q = tid; q < ne; q += NT) {
                        const int e = ie0 + (q == tid ? my_j : edge_of_lane(q));
                        int dst;
                        if (q == tid) {
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
                const bool keep_means = ir_on && (!is_last_ext_seg || lingers);
                if (role == ROLE_UV) {
                    variable_sums(s_sum, false, false);  // reads the messages of the last internal factor sweep (s_fv)
                    QSTAMP(4, qt);
                    finish(s_sum, false);
                    // the response means stay in LDS for the next segment's factor sweep (same wave: these writes follow the
                    // finish's reads of the eta sums they overwrite)
                    if (keep_means && lane < 4 * K) s_xmu[lane] = s_mu[lane] - 0.0;
                }
                // (the unary factors linearise at the means of the last INTERNAL sweep, which this external one does not touch:
                // with four waves they run beside it)
                if (prefired && (NW == 4 || role == ROLE_UV)) unary_messages(0u, s_sh, itf);
                if (prefired && is_dyn && (w.enable & 1u)) dynamic_messages(s_sh);
                QSTAMP(6, qt);
                if (keep_means) have_xmu = true;
                if (ir_on && is_last_ext_seg) {  // the plan's last external iteration: the means go to HBM (robot.rs:1842-1858)
                    __syncthreads();
                    for (int q = tid; q < ne; q += NT) {
                        const int e = ie0 + (q == tid ? my_j : edge_of_lane(q));
                        int dst;
                        if (q == tid) {
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
            if (PERSIST && radio && ir_on && (!is_last_ext_seg || lingers)) {
                if (tid < 4 * K) s_xmu[tid] = s_mu[tid] - 0.0;  // read after the barrier that opens the next segment's factor sweep
                have_xmu = true;
            }
            if (radio && ir_on && (!PERSIST || is_last_ext_seg)) {
                // responses to the foreign factors attached to our variables, routed to their inbox
                // (robot.rs:1842-1858): only the mean of that inbox entry is ever used (it sets the
                // linearisation point; eta / lam of the target side never reach the kept message).
                // Plain stores of LDS values: nothing in this launch but the storing thread itself reads them
                // (the means are next written after the barrier that ends the coming factor sweep / by nobody).
                for (int q = tid; q < ne; q += NT) {
                    const int e = ie0 + (q == tid ? my_j : edge_of_lane(q));
                    int dst;
                    if (q == tid && (do_extf || PERSIST)) {  // gate and constants of the thread's first edge are in registers
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
            adopt_early(tid, NT);
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
                if (pending) {
                    finish(s_snap, true);
                    if (NW == 4) __syncthreads();  // four waves: the unary factors' wave reads the means the UV wave has just completed
                }
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
                const bool is_last = it == n_int_k - 1 && (!PERSIST || (is_last_int_seg && (last_seg || !radio)));
                variable_sums(s_snap, true, is_last);
                pending = true;
                STAMP(t4);
                __syncthreads();
                STAMP(t5);
                STAMP_ADD(c_v, t3, t4);
                STAMP_ADD(c_vb, t4, t5);
            }
        }
        // segment 0 of a resident launch is through: go or abort (see the residency census above) before anything is published
        if (census && !CENSUS_EARLY && (seg_b & SEG_FIRST_OF_LAUNCH)) {
            const unsigned long long v = tid == 0 ? __hip_atomic_load(cold().decision, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            if (census_says_abort(v)) return;
        }
        // ======================= end of a segment of a resident schedule launch ====================
        // What this robot's variables last sent to their own factors — all that another robot's inter-robot factors read — goes out
        // for the external iteration that opens the next segment (exchange records, below).
        PSTAMP(ps4);
        if (PERSIST && !last_seg) {
            QSTAMP(9, qt);
            if (role == ROLE_UV && pending) finish(s_snap, true);
            // four waves: the means are complete (here, or in the merge of the side-by-side sweeps above) — the unary factors' wave
            // goes on to the sweep computed ahead, the dynamic factors' wave with it, the chain waves to publication and gather
            if (NW == 4) __syncthreads();
            // The robot's EXCHANGE RECORDS (mgx_dev.h) for the external iteration that opens the next segment, into the parity nobody
            // reads during this segment: item t = chunk t / K of variable t % K, at byte 16 t of the robot's block — a wave's stores are
            // one contiguous kilobyte per instruction, and its LDS reads (three payload dwords out of two neighbouring f64 rows of the
            // snapshot image) run along the variables, conflict-free.  Fire and forget: a chunk validates itself, so nothing is drained
            // and no word follows the stores on the consumers' critical path.  (The (eta, lam) chunks 0 .. 12 are final before the
            // means and could go out from the other wave meanwhile: measured 2.5 % SLOWER on the two-wave kernel, where that wave's
            // factor sweep is as long as this one's publication and unary sweep together — experiments/README.md.)
            const int ob = (w.cur + kg + 1) & 1;
            const unsigned long long next_count = plan.flag_base + (unsigned long long)kg + 1ull;
            auto chunk_of = [&](int t, uint32_t seq) __attribute__((always_inline)) {
                const int ch = t / K, i = t - ch * K, da = (3 * ch) >> 1;
                v4u32 v;
#ifdef MGX_XREC_CHECKSUM
                if (ch == XREC_CHUNKS - 1) {  // the xor of the variable's 45 payload dwords, and who published them
                    unsigned x = s_epoch[i];
                    for (int n = 0; n < 22; n++) { const double f = s_snap[n * K + i]; x ^= (unsigned)__double2loint(f) ^ (unsigned)__double2hiint(f); }
                    v.x = x; v.y = (unsigned)(r * 64 + i); v.z = 0u; v.w = seq;
                    return v;
                }
#endif
                const double A = s_snap[da * K + i], B = s_snap[(da + 1) * K + i];
                v.x = (ch & 1) ? (unsigned)__double2hiint(A) : (unsigned)__double2loint(A);
                v.y = (ch & 1) ? (unsigned)__double2loint(B) : (unsigned)__double2hiint(A);
                v.z = (ch & 1) ? (unsigned)__double2hiint(B) : (unsigned)__double2loint(B);
                if (3 * ch + 2 == XREC_EPOCH_DWORD) v.z = s_epoch[i];
                v.w = seq;
                return v;
            };
            const unsigned xbase = (unsigned)v0 * (unsigned)XREC_BYTES;
            if (role == ROLE_UV) {
                QSTAMP(10, qt);
                __builtin_amdgcn_wave_barrier();  // the wave's LDS writes (means) precede its LDS reads below
                DELAY_AT(1, true);
                TLSTAMP(k, 0);
                {
                    const uint32_t seq = xrec_seq(next_count);
                    for (int t = lane; t < XREC_CHUNKS * K; t += 64) st16_agent_raw(rs_x, xbase + 16u * (unsigned)t, ob ? xrec_bytes : 0u, chunk_of(t, seq));
                }
                QSTAMP(11, qt);
                DELAY_AT(2, true);
                TLSTAMP(k, 1);
                // "through with segment k's gather": for the one-sided readers' sake only (wait_for_peers)
                if (lane == 0) __hip_atomic_store(&w.sweep_flag[r], next_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                QSTAMP(12, qt);
                if (SHARD && xp1 > xp0) {
                    // A boundary robot of a sharded world: the same records go into the ghost area of every rank that holds this
                    // robot as a ghost (system-scope write-through stores over xGMI), numbered in THAT rank's count of segments.
                    // Nothing orders two RANKS' launches: the parity written here is the one those ranks' robots read in the LAST
                    // external iteration of the previous launch, so in segment 0 wait until they are through with it (they say so
                    // at the end of their launch, below); in later segments their records of the segment in between have.
                    if (k == 0) wait_for_progress(plan.flag_base, true);
                    for (int t = xp0; t < xp1; t++) {
                        const XPushRec xr = cold().xp_rec[t];
                        // (no indexing of the record by the parity: a copy that is indexed at run time lives in scratch)
                        const __amdgpu_buffer_rsrc_t rs_peer = uniform_rsrc(ob ? xr.xrec[1] : xr.xrec[0], (unsigned)K * (unsigned)XREC_BYTES);
                        const uint32_t seq = xrec_seq(next_count + xr.flag_delta);
                        for (int t2 = lane; t2 < XREC_CHUNKS * K; t2 += 64) {
                            v4u32 c = chunk_of(t2, seq);
                            c.w ^= xrec_mix(c.x, c.y, c.z);  // (across the fabric: the word vouches for the payload it travels with)
                            st16_system_raw(rs_peer, 16u * (unsigned)t2, c);
                        }
                        if (lane == 0)
                            __hip_atomic_store(reinterpret_cast<unsigned long long *>(xr.flag), next_count + xr.flag_delta, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
            pending = false;
            // the factor sweep that opens the next segment's internal iterations, while the records travel: the DYN wave
            // starts at once (a dynamic factor reads no mean), the UV wave after its publish.  The tracking factors' gate
            // (factorgraph.rs:701) counts the external factor sweep that the reference runs in between.
            early = (seg_b >> 24) > 0u && !idle && skip0 == 0u;
            DELAY_AT(3, role == ROLE_DYN);
            if (early) {
                if (is_dyn && (w.enable & 1u)) dynamic_messages(s_sh);
                unary_messages(0u, s_sh, itf + ((((seg_b >> 16) & 1u) && radio) ? 1 : 0));
            }
            QSTAMP(13, qt);
        }
#ifdef MGX_STAMPS
        {
            PSTAMP(ps5);
            p_wait += ps1 - ps0; p_extf += ps2 - ps1; p_extv += ps3 - ps2; p_int += ps4 - ps3; p_pub += ps5 - ps4;
        }
#endif
        // ======================= end of a plan of a LINGERING launch (mgx_dev.h) ===========================
        // The graph stays where it is.  What the NEXT plan's start needs of this one's end is put aside first — the belief (eta, lam)
        // image goes to HBM as the tail would write it (a later plan may not sweep this robot: the image is the prior's place) and
        // hands the coming prior updates their "stale (eta, lam)" (variable.rs:210-221), the prior is asked for again — then thread 0
        // waits for the go word: the next plan's number or more -> plan and prior-update record are read from the host-mapped box in
        // one round trip and the loop goes on at segment 0 of that plan, same launch-wide index; the word odd -> tail and
        // write-back, as every launch ends.  The wait is bounded (plan.linger_ticks, and the world's abort word ends it at once):
        // whoever waits it out raises the word itself — an atomic max, so one outcome for all.
        if constexpr (PLDS) {
            if (last_seg && !lingers) break;
        }
        if constexpr (LINGER) {
            if (last_seg) {
                const int plans_done = (int)ps_word(0);
                any_sweep = (ps_word(2) & 2u) != 0u;
                // (the wave that polls is the one with nothing left to do in this plan: the other one's last variable sweep — its means,
                // covariances — runs while the poll's round trip is under way)
                // Wave 0 polls: lanes 0 .. 7 the chunks of the next plan, 8 .. 10 this robot's prior-update record, lane 11 the go word —
                // ONE round trip where the post is there already (the rule: the host runs a plan or two ahead).  Chunks complete (their
                // sequence word is the post's): the plan will run — the postman moved the word before it wrote them.  The word odd at
                // this plan: the launch ends here.  Neither, and nothing promised (the word still at this plan) for the bound: this
                // workgroup raises the word itself, an atomic max, and takes what it finds.
                int *s_verdict = reinterpret_cast<int *>(lds + 1);
                const unsigned long long number = cold_plan().launch_seq + (unsigned long long)plans_done + 1ull;
                if (role == 0) {
                    const unsigned long long cur2 = 2ull * (number - 1ull);
                    unsigned long long *go = cold_plan().linger_go;
                    const uint32_t seq = xrec_seq(number);
                    const unsigned char *dslot = cold_plan().linger_dev + (size_t)(number & 1ull) * cold_plan().linger_dev_stride;
                    const __amdgpu_buffer_rsrc_t rs_s = sc1_rsrc(dslot, LINGER_SLOT_HEAD + LINGER_UPD_BYTES * (unsigned)w.R_local);
                    const unsigned my_off = lane < 8 ? 16u * (unsigned)lane : LINGER_SLOT_HEAD + LINGER_UPD_BYTES * (unsigned)r + 16u * (unsigned)(lane < 11 ? lane - 8 : 0);
                    v4u32 c = v4u32{0u, 0u, 0u, 0u};
                    unsigned long long gw = 0ull;
                    int verdict = -1;
                    long long t0 = 0;
                    for (unsigned spins = 0; verdict < 0; spins++) {
                        if (lane < 11) c = ld16_agent_raw(rs_s, my_off);
                        else if (lane == 11) gw = __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long ok = __ballot(lane < 11 && c.w == seq);
                        const uint32_t has_upd = (uint32_t)__builtin_amdgcn_readlane((int)c.y, 0);
                        unsigned long long g = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(gw >> 32), 11) << 32) |
                                               (uint32_t)__builtin_amdgcn_readlane((int)(gw & 0xffffffffull), 11);
                        if ((ok & 0xffull) == 0xffull && (!has_upd || (ok & 0x700ull) == 0x700ull)) { verdict = 1; break; }
                        if (g == cur2 + 1ull) { verdict = 0; break; }
                        if (spins == 0u) t0 = wall_clock64();
                        if (spins < 8u) __builtin_amdgcn_s_sleep(2);
                        else __builtin_amdgcn_s_sleep(24);
                        if ((spins & 7u) == 7u && g <= cur2 &&
                            (__hip_atomic_load(cold().sweep_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull || wall_clock64() - t0 > cold_plan().linger_ticks)) {
                            unsigned long long old = 0ull;
                            if (lane == 11) old = __hip_atomic_fetch_max(go, cur2 + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            g = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(old >> 32), 11) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(old & 0xffffffffull), 11);
                            if (g <= cur2 + 1ull) { verdict = 0; break; }  // (more: the postman moved first — the chunks are on their way)
                        }
                    }
                    if (verdict == 1) {  // payload: three dwords per chunk
                        if (lane < 8) { s_plan[3 * lane] = c.x; s_plan[3 * lane + 1] = c.y; s_plan[3 * lane + 2] = c.z; }
                        else if (lane < 11) {
                            uint32_t *u = reinterpret_cast<uint32_t *>(s_urec) + 3 * (lane - 8);
                            u[0] = c.x; u[1] = c.y;
                            if (lane < 10) u[2] = c.z;
                        }
                    }
                    if (lane == 0) *s_verdict = verdict;
                }
                if (role == ROLE_UV && pending) finish(s_snap, true);
                pending = false;
                __syncthreads();
                const int iu = role == 0 ? K - 1 : 0;
                // (parked behind the response means in the scratch sums' block, idle between the plans)
                if (lane < 20 && role < 2) s_tmp[4 * K + role * 20 + lane] = any_sweep ? s_prior[lane * K + iu] : blob[L.bel() + lane * K + iu];
                if (role == ROLE_DYN && any_sweep) copy_words_wave(blob + L.bel(), s_prior, 20 * K, lane);
                StageRegs<(KT > 0 ? 20 * KT : 2), NT> r_prior;
                if constexpr (KT > 0) r_prior.load(blob + L.prior(), tid);
                __syncthreads();
                if (*s_verdict != 1) break;  // the launch ends behind this plan
                if (lane == 0) s_ps[0] = (uint32_t)(plans_done + 1);  // (every wave's own copy)
                {
                    if constexpr (KT > 0) r_prior.store(s_prior, tid);
                    else copy_words(s_prior, blob + L.prior(), 20 * K, tid, NT);
                    __syncthreads();
                    // "this workgroup has picked up plan `number`": the slot it was read from may be written again once everybody has
                    if (tid == 0) __hip_atomic_store(&cold().census[blockIdx.x], number, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                k = -1;  // (the plan that begins: see the top of the loop)
            }
        }
    }
    int total_segs = nseg;  // segments the launch ran (parities, sequence numbers and progress counts moved on by as many)
    if constexpr (PLDS) {  // (see s_ps; the staging barrier, or the one where the last plan began, lies behind)
        total_segs = (int)ps_word(1);
        any_sweep = (ps_word(2) & 2u) != 0u;
    }
    {
        // Tail: the UV wave completes the last variable sweep (mean, covariance) while the DYN wave
        // already writes back what that does not touch — the factor -> variable messages and the belief
        // (eta, lam) image, three quarters of the robot's output.
        if (role == ROLE_UV) {
            if (pending) finish(s_snap, true);
        } else if (role == ROLE_DYN) {
            copy_words_wave(blob + L.fv(), s_fv, 20 * E1, lane);
            if (any_sweep && !bel_dead) copy_words_wave(blob + L.bel(), s_prior, 20 * K, lane);
        }
        __syncthreads();
#ifdef MGX_STAMPS
        if (w.dbg && lane == 0 && role < 2) {  // per wave: cycles in factor phase, its barrier, variable phase, its barrier
            unsigned long long *d = w.dbg + ((size_t)blockIdx.x * 2 + role) * 8;
            d[0] = c_f; d[1] = c_fb; d[2] = c_v; d[3] = c_vb; d[4] = __builtin_readcyclecounter() - t_loop0;
            d[5] = __builtin_amdgcn_s_memrealtime() - rt0;  // 100 MHz ticks over the same span
            d[6] = t_staged - t_k0;
            d[0] = (d[0] & 0xffffffffull) | ((t_extf - t_staged) << 32);  // external factor sweep (high word)
            d[1] = (d[1] & 0xffffffffull) | ((t_extv - t_extf) << 32);    // external variable sweep (high word)
            if (PERSIST) {
                d[0] = p_wait; d[1] = p_extf; d[2] = p_extv; d[3] = p_int; d[4] = p_pub;
                unsigned long long *d2 = w.dbg + (size_t)((census ? gridDim.x - 1 : gridDim.x) + 4) * 16 + ((size_t)blockIdx.x * 2 + role) * 16;
                for (int i = 0; i < 16; i++) d2[i] = qs[i];
                d2[14] = q_arrive;
            }
        }
#endif
    }
    if (PERSIST) snap_out = (w.cur + total_segs) & 1;  // where the records of the launch's last sweep go (the host follows)
    if (SHARD && role == ROLE_UV && lane == 0) {  // "through with this launch's reads of your records": see the end of a segment
        for (int t = xp0; t < xp1; t++) {
            const XPushRec xr = cold().xp_rec[t];
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(xr.flag), plan.flag_base + xr.flag_delta + (unsigned long long)total_segs,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }

    // ---- write back: straight copies of the LDS images ----------------------------------------------
    for (int t = tid; t < 16 * K; t += NT)  // covariance of the variables that recomputed it
        if (s_covset[t % K]) blob[L.cov() + t] = s_cov[t];
    copy_words(blob + L.mu(), s_mu, 4 * K, tid, NT);
    copy_words(blob + L.valid(), (const double *)s_valid, K, tid, NT);
    if (snap_out >= 0) {
        double *dst = w.snap[snap_out] + (size_t)v0 * SNAP_W;
        for (int t = tid; t < SNAP_W * K; t += NT) dst[t] = s_snap[(t % SNAP_W) * K + (t / SNAP_W)];
        for (int t = tid; t < K; t += NT) w.snap_epoch[snap_out][v0 + t] = s_epoch[t];
    }
    if (is_trk) {
        w.trk_record[trk_item] = trk_rec;
        w.trk_last_pos[trk_item] = trk_lp[0];
        w.trk_last_pos[(size_t)w.NT + trk_item] = trk_lp[1];
        w.trk_last_val[trk_item] = trk_lv;
    }
    if (tid == 0) w.iter_factor[r] = itf;
#ifdef MGX_STAMPS
    if (w.dbg && lane == 0 && role < 2) w.dbg[((size_t)blockIdx.x * 2 + role) * 8 + 7] = __builtin_readcyclecounter() - t_k0;  // whole kernel
#endif
}

}  // namespace mgx
