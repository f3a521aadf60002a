// mgx_sweep_inst.hip — instantiations of k_robot_sweep (mgx_sweep.h) and their launchers.
//
// ONE source, compiled into several objects (the build passes -DMGX_FLAVOR and -DMGX_KSET) so that the sixty
// instantiations compile side by side instead of one after the other:
//   MGX_FLAVOR 0  launch-per-segment kernels  k_robot_sweep<K, IR_NONE | IR_STAGED | IR_GLOBAL, false>
//   MGX_FLAVOR 1  resident schedule launches  k_robot_sweep<K, IR_STAGED, true>
//   MGX_FLAVOR 2  resident schedule launches of a SHARDED world (ghost records arrive inside the launch)
//                                             k_robot_sweep<K, IR_STAGED, true, true>
//   MGX_KSET 0 | 1 | 2  which horizon lengths: constant-K code for the horizons of BASELINE.json and of the reference's
//                       scenarios; 0 / -1 = the run-time-K fallbacks
// Every object exports `..._set<KSET>` functions that answer "not mine" (false / -1) for a horizon variant of another
// set; mgx_kernels.hip asks them in turn.
#include "mgx_sweep.h"

#ifndef MGX_FLAVOR
#error "compile with -DMGX_FLAVOR=0|1|2 -DMGX_KSET=0|1|2"
#endif

#if MGX_KSET == 0
#define MGX_K_LIST(DO) DO(10) DO(11) /* Tracking Factor Showcase */ DO(12) /* Junction Twoway */ DO(13) /* Junction Experiment */
#define MGX_SET_NAME(base) base##_set0
#elif MGX_KSET == 1
#define MGX_K_LIST(DO) DO(16) DO(17) /* Merge, Iteration Amount */ DO(20) /* Schedules Experiment */ DO(21) /* Circle Experiment */
#define MGX_SET_NAME(base) base##_set1
#else
#define MGX_K_LIST(DO) DO(32) DO(35) /* Communications Failure */ DO(0) DO(-1)
#define MGX_SET_NAME(base) base##_set2
#endif

#ifdef MGX_ONLY_K  // diagnostics (an assembly listing of ONE instantiation): -DMGX_ONLY_K=16 with the set that holds it
#undef MGX_K_LIST
#define MGX_K_LIST(DO) DO(MGX_ONLY_K)
#endif

namespace mgx {

size_t sweep_lds_bytes(int K, int ir_edges);
size_t sweep_lds_bytes(int K, int ir_edges, bool resident);
size_t sweep_resident_lds_max();

#if MGX_FLAVOR == 0

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

bool MGX_SET_NAME(sweep_plain)(int kt, const DevWorld &w, int robot0, int n_robots, uint32_t ext_mask, uint32_t int_mask, int n_int,
                               int snap_out, uint32_t hints, hipStream_t stream) {
    switch (kt) {
#define MGX_DO(KT) case KT: launch_k<KT>(w, robot0, n_robots, ext_mask, int_mask, n_int, snap_out, hints, stream); return true;
        MGX_K_LIST(MGX_DO)
#undef MGX_DO
    default: return false;
    }
}

#else  // resident schedule launches (MGX_FLAVOR 1: every robot local; 2: sharded worlds)

constexpr bool SHARDED = MGX_FLAVOR == 2;

// A workgroup may take up to the CU's whole 160 KB of LDS (MI355X_MICROARCH.md); beyond 64 KB the kernel has to be told.
template <int KT>
static bool resident_allow_lds(size_t staged) {
    if (staged <= 64 * 1024) return true;
    static size_t allowed = 0;  // per instantiation
    if (staged <= allowed) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_robot_sweep<KT, IR_STAGED, true, SHARDED>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)staged) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    allowed = staged;
    return true;
}
// How many workgroups of the resident kernel the device holds at once (0: this world's shape has no resident
// form: no inter-robot edges, or too many per robot to stage in LDS).  Every workgroup of such a launch waits for
// its neighbours INSIDE the launch, so all of them have to be resident together.
template <int KT>
static int resident_capacity_k(const DevWorld &w) {
    const size_t staged = sweep_lds_bytes(w.K, w.ir_max_edges, true);
    if (w.ir_max_edges == 0 || staged > sweep_resident_lds_max() || !resident_allow_lds<KT>(staged)) return 0;
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_robot_sweep<KT, IR_STAGED, true, SHARDED>, sweep_threads<KT, true>(), staged) != hipSuccess)
        return 0;
    return per_cu * cus;
}
template <int KT>
static hipError_t resident_launch_k(const DevWorld &w, int n_robots, const SegPlan &plan, bool cooperative, hipStream_t stream) {
    const size_t staged = sweep_lds_bytes(w.K, w.ir_max_edges, true);
    if (!resident_allow_lds<KT>(staged)) return hipErrorInvalidValue;
    const int grid = n_robots + (plan.launch_seq != 0ull ? 1 : 0);  // + the residency census' decider workgroup (mgx_sweep.h)
    if (cooperative) {  // the launch-time check of the grid against the occupancy query; same residency as a plain launch
        DevWorld wa = w;
        SegPlan pa = plan;
        int robot0 = 0, n_int = 0, snap_out = -1;
        uint32_t ext_mask = 0u, int_mask = 0u, hints = 0u;
        void *args[] = {&wa, &robot0, &ext_mask, &int_mask, &n_int, &snap_out, &hints, &pa};
        return hipLaunchCooperativeKernel(reinterpret_cast<const void *>(&k_robot_sweep<KT, IR_STAGED, true, SHARDED>), dim3(grid),
                                          dim3(sweep_threads<KT, true>()), args, (unsigned int)staged, stream);
    }
    hipLaunchKernelGGL((k_robot_sweep<KT, IR_STAGED, true, SHARDED>), dim3(grid), dim3(sweep_threads<KT, true>()), staged, stream, w, 0, 0u, 0u, 0,
                       -1, 0u, plan);
    return hipGetLastError();
}

#if MGX_FLAVOR == 1
#define MGX_RES_NAME(base) MGX_SET_NAME(resident_##base)
#else
#define MGX_RES_NAME(base) MGX_SET_NAME(sharded_##base)
#endif

int MGX_RES_NAME(capacity)(int kt, const DevWorld &w) {  // -1: not this set's horizon
    switch (kt) {
#define MGX_DO(KT) case KT: return resident_capacity_k<KT>(w);
        MGX_K_LIST(MGX_DO)
#undef MGX_DO
    default: return -1;
    }
}
bool MGX_RES_NAME(launch)(int kt, const DevWorld &w, int n_robots, const SegPlan &plan, bool cooperative, hipStream_t stream,
                          hipError_t *err) {
    switch (kt) {
#define MGX_DO(KT) case KT: *err = resident_launch_k<KT>(w, n_robots, plan, cooperative, stream); return true;
        MGX_K_LIST(MGX_DO)
#undef MGX_DO
    default: return false;
    }
}

#endif

}  // namespace mgx
