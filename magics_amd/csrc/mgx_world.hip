// mgx_world.hip — host side of the C ABI (include/mgx.h): world arena, topology -> slot tables,
// device buffers, launch sequencing.  No CPU compute path exists: every compute entry point
// needs a HIP device.
//
// Host keeps a per-robot / per-connection mirror of all mutable state only to (re)build the
// device arrays when the topology changes (`commit`): device is the source of truth between
// commits; `pull` downloads it first so that added robots / inter-robot connections never
// disturb existing state (the reference mutates its graphs in place,
// factorgraph.rs:190-226,304-353,380-436).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mgx.h"
#include "gbp_math.h"
#include "mgx_dev.h"

namespace mgx {
size_t sweep_lds_bytes(int K, int ir_edges);
size_t sweep_lds_bytes(int K, int ir_edges, bool resident);
int blob_words(int K);
bool sweep_supports(int K);
hipError_t launch_robot_sweep(const DevWorld &w, int robot0, int n_robots, uint32_t ext_mask, uint32_t int_mask, int n_int,
                              int snap_out, uint32_t hints, hipStream_t stream);
int sweep_resident_capacity(const DevWorld &w, bool sharded);
size_t sweep_resident_lds_max();
hipError_t launch_robot_schedule(const DevWorld &w, int n_robots, const SegPlan &plan, bool sharded, bool cooperative, hipStream_t stream);
hipError_t launch_agree_abort(const DevWorld &w, const SegPlan &plan, hipStream_t stream);
hipError_t launch_change_prior(const DevWorld &w, int n, const int32_t *robots, const uint32_t *vars, const double *means,
                               hipStream_t stream);
hipError_t launch_update_priors(const DevWorld &w, int n, const int32_t *robots, const double *waypoints, const double *time_scale,
                                const uint8_t *what, double max_speed, double delta_t, hipStream_t stream);
hipError_t launch_halo_pack(const DevWorld &w, int n, const int32_t *robots, double *buf, hipStream_t stream);
hipError_t launch_halo_unpack(const DevWorld &w, int n, const int32_t *ghosts, const double *buf, hipStream_t stream);
hipError_t launch_copy_bytes(uint8_t *dst, const uint8_t *src, size_t n, hipStream_t stream);
hipError_t launch_freeze(const DevWorld &w, uint32_t kinds, hipStream_t stream);
hipError_t launch_ir_freeze(const DevWorld &w, double *frozen_snap, uint32_t *frozen_epoch, hipStream_t stream);
hipError_t launch_thaw_ir(const DevWorld &w, uint8_t *gate, hipStream_t stream);
hipError_t launch_keyless_ir(const DevWorld &w, uint8_t *gate, int n, const KeylessRec *recs, hipStream_t stream);
hipError_t launch_or_bytes(uint8_t *p, int n, uint8_t keep, uint8_t set, hipStream_t stream);
hipError_t launch_thaw(const DevWorld &w, int robot0, int n_robots, uint32_t ext_mask, hipStream_t stream);
hipError_t launch_thaw_done(const DevWorld &w, int robot0, int n_robots, int clear, hipStream_t stream);
hipError_t launch_gather_variable_means(const DevWorld &w, int var, double *out, hipStream_t stream);
hipError_t launch_mission_reached(const DevWorld &w, const DevMission &m, int n, long long tick, unsigned int *ev, hipStream_t stream);
hipError_t launch_mission_positions(const DevMission &m, int n, const int32_t *alive, float *out, hipStream_t stream);
hipError_t launch_mission_prepare(const DevWorld &w, const DevMission &m, int n, const uint8_t *moving, double *rec, int32_t *robots,
                                  double *waypoints, double *time_scale, uint8_t *what, hipStream_t stream);
hipError_t launch_retopo_unpack(const void *src, size_t b_slots, size_t b_ptr, size_t b_mid, size_t b_peers, void *slots, void *in_ptr, void *mid,
                                void *peers, int R, int K, int32_t *var_ptr, int32_t *var_mid, hipStream_t stream);
hipError_t launch_edge_rebuild(const DevWorld &w, int n_slots, const IrSlotRec *slots, const int32_t *in_new, const int32_t *in_old,
                               int stride_new, IrEdgeRec *recs, double *fv_eta, double *fv_lam, double *bmu, uint8_t *gate, hipStream_t stream);
hipError_t launch_var_tables(int R, int K, const int32_t *in_ptr, const int32_t *in_mid, int32_t *var_ptr, int32_t *var_mid,
                             hipStream_t stream);
hipError_t launch_edge_gates(int n, const IrEdgeRec *recs, const uint8_t *antenna, const uint8_t *idle, uint8_t *gate, hipStream_t stream);
hipError_t launch_halo_push(const DevWorld &w, int n, const int32_t *robots, const unsigned long long *dst, int n_peers,
                            const unsigned long long *peer_flags, unsigned long long seq, unsigned int *done, hipStream_t stream, bool always);
hipError_t launch_halo_wait_unpack(const DevWorld &w, int n, const int32_t *ghosts, const double *recv, int n_sources,
                                   const unsigned long long *flags, unsigned long long seq, unsigned long long *err,
                                   long long timeout_ticks, unsigned long long *ready, unsigned long long *host_err, hipStream_t stream,
                                   bool by_slot);
// mgx_topology.hip
int env_red_plane(const mgx_env_desc *d, uint32_t resolution, float expansion, float blur_percent, bool with_blur, hipStream_t s,
                  std::vector<uint8_t> &red, uint32_t &W, uint32_t &H);  // mgx_env.hip
hipError_t neighbours_count(const float *pos, int n, float radius, bool grid, uint32_t M, int32_t *cnt, int32_t *bucket_cnt,
                            int32_t *bucket_ptr, int32_t *cursor, int32_t *members, int32_t *special, int32_t *n_special,
                            int32_t *ptr, hipStream_t s);
hipError_t neighbours_fill(const float *pos, int n, float radius, bool grid, uint32_t M, const int32_t *bucket_ptr,
                           const int32_t *members, const int32_t *special, const int32_t *n_special, const int32_t *ptr,
                           int32_t *idx, int32_t cap, hipStream_t s);
hipError_t neighbours_rows(const float *pos, int n, float radius, int32_t cap, int32_t *cnt, int32_t *rows, hipStream_t s, float *stage);
}  // namespace mgx

using namespace mgx;

static thread_local std::string g_err;
// MGX_TIMING=1: host-side stage times of the topology pass and the table rebuild on stderr (diagnostic)
struct StageTimer {
    bool on;
    double t0;
    const char *what;
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
    explicit StageTimer(const char *w) : what(w) { static const bool e = getenv("MGX_TIMING") != nullptr; on = e; t0 = on ? now() : 0.0; }
    void lap(const char *stage) { if (on) { const double t = now(); fprintf(stderr, "[mgx timing] %s: %s %.1f us\n", what, stage, t - t0); t0 = t; } }
};
static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(MGX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e));    \
    } while (0)

namespace {

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;  // owns its allocation
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void swap(DevBuf &o) {
        std::swap(p, o.p);
        std::swap(n, o.n);
        std::swap(cap, o.cap);
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = cap = 0;
    }
    // (re)allocates only when growing; the copy is enqueued on `s` from pageable memory, so the
    // caller synchronises before `h` dies
    hipError_t upload(const std::vector<T> &h, hipStream_t s) {
        const size_t want = h.size() ? h.size() : 1;
        if (want > cap) {
            release();
            hipError_t e = hipMalloc((void **)&p, sizeof(T) * want);
            if (e != hipSuccess) return e;
            cap = want;
        }
        n = h.size();
        if (n) return hipMemcpyAsync(p, h.data(), sizeof(T) * n, hipMemcpyHostToDevice, s);
        return hipSuccess;
    }
    hipError_t reserve(size_t want) {  // contents undefined afterwards
        if (want < 1) want = 1;
        if (want > cap) {  // grow with headroom: tables that follow a churning topology would otherwise be
            release();     // re-allocated (a device-wide synchronisation) at every new maximum
            const size_t room = want + want / 4 + 64;
            hipError_t e = hipMalloc((void **)&p, sizeof(T) * room);
            if (e != hipSuccess) return e;
            cap = room;
        }
        n = want;
        return hipSuccess;
    }
    hipError_t download(std::vector<T> &h, hipStream_t s) const {
        h.resize(n);
        if (!n) return hipSuccess;
        return hipMemcpyAsync(h.data(), p, sizeof(T) * n, hipMemcpyDeviceToHost, s);
    }
};

// Pinned, device-mapped host staging for the small per-tick argument lists: the kernels read them
// in place over the host link (tens of KB), so a tick enqueues no copy and never synchronises host
// and device.  A ring of slots, each guarded by an event recorded after the kernel that reads it.
struct StageRing {
    static constexpr int SLOTS = 8;
    void *host[SLOTS] = {};
    size_t cap[SLOTS] = {};
    hipEvent_t ev[SLOTS] = {};
    bool pending[SLOTS] = {};
    int next = 0;
    ~StageRing() {
        for (int i = 0; i < SLOTS; i++) {
            if (ev[i]) { (void)hipEventSynchronize(ev[i]); (void)hipEventDestroy(ev[i]); }
            if (host[i]) (void)hipHostFree(host[i]);
        }
    }
    hipError_t acquire(size_t bytes, void **p, int *slot) {
        const int i = next;
        next = (next + 1) % SLOTS;
        hipError_t e;
        if (!ev[i] && (e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)) != hipSuccess) return e;
        if (pending[i]) {
            if ((e = hipEventSynchronize(ev[i])) != hipSuccess) return e;
            pending[i] = false;
        }
        if (bytes > cap[i] || !host[i]) {  // also for a request of zero bytes (a rank that holds ghosts only): callers map the slot
            if (host[i]) (void)hipHostFree(host[i]);
            host[i] = nullptr;
            cap[i] = 0;
            const size_t want = std::max<size_t>(bytes + bytes / 2, 4096);
            if ((e = hipHostMalloc(&host[i], want, hipHostMallocMapped)) != hipSuccess) return e;
            cap[i] = want;
        }
        *p = host[i];
        *slot = i;
        return hipSuccess;
    }
    hipError_t release(int slot, hipStream_t s) {
        pending[slot] = true;
        return hipEventRecord(ev[slot], s);
    }
};

// mutable state of one robot, item-major (AoS) on the host
struct Robot {
    int K = 0;
    bool ghost = false;
    double radius = 1.0;
    uint64_t order_key = 0;
    uint8_t antenna = 1, idle = 0;
    bool removed = false;  // despawned: never iterated, nothing delivered, invisible to the neighbour search
    std::vector<double> prior_eta, prior_lam, bel_eta, bel_lam, bel_mu, bel_cov;  // [K][4|16]
    std::vector<int32_t> valid;                                                   // [K]
    std::vector<double> snap;                                                     // [K][24]
    std::vector<uint32_t> epoch;                                                  // [K]
    std::vector<double> fv_eta, fv_lam;                                           // [E][4|16]
    std::vector<double> dyn_m;                                                    // [K-1][16]
    std::vector<int32_t> trk_record;                                              // [K-2]
    std::vector<float> trk_last_pos;                                              // [K-2][2]
    std::vector<double> trk_last_val;                                             // [K-2]
    std::vector<float> path;                                                      // [n_path][2]
    int32_t iter_factor = 0;
    // petgraph StableGraph node slots of this robot's graph: K variables, K-1 dynamic, K-2 obstacle
    // and K-2 tracking factors first (robot.rs:1179-1334), inter-robot factors after; vacated slots
    // are reused last-freed-first.  Only the ORDER of the indices matters (inbox key order).
    int n_nodes = 0;
    std::vector<int> free_nodes;
    int alloc_node() {
        if (!free_nodes.empty()) { const int ix = free_nodes.back(); free_nodes.pop_back(); return ix; }
        return n_nodes++;
    }
    // MessageCount of the graph's permanent nodes (variables, dynamic / obstacle / tracking factors):
    // sent internal, sent external, received internal, received external (factorgraph/mod.rs:29-137)
    uint64_t cnt[4] = {0, 0, 0, 0};
    int64_t cnt_itf = 0;  // iteration_count.factor as far as the counters have been advanced
    std::vector<uint32_t> slot_uses;  // per node slot: entries of interrobot_factor_indices naming it
    // run-time switching of factor kinds (mgx_set_enabled): the inbox the internal factors froze with, whether each
    // entry is present, and the kinds still to take their first update from it (empty until the world needs them)
    std::vector<double> frozen;
    std::vector<uint8_t> frozen_flag;
    uint8_t thaw = 0;
    std::vector<double> ir_frozen_snap;                    // [K][24] what the variables had sent when inter-robot factors went off
    std::vector<uint32_t> ir_frozen_epoch, ir_thaw_epoch;  // [K]
};

struct IrEdge {  // one InterRobotFactor, kept at its target variable
    double fv_eta[4] = {0, 0, 0, 0}, fv_lam[16] = {0}, bmu[4] = {0, 0, 0, 0};
    uint32_t created = 0;
    bool fresh = true;  // created since the last commit: state is initialised at commit
};
struct IrConn {  // K-1 factors owner -> other
    // (what the per-tick host passes over ALL connections read — counters, table rebuild — sits in the first cache line)
    int owner, other;
    // (its slot in the target's incoming list on the device and "some edge is still fresh" live in mgx_world::conn_hot)
    uint64_t first_number;
    uint64_t cnt[4] = {0, 0, 0, 0};  // MessageCount summed over the K-1 factors
    // Sum over the factors of how often each one's node slot occurs in the owner's
    // interrobot_factor_indices: that list is never pruned (factorgraph.rs:729-733), so a factor in a
    // re-used slot is updated once per occurrence in every external sweep — same message, but every
    // update counts as sent / received.
    uint64_t updates_per_sweep = 0;
    int node_first = 0, node_last = 0;  // node[0], node.back(): what orders two connections of one owner in an inbox
    // Everything `cnt` counts is a function of what its two robots have run since the counters were last brought up to date
    // (internal / external variable sweeps, external factor sweeps, prior changes of variables that carry inter-robot factors:
    // mgx_world::cum) under flags that do not change in between — so a connection is SETTLED (settle_conn) only when somebody
    // needs its numbers: a read, a switch of flags or kinds, its deletion; the per-tick topology pass no longer walks every
    // connection for it.  base: the robots' cumulative counts when the connection was settled last.
    uint64_t base[5] = {0, 0, 0, 0, 0};  // owner's nIv, target's nEv, owner's nEf, owner's / target's prior changes
    std::vector<IrEdge> edges;  // index i-1 for variable i
    std::vector<int> node;      // node slot of each factor in the owner's graph
    // Factors created while their kind is switched off drop the two messages that would have filled their inbox
    // (factor/mod.rs:307-310), and FactorNode::update answers inbox KEYS: once enabled, such a factor sends nothing to
    // a variable that has not delivered to it yet.  The messages themselves are handled on the device (delivery
    // counts); this is the same knowledge for the counters: per factor, bit 0 = the own variable's key is there,
    // bit 1 = the foreign variable's; `uses` = the factor's share of updates_per_sweep.  Empty: every key is there.
    std::vector<uint8_t> keys;
    std::vector<uint32_t> uses;
};

// RobotConnections::robots_connected_with of every robot (robot.rs:515-531), ascending order key — ONE contiguous pool, rows of a
// fixed capacity: the topology pass walks every robot's set every tick, and a thousand separately allocated vectors are a
// thousand cache misses.  `keys`: the robots' order keys, compact, for the merges of that pass.
struct ConnSets {
    int cap = 16;
    std::vector<int32_t> ids, cnt;
    std::vector<uint64_t> keys;
    std::vector<uint8_t> ghost;   // the robots' ghost flags and radii, compact like the keys (fixed when a robot is added): the
    std::vector<double> radius;   // per-tick passes over all connections read them instead of the robots themselves
    std::vector<uint8_t> removed; // ... and Robot::removed (mgx_robot_remove)
    void ensure(size_t n) {
        if (cnt.size() < n) { cnt.resize(n, 0); ids.resize(n * (size_t)cap, 0); }
    }
    int32_t *row(size_t r) { return ids.data() + r * (size_t)cap; }
    const int32_t *row(size_t r) const { return ids.data() + r * (size_t)cap; }
    void grow() {
        const int nc = cap * 2;
        std::vector<int32_t> ni(cnt.size() * (size_t)nc, 0);
        for (size_t r = 0; r < cnt.size(); r++) std::copy(row(r), row(r) + cnt[r], ni.begin() + (long)(r * (size_t)nc));
        ids.swap(ni);
        cap = nc;
    }
    bool has(size_t r, int id) const { return std::find(row(r), row(r) + cnt[r], id) != row(r) + cnt[r]; }
    void insert_sorted(size_t r, int id) {  // keeps the row ascending in order key
        if (cnt[r] == cap) grow();
        int32_t *b = row(r), *e = b + cnt[r];
        int32_t *at = std::upper_bound(b, e, id, [&](int x, int y) { return keys[(size_t)x] < keys[(size_t)y]; });
        std::copy_backward(at, e, e + 1);
        *at = id;
        cnt[r]++;
    }
    void erase(size_t r, int id) {
        int32_t *b = row(r), *e = b + cnt[r];
        cnt[r] = (int32_t)(std::remove(b, e, id) - b);
    }
};

// RCCL, resolved at run time (no link-time dependency): the copy already in the process (a host that
// runs torch.distributed has one) or the system library.  Only what the halo exchange needs.
struct RcclApi {
    typedef int (*get_unique_id_t)(void *);
    struct Id128 { char internal[128]; };
    typedef int (*comm_destroy_t)(void *);
    typedef int (*group_t)(void);
    typedef int (*sendrecv_t)(void *, size_t, int, int, void *, hipStream_t);
    typedef const char *(*error_string_t)(int);
    get_unique_id_t get_unique_id = nullptr;
    int (*comm_init_rank)(void **, int, Id128, int) = nullptr;
    comm_destroy_t comm_destroy = nullptr;
    group_t group_start = nullptr, group_end = nullptr;
    sendrecv_t send = nullptr, recv = nullptr;
    error_string_t error_string = nullptr;
    bool tried = false, ok = false;
    bool load() {
        if (tried) return ok;
        tried = true;
        void *h = dlopen(nullptr, RTLD_NOW);  // symbols already in the process
        if (!h || !dlsym(h, "ncclCommInitRank")) {
            h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        }
        if (!h) return false;
        get_unique_id = (get_unique_id_t)dlsym(h, "ncclGetUniqueId");
        comm_init_rank = (int (*)(void **, int, Id128, int))dlsym(h, "ncclCommInitRank");
        comm_destroy = (comm_destroy_t)dlsym(h, "ncclCommDestroy");
        group_start = (group_t)dlsym(h, "ncclGroupStart");
        group_end = (group_t)dlsym(h, "ncclGroupEnd");
        send = (sendrecv_t)dlsym(h, "ncclSend");
        recv = (sendrecv_t)dlsym(h, "ncclRecv");
        error_string = (error_string_t)dlsym(h, "ncclGetErrorString");
        ok = get_unique_id && comm_init_rank && comm_destroy && group_start && group_end && send && recv;
        return ok;
    }
};
static RcclApi g_rccl;
constexpr int NCCL_FLOAT64 = 8;  // ncclFloat64 (rccl.h)

}  // namespace

// Incoming inter-robot connections of every local robot in inbox key order (graph key, node index
// — message.rs / id.rs:19-117), and the split between lower-key and higher-key owners.  Each
// connection hangs one factor on every variable 1..K-1 of its target, and the order is the same for
// all of them: by owner key, and for two connections of one owner by node slot — a connection's K-1
// slots are one block of consecutive indices (fresh, or a whole vacated block: alloc_node), so
// comparing the first slots orders the whole blocks.  Edge (variable i, list position q) of robot r
// lives at  (K-1) * in_ptr[r] + (i-1) * n_in(r) + q.
struct Incoming {
    std::vector<int32_t> in_ptr, in_list, mid;
    // on request: the resident kernel's peer table (ensure_resident_tables) from the same two passes over the connections —
    // [R + 1 row pointers | entries]: for every local robot the owners of its incoming and the targets of its outgoing connections
    std::vector<int32_t> peers;
    std::vector<int32_t> fill, pfill;  // scratch of build_incoming
    int ir_max_edges = 0;
    bool blocks_ok = true;
};
struct Launch { uint32_t ext; int n_int; uint32_t hints; };  // one [external iteration] internal* segment of a schedule

// Who points at whom, kept IN STEP with the connection list (ir_connect, ir_disconnect_batch) instead of being derived from it in
// two passes over every connection whenever a topology pass has changed something: per robot id the connections it is the TARGET
// of — in the order of its variables' inboxes (owner's order key, then node slot: build_incoming) — and the ones it OWNS (no
// order).  Entries are indices into the connection list, which closes its holes by moving the last survivors into them: a move
// rewrites the mover's two entries.  Anything the index is not told about (ir_disconnect, a robot that changes sides) just
// invalidates it: the next use builds it again from the list.
struct ConnIndex {
    bool valid = false;
    bool interleaved = false;  // node slots of two connections of one owner towards one target interleave (never: reported)
    std::vector<std::vector<int32_t>> in, out;
};

struct mgx_world {
    mgx_params p{};
    std::vector<Robot> robots;  // ids = indices; ghosts may interleave on the host, device order below
    std::vector<IrConn> conns;
    ConnSets sets;  // robots_connected_with of every robot
    std::vector<uint8_t> sdf_red;
    uint32_t sdf_w = 0, sdf_h = 0;
    double world_w = 1.0, world_h = 1.0;
    int K = 0;

    hipStream_t stream = nullptr;
    bool dirty = true;       // robots / image changed since the device arrays were built: full rebuild
    bool conns_dirty = false;  // only inter-robot connections changed: edge tables are rebuilt in place
    bool flags_dirty = true;
    bool dev_valid = false;  // device arrays hold live state
    bool frozen_live = false;     // the frozen-inbox arrays exist (a kind has been switched at run time)
    uint32_t thaw_kinds = 0;      // kinds some robot may still be thawing: k_thaw runs before sweeps with a factor phase
    DevBuf<double> frozen_buf;
    DevBuf<uint8_t> frozen_flag_buf, thaw_buf, skip0_buf;
    bool ir_frozen_live = false;  // ir_frozen_* hold what the variables had sent when inter-robot factors were switched off
    bool ir_thaw_active = false;  // inter-robot factors are back and some owner may not have delivered since
    DevBuf<double> ir_frozen_snap_buf;
    DevBuf<uint32_t> ir_frozen_epoch_buf, ir_thaw_epoch_buf;
    bool trk_ever_on = false;  // tracking factors were enabled at some point: their message columns may be non-zero
    uint32_t stale_kinds = 0;  // disabled factor kinds whose inboxes have missed a delivery (mgx_set_enabled)
    DevWorld d{};
    std::vector<int> dev_of;     // robot id -> device robot index (locals first, then ghosts)
    std::vector<int> robot_of;   // device robot index -> robot id

    DevBuf<double> blob, snap0, snap1, dyn_m, trk_last_val, ir_fv_eta, ir_fv_lam, ir_bmu;
    DevBuf<IrEdgeRec> ir_rec;
    DevBuf<double> ir_fv_eta_b, ir_fv_lam_b, ir_bmu_b;  // second set: the edge tables are rebuilt out of place
    DevBuf<IrEdgeRec> ir_rec_b;
    DevBuf<int32_t> in_ptr_dev, in_ptr_dev_b, in_mid_dev;  // per-robot slot ranges (current / being built), split index
    DevBuf<IrSlotRec> slot_recs;
    std::vector<int32_t> dev_in_ptr;  // [R_local + 1] incoming-slot ranges of the tables now on the device
    DevBuf<int32_t> trk_record, path_ptr, iter_factor, ir_var_ptr, ir_var_mid;
    DevBuf<uint32_t> epoch0, epoch1;
    DevBuf<float> trk_last_pos, path_xy;
    DevBuf<uint8_t> ir_gate, antenna, idle, sdf;
    StageRing stage;  // packed per-tick arguments
    // resident schedule launches (SegPlan, mgx_dev.h): progress words, peer lists, the abort / error words
    DevBuf<unsigned long long> sweep_flag_buf, sweep_abort_buf;
    DevBuf<unsigned char> xrec_buf;  // exchange records of the local robots' variables, two parities (mgx_dev.h)
    DevBuf<int32_t> peer_ptr_dev;  // [R + 1 row pointers | entries]
    size_t peer_idx_off = 0;
    std::vector<int32_t> peer_fill;
    unsigned long long *sweep_err_host = nullptr;  // host-mapped; non-zero once a wait inside a resident launch gave up
    unsigned long long flag_base = 0;              // every progress word is below or at this value between launches
    // mgx_batch_begin .. mgx_batch_end: the schedules mgx_iterate was handed since the last submission, one after the other (what
    // iterate(a); iterate(b) computes is what iterate(a ++ b) computes), the launches they were submitted as and how many of them
    struct Batch {
        bool open = false;
        std::vector<uint8_t> steps;
        uint32_t schedules = 0, submissions = 0, launches = 0;
    } batch;
    bool resident_off = false;                     // mgx_set_resident_launches(w, 0)
    bool resident_decline = false;                 // mgx_set_resident_launches(w, 2)
    // residency census of resident launches (SegPlan, mgx_dev.h): cumulative per-group counts the device counters reach, the
    // launch number, and the launch the host has enqueued but not yet seen decided (go / abort)
    DevBuf<unsigned long long> census_buf, decision_buf;
    unsigned long long *decision_host = nullptr;   // host-mapped
    unsigned long long launch_seq = 0;
    struct PendingResident {
        bool active = false;
        unsigned long long seq = 0;
        std::vector<std::pair<uint32_t, int>> segs;  // (external phases, internal iterations) of the launch's segments
        std::vector<uint32_t> hints;
        int cur_before = 0;
        unsigned long long flag_base_before = 0;
        bool partial = false;  // the launch is not the first of its schedule (more than MAX_SEGS segments)
        const double *upd = nullptr;  // mgx_tick: the prior updates that ride in the launch
        int upd_slot = -1;            // ... and the pinned ring slot they sit in (-1: device memory of the caller's, mgx_mission_tick):
                                      // a re-run guards it again — the event behind the declined launch completed at once
        double upd_max_speed = 0.0, upd_delta_t = 0.0;
    } pending;
    int upd_ring_slot = -1;  // mgx_tick -> run_resident: the ring slot d.upd points into
    const double *upd_host = nullptr;  // ... and the host's view of the same records (null: they live in device memory)
    // LINGERING resident launches (mgx_dev.h): the host's side of the box.  `open`: a launch that lingers is in flight — every
    // entry point but mgx_iterate / mgx_tick (and the pure queries) ends it first (MGX_ENTER, commit); those two POST their schedule
    // into it when it qualifies (run_resident).  At most one post is outstanding without the launch's word for it (`un`): what
    // is needed to take it back and run it as a launch of its own if the launch ended first.
    struct Linger {
        long long ticks = -1;  // wall-clock ticks (100 MHz) a robot's workgroup waits for the next post; -1: not asked yet, 0: off
        LingerBox *box = nullptr;
        size_t upd_stride = 0;  // f64 words per slot of prior-update records behind the box
        DevBuf<unsigned long long> go;
        DevBuf<unsigned char> dev;  // the launch's device-side slots (the postman's copies of the posts): [2][dev_stride]
        size_t dev_stride = 0;
        bool open = false, hold = false;
        unsigned long long seq0 = 0;       // number of the open launch's own plan
        uint32_t taken_in_launch = 0;      // posts the open launch has taken
        int useless = 0;                   // lingering launches in a row that ended without having taken a post
        uint32_t streak = 0;               // schedules issued back to back, this one included (no other call on the world in between)
        struct Post {
            bool active = false;
            unsigned long long number = 0;
            std::vector<Launch> plan;
            bool has_upd = false;
            double max_speed = 0.0, delta_t = 0.0;
            int cur_before = 0;
            unsigned long long flag_base_before = 0;
        } un;
        uint64_t launches = 0, posts = 0, reruns = 0, ended_by_device = 0;
    } linger;
    int sticky_rc = 0;       // a declined launch whose re-run failed inside a call that cannot report it (flush_counts): every
                             // later sweep, read-back and mgx_synchronize reports it (check_device_error)
    // after a declined launch the schedules skip the resident form for a while: counted in world-wide external iterations that
    // ran launch by launch (whoever drives them: the engine's own schedules or a host's mgx_sweep calls — on a sharded world
    // every rank runs the same ones, so every rank comes back to the resident form with the same schedule)
    int resident_backoff = 0;
    int resident_backoff_len = 0;
    uint64_t resident_aborts = 0, resident_launches = 0;
    int resident_cap = -1;                         // workgroups of the resident kernel the device holds at once (-1: not asked yet)
    int resident_cap_sharded = -1;                 // the same for the instantiation that takes ghost records in-launch
    bool peers_valid = false;
    // what the per-tick table rebuild reads of EVERY connection, 16 bytes apiece beside the connections themselves (168 bytes and
    // four vectors each): kept in step wherever the list changes (ir_connect, ir_disconnect, ir_disconnect_batch)
    struct ConnHot {
        int32_t owner, other, node_first, node_last;
        uint64_t first_number;
        int32_t dev_slot;   // slot of this connection in its target's incoming list on the device (-1: not there) — kept HERE only
        uint8_t has_fresh;  // some edge still carries `fresh` (created since the device tables were last laid out) — kept HERE only
    };
    std::vector<ConnHot> conn_hot;
    ConnIndex cidx;
    // per robot, cumulative since the world began: internal variable sweeps run, external variable sweeps, external factor sweeps,
    // prior changes of variables that carry inter-robot factors — what the connections' counters are settled against (IrConn::base)
    struct Cum { std::vector<uint64_t> nIv, nEv, nEf, on_ir; } cum;
    bool conns_unsettled = false;  // some flush since the last full one left the connections' counters behind (lazy)
    Incoming retopo_tables;               // retopo's host tables and slot records (storage kept from tick to tick)
    std::vector<IrSlotRec> retopo_slots;
    // missions on the device (mgx_mission_*): host copies of what mgx_mission_set gave, the device arrays, the
    // host-mapped event list of robots that reached their last waypoint, and the tick counter
    struct Mission {
        bool any = false, dirty = false, uploaded = false;
        std::vector<std::vector<double>> wp;  // per robot: [n][2]
        std::vector<int32_t> target;
        std::vector<uint32_t> vars;           // [R][2]
        std::vector<float> dist2, translation;  // [R][2], [R][3]
        std::vector<double> time_scale;
        std::vector<uint8_t> has;
        std::vector<long long> finished_tick;
        DevBuf<int32_t> wp_ptr_d, target_d, alive_d, robots_d;
        DevBuf<double> wp_xy_d, time_scale_d, rec_d, waypoints_d, ts_list_d;
        DevBuf<uint32_t> vars_d;
        DevBuf<float> dist2_d, translation_d;
        DevBuf<uint8_t> has_d, moving_d, what_d;
        DevBuf<long long> finished_d;
        unsigned int *ev_host = nullptr;  // mapped: [0] count, [1 ..] robot ids
        size_t ev_cap = 0;
        std::vector<int32_t> alive_host;  // the robots the search of the coming tick looks at
        bool alive_dirty = true;
        long long tick_no = 0;
        bool in_tick = false;                // between mgx_mission_tick_begin and _end
        std::vector<int32_t> last_finished;  // robots whose mission completed in the last begin, ascending
        float search_radius = 0.f;           // what the last tick's topology pass searched with: the coming tick's search is enqueued
        uint32_t search_method = 0;          //   with the same (mgx_mission_tick_end), used if the next begin asks for the same
        bool search_known = false;
        float *tr_host = nullptr;            // pinned: Transforms after the last tick's move (valid after the next synchronisation)
        size_t tr_cap = 0, tr_n = 0;
        DevMission d{};
    } mission;
    // a neighbour search that has been enqueued and not collected yet (neighbours_enqueue / neighbours_collect)
    struct PendingSearch {
        hipStream_t stream = nullptr;  // where it was enqueued
        bool valid = false, compact = false, grid = false;
        bool rows = false;  // the one-pass kernel with rows of a fixed capacity (small worlds, AUTO)
        bool from_missions = false;
        int row_cap = 0;
        int n = 0, n_all = 0;
        std::vector<int> alive;
        size_t guess = 0, off_ptr = 0, off_idx = 0;
        float radius = 0.f;
        uint32_t method = 0, M = 0;
    } mission_search;  // the coming tick's search, enqueued by mgx_mission_tick_end
    uint32_t last_sweep_launches = 0;  // sweep-kernel launches of the last mgx_iterate / mgx_tick call (mgx_last_launch_count)
    // message counters are advanced lazily: launches and prior changes are only logged here
    struct CountEntry { uint8_t ext, in; int n_int, robot; uint64_t times; };
    std::vector<CountEntry> clog;
    int n_keyless = 0;  // connections whose factors still lack inbox keys (IrConn::keys)
    std::vector<uint32_t> cp_pending;  // [robot * K + variable] change_prior calls not yet counted
    std::vector<uint32_t> cp_dirty;
    DevBuf<unsigned long long> dbg;  // diagnostic builds only
    // halo plan: local robots whose snapshots are sent / ghost robots that receive, in buffer order
    std::vector<int32_t> halo_send, halo_recv;
    DevBuf<int32_t> halo_send_dev, halo_recv_dev;
    bool halo_dirty = false;
    // direct halo exchange (peer-mapped stores): this rank's receive area and arrival counters are
    // fine-grained device memory that the producers write; `dst` / `peer_flags` are addresses inside
    // the consumers' areas
    struct DirectHalo {
        double *recv = nullptr;               // [2][recv_words]
        unsigned long long *flags = nullptr;  // [n_sources] arrival counters, then one error word
        size_t recv_words = 0;
        int n_sources = 0, n_peers = 0;
        bool connected = false;
        // a wiring that survives changes of the exchange lists (mgx_halo_direct_setup_slots): one record slot per ghost robot —
        // slot = the robot's place among this rank's ghosts — instead of one per entry of the receive list
        bool by_slot = false;
        size_t slot_cap = 0;
        // ... whose push destinations (dst[], one per entry of the send list, in the consumers' slot numbering) are only as good as
        // the lists and the device layout they were made for: any change of either (mgx_halo_plan*, a robot added or released)
        // takes the aim away until mgx_halo_direct_connect_slots has run again — an exchange in between is refused, not run
        // against tables of another length
        bool aimed = false;
        unsigned long long seq = 0, push_seq = 0;  // exchanges waited for / pushed
        long long timeout_ticks = 500000000ll;  // 5 s of the 100 MHz wall clock
        DevBuf<unsigned long long> dst[2], peer_flags, ready;  // ready: the exchange workgroup 0 of the wait kernel has announced
        DevBuf<unsigned int> done;
    } direct;
    // resident schedule launches of a sharded world (mgx_halo_resident_*): this rank's ghost area (fine-grained; the ghosts'
    // owner ranks store into it from inside their launches) and where the records of this rank's boundary robots go
    struct ResidentHalo {
        void *area = nullptr;
        size_t bytes = 0;
        int n_ghosts = 0;
        bool connected = false;
        bool wired = false;  // connect has run and disconnect has not: the peers may hold `area` mapped and store into it
        // a wiring that outlives the exchange lists (mgx_halo_resident_connect_peers / _aim): what translates this rank's parity and
        // segment count into each peer's, settled once when the ranks connect
        struct Peer { unsigned long long base = 0; size_t n_slots = 0; unsigned x = 0; unsigned long long flag_delta = 0; };
        std::vector<Peer> peers;
        DevBuf<int32_t> xp_ptr;
        DevBuf<XPushRec> xp_rec;
        // the ranks' agreement on every schedule's launches (SegPlan::agree_seq): the word (in rank 0's area), the number of
        // ranks that sign in on it, and the number of the last schedule this rank took there — the same on every rank
        unsigned long long *agree = nullptr;
        int n_ranks = 0;
        unsigned long long agree_seq = 0;
    } xres;
    // halo exchange through RCCL inside the library (grouped ncclSend / ncclRecv on the world's stream)
    struct RcclHalo {
        void *comm = nullptr;
        bool connected = false;
        std::vector<int> peer_rank;
        std::vector<uint32_t> send_first, recv_first;  // [n_peers + 1] into halo_send / halo_recv
        DevBuf<double> send_buf, recv_buf;
    } rccl;
    // neighbour search scratch (mgx_topology.hip)
    DevBuf<float> nb_pos;
    DevBuf<int32_t> nb_cnt, nb_bucket_cnt, nb_bucket_ptr, nb_cursor, nb_members, nb_special, nb_nspecial, nb_ptr, nb_idx;
    size_t nb_last_total = 0;  // rows of the last search: sizes the speculative second pass of the next one
    int nb_row_cap = 16;       // one-pass searches: capacity of a row (grown to what the largest row needed)
    hipStream_t search_stream = nullptr, nb_last_stream = nullptr;  // searches over host-supplied positions run beside the world's stream
    bool nb_last_stream_set = false;
    // pinned host memory the search's positions go up from and its rows come back into: copies to and from pageable memory
    // (std::vector) are staged by the runtime, tens of microseconds each
    struct PinBuf {
        void *p = nullptr;
        size_t cap = 0;
        ~PinBuf() { if (p) (void)hipHostFree(p); }
        hipError_t reserve(size_t bytes) {
            if (bytes <= cap) return hipSuccess;
            if (p) (void)hipHostFree(p);
            p = nullptr;
            cap = 0;
            const size_t want = bytes + bytes / 2 + 4096;
            const hipError_t e = hipHostMalloc(&p, want, hipHostMallocMapped);  // the one-pass search reads and writes it in place
            if (e == hipSuccess) cap = want;
            return e;
        }
    } nb_pin;
};

static size_t edge_index(const std::vector<int32_t> &in_ptr, int K, int r, int j, int slot);
static void flush_counts(mgx_world *w, bool lazy = false);
static void conn_index_ensure(mgx_world *w);

// mgx_batch_begin .. mgx_batch_end (below, in front of mgx_iterate): every other call on the world first submits the schedules
// recorded so far, so that it finds the world as if each one had run when it was issued
static int iterate_now(mgx_world *w, const uint8_t *steps, uint32_t n);
static int submit_batch(mgx_world *w);
static int linger_close(mgx_world *w);
// MGX_ENTER_SCHEDULE: mgx_tick (mgx_iterate has the batch's own logic) and the pure queries — recorded schedules are submitted,
// a lingering launch stays open.  MGX_ENTER: everything else — it also ends a lingering launch (what the call does would sit
// behind it in the stream, or read what it has not written back) and breaks the streak of back-to-back schedules.
#define MGX_ENTER_SCHEDULE(w)                                           \
    do {                                                                \
        if ((w) && !(w)->batch.steps.empty()) {                         \
            const int rc_enter_ = submit_batch(w);                      \
            if (rc_enter_ != MGX_OK) return rc_enter_;                  \
        }                                                               \
    } while (0)
#define MGX_ENTER(w)                                                    \
    do {                                                                \
        MGX_ENTER_SCHEDULE(w);                                          \
        if (w) {                                                        \
            (w)->linger.streak = 0;                                     \
            if ((w)->linger.open) {                                     \
                const int rc_enter_ = linger_close(w);                  \
                if (rc_enter_ != MGX_OK) return rc_enter_;              \
            }                                                           \
        }                                                               \
    } while (0)

static bool device_ok() {
    static int state = 0;  // 0 unknown, 1 ok, -1 none
    if (state == 0) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        state = (e == hipSuccess && n > 0) ? 1 : -1;
    }
    return state == 1;
}

// ---- SoA helpers --------------------------------------------------------------------------------
template <class T>
static void scatter(std::vector<T> &dst, size_t stride, size_t item, const T *src, int comps) {
    for (int c = 0; c < comps; c++) dst[(size_t)c * stride + item] = src[c];
}
template <class T>
static void gather(const std::vector<T> &src, size_t stride, size_t item, T *dst, int comps) {
    for (int c = 0; c < comps; c++) dst[c] = src[(size_t)c * stride + item];
}

// J^T Q J of a dynamic factor in the reference's evaluation order (dynamic.rs:22-52,
// factor/mod.rs:391-394), compacted to the 4x4 M with lam_p = M (x) I2.
static void dynamic_potential(double dt, double sigma, double *M /*16*/) {
    const double qc = 1.0 / (sigma * sigma);
    const double q11 = 12.0 * (1.0 / (dt * dt * dt)) * qc, q12 = -6.0 * (1.0 / (dt * dt)) * qc, q22 = (4.0 / dt) * qc;
    double Q[16] = {0}, J[32] = {0};
    for (int a = 0; a < 2; a++) {
        Q[a * 4 + a] = q11;
        Q[a * 4 + a + 2] = q12;
        Q[(a + 2) * 4 + a] = q12;
        Q[(a + 2) * 4 + a + 2] = q22;
        J[a * 8 + a] = 1.0;
        J[a * 8 + a + 2] = dt;
        J[a * 8 + a + 4] = -1.0;
        J[(a + 2) * 8 + a + 2] = 1.0;
        J[(a + 2) * 8 + a + 6] = -1.0;
    }
    double JtQ[32], L[64];
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += J[k * 8 + i] * Q[k * 4 + j];
            JtQ[i * 4 + j] = s;
        }
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += JtQ[i * 4 + k] * J[k * 8 + j];
            L[i * 8 + j] = s;
        }
    for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++) M[a * 4 + b] = L[(2 * a) * 8 + (2 * b)];
}

// ---- blob <-> host mirror -----------------------------------------------------------------------------
// rows of a robot's blob are component-major ([c][K] / [c][E+1]); the host mirror is item-major
static void blob_pack(const Robot &rb, double *b) {
    const int K = rb.K, E = 4 * K - 6, E1 = E + 1;
    const BlobLayout L(K);
    for (int i = 0; i < K; i++) {
        for (int c = 0; c < 4; c++) {
            b[L.prior() + c * K + i] = rb.prior_eta[4 * i + c];
            b[L.bel() + c * K + i] = rb.bel_eta[4 * i + c];
            b[L.mu() + c * K + i] = rb.bel_mu[4 * i + c];
        }
        for (int c = 0; c < 16; c++) {
            b[L.prior() + (4 + c) * K + i] = rb.prior_lam[16 * i + c];
            b[L.bel() + (4 + c) * K + i] = rb.bel_lam[16 * i + c];
            b[L.cov() + c * K + i] = rb.bel_cov[16 * i + c];
        }
    }
    for (int c = 0; c < 20; c++) b[L.fv() + c * E1 + E] = 0.0;  // the all-zero column
    if (!rb.ghost)
        for (int e = 0; e < E; e++) {
            for (int c = 0; c < 4; c++) b[L.fv() + c * E1 + e] = rb.fv_eta[4 * e + c];
            for (int c = 0; c < 16; c++) b[L.fv() + (4 + c) * E1 + e] = rb.fv_lam[16 * e + c];
        }
    int32_t *valid = reinterpret_cast<int32_t *>(b + L.valid());
    for (int i = 0; i < K; i++) valid[i] = rb.valid[i];
}
static void blob_unpack(Robot &rb, const double *b) {
    const int K = rb.K, E = 4 * K - 6, E1 = E + 1;
    const BlobLayout L(K);
    for (int i = 0; i < K; i++) {
        for (int c = 0; c < 4; c++) {
            rb.prior_eta[4 * i + c] = b[L.prior() + c * K + i];
            rb.bel_eta[4 * i + c] = b[L.bel() + c * K + i];
            rb.bel_mu[4 * i + c] = b[L.mu() + c * K + i];
        }
        for (int c = 0; c < 16; c++) {
            rb.prior_lam[16 * i + c] = b[L.prior() + (4 + c) * K + i];
            rb.bel_lam[16 * i + c] = b[L.bel() + (4 + c) * K + i];
            rb.bel_cov[16 * i + c] = b[L.cov() + c * K + i];
        }
    }
    if (!rb.ghost)
        for (int e = 0; e < E; e++) {
            for (int c = 0; c < 4; c++) rb.fv_eta[4 * e + c] = b[L.fv() + c * E1 + e];
            for (int c = 0; c < 16; c++) rb.fv_lam[16 * e + c] = b[L.fv() + (4 + c) * E1 + e];
        }
    const int32_t *valid = reinterpret_cast<const int32_t *>(b + L.valid());
    for (int i = 0; i < K; i++) rb.valid[i] = valid[i];
}

// ---- pull: device -> host mirror ----------------------------------------------------------------
static int confirm_resident(mgx_world *w, bool rerun = true, int32_t *outcome = nullptr);
static int pull(mgx_world *w) {
    if (!w->dev_valid) return MGX_OK;
    if (w->pending.active) { const int rcc = confirm_resident(w); if (rcc != MGX_OK) return rcc; }
    const int K = w->K;
    const size_t NT = (size_t)w->d.NT, NI = (size_t)w->d.NI, BS = (size_t)w->d.BS;
    std::vector<double> bl, sn, tlv, ife, ifl, ibm;
    std::vector<int32_t> trc, itf;
    std::vector<uint32_t> ep;
    std::vector<float> tlp;
    hipStream_t s = w->stream;
    HIP_TRY(w->blob.download(bl, s));
    HIP_TRY((w->d.cur ? w->snap1 : w->snap0).download(sn, s));
    HIP_TRY((w->d.cur ? w->epoch1 : w->epoch0).download(ep, s));
    HIP_TRY(w->trk_record.download(trc, s));
    HIP_TRY(w->trk_last_pos.download(tlp, s));
    HIP_TRY(w->trk_last_val.download(tlv, s));
    HIP_TRY(w->iter_factor.download(itf, s));
    HIP_TRY(w->ir_fv_eta.download(ife, s));
    HIP_TRY(w->ir_fv_lam.download(ifl, s));
    HIP_TRY(w->ir_bmu.download(ibm, s));
    std::vector<IrEdgeRec> irc;
    HIP_TRY(w->ir_rec.download(irc, s));
    std::vector<double> fzd;
    std::vector<uint8_t> fzf, thw;
    if (w->frozen_live) {
        HIP_TRY(w->frozen_buf.download(fzd, s));
        HIP_TRY(w->frozen_flag_buf.download(fzf, s));
        HIP_TRY(w->thaw_buf.download(thw, s));
    }
    std::vector<double> ifs;
    std::vector<uint32_t> ife_, ite;
    if (w->ir_frozen_live) {
        HIP_TRY(w->ir_frozen_snap_buf.download(ifs, s));
        HIP_TRY(w->ir_frozen_epoch_buf.download(ife_, s));
        if (w->ir_thaw_active) HIP_TRY(w->ir_thaw_epoch_buf.download(ite, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    for (size_t dr = 0; dr < w->robot_of.size(); dr++) {
        Robot &rb = w->robots[(size_t)w->robot_of[dr]];
        blob_unpack(rb, &bl[dr * BS]);
        if (w->ir_frozen_live) {
            rb.ir_frozen_snap.assign(ifs.begin() + (long)(dr * K * 24), ifs.begin() + (long)((dr + 1) * K * 24));
            rb.ir_frozen_epoch.assign(ife_.begin() + (long)(dr * K), ife_.begin() + (long)((dr + 1) * K));
            if (w->ir_thaw_active) rb.ir_thaw_epoch.assign(ite.begin() + (long)(dr * K), ite.begin() + (long)((dr + 1) * K));
        }
        for (int i = 0; i < K; i++) {
            const size_t v = dr * K + i;
            memcpy(&rb.snap[24 * i], &sn[v * 24], 24 * sizeof(double));
            rb.epoch[i] = ep[v];
        }
        if (rb.ghost) continue;
        for (int j = 0; j < K - 2; j++) {
            const size_t t = dr * (K - 2) + j;
            rb.trk_record[j] = trc[t];
            rb.trk_last_pos[2 * j] = tlp[t];
            rb.trk_last_pos[2 * j + 1] = tlp[NT + t];
            rb.trk_last_val[j] = tlv[t];
        }
        rb.iter_factor = itf[dr];
        if (w->frozen_live) {
            const size_t FZ = (size_t)frozen_words(K), E = (size_t)(4 * K - 6);
            rb.frozen.assign(fzd.begin() + (long)(dr * FZ), fzd.begin() + (long)((dr + 1) * FZ));
            rb.frozen_flag.assign(fzf.begin() + (long)(dr * E), fzf.begin() + (long)((dr + 1) * E));
            rb.thaw = thw[dr];
        }
    }
    for (size_t ci = 0; ci < w->conns.size(); ci++) {
        IrConn &c = w->conns[ci];
        const int32_t dev_slot = w->conn_hot[ci].dev_slot;
        for (size_t j = 0; j < c.edges.size(); j++) {
            if (dev_slot < 0) continue;  // created since the tables were built: nothing on the device yet
            const size_t e = edge_index(w->dev_in_ptr, w->K, w->dev_of[(size_t)c.other], (int)j, dev_slot);
            IrEdge &ed = c.edges[j];
            gather(ife, NI, e, ed.fv_eta, 4);
            gather(ifl, NI, e, ed.fv_lam, 16);
            gather(ibm, NI, e, ed.bmu, 4);
            ed.created = irc[e].created;  // edges born in an in-place rebuild got theirs on the device
            // compact messages: only eta[0..2) and lam[0..2)x[0..2) are kept on the device
            ed.fv_eta[2] = ed.fv_eta[3] = 0.0;
            for (int q = 0; q < 16; q++)
                if ((q >> 2) >= 2 || (q & 3) >= 2) ed.fv_lam[q] = 0.0;
        }
    }
    return MGX_OK;
}

// ---- message counters (factorgraph/mod.rs:29-137, factorgraph.rs:876-890) ----------------------------
// Every count is structural: who sends to whom is fixed by the topology, the enabled kinds, the
// antenna / idle flags and (for tracking factors) iteration_count.factor — never by message
// contents (a skipped factor still "sends" its empty messages, factor/mod.rs:353-367).  So the
// launches are only logged, and the counters are brought up to date whenever one of those inputs
// is about to change or a count is asked for.

// A connection's counters brought up to date against its robots' cumulative counts (IrConn::base; the flags and the enabled kinds
// have not changed since it was settled last: whoever changes them settles everything first — flush_counts in full).
static void settle_conn(mgx_world *w, IrConn &c) {
    const size_t o = (size_t)c.owner, t = (size_t)c.other;
    const mgx_world::Cum &cu = w->cum;
    const uint64_t now[5] = {cu.nIv[o], cu.nEv[t], cu.nEf[o], cu.on_ir[o], cu.on_ir[t]};
    uint64_t d[5];
    for (int q = 0; q < 5; q++) { d[q] = now[q] - c.base[q]; c.base[q] = now[q]; }
    if (!(w->p.enable_mask & 2u)) return;
    const uint64_t K1 = (uint64_t)(w->K - 1);
    const Robot &ra = w->robots[o], &rb = w->robots[t];
    const bool radio_a = ra.antenna && !ra.idle, radio_b = rb.antenna && !rb.idle;
    c.cnt[2] += d[3];                          // the owner's own variable delivers to its factor (change_prior)
    c.cnt[3] += d[4];                          // the foreign variable delivers to it
    c.cnt[2] += d[0] * K1;                     // own variables' responses (internal sweeps)
    if (radio_a) c.cnt[3] += d[1] * K1;        // the foreign variables' responses (robot.rs:1842-1858)
    const uint64_t sent = d[2] * c.updates_per_sweep;  // external factor sweeps: one message per key
    c.cnt[0] += sent;
    c.cnt[1] += sent;
    if (radio_b) w->robots[t].cnt[3] += sent;  // delivered (robot.rs:1813-1831)
}

// lazy: the robots' own counters and cumulative counts only — the connections stay as they are until somebody needs them settled
// (the per-tick topology pass: what it deletes it settles itself).  Not while factors still lack inbox keys (their counting
// replays the log per connection).
static void flush_counts(mgx_world *w, bool lazy) {
    if (w->pending.active) {  // the launch's entries join the log once it is known to have run
        const int rcc = confirm_resident(w);
        if (rcc != MGX_OK && w->sticky_rc == MGX_OK) w->sticky_rc = rcc;  // (a re-run that failed: reported by whatever runs or reads next)
    }
    const size_t n = w->robots.size();
    mgx_world::Cum &cu = w->cum;
    if (cu.nIv.size() < n) { cu.nIv.resize(n, 0); cu.nEv.resize(n, 0); cu.nEf.resize(n, 0); cu.on_ir.resize(n, 0); }
    const bool keyless = w->n_keyless > 0;
    if (keyless) lazy = false;
    if (w->clog.empty() && w->cp_dirty.empty()) {
        if (!lazy && w->conns_unsettled) {
            for (IrConn &c : w->conns) settle_conn(w, c);
            w->conns_unsettled = false;
        }
        return;
    }
    const int K = w->K;
    const uint32_t en = w->p.enable_mask;
    const uint64_t dynf = (en & 1u) ? 2ull * (K - 1) : 0, obsf = (en & 4u) ? (uint64_t)(K - 2) : 0, trkf = (en & 8u) ? (uint64_t)(K - 2) : 0;
    // The robots (what each one's sweeps were, from the log: their own counters, and the cumulative counts the connections are
    // settled against), then — in full — the connections.  What the passes need of a robot sits in compact arrays: a Robot is a
    // dozen vectors wide.
    std::vector<uint64_t> nIv(n, 0), nEf(n, 0), nEv(n, 0), nIfv(n, 0), trkv(n, 0);
    std::vector<uint8_t> flags(n, 0);  // bit 0: on air (antenna and not idle), bit 1: idle
    for (size_t r = 0; r < n; r++) {
        Robot &rb = w->robots[r];
        // ghosts too: on a sharded world every rank sees the same launches and holds every robot's flags, so the sweeps a
        // ghost has run (what its variables answered to the factors local robots own) are known here; only the totals of
        // the ghost's own graph are its owner's business
        const bool idle = rb.idle != 0, radio = rb.antenna && !idle;
        flags[r] = (uint8_t)((radio ? 1 : 0) | (idle ? 2 : 0));
        uint64_t nIf = 0, trk = 0;
        int64_t itf = rb.cnt_itf;
        for (const mgx_world::CountEntry &e : w->clog) {
            if (e.robot >= 0 && (size_t)e.robot != r) continue;
            // one repetition = [external factor][external variable] n_int x ([internal factor][internal variable])
            const uint64_t fac_per_rep = ((e.ext & 1u) && radio ? 1u : 0u) + ((e.in & 1u) && !idle ? (uint64_t)e.n_int : 0u);
            uint64_t rep = 0;
            for (; rep < e.times && itf < 10 && fac_per_rep; rep++) {  // tracking gate still closed: step by step
                if ((e.ext & 1u) && radio) { nEf[r]++; itf++; }
                if ((e.ext & 2u) && radio) nEv[r]++;
                if (!idle)
                    for (int q = 0; q < e.n_int; q++) {
                        if (e.in & 1u) { nIf++; if (itf >= 10) trk++; itf++; }
                        if (e.in & 2u) nIv[r]++;
                    }
            }
            const uint64_t left = e.times - rep;
            if ((e.ext & 1u) && radio) { nEf[r] += left; itf += (int64_t)left; }
            if ((e.ext & 2u) && radio) nEv[r] += left;
            if (!idle) {
                if (e.in & 1u) { const uint64_t k = left * (uint64_t)e.n_int; nIf += k; if (fac_per_rep) trk += k; itf += (int64_t)k; }
                if (e.in & 2u) nIv[r] += left * (uint64_t)e.n_int;
            }
        }
        rb.cnt_itf = itf;
        nIfv[r] = nIf;
        trkv[r] = trk;
    }
    // change_prior (variable.rs:203-230): its sends stay in a local counter; the connected factors receive
    if (!w->cp_dirty.empty()) {
        for (uint32_t key : w->cp_dirty) {
            const size_t r = key / (uint32_t)K;
            const int i = (int)(key % (uint32_t)K);
            const uint64_t c = w->cp_pending[key];
            w->cp_pending[key] = 0;
            Robot &rb = w->robots[r];
            const uint64_t dyn_here = (uint64_t)((i >= 1) + (i <= K - 2));
            rb.cnt[2] += c * (((en & 1u) ? dyn_here : 0) + ((i >= 1 && i <= K - 2) ? (uint64_t)(((en & 4u) != 0) + ((en & 8u) != 0)) : 0));
            if (i >= 1) cu.on_ir[r] += c;  // one inter-robot factor per connection hangs on this variable
        }
        w->cp_dirty.clear();
    }
    const bool log_any = !w->clog.empty();
    if (keyless) {
        // Some factors still lack inbox keys: what they send follows the log in order, per connection (the connections WITH all
        // their keys are settled as ever, against the cumulative counts BEFORE this log joins them — their share of it is added here).
        for (IrConn &c : w->conns) {
            settle_conn(w, c);  // (prior changes up to now; sweeps up to the last flush)
            if (!(en & 2u) || !log_any) continue;
            const size_t o = (size_t)c.owner, t = (size_t)c.other;
            const bool radio_a = (flags[o] & 1u) != 0, radio_b = (flags[t] & 1u) != 0;
            c.cnt[2] += nIv[o] * (uint64_t)(K - 1);
            if (radio_a) c.cnt[3] += nEv[t] * (uint64_t)(K - 1);
            uint64_t to_own = nEf[o] * c.updates_per_sweep, to_foreign = to_own;
            if (!c.keys.empty()) {  // some keys are still missing: replay the log in order until they are all there
                const bool a_idle = (flags[o] & 2u) != 0;
                to_own = to_foreign = 0;
                for (const mgx_world::CountEntry &e : w->clog) {
                    if (e.robot >= 0 && e.robot != c.owner) continue;  // per-robot launches run internal sweeps only
                    for (uint64_t rep = 0; rep < e.times; rep++) {
                        if (e.robot < 0 && (e.ext & 1u) && radio_a)
                            for (size_t f = 0; f < c.keys.size(); f++) { to_own += c.uses[f] * (c.keys[f] & 1u); to_foreign += c.uses[f] * ((c.keys[f] >> 1) & 1u); }
                        if (e.robot < 0 && (e.ext & 2u) && radio_a && radio_b)
                            for (uint8_t &k : c.keys) k |= 2u;
                        if ((e.in & 2u) && e.n_int > 0 && !a_idle)
                            for (uint8_t &k : c.keys) k |= 1u;
                    }
                }
                bool all = true;
                for (uint8_t k : c.keys) all = all && k == 3u;
                if (all) { c.keys.clear(); c.uses.clear(); w->n_keyless--; }
            }
            c.cnt[0] += to_own;
            c.cnt[1] += to_foreign;
            if (radio_b) w->robots[t].cnt[3] += to_foreign;  // delivered (robot.rs:1813-1831)
        }
    }
    if (log_any) {
        conn_index_ensure(w);  // (how many connections a robot owns / is the target of: the index' lists)
        for (size_t r = 0; r < n; r++) {
            cu.nIv[r] += nIv[r]; cu.nEv[r] += nEv[r]; cu.nEf[r] += nEf[r];
            Robot &rb = w->robots[r];
            if (rb.ghost) continue;
            const uint64_t own = w->cidx.out[r].size(), foreign = w->cidx.in[r].size();
            const uint64_t s_int = 2ull * (K - 1) + 2ull * (K - 2) + (uint64_t)(K - 1) * own, s_ext = (uint64_t)(K - 1) * foreign;
            // internal factor sweeps: one message per inbox key of every updated factor, received by the variables
            rb.cnt[0] += nIfv[r] * (dynf + obsf) + trkv[r] * trkf;
            rb.cnt[2] += nIfv[r] * (dynf + obsf) + trkv[r] * trkf;
            // variable sweeps answer every inbox key; only own-graph, enabled factors receive (internal sweeps)
            rb.cnt[0] += (nIv[r] + nEv[r]) * s_int;
            rb.cnt[1] += (nIv[r] + nEv[r]) * s_ext;
            rb.cnt[2] += nIv[r] * (dynf + obsf + trkf);
        }
    }
    if (keyless) {  // (their share of this log was added by hand above: the bases move with the cumulative counts)
        for (IrConn &c : w->conns) {
            const size_t o = (size_t)c.owner, t = (size_t)c.other;
            c.base[0] = cu.nIv[o]; c.base[1] = cu.nEv[t]; c.base[2] = cu.nEf[o];
        }
        w->conns_unsettled = false;
    } else if (!lazy) {
        for (IrConn &c : w->conns) settle_conn(w, c);
        w->conns_unsettled = false;
    } else {
        w->conns_unsettled = true;
    }
    w->clog.clear();
}
static void log_launch(mgx_world *w, int robot, uint32_t ext_mask, uint32_t int_mask, int n_int) {
    const uint8_t ext = (uint8_t)((ext_mask & PH_EXT_FACTOR ? 1 : 0) | (ext_mask & PH_EXT_VARIABLE ? 2 : 0));
    const uint8_t in = (uint8_t)((int_mask & PH_INT_FACTOR ? 1 : 0) | (int_mask & PH_INT_VARIABLE ? 2 : 0));
    if (!ext && (!in || n_int <= 0)) return;
    if (!w->clog.empty()) {
        mgx_world::CountEntry &b = w->clog.back();
        if (b.ext == ext && b.in == in && b.n_int == n_int && b.robot == robot) { b.times++; return; }
    }
    w->clog.push_back({ext, in, in ? n_int : 0, robot, 1});
    if (w->clog.size() > 4096) flush_counts(w);
}
static void log_change_prior(mgx_world *w, int robot, int var) {
    if (w->n_keyless > 0 && var >= 1 && (w->p.enable_mask & 2u)) {  // the delivery fills inbox keys of factors that lack them
        flush_counts(w);                                           // (what was logged so far saw them missing)
        for (IrConn &c : w->conns) {
            if (c.keys.empty()) continue;
            if (c.owner == robot) c.keys[(size_t)var - 1] |= 1u;
            if (c.other == robot) c.keys[(size_t)var - 1] |= 2u;
        }
    }
    const size_t key = (size_t)robot * (size_t)w->K + (size_t)var;
    if (w->cp_pending.size() <= key) w->cp_pending.resize(w->robots.size() * (size_t)w->K, 0);
    if (w->cp_pending[key]++ == 0) w->cp_dirty.push_back((uint32_t)key);
}

// ---- commit: host mirror -> device arrays ---------------------------------------------------------
static int upload_flags(mgx_world *w) {
    // antenna[R] | idle[R] staged in one pinned block and moved by a copy kernel, the edge gates
    // derived from them on the device: no blocking copy, no synchronisation (update_failed_comms
    // rewrites every antenna each tick)
    const size_t R = w->robot_of.size(), NE = (size_t)std::max(w->d.NI, 1);
    void *hp = nullptr;
    int slot = 0;
    HIP_TRY(w->stage.acquire(2 * R, &hp, &slot));
    uint8_t *an = (uint8_t *)hp, *id = an + R;
    for (size_t dr = 0; dr < R; dr++) {
        an[dr] = w->robots[(size_t)w->robot_of[dr]].antenna;
        id[dr] = w->robots[(size_t)w->robot_of[dr]].idle;
    }
    HIP_TRY(w->antenna.reserve(R));
    HIP_TRY(w->idle.reserve(R));
    HIP_TRY(w->ir_gate.reserve(NE));
    void *dp = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dp, hp, 0));
    const uint8_t *src = (const uint8_t *)dp;
    HIP_TRY(launch_copy_bytes(w->antenna.p, src, R, w->stream));
    HIP_TRY(launch_copy_bytes(w->idle.p, src + R, R, w->stream));
    HIP_TRY(w->stage.release(slot, w->stream));
    // edge gates: the owner of the edge is on air
    const int n_edges = w->dev_in_ptr.empty() ? 0 : w->dev_in_ptr.back() * (w->K - 1);
    if (n_edges == 0) HIP_TRY(hipMemsetAsync(w->ir_gate.p, 0, NE, w->stream));
    HIP_TRY(launch_edge_gates(n_edges, w->ir_rec.p, w->antenna.p, w->idle.p, w->ir_gate.p, w->stream));
    w->d.antenna = w->antenna.p;
    w->d.idle = w->idle.p;
    w->d.ir_gate = w->ir_gate.p;
    w->flags_dirty = false;
    return MGX_OK;
}

static size_t edge_index(const std::vector<int32_t> &in_ptr, int K, int r, int j, int slot) {
    const int n_in = in_ptr[(size_t)r + 1] - in_ptr[(size_t)r];
    return (size_t)(K - 1) * (size_t)in_ptr[(size_t)r] + (size_t)j * (size_t)n_in + (size_t)(slot - in_ptr[(size_t)r]);
}
// ---- the connection index (ConnIndex) ------------------------------------------------------------------------------------------
static bool conn_before(const mgx_world *w, int32_t a, int32_t b, bool *interleaved) {  // the inbox order of two connections of one target
    const mgx_world::ConnHot &ca = w->conn_hot[(size_t)a], &cb = w->conn_hot[(size_t)b];
    const uint64_t ka = w->sets.keys[(size_t)ca.owner], kb = w->sets.keys[(size_t)cb.owner];
    if (ka != kb) return ka < kb;
    if ((ca.node_first < cb.node_first) != (ca.node_last < cb.node_last)) *interleaved = true;
    return ca.node_first < cb.node_first;
}
static void conn_index_add(mgx_world *w, int32_t ci) {
    ConnIndex &x = w->cidx;
    const mgx_world::ConnHot &c = w->conn_hot[(size_t)ci];
    std::vector<int32_t> &lst = x.in[(size_t)c.other];
    size_t at = lst.size();
    while (at > 0 && conn_before(w, ci, lst[at - 1], &x.interleaved)) at--;
    lst.insert(lst.begin() + (long)at, ci);
    x.out[(size_t)c.owner].push_back(ci);
}
static void conn_index_rebuild(mgx_world *w) {
    ConnIndex &x = w->cidx;
    const size_t n = w->robots.size();
    x.in.assign(n, {});
    x.out.assign(n, {});
    x.interleaved = false;
    for (size_t ci = 0; ci < w->conn_hot.size(); ci++) conn_index_add(w, (int32_t)ci);
    x.valid = true;
}
static void conn_index_ensure(mgx_world *w) {
    if (!w->cidx.valid || w->cidx.in.size() != w->robots.size()) conn_index_rebuild(w);
}
static void conn_index_drop(mgx_world *w, int32_t ci) {  // connection ci leaves the list
    ConnIndex &x = w->cidx;
    const mgx_world::ConnHot &c = w->conn_hot[(size_t)ci];
    std::vector<int32_t> &a = x.in[(size_t)c.other], &b = x.out[(size_t)c.owner];
    a.erase(std::find(a.begin(), a.end(), ci));
    b.erase(std::find(b.begin(), b.end(), ci));
}
static void conn_index_moved(mgx_world *w, int32_t from, int32_t to) {  // connection `from` now sits at `to` (conn_hot[to] holds it already)
    ConnIndex &x = w->cidx;
    const mgx_world::ConnHot &c = w->conn_hot[(size_t)to];
    std::vector<int32_t> &a = x.in[(size_t)c.other], &b = x.out[(size_t)c.owner];
    *std::find(a.begin(), a.end(), from) = to;
    *std::find(b.begin(), b.end(), from) = to;
}
// build_incoming from the index: O(local robots + their connections), no pass over the whole list, no sorting
static void build_incoming_indexed(mgx_world *w, int R_local, Incoming &t, bool want_peers) {
    const int K = w->K;
    const ConnIndex &x = w->cidx;
    const uint64_t *key = w->sets.keys.data();
    const int32_t *dev_of = w->dev_of.data();
    const mgx_world::ConnHot *conns = w->conn_hot.data();
    const size_t R = (size_t)R_local;
    t.in_ptr.assign(R + 1, 0);
    t.mid.assign((size_t)std::max(R_local, 1), 0);
    t.blocks_ok = !x.interleaved;
    t.ir_max_edges = 0;
    size_t n_in_all = 0, n_peer_all = 0;
    for (size_t r = 0; r < R; r++) {
        const size_t id = (size_t)w->robot_of[r];
        n_in_all += x.in[id].size();
        n_peer_all += x.in[id].size() + x.out[id].size();
        t.in_ptr[r + 1] = (int32_t)n_in_all;
    }
    t.in_list.resize(n_in_all);
    t.peers.clear();
    int32_t *pp = nullptr, *pidx = nullptr;
    if (want_peers) {
        t.peers.assign(R + 1 + std::max<size_t>(n_peer_all, 1), 0);
        pp = t.peers.data();
        pidx = pp + R + 1;
    }
    size_t pw = 0;
    for (size_t r = 0; r < R; r++) {
        const size_t id = (size_t)w->robot_of[r];
        const std::vector<int32_t> &lst = x.in[id];
        const int n_in = (int)lst.size();
        if (n_in) memcpy(t.in_list.data() + t.in_ptr[r], lst.data(), sizeof(int32_t) * (size_t)n_in);
        const uint64_t own_key = key[id];
        int mid = n_in;  // first connection whose owner has a HIGHER key than the target
        for (int q = n_in - 1; q >= 0; q--)
            if (key[(size_t)conns[(size_t)lst[(size_t)q]].owner] > own_key) mid = q;
        t.mid[r] = mid;
        t.ir_max_edges = std::max(t.ir_max_edges, n_in * (K - 1));
        if (pp) {
            for (int32_t ci : lst) pidx[pw++] = dev_of[(size_t)conns[(size_t)ci].owner];
            for (int32_t ci : x.out[id]) pidx[pw++] = dev_of[(size_t)conns[(size_t)ci].other];
            pp[r + 1] = (int32_t)pw;
        }
    }
    if (pp) t.peers.resize(R + 1 + std::max<size_t>(pw, 1));
}

static void build_incoming(const mgx_world *w, int R_local, Incoming &t, bool want_peers = false) {
    const int K = w->K;
    const uint64_t *key = w->sets.keys.data();  // (compact copies of the robots' order keys and ghost flags: a Robot is a dozen vectors wide)
    const uint8_t *ghost = w->sets.ghost.data();
    const int32_t *dev_of = w->dev_of.data();
    const mgx_world::ConnHot *conns = w->conn_hot.data();  // (owner, other, first and last node slot of every connection)
    const size_t n_conns = w->conns.size(), R = (size_t)R_local;
    t.in_ptr.assign(R + 1, 0);
    t.mid.assign((size_t)std::max(R_local, 1), 0);
    int32_t *pp = nullptr;
    t.blocks_ok = true;
    t.peers.clear();
    if (want_peers) {
        t.peers.assign(R + 1 + std::max<size_t>(2 * n_conns, 1), 0);
        pp = t.peers.data();
    }
    for (size_t ci = 0; ci < n_conns; ci++) {
        const size_t o = (size_t)dev_of[(size_t)conns[ci].owner], tt = (size_t)dev_of[(size_t)conns[ci].other];
        if (!ghost[(size_t)conns[ci].other]) t.in_ptr[tt + 1]++;  // ghost target: another rank's
        if (pp) {
            if (o < R) pp[o + 1]++;
            if (tt < R) pp[tt + 1]++;
        }
    }
    for (size_t r = 0; r < R; r++) t.in_ptr[r + 1] += t.in_ptr[r];
    t.in_list.assign((size_t)t.in_ptr[R], 0);
    std::vector<int32_t> &fill = t.fill, &pfill = t.pfill;
    fill.assign(t.in_ptr.begin(), t.in_ptr.end() - 1);
    if (pp) {
        for (size_t r = 0; r < R; r++) pp[r + 1] += pp[r];
        pfill.assign(pp, pp + R);
    }
    int32_t *pidx = pp ? pp + R + 1 : nullptr;
    for (size_t ci = 0; ci < n_conns; ci++) {
        const int o = dev_of[(size_t)conns[ci].owner], tt = dev_of[(size_t)conns[ci].other];
        if (!ghost[(size_t)conns[ci].other]) t.in_list[(size_t)fill[(size_t)tt]++] = (int32_t)ci;
        if (pp) {
            if ((size_t)o < R) pidx[(size_t)pfill[(size_t)o]++] = tt;
            if ((size_t)tt < R) pidx[(size_t)pfill[(size_t)tt]++] = o;
        }
    }
    if (pp) t.peers.resize(R + 1 + (size_t)std::max(pp[R], 1));
    t.ir_max_edges = 0;
    for (int r = 0; r < R_local; r++) {
        int32_t *seg = t.in_list.data() + t.in_ptr[(size_t)r];
        const int n_in = t.in_ptr[(size_t)r + 1] - t.in_ptr[(size_t)r];
        auto before = [&](int32_t a, int32_t b) {
            const mgx_world::ConnHot &ca = conns[(size_t)a], &cb = conns[(size_t)b];
            const uint64_t ka = key[(size_t)ca.owner], kb = key[(size_t)cb.owner];
            if (ka != kb) return ka < kb;
            if ((ca.node_first < cb.node_first) != (ca.node_last < cb.node_last)) t.blocks_ok = false;
            return ca.node_first < cb.node_first;
        };
        for (int a = 1; a < n_in; a++) {  // insertion sort: a handful of connections per robot
            const int32_t v = seg[a];
            int b = a;
            while (b > 0 && before(v, seg[b - 1])) { seg[b] = seg[b - 1]; b--; }
            seg[b] = v;
        }
        const uint64_t own_key = key[(size_t)w->robot_of[(size_t)r]];
        int mid = n_in;  // first connection whose owner has a HIGHER key than the target
        for (int q = n_in - 1; q >= 0; q--)
            if (key[(size_t)conns[(size_t)seg[q]].owner] > own_key) mid = q;
        t.mid[(size_t)r] = mid;
        t.ir_max_edges = std::max(t.ir_max_edges, n_in * (K - 1));
    }
}
// the constant record of edge (connection c, variable j + 1); `created` is the caller's
static IrEdgeRec edge_record(const mgx_world *w, const IrConn &c, int j) {
    const Robot &ow = w->robots[(size_t)c.owner], &ot = w->robots[(size_t)c.other];
    IrEdgeRec rec;
    rec.src_var = w->dev_of[(size_t)c.owner] * w->K + j + 1;
    rec.src_robot = w->dev_of[(size_t)c.owner];
    rec.created = 0;
    rec.dst = (int32_t)(j + 1) | ((ot.order_key > ow.order_key) ? (1 << 16) : 0);
    rec.d_safe = w->p.safety_multiplier * ow.radius;                      // interrobot.rs:64
    rec.offset = (double)1e-6f * (double)(c.first_number + (uint64_t)j);  // interrobot.rs:52,75
    return rec;
}

static int upload_flags(mgx_world *w);

// Only inter-robot connections changed (create_ / delete_interrobot_factors between two ticks):
// the edge tables are rebuilt on the device from the old ones — surviving edges keep their state
// (message, response mean, creation epoch), new edges are initialised from the owner's current
// delivery count and the target's current belief mean (robot.rs:1549-1585) — without moving any
// robot state between host and device.  The host sends one record per CONNECTION; the per-edge
// arrays are laid out by k_edge_rebuild.
static int retopo(mgx_world *w) {
    const int K = w->K, R_local = w->d.R_local;
    StageTimer tm("retopo");
    // (the tables are rebuilt every tick by a world that follows its topology: their storage stays with the world — a quarter of a
    // megabyte of slot records from the allocator every tick is page faults)
    Incoming &t = w->retopo_tables;
    // (a world whose schedules can run as resident launches gets that kernel's peer table from the same passes, in the same block)
    const bool want_peers = (w->d.R_total == w->d.R_local || w->xres.connected) && (w->p.enable_mask & 2u) && w->sweep_flag_buf.p;
    conn_index_ensure(w);
    build_incoming_indexed(w, R_local, t, want_peers);
    static const bool check_index = getenv("MGX_CHECK_INDEX") != nullptr;
    if (check_index) {  // (diagnostics: the tables of the index against the two passes over the list they replace)
        Incoming full;
        build_incoming(w, R_local, full, want_peers);
        bool same = full.in_ptr == t.in_ptr && full.in_list == t.in_list && full.mid == t.mid && full.ir_max_edges == t.ir_max_edges &&
                    full.blocks_ok == t.blocks_ok && full.peers.size() == t.peers.size();
        if (same && want_peers) {
            for (size_t r = 0; r < (size_t)R_local && same; r++) {  // (a row's order carries no meaning: compared as multisets)
                same = full.peers[r + 1] == t.peers[r + 1];
                if (!same) break;
                std::vector<int32_t> a(full.peers.begin() + R_local + 1 + full.peers[r], full.peers.begin() + R_local + 1 + full.peers[r + 1]),
                    b(t.peers.begin() + R_local + 1 + t.peers[r], t.peers.begin() + R_local + 1 + t.peers[r + 1]);
                std::sort(a.begin(), a.end());
                std::sort(b.begin(), b.end());
                same = a == b;
            }
        }
        if (!same) return fail(MGX_ERR_STATE, "internal: the connection index and the connection list disagree (MGX_CHECK_INDEX)");
    }
    tm.lap("build_incoming");
    if (!t.blocks_ok) return fail(MGX_ERR_STATE, "internal: node slots of two connections interleave");
    const size_t n_slots = t.in_list.size(), NI = n_slots * (size_t)(K - 1), NIs = std::max<size_t>(NI, 1);
    std::vector<IrSlotRec> &slots = w->retopo_slots;
    slots.resize(std::max<size_t>(n_slots, 1));
    mgx_world::ConnHot *hot = w->conn_hot.data();  // (slot words, robot numbers and the "fresh" marks sit in the connections' mirror)
    if (n_slots != w->conns.size())  // (connections towards a ghost are another rank's: no slot here; all others are rewritten below)
        for (size_t ci = 0; ci < w->conn_hot.size(); ci++)
            if (w->sets.ghost[(size_t)hot[ci].other]) hot[ci].dev_slot = -1;
    for (size_t g = 0; g < n_slots; g++) {
        mgx_world::ConnHot &c = hot[(size_t)t.in_list[g]];
        IrSlotRec &sl = slots[g];
        sl.tgt_robot = w->dev_of[(size_t)c.other];
        sl.src_robot = w->dev_of[(size_t)c.owner];
        sl.old_slot = c.dev_slot;
        sl.flags = (w->sets.keys[(size_t)c.other] > w->sets.keys[(size_t)c.owner]) ? 1 : 0;
        sl.d_safe = w->p.safety_multiplier * w->sets.radius[(size_t)c.owner];
        sl.first_number = c.first_number;
        c.dev_slot = (int32_t)g;  // (from here on the connection's slot in the tables being built)
        if (c.has_fresh) {
            for (IrEdge &ed : w->conns[(size_t)t.in_list[g]].edges) ed.fresh = false;
            c.has_fresh = 0;
        }
    }
    tm.lap("slot records");
    hipStream_t s = w->stream;
    // the three host tables travel through one pinned block of the argument ring (every part padded to 16 bytes); ONE kernel
    // takes it apart into the device arrays and derives the per-variable tables from it (k_retopo_unpack), a second one lays the
    // edges out and — the robots' flags unchanged — writes their gate bytes along: two launches, no copies, no synchronisation
    // (the block is released by an event after the kernels)
    auto pad16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
    const size_t b_slots = pad16(sizeof(IrSlotRec) * slots.size()), b_ptr = pad16(sizeof(int32_t) * t.in_ptr.size()),
                 b_mid = pad16(sizeof(int32_t) * t.mid.size()), b_peers = pad16(sizeof(int32_t) * t.peers.size());
    void *hp = nullptr, *dp = nullptr;
    int ring_slot = 0;
    HIP_TRY(w->stage.acquire(b_slots + b_ptr + b_mid + b_peers, &hp, &ring_slot));
    memcpy(hp, slots.data(), sizeof(IrSlotRec) * slots.size());
    memcpy((char *)hp + b_slots, t.in_ptr.data(), sizeof(int32_t) * t.in_ptr.size());
    memcpy((char *)hp + b_slots + b_ptr, t.mid.data(), sizeof(int32_t) * t.mid.size());
    if (b_peers) {
        memcpy((char *)hp + b_slots + b_ptr + b_mid, t.peers.data(), sizeof(int32_t) * t.peers.size());
        HIP_TRY(w->peer_ptr_dev.reserve(b_peers / sizeof(int32_t)));
    }
    HIP_TRY(hipHostGetDevicePointer(&dp, hp, 0));
    HIP_TRY(w->slot_recs.reserve(b_slots / sizeof(IrSlotRec) + 1));
    HIP_TRY(w->in_ptr_dev_b.reserve(b_ptr / sizeof(int32_t)));
    HIP_TRY(w->in_mid_dev.reserve(b_mid / sizeof(int32_t)));
    HIP_TRY(w->ir_rec_b.reserve(NIs));
    HIP_TRY(w->ir_fv_eta_b.reserve(4 * NIs));
    HIP_TRY(w->ir_fv_lam_b.reserve(16 * NIs));
    HIP_TRY(w->ir_bmu_b.reserve(4 * NIs));
    const bool gates_along = !w->flags_dirty;
    if (gates_along) HIP_TRY(w->ir_gate.reserve(NIs));
    HIP_TRY(launch_retopo_unpack(dp, b_slots, b_ptr, b_mid, b_peers, w->slot_recs.p, w->in_ptr_dev_b.p, w->in_mid_dev.p,
                                 b_peers ? w->peer_ptr_dev.p : nullptr, R_local, K, w->ir_var_ptr.p, w->ir_var_mid.p, s));
    HIP_TRY(launch_edge_rebuild(w->d, (int)n_slots, w->slot_recs.p, w->in_ptr_dev_b.p, w->in_ptr_dev.p, (int)NIs, w->ir_rec_b.p,
                                w->ir_fv_eta_b.p, w->ir_fv_lam_b.p, w->ir_bmu_b.p, gates_along ? w->ir_gate.p : nullptr, s));
    HIP_TRY(w->stage.release(ring_slot, s));
    tm.lap("uploads + launches");
    w->ir_rec.swap(w->ir_rec_b);
    w->ir_fv_eta.swap(w->ir_fv_eta_b);
    w->ir_fv_lam.swap(w->ir_fv_lam_b);
    w->ir_bmu.swap(w->ir_bmu_b);
    w->in_ptr_dev.swap(w->in_ptr_dev_b);
    DevWorld &d = w->d;
    d.NI = (int)NIs;
    d.ir_rec = w->ir_rec.p; d.ir_fv_eta = w->ir_fv_eta.p; d.ir_fv_lam = w->ir_fv_lam.p; d.ir_bmu = w->ir_bmu.p;
    // the resident kernel's LDS per workgroup (hence workgroups per CU) follows the largest number of edges on one robot:
    // a capacity asked for a sparser topology says nothing about this one
    if (t.ir_max_edges != d.ir_max_edges) w->resident_cap = w->resident_cap_sharded = -1;
    d.ir_max_edges = t.ir_max_edges;
    w->dev_in_ptr = t.in_ptr;
    w->conns_dirty = false;
    w->peers_valid = b_peers != 0;
    if (b_peers) {
        w->peer_idx_off = (size_t)R_local + 1;
        w->d.peer_ptr = w->peer_ptr_dev.p;
        w->d.peer_idx = w->peer_ptr_dev.p + w->peer_idx_off;
    }
    tm.lap("slot bookkeeping");
    int rc_flags = MGX_OK;  // the gate bytes follow the edges (the robots' own flags only when they changed too)
    if (w->flags_dirty) {
        rc_flags = upload_flags(w);
    } else {
        if (NI == 0) HIP_TRY(hipMemsetAsync(w->ir_gate.p, 0, NIs, s));
        d.ir_gate = w->ir_gate.p;  // (written by k_edge_rebuild)
    }
    tm.lap("flags + gates");
    return rc_flags;
}


static int commit(mgx_world *w) {
    if (!device_ok()) return fail(MGX_ERR_NO_DEVICE, "no usable HIP device (this library has no CPU path)");
    if (w->pending.active) { const int rcc = confirm_resident(w); if (rcc != MGX_OK) return rcc; }
    // (whoever comes through here enqueues work or reads state: a lingering launch ends first — unless the caller is on its way
    // to post a schedule into it)
    if (w->linger.open && !w->linger.hold) { const int rcl = linger_close(w); if (rcl != MGX_OK) return rcl; }
    if (!w->dirty) {
        if (w->conns_dirty) return retopo(w);
        if (w->flags_dirty) return upload_flags(w);
        return MGX_OK;
    }
    if (w->robots.empty()) return fail(MGX_ERR_STATE, "world has no robots");
    int rc = pull(w);
    if (rc != MGX_OK) return rc;
    const int K = w->K, E = 4 * K - 6;
    if (!sweep_supports(K) || sweep_lds_bytes(K, 0) > 60 * 1024) return fail(MGX_ERR_INVALID, "K = %d not supported (3 <= K <= 45: the robot's graph has to fit 60 KB of LDS)", K);

    // device robot order: locals (id order), then ghosts
    w->robot_of.clear();
    w->dev_of.assign(w->robots.size(), -1);
    for (size_t r = 0; r < w->robots.size(); r++)
        if (!w->robots[r].ghost) { w->dev_of[r] = (int)w->robot_of.size(); w->robot_of.push_back((int)r); }
    const int R_local = (int)w->robot_of.size();
    for (size_t r = 0; r < w->robots.size(); r++)
        if (w->robots[r].ghost) { w->dev_of[r] = (int)w->robot_of.size(); w->robot_of.push_back((int)r); }
    const int R_total = (int)w->robot_of.size();
    const size_t V = (size_t)R_total * K, EI = (size_t)std::max(R_local, 1) * E, ND = (size_t)std::max(R_local, 1) * (K - 1),
                 NT = (size_t)std::max(R_local, 1) * std::max(K - 2, 1);

    // initialise fresh inter-robot edges from the (pulled) current state: created epoch of the
    // owner's variable (its inbox slot stays empty until the owner's next delivery) and the target
    // variable's current belief mean (robot.rs:1549-1585)
    for (IrConn &c : w->conns)
        for (size_t j = 0; j < c.edges.size(); j++) {
            IrEdge &ed = c.edges[j];
            if (!ed.fresh) continue;
            const Robot &ow = w->robots[(size_t)c.owner], &ot = w->robots[(size_t)c.other];
            ed.created = ow.epoch[j + 1];
            for (int q = 0; q < 4; q++) ed.bmu[q] = (w->p.enable_mask & 2u) ? ot.bel_mu[4 * (j + 1) + q] : 0.0;  // dropped while the kind is off
            ed.fresh = false;
        }

    Incoming t;
    build_incoming(w, R_local, t);
    if (!t.blocks_ok) return fail(MGX_ERR_STATE, "internal: node slots of two connections interleave");
    const size_t n_slots = t.in_list.size(), NI = n_slots * (size_t)(K - 1), NIs = std::max<size_t>(NI, 1);
    std::vector<IrEdgeRec> recs(NIs, IrEdgeRec{0, 0, 0, 0, 0.0, 0.0});
    std::vector<double> ife(4 * NIs, 0.0), ifl(16 * NIs, 0.0), ibm(4 * NIs, 0.0);
    for (mgx_world::ConnHot &h : w->conn_hot) h.dev_slot = -1;
    for (size_t g = 0; g < n_slots; g++) {
        IrConn &c = w->conns[(size_t)t.in_list[g]];
        w->conn_hot[(size_t)t.in_list[g]].dev_slot = (int32_t)g;
        const int r = w->dev_of[(size_t)c.other];
        for (int j = 0; j < K - 1; j++) {
            const size_t e = edge_index(t.in_ptr, K, r, j, (int)g);
            const IrEdge &ed = c.edges[(size_t)j];
            recs[e] = edge_record(w, c, j);
            recs[e].created = ed.created;
            scatter(ife, NIs, e, ed.fv_eta, 4);
            scatter(ifl, NIs, e, ed.fv_lam, 16);
            scatter(ibm, NIs, e, ed.bmu, 4);
        }
    }
    w->dev_in_ptr = t.in_ptr;
    const int ir_max_edges = t.ir_max_edges;

    const size_t BS = (size_t)blob_words(K);
    std::vector<double> blb(BS * (size_t)R_total, 0.0), sn(24 * V), dm(16 * ND, 0.0), tlv(NT, 0.0);
    std::vector<int32_t> trc(NT, 0), pptr((size_t)R_local + 1, 0), itf((size_t)std::max(R_local, 1), 0);
    std::vector<uint32_t> ep(V);
    std::vector<float> tlp(2 * NT, 0.f), pxy;
    for (int dr = 0; dr < R_total; dr++) {
        const Robot &rb = w->robots[(size_t)w->robot_of[(size_t)dr]];
        blob_pack(rb, &blb[(size_t)dr * BS]);
        for (int i = 0; i < K; i++) {
            const size_t v = (size_t)dr * K + i;
            memcpy(&sn[v * 24], &rb.snap[24 * i], 24 * sizeof(double));
            ep[v] = rb.epoch[i];
        }
        if (rb.ghost) continue;
        for (int f = 0; f < K - 1; f++) scatter(dm, ND, (size_t)dr * (K - 1) + f, &rb.dyn_m[16 * f], 16);
        for (int j = 0; j < K - 2; j++) {
            const size_t t = (size_t)dr * (K - 2) + j;
            trc[t] = rb.trk_record[j];
            tlp[t] = rb.trk_last_pos[2 * j];
            tlp[NT + t] = rb.trk_last_pos[2 * j + 1];
            tlv[t] = rb.trk_last_val[j];
        }
        pptr[(size_t)dr] = (int32_t)(pxy.size() / 2);
        pxy.insert(pxy.end(), rb.path.begin(), rb.path.end());
        pptr[(size_t)dr + 1] = (int32_t)(pxy.size() / 2);
        itf[(size_t)dr] = rb.iter_factor;
    }
    if (pxy.empty()) pxy.assign(2, 0.f);
    if (recs.empty()) recs.push_back(IrEdgeRec{0, 0, 0, 0, 0.0, 0.0});
    if (w->sdf_red.empty()) {  // no image: every lookup is "outside" => h = 0
        w->sdf_red.assign(1, 255);
        w->sdf_w = w->sdf_h = 0;
    }

    hipStream_t s = w->stream;
    HIP_TRY(w->blob.upload(blb, s));
    HIP_TRY(w->snap0.upload(sn, s));
    HIP_TRY(w->snap1.upload(sn, s));
    HIP_TRY(w->epoch0.upload(ep, s));
    HIP_TRY(w->epoch1.upload(ep, s));
    HIP_TRY(w->dyn_m.upload(dm, s));
    HIP_TRY(w->trk_record.upload(trc, s));
    HIP_TRY(w->trk_last_pos.upload(tlp, s));
    HIP_TRY(w->trk_last_val.upload(tlv, s));
    HIP_TRY(w->path_ptr.upload(pptr, s));
    HIP_TRY(w->path_xy.upload(pxy, s));
    HIP_TRY(w->iter_factor.upload(itf, s));
    HIP_TRY(w->in_ptr_dev.upload(t.in_ptr, s));
    HIP_TRY(w->in_mid_dev.upload(t.mid, s));
    HIP_TRY(w->ir_var_ptr.reserve((size_t)R_local * K + 1));
    HIP_TRY(w->ir_var_mid.reserve((size_t)std::max(R_local, 1) * K));
    HIP_TRY(launch_var_tables(R_local, K, w->in_ptr_dev.p, w->in_mid_dev.p, w->ir_var_ptr.p, w->ir_var_mid.p, s));
    HIP_TRY(w->ir_rec.upload(recs, s));
    HIP_TRY(w->ir_fv_eta.upload(ife, s));
    HIP_TRY(w->ir_fv_lam.upload(ifl, s));
    HIP_TRY(w->ir_bmu.upload(ibm, s));
    HIP_TRY(w->sdf.upload(w->sdf_red, s));
    std::vector<double> fzd;
    std::vector<uint8_t> fzf, thw, zero_bytes;
    if (w->frozen_live) {  // robots that joined since start with empty frozen inboxes and nothing to thaw
        const size_t FZ = (size_t)frozen_words(K), RL = (size_t)std::max(R_local, 1);
        fzd.assign(RL * FZ, 0.0);
        fzf.assign(RL * (size_t)E, 0);
        thw.assign(RL, 0);
        zero_bytes.assign(RL, 0);
        for (int dr = 0; dr < R_local; dr++) {
            const Robot &rb = w->robots[(size_t)w->robot_of[(size_t)dr]];
            if (rb.frozen.size() == FZ) std::copy(rb.frozen.begin(), rb.frozen.end(), fzd.begin() + (long)((size_t)dr * FZ));
            if (rb.frozen_flag.size() == (size_t)E) std::copy(rb.frozen_flag.begin(), rb.frozen_flag.end(), fzf.begin() + (long)((size_t)dr * (size_t)E));
            thw[(size_t)dr] = rb.thaw;
        }
        HIP_TRY(w->frozen_buf.upload(fzd, s));
        HIP_TRY(w->frozen_flag_buf.upload(fzf, s));
        HIP_TRY(w->thaw_buf.upload(thw, s));
        HIP_TRY(w->skip0_buf.upload(zero_bytes, s));
    }
    std::vector<double> ifs;
    std::vector<uint32_t> ife_, ite;
    if (w->ir_frozen_live) {  // robots that joined since have sent nothing their (disabled) factors could have kept
        ifs.assign(24 * V, 0.0);
        ife_.assign(V, 0);
        ite.assign(V, 0xffffffffu);  // "has delivered since": a joiner's live record is its inbox
        for (int dr = 0; dr < R_total; dr++) {
            const Robot &rb = w->robots[(size_t)w->robot_of[(size_t)dr]];
            if (rb.ir_frozen_snap.size() == (size_t)24 * K) std::copy(rb.ir_frozen_snap.begin(), rb.ir_frozen_snap.end(), ifs.begin() + (long)((size_t)dr * K * 24));
            if (rb.ir_frozen_epoch.size() == (size_t)K) std::copy(rb.ir_frozen_epoch.begin(), rb.ir_frozen_epoch.end(), ife_.begin() + (long)((size_t)dr * K));
            if (rb.ir_thaw_epoch.size() == (size_t)K) std::copy(rb.ir_thaw_epoch.begin(), rb.ir_thaw_epoch.end(), ite.begin() + (long)((size_t)dr * K));
        }
        HIP_TRY(w->ir_frozen_snap_buf.upload(ifs, s));
        HIP_TRY(w->ir_frozen_epoch_buf.upload(ife_, s));
        HIP_TRY(w->ir_thaw_epoch_buf.upload(ite, s));
    }
    HIP_TRY(hipStreamSynchronize(s));  // host staging vectors die at scope exit

    DevWorld &d = w->d;
    d.R_local = R_local; d.R_total = R_total; d.K = K; d.E = E;
    d.V = (int)V; d.EI = (int)EI; d.ND = (int)ND; d.NT = (int)NT; d.NI = (int)NIs;
    d.cur = 0;
    d.ir_max_edges = ir_max_edges;
    d.trk_cols = w->trk_ever_on ? 1 : 0;
    d.upd = nullptr; d.upd_max_speed = 0.0; d.upd_delta_t = 0.0;
    d.ir_frozen_snap = w->ir_thaw_active ? w->ir_frozen_snap_buf.p : nullptr;
    d.ir_frozen_epoch = w->ir_thaw_active ? w->ir_frozen_epoch_buf.p : nullptr;
    d.ir_thaw_epoch = w->ir_thaw_active ? w->ir_thaw_epoch_buf.p : nullptr;
    d.frozen = w->frozen_live ? w->frozen_buf.p : nullptr;
    d.frozen_flag = w->frozen_live ? w->frozen_flag_buf.p : nullptr;
    d.thaw = w->frozen_live ? w->thaw_buf.p : nullptr;
    d.skip0 = (w->frozen_live && w->thaw_kinds) ? w->skip0_buf.p : nullptr;
    d.enable = w->p.enable_mask;
    d.blob = w->blob.p; d.BS = (int)BS;
    d.snap[0] = w->snap0.p; d.snap[1] = w->snap1.p;
    d.snap_epoch[0] = w->epoch0.p; d.snap_epoch[1] = w->epoch1.p;
    d.dyn_m = w->dyn_m.p;
    d.trk_record = w->trk_record.p; d.trk_last_pos = w->trk_last_pos.p; d.trk_last_val = w->trk_last_val.p;
    d.path_ptr = w->path_ptr.p; d.path_xy = w->path_xy.p; d.iter_factor = w->iter_factor.p;
    d.ir_var_ptr = w->ir_var_ptr.p; d.ir_var_mid = w->ir_var_mid.p; d.ir_rec = w->ir_rec.p;
    d.ir_fv_eta = w->ir_fv_eta.p; d.ir_fv_lam = w->ir_fv_lam.p; d.ir_bmu = w->ir_bmu.p;
    d.sdf = w->sdf.p; d.sdf_w = w->sdf_w; d.sdf_h = w->sdf_h; d.world_w = w->world_w; d.world_h = w->world_h;
    // ObstacleFactor::new jacobian_delta (obstacle.rs:98-102)
    d.obs_delta = (w->sdf_w && w->sdf_h) ? (w->world_w / (double)w->sdf_w + w->world_h / (double)w->sdf_h) / 2.0 : 1.0;
    d.inv_s2_obs = 1.0 / (w->p.sigma_obstacle * w->p.sigma_obstacle);    // FactorState::new, factor/mod.rs:631-632
    d.inv_s2_ir = 1.0 / (w->p.sigma_interrobot * w->p.sigma_interrobot);
    d.inv_s2_trk = 1.0 / (w->p.sigma_tracking * w->p.sigma_tracking);
    d.trk_pad = w->p.tracking_switch_padding;
    d.trk_attr = w->p.tracking_attraction_distance;
#ifdef MGX_STAMPS
    {
        // [.. * 16): stages per wave, then 16 sub-stage sums per wave; behind them the hand-off timeline, 16 segments x 4 per robot
        std::vector<unsigned long long> z((size_t)(R_local + 4) * 48 + (size_t)R_local * 64, 0ull);
        HIP_TRY(w->dbg.upload(z, s));
        HIP_TRY(hipStreamSynchronize(s));
        d.dbg = w->dbg.p;
    }
#else
    d.dbg = nullptr;
#endif
    w->dirty = false;
    w->conns_dirty = false;
    w->dev_valid = true;
    w->halo_dirty = true;
    w->peers_valid = false;
    w->sweep_flag_buf.n = 0;  // progress words of resident launches: re-created (zero) for the new device arrays
    w->flag_base = 0;
    w->resident_cap = w->resident_cap_sharded = -1;
    w->xres.connected = false;  // ghost slots and progress words belonged to the old layout: the ranks wire them again
    w->direct.aimed = false;    // ... and a slot-wired direct exchange names device indices and slots of the old layout
    d.gxrec[0] = d.gxrec[1] = nullptr; d.gflag = nullptr; d.xp_ptr = nullptr; d.xp_rec = nullptr;
    w->xres.agree = nullptr; d.agree = nullptr; d.n_ranks = 0;
    if (!w->sweep_err_host) {  // the word device code reports a wait that gave up in (resident launches, direct halo waits)
        HIP_TRY(hipHostMalloc((void **)&w->sweep_err_host, sizeof(unsigned long long), hipHostMallocMapped));
        *w->sweep_err_host = 0ull;
    }
    {
        void *dp = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&dp, w->sweep_err_host, 0));
        w->d.sweep_err = (unsigned long long *)dp;
    }
    return upload_flags(w);
}

// A wait on the device gave up — a direct halo exchange whose producer never published, or a resident schedule
// launch whose neighbour never did: the waiting kernels end (never a hung GPU) but what they computed from stale
// records is wrong.  Sticky: every later sweep, synchronisation and read-back reports it.
static int check_device_error(mgx_world *w) {
    if (w->sticky_rc != MGX_OK)
        return fail(w->sticky_rc, "a schedule that had to be run again launch by launch (its resident launch was declined) failed with code %d "
                    "inside a call that edits the graph: the world is behind the schedules issued", w->sticky_rc);
    if (w->sweep_err_host && *w->sweep_err_host != 0ull)
        return fail(MGX_ERR_STATE, "a wait on the device timed out (exchange / progress word %llu): a peer rank or a neighbouring "
                    "workgroup never published its records, the world's beliefs are invalid", *w->sweep_err_host);
    return MGX_OK;
}

// ---- launches -----------------------------------------------------------------------------------------
static int direct_exchange(mgx_world *w);
static int rccl_exchange(mgx_world *w);
static int sweep(mgx_world *w, int32_t robot, uint32_t ext_mask, uint32_t int_mask, int n_int, uint32_t hints = 0);
static void log_launch(mgx_world *w, int robot, uint32_t ext_mask, uint32_t int_mask, int n_int);

// A resident schedule launch decides for itself, before it writes anything, whether all its workgroups are on the device
// together (SegPlan: residency census) — and if another tenant of the GPU holds the CUs its tail needs, it returns at once
// and leaves the world untouched.  The host must not put anything behind a launch whose decision it has not seen (what
// follows would run on the wrong state), so every entry point that enqueues work or reads state comes through here first:
// the decision falls within microseconds of the launch's START, so in a stream of ticks the host simply stays ONE launch
// ahead of the device instead of many.  An aborted launch is run again on the launch-per-segment path, and the next few
// schedules skip the resident form (doubling back-off while the GPU stays shared).
static int confirm_resident(mgx_world *w, bool rerun, int32_t *outcome) {
    mgx_world::PendingResident &pd = w->pending;
    if (outcome) *outcome = MGX_RESIDENT_NONE;
    if (!pd.active) return MGX_OK;
    StageTimer clock("confirm");
    const double t0 = StageTimer::now();
    unsigned long long v = 0;
    for (unsigned spins = 0;; spins++) {
        v = __atomic_load_n(w->decision_host, __ATOMIC_ACQUIRE);
        if ((v >> 2) >= pd.seq) break;
        if ((spins & 0xfffffu) == 0xfffffu && StageTimer::now() - t0 > 30e6) {
            // 30 s: the launch never started (a stuck stream): nothing sensible is left to do.  (No runtime call in this loop:
            // a stream query per look made the runtime put markers between the launches.)
            pd.active = false;
            return fail(MGX_ERR_STATE, "resident launch %llu was never decided (the stream does not advance)", pd.seq);
        }
    }
    pd.active = false;
    clock.lap((v & 3ull) == RESIDENT_ABORT ? "launch ABORTED" : "launch decided: go");
    if ((v >> 2) == pd.seq && (v & 3ull) == RESIDENT_ABORT) {
        // nothing happened on the device: take the host's bookkeeping back and run the same schedule launch by launch
        w->resident_aborts++;
        if (w->linger.open && w->linger.seq0 == pd.seq) w->linger.open = false;  // (its decider left with the verdict: nobody lingers)
        w->resident_backoff_len = std::min(std::max(2 * w->resident_backoff_len, 64), 32768);
        w->resident_backoff = w->resident_backoff_len + (int)pd.segs.size();  // (+ this schedule's own re-run)
        w->d.cur = pd.cur_before;
        w->flag_base = pd.flag_base_before;
        w->last_sweep_launches = 0;
        if (!rerun && !pd.upd && !pd.partial) {  // mgx_resident_outcome: the caller issues the schedule again
            if (outcome) *outcome = MGX_RESIDENT_DECLINED;
            return MGX_OK;
        }
        if (outcome) *outcome = MGX_RESIDENT_RAN;  // (by the time the caller looks, it has: launch by launch)
        bool first = true;
        for (size_t k = 0; k < pd.segs.size(); k++) {
            if (first && pd.upd) { w->d.upd = pd.upd; w->d.upd_max_speed = pd.upd_max_speed; w->d.upd_delta_t = pd.upd_delta_t; }
            const int rc = sweep(w, -1, pd.segs[k].first, pd.segs[k].second ? (PH_INT_FACTOR | PH_INT_VARIABLE) : 0, pd.segs[k].second, pd.hints[k]);
            w->d.upd = nullptr;
            if (first && pd.upd && pd.upd_slot >= 0) {
                // the slot was released behind the declined launch, which returned at once: without this the ring would hand it
                // out again (or free it) while the kernel just enqueued has yet to read the prior updates
                const hipError_t e = w->stage.release(pd.upd_slot, w->stream);
                if (rc == MGX_OK && e != hipSuccess) return fail(MGX_ERR_HIP, "event record: %s", hipGetErrorString(e));
            }
            first = false;
            if (rc != MGX_OK) return rc;
        }
        return MGX_OK;
    }
    w->resident_backoff_len = 0;
    if (outcome) *outcome = MGX_RESIDENT_RAN;
    for (const auto &sg : pd.segs) log_launch(w, -1, sg.first, sg.second ? (PH_INT_FACTOR | PH_INT_VARIABLE) : 0, sg.second);
    return MGX_OK;
}
static int sweep(mgx_world *w, int32_t robot, uint32_t ext_mask, uint32_t int_mask, int n_int, uint32_t hints) {
    int rc = commit(w);
    if (rc != MGX_OK) return rc;
    if ((rc = check_device_error(w)) != MGX_OK) return rc;
    if (!w->robots.empty()) w->stale_kinds |= ~w->p.enable_mask & 15u;  // disabled factors miss what this sweep delivers
    const bool writes_snap = (int_mask & PH_INT_VARIABLE) && n_int > 0;
    if (robot < 0 && ext_mask && w->thaw_kinds && (int_mask & PH_INT_FACTOR) && n_int > 0) {
        // Factors that come back from being switched off take their first update in front of the sweep launch (k_thaw
        // writes their messages into the robots' images).  An external variable sweep at the head of the same launch
        // would sum those new messages where the reference still sums the stale ones — its beliefs are overwritten by
        // the internal sweep that follows, but the means it hands to the neighbours' factors are not.  So the external
        // iteration runs as a launch of its own first.
        rc = sweep(w, -1, ext_mask, 0, 0, 0);
        return rc != MGX_OK ? rc : sweep(w, -1, 0, int_mask, n_int, hints & ~HINT_IR_DEAD);
    }
    if (robot < 0) {
        if (ext_mask && w->resident_backoff > 0) w->resident_backoff--;
        if (w->direct.connected && (ext_mask & PH_EXT_FACTOR)) {  // the inter-robot factors read the ghosts' snapshots
            rc = direct_exchange(w);
            if (rc != MGX_OK) return rc;
        } else if (w->rccl.connected && (ext_mask & PH_EXT_FACTOR)) {
            rc = rccl_exchange(w);
            if (rc != MGX_OK) return rc;
        }
        const int out = writes_snap ? 1 - w->d.cur : -1;
        const bool thawing = w->thaw_kinds && (int_mask & PH_INT_FACTOR) && n_int > 0;
        if (thawing) HIP_TRY(launch_thaw(w->d, 0, w->d.R_local, ext_mask, w->stream));
        if (w->n_keyless > 0 && (ext_mask & PH_EXT_FACTOR) && (w->p.enable_mask & 2u)) {
            // factors created while their kind was switched off and not yet in possession of both inbox keys (KeylessRec):
            // the keys fill structurally, so the log says which ones are there now
            flush_counts(w);
            std::vector<KeylessRec> recs;
            for (size_t ci = 0; ci < w->conns.size(); ci++) {
                const IrConn &c = w->conns[ci];
                const int32_t dev_slot = w->conn_hot[ci].dev_slot;
                if (c.keys.empty() || dev_slot < 0) continue;
                const int tr = w->dev_of[(size_t)c.other];
                for (size_t j = 0; j < c.keys.size(); j++)
                    recs.push_back(KeylessRec{(int32_t)edge_index(w->dev_in_ptr, w->K, tr, (int)j, dev_slot), tr, (uint32_t)c.keys[j], 0u});
            }
            if (!recs.empty()) {
                void *hp = nullptr, *dp = nullptr;
                int slot = 0;
                HIP_TRY(w->stage.acquire(sizeof(KeylessRec) * recs.size(), &hp, &slot));
                memcpy(hp, recs.data(), sizeof(KeylessRec) * recs.size());
                HIP_TRY(hipHostGetDevicePointer(&dp, hp, 0));
                HIP_TRY(launch_keyless_ir(w->d, w->ir_gate.p, (int)recs.size(), (const KeylessRec *)dp, w->stream));
                HIP_TRY(w->stage.release(slot, w->stream));
                w->flags_dirty = true;  // the gate bytes go back to 0 / 1 in front of the next launch
            }
        }
        if (w->ir_thaw_active && (ext_mask & PH_EXT_FACTOR) && w->d.NI > 0 && !w->conns.empty())
            HIP_TRY(launch_thaw_ir(w->d, w->ir_gate.p, w->stream));
        HIP_TRY(launch_robot_sweep(w->d, 0, w->d.R_local, ext_mask, int_mask, n_int, out, hints, w->stream));
        w->last_sweep_launches++;
        if (w->ir_thaw_active && writes_snap) {  // once every robot has run an internal variable sweep, every owner has delivered
            bool all_take_part = true;  // ghosts count: their owners' flags are kept here too, and their records arrive by exchange
            for (const Robot &rb : w->robots) all_take_part = all_take_part && (rb.removed || !rb.idle);
            if (all_take_part) {
                w->ir_thaw_active = false;
                w->d.ir_frozen_snap = nullptr; w->d.ir_frozen_epoch = nullptr; w->d.ir_thaw_epoch = nullptr;
                for (Robot &rb : w->robots) rb.ir_thaw_epoch.clear();
                w->flags_dirty = true;  // gate bytes back to 0 / 1
            }
        }
        if (w->thaw_kinds && (thawing || writes_snap)) {
            HIP_TRY(launch_thaw_done(w->d, 0, w->d.R_local, writes_snap ? 1 : 0, w->stream));
            bool all_take_part = true;  // idle robots keep thawing until they iterate again
            for (const Robot &rb : w->robots) all_take_part = all_take_part && (rb.ghost || rb.removed || !rb.idle);
            if (writes_snap && all_take_part) { w->thaw_kinds = 0; w->d.skip0 = nullptr; }
        }
        if (writes_snap) w->d.cur ^= 1;
        log_launch(w, -1, ext_mask, int_mask, n_int);
    } else {
        if ((size_t)robot >= w->robots.size() || w->robots[(size_t)robot].ghost) return fail(MGX_ERR_INVALID, "bad robot id %d", robot);
        if (ext_mask) return fail(MGX_ERR_INVALID, "external sweeps are world-wide (robot must be -1)");
        // single workgroup: nobody else reads the snapshot buffer concurrently => update in place
        const bool thawing = w->thaw_kinds && (int_mask & PH_INT_FACTOR) && n_int > 0;
        if (thawing) HIP_TRY(launch_thaw(w->d, w->dev_of[(size_t)robot], 1, 0, w->stream));
        HIP_TRY(launch_robot_sweep(w->d, w->dev_of[(size_t)robot], 1, 0, int_mask, n_int, writes_snap ? w->d.cur : -1, 0, w->stream));
        if (w->thaw_kinds && (thawing || writes_snap)) HIP_TRY(launch_thaw_done(w->d, w->dev_of[(size_t)robot], 1, writes_snap ? 1 : 0, w->stream));
        log_launch(w, robot, 0, int_mask, n_int);
    }
    return MGX_OK;
}

// ---- resident schedule launches: a whole mgx_iterate / mgx_tick schedule in ONE launch -----------------------

static bool resident_enabled() {  // MGX_PERSISTENT=0 keeps every schedule on the launch-per-segment path
    static int v = -1;
    if (v < 0) { const char *e = getenv("MGX_PERSISTENT"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}
// a wait inside a resident launch gave up (a neighbour's workgroup never published): the beliefs are not to be trusted
// the robots each local robot exchanges snapshot records with: owners of its incoming connections and targets of
// its outgoing ones (the latter matter when the reference's bookkeeping has left a connection one-sided)
static int ensure_resident_tables(mgx_world *w) {
    hipStream_t s = w->stream;
    const size_t R = (size_t)w->d.R_local;
    if (!w->sweep_abort_buf.p) {
        std::vector<unsigned long long> z(1, 0ull);
        HIP_TRY(w->sweep_abort_buf.upload(z, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    if (w->sweep_flag_buf.n != R) {
        HIP_TRY(w->sweep_flag_buf.reserve(R));
        HIP_TRY(hipMemsetAsync(w->sweep_flag_buf.p, 0, sizeof(unsigned long long) * R, s));
        w->flag_base = 0;
        // ... and with the segment count the exchange records start over: zeroed, so that no sequence word of an earlier life
        // of the device arrays validates (a valid word has its top bit set, mgx_dev.h)
        const size_t xb = R * (size_t)w->K * (size_t)XREC_BYTES;
        HIP_TRY(w->xrec_buf.reserve(2 * xb));
        HIP_TRY(hipMemsetAsync(w->xrec_buf.p, 0, 2 * xb, s));
    }
    {
        const size_t xb = R * (size_t)w->K * (size_t)XREC_BYTES;
        w->d.xrec[0] = w->xrec_buf.p;
        w->d.xrec[1] = w->xrec_buf.p ? w->xrec_buf.p + xb : nullptr;
    }
    if (w->census_buf.n < R + 1) {  // residency census: one word per workgroup of a launch (never reset: monotonic in the launch number)
        std::vector<unsigned long long> z(R + 1 + R / 4 + 64, 0ull);
        HIP_TRY(w->census_buf.upload(z, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    if (!w->decision_buf.p) {  // the decision word and its host-mapped copy
        std::vector<unsigned long long> z1(1, 0ull);
        HIP_TRY(w->decision_buf.upload(z1, s));
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipHostMalloc((void **)&w->decision_host, sizeof(unsigned long long), hipHostMallocMapped));
        *w->decision_host = 0ull;
    }
    {
        void *dp = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&dp, w->decision_host, 0));
        w->d.census = w->census_buf.p;
        w->d.decision = w->decision_buf.p;
        w->d.decision_host = (unsigned long long *)dp;
    }
    if (!w->peers_valid) {
        // (lists of the LOCAL robots; a ghost — device index >= R — appears in them as a peer, its word lives in the ghost area.)
        // A list is what its robot's polling lanes walk, nothing more: its order carries no meaning and a robot that is both the
        // owner of an incoming and the target of an outgoing connection — the rule — may stand in it twice (two lanes look at
        // the same word).  So the table is two passes over the connections, written straight into the pinned block it travels
        // in ([R + 1 row pointers | entries], ONE copy): a world that follows its topology builds it every tick.
        const size_t n_conns = w->conns.size();
        const IrConn *conns = w->conns.data();
        const int32_t *dev_of = w->dev_of.data();
        void *hp = nullptr;
        int slot = 0;
        const size_t words = R + 1 + std::max<size_t>(2 * n_conns, 1);
        HIP_TRY(w->stage.acquire(sizeof(int32_t) * words, &hp, &slot));
        int32_t *ptr = (int32_t *)hp, *idx = ptr + R + 1;
        std::fill(ptr, ptr + R + 1, 0);
        for (size_t ci = 0; ci < n_conns; ci++) {
            const size_t o = (size_t)dev_of[(size_t)conns[ci].owner], t = (size_t)dev_of[(size_t)conns[ci].other];
            if (o < R) ptr[o + 1]++;
            if (t < R) ptr[t + 1]++;
        }
        for (size_t r = 0; r < R; r++) ptr[r + 1] += ptr[r];
        std::vector<int32_t> &fill = w->peer_fill;
        fill.assign(ptr, ptr + R);
        for (size_t ci = 0; ci < n_conns; ci++) {
            const int o = dev_of[(size_t)conns[ci].owner], t = dev_of[(size_t)conns[ci].other];
            if ((size_t)o < R) idx[(size_t)fill[(size_t)o]++] = t;
            if ((size_t)t < R) idx[(size_t)fill[(size_t)t]++] = o;
        }
        if (ptr[R] == 0) idx[0] = 0;
        HIP_TRY(w->peer_ptr_dev.reserve(words));
        HIP_TRY(hipMemcpyAsync(w->peer_ptr_dev.p, hp, sizeof(int32_t) * (R + 1 + (size_t)std::max(ptr[R], 1)), hipMemcpyHostToDevice, s));
        HIP_TRY(w->stage.release(slot, s));
        w->peer_idx_off = R + 1;
        w->peers_valid = true;
    }
    w->d.sweep_flag = w->sweep_flag_buf.p;
    w->d.sweep_abort = w->sweep_abort_buf.p;
    w->d.peer_ptr = w->peer_ptr_dev.p;
    w->d.peer_idx = w->peer_ptr_dev.p + w->peer_idx_off;
    return MGX_OK;
}
// Runs the schedule as resident launches if this world qualifies: 1 = done, 0 = not eligible (the caller takes the
// launch-per-segment path), negative = error.  Eligible: inter-robot factors enabled and staged in LDS, every robot
// local (no ghosts: their records arrive between launches), nothing thawing, and every workgroup co-resident.
// the conditions every rank of a sharded world decides alike on: same schedule, same world-wide switches, the same back-off
// (aborts are the ranks' common answer)
static bool resident_gate(const mgx_world *w, const std::vector<Launch> &plan) {
    if (!resident_enabled() || w->resident_off || plan.size() < 2) return false;
    if (w->resident_backoff > 0) return false;  // a recent launch found the GPU shared (residency census): launch by launch for a while
    const DevWorld &d = w->d;
    const bool sharded = w->xres.connected;  // the ranks have agreed (mgx_halo_resident_connect) that ghost records travel inside the launches
    if ((d.R_total != d.R_local && !sharded) || !(w->p.enable_mask & 2u)) return false;
    if (((w->direct.connected || w->rccl.connected) && !sharded)) return false;
    for (const Launch &l : plan)
        if (l.n_int > 255) return false;
    if (sharded && plan[0].ext && !w->direct.connected) return false;  // the exchange in front of the launch is the direct one
    return true;
}

// ---- lingering resident launches: the host's side (mgx_dev.h; the device's side is in mgx_sweep.h) --------------------------
static int run_resident(mgx_world *w, const std::vector<Launch> &plan);
static long long linger_ticks(mgx_world *w) {
    mgx_world::Linger &lg = w->linger;
    if (lg.ticks < 0) {
        const char *off = getenv("MGX_LINGER"), *us = getenv("MGX_LINGER_US");
        long long v = us ? atoll(us) : 300;  // microseconds a workgroup waits for the next schedule before it ends the launch
        if (off && off[0] == '0') v = 0;
        lg.ticks = (v > 0 ? std::min<long long>(v, 1000000) : 0) * 100ll;  // 100 MHz wall clock
    }
    return lg.ticks;
}
static double *linger_upd_slot(mgx_world *w, unsigned long long number) {
    mgx_world::Linger &lg = w->linger;
    return reinterpret_cast<double *>(reinterpret_cast<char *>(lg.box) + sizeof(LingerBox)) + (size_t)(number & 1ull) * lg.upd_stride;
}
// the box (host-mapped: header, two plan slots, two blocks of prior-update records) and the go word; never while a launch lingers
static int ensure_linger_box(mgx_world *w) {
    mgx_world::Linger &lg = w->linger;
    const size_t stride = ((size_t)4 * (size_t)std::max(w->d.R_local, 1) + 7) & ~(size_t)7;
    if (!lg.go.p) {
        std::vector<unsigned long long> z(1, 0ull);
        HIP_TRY(lg.go.upload(z, w->stream));
        HIP_TRY(hipStreamSynchronize(w->stream));
    }
    if (lg.box && lg.upd_stride >= stride) return MGX_OK;
    if (lg.box) { (void)hipHostFree(lg.box); lg.box = nullptr; }
    const size_t grown = (stride + stride / 2 + 7) & ~(size_t)7, bytes = sizeof(LingerBox) + 2 * grown * sizeof(double);
    HIP_TRY(hipHostMalloc((void **)&lg.box, bytes, hipHostMallocMapped));
    memset(lg.box, 0, bytes);
    lg.upd_stride = grown;
    lg.dev_stride = LINGER_SLOT_HEAD + (size_t)LINGER_UPD_BYTES * (grown / 4);  // the plan's chunks, then three chunks per robot (mgx_dev.h)
    HIP_TRY(lg.dev.reserve(2 * lg.dev_stride));
    HIP_TRY(hipMemsetAsync(lg.dev.p, 0, 2 * lg.dev_stride, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    return MGX_OK;
}
// Spins (bounded) until `pred` holds: 0; or until the launch has ended (its go word went odd and the postman said so): 1.
template <class F>
static int linger_wait(mgx_world *w, F pred, const char *what) {
    mgx_world::Linger &lg = w->linger;
    const double t0 = StageTimer::now();
    for (unsigned spins = 0;; spins++) {
        if (pred()) return 0;
        const unsigned long long c = __atomic_load_n(&lg.box->closed, __ATOMIC_ACQUIRE);
        if ((c & 1ull) && (c >> 1) >= lg.seq0) return 1;
        if ((spins & 0xfffffu) == 0xfffffu && StageTimer::now() - t0 > 30e6)
            return fail(MGX_ERR_STATE, "lingering launch %llu: no answer from the device while waiting for %s (the stream does not advance)", lg.seq0, what);
    }
}
// a post the launch has taken: it runs (or has run) behind everything before it — its launches enter the counters' log
static void linger_confirm(mgx_world *w) {
    mgx_world::Linger &lg = w->linger;
    for (const Launch &l : lg.un.plan) log_launch(w, -1, l.ext, l.n_int ? (PH_INT_FACTOR | PH_INT_VARIABLE) : 0, l.n_int);
    lg.un.active = false;
    lg.posts++;
    lg.taken_in_launch++;
}
// The launch has ended (closed word c = 2 S + 1: behind plan S).  A post it never took is taken back and run as a launch of its
// own — from the same records, nothing lost and nothing twice.
static int linger_settle(mgx_world *w, bool nested = false) {  // nested: on the way to the NEXT schedule (whose launches are counted apart)
    mgx_world::Linger &lg = w->linger;
    const uint32_t count_before = w->last_sweep_launches;
    const unsigned long long c = __atomic_load_n(&lg.box->closed, __ATOMIC_ACQUIRE);
    lg.open = false;
    int rc = MGX_OK;
    if (lg.un.active) {
        if ((c >> 1) >= lg.un.number) {
            linger_confirm(w);
        } else {
            mgx_world::Linger::Post un = std::move(lg.un);
            lg.un = mgx_world::Linger::Post{};
            lg.reruns++;
            w->d.cur = un.cur_before;
            w->flag_base = un.flag_base_before;
            w->last_sweep_launches = 0;  // (how the schedule that was posted ran after all: what mgx_last_launch_count says of it)
            void *hp = nullptr, *dp = nullptr;
            int slot = -1;
            if (un.has_upd) {  // (a copy in the pinned ring: the box's slot belongs to the posts of the launch that follows)
                const size_t bytes = 4 * (size_t)w->d.R_local * sizeof(double);
                HIP_TRY(w->stage.acquire(bytes, &hp, &slot));
                memcpy(hp, linger_upd_slot(w, un.number), bytes);
                HIP_TRY(hipHostGetDevicePointer(&dp, hp, 0));
            }
            bool first = true;
            auto with_upd = [&]() { if (first && un.has_upd) { w->d.upd = (const double *)dp; w->d.upd_max_speed = un.max_speed; w->d.upd_delta_t = un.delta_t; w->upd_ring_slot = slot; } };
            with_upd();
            w->upd_host = (const double *)hp;
            const int resident = run_resident(w, un.plan);
            w->upd_host = nullptr;
            w->upd_ring_slot = -1;
            w->d.upd = nullptr;
            if (resident < 0) rc = resident;
            if (resident == 0)
                for (const Launch &l : un.plan) {
                    with_upd();
                    rc = sweep(w, -1, l.ext, l.n_int ? (PH_INT_FACTOR | PH_INT_VARIABLE) : 0, l.n_int, l.hints);
                    w->d.upd = nullptr;
                    first = false;
                    if (rc != MGX_OK) break;
                }
            if (slot >= 0) {
                const hipError_t e = w->stage.release(slot, w->stream);
                if (rc == MGX_OK && e != hipSuccess) rc = fail(MGX_ERR_HIP, "event record: %s", hipGetErrorString(e));
            }
        }
    }
    if (nested) w->last_sweep_launches = count_before;
    if (lg.taken_in_launch == 0) lg.useless++;
    else lg.useless = 0;
    return rc;
}
// Ends the lingering launch: the postman turns the go word odd behind everything posted, every workgroup writes back as at the
// end of any launch.  What follows in the stream finds the world as after a plain launch.
static int linger_close(mgx_world *w) {
    mgx_world::Linger &lg = w->linger;
    if (!lg.open) return MGX_OK;
    if (w->pending.active) {  // (the launch's census: an aborted launch does not linger)
        const bool hold = lg.hold;
        lg.hold = true;
        const int rc = confirm_resident(w);
        lg.hold = hold;
        if (rc != MGX_OK) return rc;
        if (!lg.open) return MGX_OK;
    }
    StageTimer clock("linger");
    __atomic_store_n(&lg.box->close_req, lg.seq0, __ATOMIC_RELEASE);
    const int r = linger_wait(w, [] { return false; }, "the launch to end");
    if (r < 0) return r;
    clock.lap("closed");
    return linger_settle(w);
}
// On the way to post `plan` into the open launch: 1 = the slot of the next number may be written and posted; 0 = no launch
// lingers any more (it had ended, or the plan does not qualify and it was ended): the caller launches.
static bool linger_plan_fits(const std::vector<Launch> &plan) {
    if (plan.empty() || plan.size() > (size_t)MAX_SEGS || plan[0].ext) return false;  // a post CONTINUES the last segment of the plan before
    for (const Launch &l : plan)
        if (l.n_int > 255) return false;
    return true;
}
static int linger_prepare_post(mgx_world *w, const std::vector<Launch> &plan) {
    mgx_world::Linger &lg = w->linger;
    if (!lg.open) return 0;
    if (w->pending.active) {  // the launch's own census first (one launch of run-ahead, as ever)
        const uint32_t count_before = w->last_sweep_launches;  // (a declined launch is run again here: not the coming schedule's launches)
        lg.hold = true;
        const int rc = confirm_resident(w);
        lg.hold = false;
        w->last_sweep_launches = count_before;
        if (rc != MGX_OK) return rc;
        if (!lg.open) return 0;
    }
    const bool fits = linger_plan_fits(plan) && !w->dirty && !w->conns_dirty && !w->flags_dirty && !w->thaw_kinds && !w->ir_thaw_active &&
                      w->n_keyless == 0 && !w->resident_decline && !w->resident_off && (w->p.enable_mask & 2u) && w->resident_backoff == 0;
    if (!fits) {
        const int rc = linger_close(w);
        return rc != MGX_OK ? rc : 0;
    }
    int r = 0;
    if (lg.un.active) {  // the post before: taken?
        const unsigned long long n = lg.un.number;
        r = linger_wait(w, [&] { return __atomic_load_n(&lg.box->taken, __ATOMIC_ACQUIRE) >= n; }, "the last post to be taken");
        if (r == 0) linger_confirm(w);
    }
    // (the box's slot of the coming number held the post two before it, which the postman has copied to the device: `taken`)
    if (r < 0) return r;
    if (r == 1) {  // the launch ended meanwhile (its workgroups waited out their bound)
        lg.ended_by_device++;
        const int rc = linger_settle(w, true);
        return rc != MGX_OK ? rc : 0;
    }
    return 1;
}
// Posts `plan` (prepared: linger_prepare_post returned 1; the prior-update records, if any, are in the coming number's slot).
static int linger_post(mgx_world *w, const std::vector<Launch> &plan, bool has_upd, double max_speed, double delta_t) {
    mgx_world::Linger &lg = w->linger;
    const unsigned long long P = ++w->launch_seq;
    LingerPlan &lp = lg.box->plan[P & 1ull];
    lp.n = (uint32_t)plan.size();
    lp.has_upd = has_upd ? 1u : 0u;
    uint8_t ext[MAX_SEGS] = {}, n_int[MAX_SEGS] = {};
    for (size_t k = 0; k < plan.size(); k++) { ext[k] = plan[k].ext ? 1 : 0; n_int[k] = (uint8_t)plan[k].n_int; }
    memcpy(lp.ext, ext, sizeof ext);
    memcpy(lp.n_int, n_int, sizeof n_int);
    lp.upd_max_speed = max_speed;
    lp.upd_delta_t = delta_t;
    lp.number = P;
    lg.un.active = true;
    lg.un.number = P;
    lg.un.plan = plan;
    lg.un.has_upd = has_upd; lg.un.max_speed = max_speed; lg.un.delta_t = delta_t;
    lg.un.cur_before = w->d.cur;
    lg.un.flag_base_before = w->flag_base;
    __atomic_store_n(&lg.box->posted, P, __ATOMIC_RELEASE);
    // (segment 0 of the post continues the last segment of the plan before: one launch-wide index less than a launch of its own)
    w->d.cur = (w->d.cur + (int)plan.size() - 1) & 1;
    w->flag_base += (unsigned long long)plan.size() - 1ull;
    w->stale_kinds |= ~w->p.enable_mask & 15u;  // disabled factors miss what these sweeps deliver
    w->last_sweep_launches++;  // (one submission: the schedule runs inside the launch that is there)
    return MGX_OK;
}
static int run_resident(mgx_world *w, const std::vector<Launch> &plan) {
    if (w->linger.open) {
        // A launch lingers: the schedule is posted into it if it qualifies.  (mgx_tick comes here prepared, its records in the box
        // already — unless the launch it found had ended and the post taken back became THIS lingering launch: then the tick's
        // prior updates sit in the pinned ring, and are copied over; records in device memory, mgx_mission_tick_end's, cannot ride
        // in a post: the launch ends first.)
        if (w->d.upd && !w->upd_host) {
            const int rc = linger_close(w);
            if (rc != MGX_OK) return rc;
        } else {
            const int r = linger_prepare_post(w, plan);
            if (r < 0) return r;
            if (r == 1) {
                const bool has_upd = w->d.upd != nullptr;
                if (has_upd) memcpy(linger_upd_slot(w, w->launch_seq + 1ull), w->upd_host, 4 * (size_t)w->d.R_local * sizeof(double));
                (void)linger_post(w, plan, has_upd, w->d.upd_max_speed, w->d.upd_delta_t);
                return 1;
            }
        }
    }
    if (!resident_enabled() || w->resident_off || plan.size() < 2) return 0;
    int rc = commit(w);
    if (rc != MGX_OK) return rc;
    if (!resident_gate(w, plan)) return 0;
    StageTimer tr("resident");
    const DevWorld &d = w->d;
    const bool sharded = w->xres.connected;
    // What follows is this rank's own: where the ranks agree on every schedule (xres.agree) a rank
    // that cannot take part says so THERE — its launch is a single vote, and everybody takes the launch-by-launch path.
    const bool ranks_agree = sharded && w->xres.agree != nullptr;
    bool can = d.ir_max_edges > 0 && !w->conns.empty() && !w->thaw_kinds && !w->ir_thaw_active && w->n_keyless == 0 && !w->resident_decline;
    if (!can && !ranks_agree) return 0;
    if (can && sweep_lds_bytes(w->K, d.ir_max_edges, true) > sweep_resident_lds_max()) {
        if (!sharded) return 0;
        if (!ranks_agree) return fail(MGX_ERR_STATE, "resident launches were agreed on with the other ranks, but this rank's robots no longer fit LDS");
        can = false;
    }
    int &cap = sharded ? w->resident_cap_sharded : w->resident_cap;
    if (can) {
        if (cap < 0) cap = sweep_resident_capacity(d, sharded);
        if (d.R_local + 1 > cap) {  // (+ the residency census' decider workgroup: one slot kept free for it)
            if (!sharded) return 0;
            if (!ranks_agree && d.R_local > cap)
                return fail(MGX_ERR_STATE, "resident launches were agreed on with the other ranks, but only %d of this rank's %d workgroups "
                                           "are resident at once", cap, d.R_local);
            if (ranks_agree) can = false;
        }
    }
    tr.lap("gate + capacity");
    rc = ensure_resident_tables(w);
    if (rc != MGX_OK) return rc;
    tr.lap("peer tables");
    static const long long timeout_ticks = [] {
        const char *e = getenv("MGX_RESIDENT_TIMEOUT_MS");
        const long long ms = e ? atoll(e) : 2000;
        return (ms > 0 ? ms : 2000) * 100000ll;  // 100 MHz wall clock
    }();
    w->stale_kinds |= ~w->p.enable_mask & 15u;  // disabled factors miss what these sweeps deliver
    for (size_t i0 = 0; i0 < plan.size(); i0 += MAX_SEGS) {
        SegPlan sp{};
        sp.n = (int32_t)std::min<size_t>(MAX_SEGS, plan.size() - i0);
        for (int k = 0; k < sp.n; k++) {
            const Launch &l = plan[i0 + (size_t)k];
            sp.ext[k] = l.ext ? 1 : 0;
            sp.n_int[k] = (uint8_t)l.n_int;
        }
        sp.flag_base = w->flag_base;
        sp.timeout_ticks = timeout_ticks;
        // residency census + clean abort (SegPlan); the ranks of a sharded world abort together, on the word they agree on
        // (without one — mgx_halo_resident_connect without a coordinator — they keep the plain bound on every wait)
        static const long long census_ticks = [] {
            const char *e = getenv("MGX_RESIDENT_CENSUS_US");
            const long long us = e ? atoll(e) : 200;
            return (us > 0 ? us : 200) * 100ll;  // 100 MHz wall clock
        }();
        static const long long census_ticks_sharded = [] {  // the ranks' hosts do not launch at the same instant
            const char *e = getenv("MGX_RESIDENT_CENSUS_SHARDED_US");
            const long long us = e ? atoll(e) : 20000;
            return (us > 0 ? us : 20000) * 100ll;
        }();
        static const bool census_on = [] { const char *e = getenv("MGX_RESIDENT_CENSUS"); return !(e && e[0] == '0'); }();
        const bool census = sharded ? ranks_agree : census_on;  // MGX_RESIDENT_CENSUS=0: plain bound on every wait
        if (census) {
            if (i0 > 0 && (rc = confirm_resident(w)) != MGX_OK) return rc;  // the previous part of this schedule
            if (i0 > 0 && w->resident_backoff > 0) {  // ... was sent back: the rest follows it launch by launch
                for (size_t i = i0; i < plan.size(); i++)
                    if ((rc = sweep(w, -1, plan[i].ext, plan[i].n_int ? (PH_INT_FACTOR | PH_INT_VARIABLE) : 0, plan[i].n_int, plan[i].hints)) != MGX_OK) return rc;
                return 1;
            }
            sp.launch_seq = ++w->launch_seq;
            sp.census_ticks = sharded ? census_ticks_sharded : census_ticks;
            if (ranks_agree) sp.agree_seq = ++w->xres.agree_seq;
        }
        // Lingering (mgx_dev.h): the launch that runs the END of the schedule stays for the schedules that follow — when the caller's
        // pattern promises some (schedules back to back, or no evidence yet that they are not: two lingering launches in a row that
        // ended without a post switch it off until schedules come back to back again)
        if (census && can && !sharded && i0 + (size_t)MAX_SEGS >= plan.size() && linger_ticks(w) > 0 &&
            (w->linger.useless < 2 || w->linger.streak >= 2)) {
            if ((rc = ensure_linger_box(w)) != MGX_OK) return rc;
            void *bd = nullptr;
            HIP_TRY(hipHostGetDevicePointer(&bd, w->linger.box, 0));
            sp.linger_ticks = linger_ticks(w);
            sp.linger_box = (const LingerBox *)bd;
            sp.linger_upd = reinterpret_cast<const double *>(reinterpret_cast<const char *>(bd) + sizeof(LingerBox));
            sp.linger_upd_stride = (unsigned long long)w->linger.upd_stride;
            sp.linger_dev = w->linger.dev.p;
            sp.linger_dev_stride = (unsigned long long)w->linger.dev_stride;
            sp.linger_go = w->linger.go.p;
        }
        if (sharded && sp.ext[0]) {  // segment 0 reads the ghosts' plain copies: one direct exchange in front of the launch
            rc = direct_exchange(w);
            if (rc != MGX_OK) return rc;
        }
        // MGX_COOPERATIVE=1: hipLaunchCooperativeKernel — the runtime checks the grid against the occupancy query at launch
        // time (same residency as a plain launch, +15..19 us of host time per launch: MI355X_MICROARCH.md); a grid it turns
        // down takes the launch-per-segment path from now on instead of waiting for workgroups that never become resident
        static const bool cooperative = [] { const char *e = getenv("MGX_COOPERATIVE"); return e && e[0] == '1'; }();
        const hipError_t le = can ? launch_robot_schedule(w->d, w->d.R_local, sp, sharded, cooperative, w->stream)
                                  : launch_agree_abort(w->d, sp, w->stream);
        if (le != hipSuccess) {
            (void)hipGetLastError();
            if (cooperative && le == hipErrorCooperativeLaunchTooLarge && i0 == 0 && !sharded) {
                cap = 0;  // until the topology (hence the workgroup's LDS) changes
                return 0;
            }
            return fail(MGX_ERR_HIP, "resident schedule launch: %s", hipGetErrorString(le));
        }
        w->last_sweep_launches++;
        w->resident_launches++;
        if (sp.linger_ticks > 0) {
            mgx_world::Linger &lg = w->linger;
            lg.open = true;
            lg.seq0 = sp.launch_seq;
            lg.taken_in_launch = 0;
            lg.un.active = false;
            lg.launches++;
        }
        if (census) {  // what confirm_resident needs to take the launch back and run it again launch by launch
            mgx_world::PendingResident &pd = w->pending;
            pd.active = true;
            pd.seq = sp.launch_seq;
            pd.partial = i0 > 0;
            pd.segs.clear();
            pd.hints.clear();
            for (int k = 0; k < sp.n; k++) {
                const Launch &l = plan[i0 + (size_t)k];
                pd.segs.emplace_back(l.ext, l.n_int);
                pd.hints.push_back(l.hints);
            }
            pd.cur_before = w->d.cur;
            pd.flag_base_before = w->flag_base;
            pd.upd = w->d.upd; pd.upd_max_speed = w->d.upd_max_speed; pd.upd_delta_t = w->d.upd_delta_t;
            pd.upd_slot = w->d.upd ? w->upd_ring_slot : -1;
        }
        w->d.upd = nullptr;  // mgx_tick's prior updates ride in the first launch only
        w->d.cur = (w->d.cur + sp.n) & 1;
        w->flag_base += (unsigned long long)sp.n;
        if (!census)
            for (int k = 0; k < sp.n; k++) {
                const Launch &l = plan[i0 + (size_t)k];
                log_launch(w, -1, l.ext, l.n_int ? (PH_INT_FACTOR | PH_INT_VARIABLE) : 0, l.n_int);
            }
    }
    return 1;
}

// =========================================================================================================
//                                            C ABI
// =========================================================================================================
extern "C" {

const char *mgx_last_error(void) { return g_err.c_str(); }
// for the other translation units of the library (mgx_linalg.cpp): same thread-local text
int mgx_set_error_(int code, const char *text) {
    g_err = text ? text : "";
    return code;
}

int mgx_world_create(const mgx_params *params, mgx_world **out) {
    if (!params || !out) return fail(MGX_ERR_INVALID, "null argument");
    if (!(params->sigma_dynamics > 0) || !(params->sigma_interrobot > 0) || !(params->sigma_obstacle > 0) ||
        !(params->sigma_tracking > 0) || !(params->safety_multiplier > 0))
        return fail(MGX_ERR_INVALID, "sigmas and safety multiplier must be positive");
    if (!device_ok()) return fail(MGX_ERR_NO_DEVICE, "no usable HIP device (this library has no CPU path)");
    mgx_world *w = new (std::nothrow) mgx_world();
    if (!w) return fail(MGX_ERR_NOMEM, "out of memory");
    w->p = *params;
    w->trk_ever_on = (params->enable_mask & 8u) != 0;
    *out = w;
    return MGX_OK;
}

int mgx_world_destroy(mgx_world *w) {
    if (!w) return MGX_OK;
    w->batch.steps.clear();
    if (w->linger.open) (void)linger_close(w);  // (nobody is left to post: the launch would wait out its bound)
    if (w->dev_valid) (void)hipStreamSynchronize(w->stream);
    if (w->linger.box) (void)hipHostFree(w->linger.box);
    if (w->rccl.comm && g_rccl.ok) (void)g_rccl.comm_destroy(w->rccl.comm);
    if (w->direct.recv) (void)hipFree(w->direct.recv);
    if (w->direct.flags) (void)hipFree(w->direct.flags);
    if (w->xres.area) (void)hipFree(w->xres.area);
    if (w->search_stream) { (void)hipStreamSynchronize(w->search_stream); (void)hipStreamDestroy(w->search_stream); }
    if (w->decision_host) (void)hipHostFree(w->decision_host);
    if (w->sweep_err_host) (void)hipHostFree(w->sweep_err_host);
    if (w->mission.ev_host) (void)hipHostFree(w->mission.ev_host);
    if (w->mission.tr_host) (void)hipHostFree(w->mission.tr_host);
    delete w;
    return MGX_OK;
}

int mgx_set_stream(mgx_world *w, void *hip_stream) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    if (w->dev_valid) HIP_TRY(hipStreamSynchronize(w->stream));
    w->stream = (hipStream_t)hip_stream;
    return MGX_OK;
}

int mgx_synchronize(mgx_world *w) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    if (!device_ok()) return fail(MGX_ERR_NO_DEVICE, "no usable HIP device");
    if (w->pending.active) { const int rcc = confirm_resident(w); if (rcc != MGX_OK) return rcc; }
    HIP_TRY(hipStreamSynchronize(w->stream));
    return check_device_error(w);
}

// Everything issued so far is ENQUEUED: recorded schedules are submitted and a lingering launch is told to end (it writes back and
// leaves the stream to what the caller puts behind it) — without waiting for the stream.
int mgx_flush(mgx_world *w) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    return MGX_OK;
}
int mgx_set_linger(mgx_world *w, int32_t microseconds) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    w->linger.ticks = -1;
    if (microseconds >= 0) w->linger.ticks = (long long)std::min(microseconds, 1000000) * 100ll;
    w->linger.useless = 0;
    return MGX_OK;
}
int mgx_linger_stats(mgx_world *w, uint64_t *launches, uint64_t *posts, uint64_t *reruns, uint64_t *ended_by_device) {
    MGX_ENTER_SCHEDULE(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    const mgx_world::Linger &lg = w->linger;
    if (launches) *launches = lg.launches;
    if (posts) *posts = lg.posts + (lg.un.active ? 1u : 0u);
    if (reruns) *reruns = lg.reruns;
    if (ended_by_device) *ended_by_device = lg.ended_by_device;
    return MGX_OK;
}

int mgx_world_set_sdf(mgx_world *w, const uint8_t *rgb, uint32_t width, uint32_t height, double world_w, double world_h) {
    MGX_ENTER(w);
    if (!w || !rgb || !width || !height || !(world_w > 0) || !(world_h > 0)) return fail(MGX_ERR_INVALID, "bad sdf arguments");
    w->sdf_red.resize((size_t)width * height);
    for (size_t i = 0; i < w->sdf_red.size(); i++) w->sdf_red[i] = rgb[3 * i];  // pixel[0], obstacle.rs:178
    w->sdf_w = width; w->sdf_h = height; w->world_w = world_w; w->world_h = world_h;
    w->dirty = true;
    return MGX_OK;
}

int mgx_world_set_environment(mgx_world *w, const mgx_env_desc *env) {
    MGX_ENTER(w);
    if (!w || !env) return fail(MGX_ERR_INVALID, "null argument");
    uint32_t W = 0, H = 0;
    const int rc = env_red_plane(env, env->sdf_resolution, env->sdf_expansion, env->sdf_blur, true, w->stream, w->sdf_red, W, H);
    if (rc != MGX_OK) return rc;
    w->sdf_w = W; w->sdf_h = H;
    w->world_w = (double)env->tile_size * (double)env->n_cols;  // robot.rs:1259-1264
    w->world_h = (double)env->tile_size * (double)env->n_rows;
    w->dirty = true;
    return MGX_OK;
}

int mgx_robot_add(mgx_world *w, const mgx_robot_desc *d, int32_t *robot_id) {
    MGX_ENTER(w);
    if (!w || !d || !d->mean0 || !d->prior_diag || !d->dt) return fail(MGX_ERR_INVALID, "null argument");
    if (d->K < 3) return fail(MGX_ERR_INVALID, "K must be >= 3");
    if (w->K && (int)d->K != w->K) return fail(MGX_ERR_INVALID, "all robots of a world share K (%d), got %u", w->K, d->K);
    if (!(d->radius > 0)) return fail(MGX_ERR_INVALID, "radius must be positive");
    for (const Robot &o : w->robots)
        if (o.order_key == d->order_key) return fail(MGX_ERR_INVALID, "duplicate order_key");
    const int K = (int)d->K, E = 4 * K - 6;
    Robot rb;
    rb.K = K; rb.ghost = d->ghost != 0; rb.radius = d->radius; rb.order_key = d->order_key;
    rb.n_nodes = K + (K - 1) + 2 * (K - 2);
    flush_counts(w);  // launches logged so far do not concern the new robot
    {   // add_internal_edge (factorgraph.rs:304-330): the variable receives an (empty) message, the factor one if enabled
        const uint32_t en = w->p.enable_mask;
        rb.cnt[2] = (uint64_t)(2 * (K - 1) + 2 * (K - 2)) + ((en & 1u) ? 2 * (K - 1) : 0) + ((en & 4u) ? K - 2 : 0) + ((en & 8u) ? K - 2 : 0);
    }
    rb.prior_eta.assign(4 * K, 0.0); rb.prior_lam.assign(16 * K, 0.0);
    rb.bel_eta.assign(4 * K, 0.0); rb.bel_lam.assign(16 * K, 0.0); rb.bel_mu.assign(4 * K, 0.0); rb.bel_cov.assign(16 * K, 0.0);
    rb.valid.assign(K, 1); rb.snap.assign(24 * K, 0.0); rb.epoch.assign(K, 0);
    rb.fv_eta.assign(4 * E, 0.0); rb.fv_lam.assign(16 * E, 0.0); rb.dyn_m.assign(16 * (K - 1), 0.0);
    rb.trk_record.assign(K - 2, 0); rb.trk_last_pos.assign(2 * (K - 2), 0.f); rb.trk_last_val.assign(K - 2, 0.0);
    for (int i = 0; i < K; i++) {  // VariableNode::new, variable.rs:140-166
        double pd = d->prior_diag[i];
        if (!std::isfinite(pd)) pd = 0.0;  // :146-148
        double lam[16] = {0}, cov[16] = {0};
        for (int a = 0; a < 4; a++) lam[a * 5] = pd;
        const double *m = d->mean0 + 4 * i;
        for (int a = 0; a < 4; a++) {
            double s = 0.0;
            for (int b = 0; b < 4; b++) s += lam[a * 4 + b] * m[b];
            rb.prior_eta[4 * i + a] = s;
            rb.bel_eta[4 * i + a] = s;
            rb.bel_mu[4 * i + a] = m[a];
        }
        memcpy(&rb.prior_lam[16 * i], lam, sizeof lam);
        memcpy(&rb.bel_lam[16 * i], lam, sizeof lam);
        if (!inv4(lam, cov)) memset(cov, 0, sizeof cov);  // :152-154
        memcpy(&rb.bel_cov[16 * i], cov, sizeof cov);
        bool fin = true;
        for (double c : cov) fin = fin && std::isfinite(c);
        rb.valid[i] = fin;
        // what prepare_message() would send (variable.rs:234-240): seeds the tracking factor inbox
        // (factorgraph.rs:315-317); all other inboxes start empty (epoch 0)
        memcpy(&rb.snap[24 * i], &rb.bel_eta[4 * i], 4 * sizeof(double));
        memcpy(&rb.snap[24 * i + 4], lam, sizeof lam);
        memcpy(&rb.snap[24 * i + 20], m, 4 * sizeof(double));
    }
    for (int f = 0; f < K - 1; f++) {
        if (!(d->dt[f] > 0)) return fail(MGX_ERR_INVALID, "dt must be positive");
        dynamic_potential(d->dt[f], w->p.sigma_dynamics, &rb.dyn_m[16 * f]);
    }
    for (int j = 0; j < K - 2; j++) {  // new_tracking_factor, factor/mod.rs:269-274
        rb.trk_last_pos[2 * j] = (float)d->mean0[4 * (j + 1)];
        rb.trk_last_pos[2 * j + 1] = (float)d->mean0[4 * (j + 1) + 1];
    }
    if (d->n_path && d->path_xy) rb.path.assign(d->path_xy, d->path_xy + 2 * (size_t)d->n_path);
    w->sets.keys.push_back(rb.order_key);
    w->sets.ghost.push_back(rb.ghost ? 1 : 0);
    w->sets.radius.push_back(rb.radius);
    w->sets.removed.push_back(0);
    w->robots.push_back(std::move(rb));
    w->sets.ensure(w->robots.size());
    w->K = K;
    w->dirty = true;
    if (robot_id) *robot_id = (int32_t)w->robots.size() - 1;
    return MGX_OK;
}

static int ir_connect(mgx_world *w, int32_t owner, int32_t other, uint64_t first_robot_number) {
    if (!w || owner < 0 || other < 0 || (size_t)owner >= w->robots.size() || (size_t)other >= w->robots.size() || owner == other)
        return fail(MGX_ERR_INVALID, "bad robot ids");
    if (first_robot_number == 0) return fail(MGX_ERR_INVALID, "robot_number is NonZeroUsize");
    // no "already connected" check: the reference creates whatever its connection sets ask for, and
    // a pair can legitimately hold two sets of factors (robot.rs:1391-1404, see mgx_update_topology)
    flush_counts(w, true);  // (the robots' counters: what they answer changes with the connection; the other connections' can wait)
    IrConn c;
    c.owner = owner; c.other = other; c.first_number = first_robot_number;
    c.base[0] = w->cum.nIv[(size_t)owner]; c.base[1] = w->cum.nEv[(size_t)other]; c.base[2] = w->cum.nEf[(size_t)owner];
    c.base[3] = w->cum.on_ir[(size_t)owner]; c.base[4] = w->cum.on_ir[(size_t)other];
    c.edges.resize((size_t)w->K - 1);
    c.node.resize((size_t)w->K - 1);
    // add_internal_edge + add_external_edge + the other variable's belief into the new factor
    // (factorgraph.rs:304-353, robot.rs:1557-1585)
    w->robots[(size_t)owner].cnt[2] += (uint64_t)(w->K - 1);
    w->robots[(size_t)other].cnt[3] += (uint64_t)(w->K - 1);
    if (w->p.enable_mask & 2u) { c.cnt[2] = (uint64_t)(w->K - 1); c.cnt[3] = (uint64_t)(w->K - 1); }
    const bool keyless = !(w->p.enable_mask & 2u);  // created switched off: the two inbox-filling messages are dropped
    for (int &nd : c.node) {  // add_factor, ascending i
        Robot &ow = w->robots[(size_t)owner];
        nd = ow.alloc_node();
        if (ow.slot_uses.size() <= (size_t)nd) ow.slot_uses.resize((size_t)nd + 1, 0);
        const uint32_t u = ++ow.slot_uses[(size_t)nd];
        c.updates_per_sweep += u;
        if (keyless) { c.keys.push_back(0); c.uses.push_back(u); }
    }
    if (keyless) w->n_keyless++;
    c.node_first = c.node.front();
    c.node_last = c.node.back();
    w->conn_hot.push_back(mgx_world::ConnHot{c.owner, c.other, c.node_first, c.node_last, c.first_number, -1, 1});
    w->conns.push_back(std::move(c));
    if (w->cidx.valid && w->cidx.in.size() == w->robots.size()) conn_index_add(w, (int32_t)w->conn_hot.size() - 1);
    else w->cidx.valid = false;
    w->conns_dirty = true;
    return MGX_OK;
}

static int ir_disconnect(mgx_world *w, int32_t a, int32_t b) {
    if (!w || a < 0 || b < 0 || (size_t)a >= w->robots.size() || (size_t)b >= w->robots.size() || a == b)
        return fail(MGX_ERR_INVALID, "bad robot ids");
    flush_counts(w);  // the deleted factors take their counts with them (factorgraph.rs:876-890)
    // delete_interrobot_factors_connected_to on both graphs (factorgraph.rs:380-436): the node
    // slots are vacated in ascending index order
    for (int side = 0; side < 2; side++) {
        const int self = side ? b : a, other = side ? a : b;
        std::vector<int> gone;
        for (const IrConn &c : w->conns)
            if (c.owner == self && c.other == other) gone.insert(gone.end(), c.node.begin(), c.node.end());
        std::sort(gone.begin(), gone.end());
        std::vector<int> &fr = w->robots[(size_t)self].free_nodes;
        fr.insert(fr.end(), gone.begin(), gone.end());
    }
    w->cidx.valid = false;  // (the list closes up: every index behind the holes moves)
    w->conns.erase(std::remove_if(w->conns.begin(), w->conns.end(),
                                  [&](const IrConn &c) { return (c.owner == a && c.other == b) || (c.owner == b && c.other == a); }),
                   w->conns.end());
    w->conn_hot.erase(std::remove_if(w->conn_hot.begin(), w->conn_hot.end(),
                                     [&](const mgx_world::ConnHot &c) { return (c.owner == a && c.other == b) || (c.owner == b && c.other == a); }),
                      w->conn_hot.end());
    if (w->n_keyless > 0) {  // some of them may just have gone
        w->n_keyless = 0;
        for (const IrConn &c : w->conns) w->n_keyless += c.keys.empty() ? 0 : 1;
    }
    w->conns_dirty = true;  // the surviving connections' state stays on the device
    return MGX_OK;
}

// Several (a, b) deletions in one sweep over the connections (a topology pass deletes dozens):
// same effect as ir_disconnect(a, b) for each pair in order.
// the connections listed by owner (what a batch of deletions looks its pairs up in)
static void ir_disconnect_batch(mgx_world *w, const std::vector<std::pair<int, int>> &pairs) {
    if (pairs.empty()) return;
    StageTimer tm("ir_disconnect_batch");
    flush_counts(w, true);
    tm.lap("flush_counts");
    conn_index_ensure(w);  // (who owns what: the index' `out` lists)
    std::vector<uint8_t> dead(w->conns.size(), 0);
    std::vector<int32_t> dead_list;
    for (const auto &pr : pairs)
        for (int side = 0; side < 2; side++) {
            const int self = side ? pr.second : pr.first, other = side ? pr.first : pr.second;
            std::vector<int> gone;  // node slots are vacated in ascending index order (factorgraph.rs:380-436)
            for (const int32_t ci : w->cidx.out[(size_t)self]) {
                const IrConn &c = w->conns[(size_t)ci];
                if (dead[(size_t)ci] || c.other != other) continue;
                dead[(size_t)ci] = 1;
                dead_list.push_back(ci);
                gone.insert(gone.end(), c.node.begin(), c.node.end());
            }
            std::sort(gone.begin(), gone.end());
            std::vector<int> &fr = w->robots[(size_t)self].free_nodes;
            fr.insert(fr.end(), gone.begin(), gone.end());
        }
    for (const int32_t ci : dead_list) {
        settle_conn(w, w->conns[(size_t)ci]);  // (what it delivered to its target stays in that robot's counters; its own go with it)
        conn_index_drop(w, ci);
    }
    // the list's order carries no meaning (inbox order comes from order keys and node slots, build_incoming): the last
    // survivors fill the holes
    {
        size_t lo = 0, hi = w->conns.size();
        for (;;) {
            while (lo < hi && !dead[lo]) lo++;
            while (hi > lo && dead[hi - 1]) hi--;
            if (lo >= hi) break;
            w->conns[lo] = std::move(w->conns[hi - 1]);  // dead[lo], alive[hi - 1]
            w->conn_hot[lo] = w->conn_hot[hi - 1];
            conn_index_moved(w, (int32_t)(hi - 1), (int32_t)lo);
            dead[lo] = 0;
            hi--;
        }
        w->conns.resize(hi);
        w->conn_hot.resize(hi);
    }
    tm.lap("connection list");
    if (w->n_keyless > 0) {  // some of them may just have gone
        w->n_keyless = 0;
        for (const IrConn &c : w->conns) w->n_keyless += c.keys.empty() ? 0 : 1;
    }
    w->conns_dirty = true;
}

// Entity despawn (robot.rs:2172 + despawn_entity_after): the graph leaves every Bevy query, so it
// is never iterated again and whatever is addressed to it is dropped (robot.rs:1815,1844: the
// `query.get_mut` fails) — the same dataflow as idle with the antenna off, for good.  The other
// robots drop their factors towards it in the following topology passes.
int mgx_robot_remove(mgx_world *w, int32_t robot) {
    MGX_ENTER(w);
    if (!w || robot < 0 || (size_t)robot >= w->robots.size()) return fail(MGX_ERR_INVALID, "bad robot id");
    Robot &rb = w->robots[(size_t)robot];
    if (rb.removed) return fail(MGX_ERR_STATE, "robot %d already removed", robot);
    flush_counts(w);
    rb.removed = true;
    w->sets.removed[(size_t)robot] = 1;
    rb.idle = 1;
    rb.antenna = 0;
    w->sets.cnt[(size_t)robot] = 0;
    w->flags_dirty = true;
    w->mission.alive_dirty = true;
    return MGX_OK;
}

// FactorGraph::change_factor_enabled for every graph (factorgraph.rs:1529-1539, ui/settings.rs:491-496).
// Disabling is exact as it stands: a disabled factor is never updated (its last message stays in the
// variable's inbox and keeps being summed, factorgraph.rs:695,734) and drops whatever is sent to it
// (FactorNode::receive_message_from returns early, factor/mod.rs:307-310) while change_prior still
// empties the variables' inboxes (variable.rs:224-227); the counters stop counting it.  Re-enabling a
// kind that has missed deliveries would need the inbox its factors froze with (the engine derives factor
// inboxes from the variables' current snapshots) and is refused with MGX_ERR_STATE.
static int ensure_frozen(mgx_world *w) {
    if (w->frozen_live) return MGX_OK;
    const size_t RL = (size_t)std::max(w->d.R_local, 1);
    std::vector<double> z((size_t)frozen_words(w->K) * RL, 0.0);
    std::vector<uint8_t> zf((size_t)(4 * w->K - 6) * RL, 0), zb(RL, 0);
    HIP_TRY(w->frozen_buf.upload(z, w->stream));
    HIP_TRY(w->frozen_flag_buf.upload(zf, w->stream));
    HIP_TRY(w->thaw_buf.upload(zb, w->stream));
    HIP_TRY(w->skip0_buf.upload(zb, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    w->frozen_live = true;
    w->d.frozen = w->frozen_buf.p; w->d.frozen_flag = w->frozen_flag_buf.p; w->d.thaw = w->thaw_buf.p;
    return MGX_OK;
}

int mgx_set_enabled(mgx_world *w, uint32_t kind_mask) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    if (kind_mask & ~15u) return fail(MGX_ERR_INVALID, "unknown factor kind bits 0x%x", kind_mask);
    const uint32_t on = kind_mask & ~w->p.enable_mask, off = w->p.enable_mask & ~kind_mask;
    if (kind_mask == w->p.enable_mask) return MGX_OK;
    flush_counts(w);  // what was logged so far was sent under the old flags
    // Internal kinds (dynamic, obstacle, tracking): a factor switched off keeps the inbox it has now and
    // resumes from it when it is switched on again (FactorNode::receive_message_from drops everything in
    // between, factor/mod.rs:307-310).  The engine derives factor inboxes from the variables' snapshots, so
    // the inboxes are materialised here (k_freeze) and consumed by k_thaw in front of the first factor sweep.
    const uint32_t off_int = off & 13u, thaw_int = on & 13u & w->stale_kinds;
    if ((off_int || thaw_int) && !w->robots.empty()) {
        int rc = commit(w);  // the device holds the state the inboxes are derived from
        if (rc != MGX_OK) return rc;
        rc = ensure_frozen(w);
        if (rc != MGX_OK) return rc;
        if (off_int) {
            HIP_TRY(launch_freeze(w->d, off_int, w->stream));
            HIP_TRY(launch_or_bytes(w->thaw_buf.p, w->d.R_local, (uint8_t)~off_int, 0, w->stream));  // off again: nothing to thaw
        }
        if (thaw_int) {
            HIP_TRY(launch_or_bytes(w->thaw_buf.p, w->d.R_local, 0xff, (uint8_t)thaw_int, w->stream));
            w->thaw_kinds |= thaw_int;
            w->d.skip0 = w->skip0_buf.p;
        }
    }
    // Inter-robot factors: F_AB's inbox entry from A's variable is what that variable last sent to its own
    // factors; it is kept per variable when the kind goes off (k_ir_freeze) and used by k_thaw_ir, in front of
    // launches with an external factor sweep, for as long as the variable has not delivered again.
    if (((off & 2u) || ((on & 2u) & w->stale_kinds)) && !w->robots.empty()) {
        int rc = commit(w);
        if (rc != MGX_OK) return rc;
        // A sharded world with the exchange inside the engine: the records frozen here and the delivery counts thawed against
        // include the ghosts', which must be their owners' CURRENT ones — the plain copies are as old as the last exchange
        // kernel (resident launches never touch them: the ghosts' records travel inside those).  Every rank switches
        // together (ShardedWorld.set_enabled), so the exchange is one all ranks take part in.
        if (w->direct.connected) {
            rc = direct_exchange(w);
            if (rc != MGX_OK) return rc;
        } else if (w->rccl.connected) {
            rc = rccl_exchange(w);
            if (rc != MGX_OK) return rc;
        }
        const size_t V = (size_t)w->d.V;
        if (!w->ir_frozen_live) {  // never frozen before: nothing was ever received (kind off since the world began)
            std::vector<double> z(24 * V, 0.0);
            std::vector<uint32_t> ze(V, 0);
            HIP_TRY(w->ir_frozen_snap_buf.upload(z, w->stream));
            HIP_TRY(w->ir_frozen_epoch_buf.upload(ze, w->stream));
            HIP_TRY(w->ir_thaw_epoch_buf.upload(ze, w->stream));
            HIP_TRY(hipStreamSynchronize(w->stream));
            w->ir_frozen_live = true;
        }
        if (off & 2u) {
            HIP_TRY(launch_ir_freeze(w->d, w->ir_frozen_snap_buf.p, w->ir_frozen_epoch_buf.p, w->stream));
            w->ir_thaw_active = false;
            w->d.ir_frozen_snap = nullptr; w->d.ir_frozen_epoch = nullptr; w->d.ir_thaw_epoch = nullptr;
            w->flags_dirty = true;
        } else {
            HIP_TRY(hipMemcpyAsync(w->ir_thaw_epoch_buf.p, w->d.snap_epoch[w->d.cur], sizeof(uint32_t) * V, hipMemcpyDeviceToDevice, w->stream));
            w->ir_thaw_active = true;
            w->d.ir_frozen_snap = w->ir_frozen_snap_buf.p; w->d.ir_frozen_epoch = w->ir_frozen_epoch_buf.p;
            w->d.ir_thaw_epoch = w->ir_thaw_epoch_buf.p;
        }
    }
    w->p.enable_mask = kind_mask;
    w->d.enable = kind_mask;
    if (kind_mask & 8u) { w->trk_ever_on = true; w->d.trk_cols = 1; }
    w->stale_kinds &= ~kind_mask;
    return MGX_OK;
}

int mgx_set_antenna(mgx_world *w, int32_t robot, int32_t active) {
    MGX_ENTER(w);
    if (!w || robot < 0 || (size_t)robot >= w->robots.size()) return fail(MGX_ERR_INVALID, "bad robot id");
    if (w->robots[(size_t)robot].removed) return fail(MGX_ERR_STATE, "robot %d was removed", robot);
    if (w->robots[(size_t)robot].antenna != (active ? 1 : 0)) flush_counts(w);
    w->robots[(size_t)robot].antenna = active ? 1 : 0;
    w->flags_dirty = true;
    return MGX_OK;
}
int mgx_set_idle(mgx_world *w, int32_t robot, int32_t idle) {
    MGX_ENTER(w);
    if (!w || robot < 0 || (size_t)robot >= w->robots.size()) return fail(MGX_ERR_INVALID, "bad robot id");
    if (w->robots[(size_t)robot].removed) return fail(MGX_ERR_STATE, "robot %d was removed", robot);
    if (w->robots[(size_t)robot].idle != (idle ? 1 : 0)) flush_counts(w);
    w->robots[(size_t)robot].idle = idle ? 1 : 0;
    w->flags_dirty = true;
    return MGX_OK;
}

int mgx_set_antennas(mgx_world *w, uint32_t n, const int32_t *robots, const uint8_t *active) {
    MGX_ENTER(w);
    if (!w || (n && (!robots || !active))) return fail(MGX_ERR_INVALID, "null argument");
    for (uint32_t i = 0; i < n; i++) {
        if (robots[i] < 0 || (size_t)robots[i] >= w->robots.size()) return fail(MGX_ERR_INVALID, "bad robot id");
        if (w->robots[(size_t)robots[i]].removed) return fail(MGX_ERR_STATE, "robot %d was removed", robots[i]);
    }
    for (uint32_t i = 0; i < n; i++)
        if (w->robots[(size_t)robots[i]].antenna != (active[i] ? 1 : 0)) { flush_counts(w); break; }
    for (uint32_t i = 0; i < n; i++) w->robots[(size_t)robots[i]].antenna = active[i] ? 1 : 0;
    w->flags_dirty = true;
    return MGX_OK;
}

// ---- dynamic inter-robot topology (robot.rs:1362-1586) -----------------------------------------------

// The fine-grained calls keep robots_connected_with in step, as create_/delete_interrobot_factors
// do (robot.rs:1406-1408,1546), so that they can be mixed with mgx_update_topology.
int mgx_ir_connect(mgx_world *w, int32_t owner, int32_t other, uint64_t first_robot_number) {
    MGX_ENTER(w);
    int rc = ir_connect(w, owner, other, first_robot_number);
    if (rc != MGX_OK) return rc;
    if (!w->sets.has((size_t)owner, other)) w->sets.insert_sorted((size_t)owner, other);
    return MGX_OK;
}
int mgx_ir_disconnect(mgx_world *w, int32_t a, int32_t b) {
    MGX_ENTER(w);
    int rc = ir_disconnect(w, a, b);
    if (rc != MGX_OK) return rc;
    w->sets.erase((size_t)a, b);
    w->sets.erase((size_t)b, a);
    return MGX_OK;
}

// device neighbour search -> host CSR, rows ascending in order key
// pos == nullptr: the positions come from the device-resident Transforms of the missions (mgx_mission_tick)
// Two halves: everything that is enqueued (positions up, the counting and filling kernels, rows down into pinned memory) and,
// behind a synchronisation of the stream, the host side (a second filling pass if the rows outgrew the guess, ids and order).
// mgx_mission_tick enqueues the search of the NEXT tick in front of this tick's GBP schedule — the Transforms it looks at are
// final once the prior updates have moved them — so its rows are on the host long before that tick's one synchronisation.
static int neighbours_enqueue(mgx_world *w, const float *pos, float radius, uint32_t method, mgx_world::PendingSearch &ps) {
    if (!device_ok()) return fail(MGX_ERR_NO_DEVICE, "no HIP device");
    w->mission_search.valid = false;  // the buffers below are shared: whatever was waiting in them is gone
    ps.valid = false;
    const int n_all = (int)w->robots.size();
    const bool from_missions = pos == nullptr;
    std::vector<int> &alive = ps.alive;  // removed robots are in no query: search the others, map back
    alive.clear();
    std::vector<float> packed;
    for (int r = 0; r < n_all; r++) {
        // ghosts take part: a sharded world that follows a changing topology holds EVERY robot of the
        // scenario (its own ones and ghost copies of all others) and is handed all positions, so that
        // the connection bookkeeping below runs identically on every rank
        if (!w->sets.removed[(size_t)r]) alive.push_back(r);
    }
    const bool compact = (int)alive.size() != n_all;
    if (compact && !from_missions) {
        packed.resize(3 * alive.size());
        for (size_t a = 0; a < alive.size(); a++) memcpy(&packed[3 * a], pos + 3 * (size_t)alive[a], 3 * sizeof(float));
        pos = packed.data();
    }
    const int n = (int)alive.size();
    const bool usable_radius = std::isfinite(radius) && radius > 0.f;
    bool grid = method == MGX_NEIGHBOURS_GRID || (method == MGX_NEIGHBOURS_AUTO && n >= 512);
    if (!usable_radius) grid = false;  // radius <= 0 / NaN / inf: every pair has to see the predicate
    uint32_t M = 64;
    while (M < 2u * (uint32_t)std::max(n, 1)) M <<= 1;
    // A search over positions the CALLER hands in reads nothing of the world's device state: it runs on a stream of its own,
    // next to whatever the world's stream is still busy with (the previous tick's GBP schedule), instead of behind it.
    // (The missions' search reads the device's Transforms, which the tick's kernels move: that one stays in stream order.)
    hipStream_t s = w->stream;
    if (!from_missions) {
        // Never beside a resident launch that is still getting onto the device, unless both fit: the search's waves would take
        // slots the launch's last workgroups need (see mgx_update_topology).  The grid search of a small world is one workgroup
        // per 64 robots, each good for one of the launch's workgroup slots: with room for all of them the search goes out at once.
        const bool grid_rows = method == MGX_NEIGHBOURS_AUTO && n > 0 && n <= 1024 && usable_radius && w->nb_row_cap <= 32;
        const bool fits_beside = grid_rows && !w->xres.connected && w->resident_cap > 0 && w->d.R_local + 1 + (n + 63) / 64 <= w->resident_cap;
        if (w->pending.active && !fits_beside) { const int rcc = confirm_resident(w); if (rcc != MGX_OK) return rcc; }
        if (!w->search_stream) HIP_TRY(hipStreamCreateWithFlags(&w->search_stream, hipStreamNonBlocking));
        s = w->search_stream;
    }
    if (w->nb_last_stream_set && w->nb_last_stream != s) HIP_TRY(hipStreamSynchronize(w->nb_last_stream));  // the scratch buffers are shared
    w->nb_last_stream = s;
    w->nb_last_stream_set = true;
    ps.stream = s;
    HIP_TRY(w->nb_pos.reserve((size_t)3 * std::max(n, 1)));
    HIP_TRY(w->nb_cnt.reserve((size_t)std::max(n, 1)));
    HIP_TRY(w->nb_ptr.reserve((size_t)n + 1));
    HIP_TRY(w->nb_members.reserve((size_t)std::max(n, 1)));
    HIP_TRY(w->nb_special.reserve((size_t)std::max(n, 1)));
    HIP_TRY(w->nb_nspecial.reserve(1));
    HIP_TRY(w->nb_bucket_cnt.reserve(M));
    HIP_TRY(w->nb_bucket_ptr.reserve((size_t)M + 1));
    HIP_TRY(w->nb_cursor.reserve(M));
    if (from_missions) {
        mgx_world::Mission &ms = w->mission;
        if (ms.alive_dirty || ms.alive_host.size() != alive.size()) {
            ms.alive_host.assign(alive.begin(), alive.end());
            if (ms.alive_host.empty()) ms.alive_host.push_back(0);
            HIP_TRY(ms.alive_d.upload(ms.alive_host, s));
            HIP_TRY(hipStreamSynchronize(s));
            ms.alive_host.resize(alive.size());
            ms.alive_dirty = false;
        }
        HIP_TRY(launch_mission_positions(ms.d, n, ms.alive_d.p, w->nb_pos.p, s));
    }
    // small worlds (AUTO): ONE small kernel, rows of a fixed capacity written in place, no scans (mgx_topology.hip)
    const bool rows_mode = method == MGX_NEIGHBOURS_AUTO && n > 0 && n <= 4096;
    if (rows_mode) {
        const int cap = w->nb_row_cap;
        const size_t off_cnt = sizeof(float) * 3 * (size_t)n, off_rows = off_cnt + sizeof(int32_t) * (size_t)n;
        HIP_TRY(w->nb_pin.reserve(off_rows + sizeof(int32_t) * (size_t)n * (size_t)cap));
        HIP_TRY(w->nb_idx.reserve((size_t)n * (size_t)cap));
        char *pin = static_cast<char *>(w->nb_pin.p);
        // no copies at all: the kernel reads the callers' positions from the pinned block and writes counts and rows into it
        // (a copy is a launch of its own — a blit kernel too big to find room beside a resident schedule launch)
        void *dpin = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&dpin, pin, 0));
        char *dp = static_cast<char *>(dpin);
        if (!from_missions) memcpy(pin, pos, sizeof(float) * 3 * (size_t)n);
        HIP_TRY(neighbours_rows(from_missions ? w->nb_pos.p : reinterpret_cast<const float *>(dp), n, radius, cap,
                                reinterpret_cast<int32_t *>(dp + off_cnt), reinterpret_cast<int32_t *>(dp + off_rows), s, nullptr));
        ps.n = n; ps.n_all = n_all; ps.compact = compact; ps.guess = 0; ps.off_ptr = off_cnt; ps.off_idx = off_rows;
        ps.radius = radius; ps.method = method; ps.grid = false; ps.M = M; ps.rows = true; ps.row_cap = cap; ps.from_missions = from_missions;
        ps.valid = true;
        return MGX_OK;
    }
    ps.rows = false;
    const size_t guess = std::min(w->nb_idx.cap, w->nb_last_total + w->nb_last_total / 4 + 64);
    // pinned layout: [3 n floats: positions up] [n + 1 ints: row pointers down] [guess ints: rows down]
    const size_t off_ptr = sizeof(float) * 3 * (size_t)std::max(n, 1), off_idx = off_ptr + sizeof(int32_t) * ((size_t)n + 1);
    HIP_TRY(w->nb_pin.reserve(off_idx + sizeof(int32_t) * guess));
    char *pin = static_cast<char *>(w->nb_pin.p);
    if (!from_missions && n) {
        memcpy(pin, pos, sizeof(float) * 3 * (size_t)n);
        HIP_TRY(hipMemcpyAsync(w->nb_pos.p, pin, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(neighbours_count(w->nb_pos.p, n, radius, grid, M, w->nb_cnt.p, w->nb_bucket_cnt.p, w->nb_bucket_ptr.p, w->nb_cursor.p,
                             w->nb_members.p, w->nb_special.p, w->nb_nspecial.p, w->nb_ptr.p, s));
    // The second pass needs the total to size its output — one more host round trip.  Instead it runs right
    // away into the buffer left from the last search (the kernels leave it alone if the rows do not fit), and
    // rows and counts come back together; only a total beyond the guess costs the second trip.
    if (guess > 0 && w->nb_idx.p)
        HIP_TRY(neighbours_fill(w->nb_pos.p, n, radius, grid, M, w->nb_bucket_ptr.p, w->nb_members.p, w->nb_special.p,
                                w->nb_nspecial.p, w->nb_ptr.p, w->nb_idx.p, (int32_t)guess, s));
    HIP_TRY(hipMemcpyAsync(pin + off_ptr, w->nb_ptr.p, sizeof(int32_t) * ((size_t)n + 1), hipMemcpyDeviceToHost, s));
    if (guess > 0 && w->nb_idx.p) HIP_TRY(hipMemcpyAsync(pin + off_idx, w->nb_idx.p, sizeof(int32_t) * guess, hipMemcpyDeviceToHost, s));
    ps.n = n; ps.n_all = n_all; ps.compact = compact; ps.guess = guess; ps.off_ptr = off_ptr; ps.off_idx = off_idx;
    ps.radius = radius; ps.method = method; ps.grid = grid; ps.M = M;
    ps.valid = true;
    return MGX_OK;
}
static int neighbours_collect(mgx_world *w, mgx_world::PendingSearch &ps, std::vector<int32_t> &ptr, std::vector<int32_t> &idx) {
    hipStream_t s = ps.stream;
    const int n = ps.n, n_all = ps.n_all;
    const std::vector<int> &alive = ps.alive;
    const size_t guess = ps.guess;
    char *pin = static_cast<char *>(w->nb_pin.p);
    ps.valid = false;
    HIP_TRY(hipStreamSynchronize(s));
    if (ps.rows) {
        int cap = ps.row_cap;
        const int32_t *cnt = reinterpret_cast<const int32_t *>(pin + ps.off_ptr);
        int32_t longest = 0;
        for (int i = 0; i < n; i++) longest = std::max(longest, cnt[i]);
        if (longest > cap) {  // a row outgrew its capacity: once more with room (the world remembers)
            while (cap < longest) cap *= 2;
            w->nb_row_cap = cap;
            const size_t off_rows = ps.off_ptr + sizeof(int32_t) * (size_t)n;
            // the pinned block grows: the callers' positions it holds move along (a mission's are on the device)
            std::vector<float> keep;
            if (!ps.from_missions) keep.assign(reinterpret_cast<const float *>(pin), reinterpret_cast<const float *>(pin) + 3 * (size_t)n);
            HIP_TRY(w->nb_pin.reserve(off_rows + sizeof(int32_t) * (size_t)n * (size_t)cap));
            pin = static_cast<char *>(w->nb_pin.p);
            if (!ps.from_missions) memcpy(pin, keep.data(), sizeof(float) * keep.size());
            void *dpin = nullptr;
            HIP_TRY(hipHostGetDevicePointer(&dpin, pin, 0));
            char *dp = static_cast<char *>(dpin);
            HIP_TRY(neighbours_rows(ps.from_missions ? w->nb_pos.p : reinterpret_cast<const float *>(dp), n, ps.radius, cap,
                                    reinterpret_cast<int32_t *>(dp + ps.off_ptr), reinterpret_cast<int32_t *>(dp + off_rows), s, nullptr));
            HIP_TRY(hipStreamSynchronize(s));
            cnt = reinterpret_cast<const int32_t *>(pin + ps.off_ptr);
        }
        const int32_t *rows = reinterpret_cast<const int32_t *>(pin + ps.off_ptr + sizeof(int32_t) * (size_t)n);
        ptr.assign((size_t)n + 1, 0);
        for (int i = 0; i < n; i++) ptr[(size_t)i + 1] = ptr[(size_t)i] + cnt[i];
        idx.resize((size_t)ptr[(size_t)n]);
        for (int i = 0; i < n; i++)
            if (cnt[i]) memcpy(idx.data() + ptr[(size_t)i], rows + (size_t)i * (size_t)cap, sizeof(int32_t) * (size_t)cnt[i]);
        w->nb_last_total = idx.size();
    } else {
    ptr.assign((size_t)n + 1, 0);
    memcpy(ptr.data(), pin + ps.off_ptr, sizeof(int32_t) * ((size_t)n + 1));
    const size_t total = (size_t)ptr[(size_t)n];
    w->nb_last_total = total;
    if (total <= guess && (w->nb_idx.p || total == 0)) {
        idx.resize(total);
        if (total) memcpy(idx.data(), pin + ps.off_idx, sizeof(int32_t) * total);
    } else {
        idx.assign(total, 0);
        HIP_TRY(w->nb_idx.reserve(total));
        HIP_TRY(neighbours_fill(w->nb_pos.p, n, ps.radius, ps.grid, ps.M, w->nb_bucket_ptr.p, w->nb_members.p, w->nb_special.p,
                                w->nb_nspecial.p, w->nb_ptr.p, w->nb_idx.p, (int32_t)total, s));
        HIP_TRY(hipMemcpyAsync(idx.data(), w->nb_idx.p, sizeof(int32_t) * total, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    }
    if (ps.compact) {  // back to world robot ids, empty rows for the removed ones
        for (int32_t &j : idx) j = alive[(size_t)j];
        std::vector<int32_t> full((size_t)n_all + 1, 0);
        for (int a = 0; a < n; a++) full[(size_t)alive[(size_t)a] + 1] = ptr[(size_t)a + 1] - ptr[(size_t)a];
        for (int r = 0; r < n_all; r++) full[(size_t)r + 1] += full[(size_t)r];
        ptr.swap(full);
    }
    bool monotone = true;  // ids ascending == keys ascending?  (the keys' compact copies: fixed when a robot is added)
    for (int r = 1; r < n_all && monotone; r++) monotone = w->sets.keys[(size_t)r - 1] < w->sets.keys[(size_t)r];
    if (!monotone)
        for (int r = 0; r < n_all; r++)
            std::sort(idx.begin() + ptr[(size_t)r], idx.begin() + ptr[(size_t)r + 1],
                      [&](int a, int b) { return w->robots[(size_t)a].order_key < w->robots[(size_t)b].order_key; });
    return MGX_OK;
}
static int neighbours(mgx_world *w, const float *pos, float radius, uint32_t method, std::vector<int32_t> &ptr,
                      std::vector<int32_t> &idx) {
    mgx_world::PendingSearch ps;
    const int rc = neighbours_enqueue(w, pos, radius, method, ps);
    return rc != MGX_OK ? rc : neighbours_collect(w, ps, ptr, idx);
}

int mgx_neighbours(mgx_world *w, const float *positions_xyz, float radius, uint32_t method, int32_t *row_ptr, int32_t *neighbours_out,
                   uint64_t capacity, uint64_t *needed) {
    MGX_ENTER(w);
    if (!w || !positions_xyz || !row_ptr) return fail(MGX_ERR_INVALID, "null argument");
    if (method > MGX_NEIGHBOURS_GRID) return fail(MGX_ERR_INVALID, "bad method");
    std::vector<int32_t> ptr, idx;
    int rc = neighbours(w, positions_xyz, radius, method, ptr, idx);
    if (rc != MGX_OK) return rc;
    memcpy(row_ptr, ptr.data(), sizeof(int32_t) * ptr.size());
    if (needed) *needed = idx.size();
    if (!neighbours_out) return MGX_OK;  // sizing call
    if (idx.size() > capacity) return fail(MGX_ERR_INVALID, "neighbour list needs %zu entries, capacity %llu", idx.size(), (unsigned long long)capacity);
    if (!idx.empty()) memcpy(neighbours_out, idx.data(), sizeof(int32_t) * idx.size());
    return MGX_OK;
}

int mgx_connections(mgx_world *w, int32_t robot, int32_t *others, uint32_t capacity, uint32_t *n) {
    MGX_ENTER(w);
    if (!w || !n || robot < 0 || (size_t)robot >= w->robots.size()) return fail(MGX_ERR_INVALID, "bad argument");
    const int32_t *c = w->sets.row((size_t)robot);
    const size_t nc = (size_t)w->sets.cnt[(size_t)robot];
    *n = (uint32_t)nc;
    if (!others) return MGX_OK;
    if (nc > capacity) return fail(MGX_ERR_INVALID, "capacity too small");
    for (size_t i = 0; i < nc; i++) others[i] = c[i];
    return MGX_OK;
}

static int topology_bookkeeping(mgx_world *w, std::vector<int32_t> &ptr, std::vector<int32_t> &idx, uint64_t *robot_number_next,
                                uint32_t *stats, StageTimer &tm);
int mgx_update_topology(mgx_world *w, const float *positions_xyz, float radius, uint32_t method, uint64_t *robot_number_next,
                        uint32_t *stats) {
    MGX_ENTER(w);
    if (!w || !positions_xyz || !robot_number_next) return fail(MGX_ERR_INVALID, "null argument");
    if (*robot_number_next == 0) return fail(MGX_ERR_INVALID, "robot_number is NonZeroUsize");
    if (method > MGX_NEIGHBOURS_GRID) return fail(MGX_ERR_INVALID, "bad method");
    std::vector<int32_t> ptr, idx;
    StageTimer tm("update_topology");
    // update_robot_neighbours (robot.rs:1362-1384).  A small world's search runs BESIDE the GBP schedule of the tick before, on a
    // stream of its own — but it must not get onto the device before that schedule's resident launch has all its workgroups
    // there: enqueued a few microseconds behind the launch, its waves took slots the launch's last workgroups needed, the
    // residency census said no and the tick ran launch by launch (seen: three of sixty ticks, 0.8 ms each).  So the search is
    // enqueued only with the launch decided (microseconds after its start); the message counters are brought up to date
    // under it (the pass is about to change who sends to whom).  (Who owns which connection — what the deletions walk —
    // comes from the connection index, kept in step with the list: round 4 listed the connections by owner here, every tick.)
    mgx_world::PendingSearch ps;
    int rc = neighbours_enqueue(w, positions_xyz, radius, method, ps);  // (waits for the launch to be decided first if it has to)
    if (rc != MGX_OK) return rc;
    tm.lap("search enqueued");
    flush_counts(w, true);  // (lazy: the robots' counters and cumulative counts; connections are settled when deleted or read)
    tm.lap("message counters (under the search)");
    rc = neighbours_collect(w, ps, ptr, idx);
    if (rc != MGX_OK) return rc;
    tm.lap("neighbour search");
    return topology_bookkeeping(w, ptr, idx, robot_number_next, stats, tm);
}
// delete_interrobot_factors + create_interrobot_factors on the search's result (rows per robot id, ascending)
static int topology_bookkeeping(mgx_world *w, std::vector<int32_t> &ptr, std::vector<int32_t> &idx, uint64_t *robot_number_next,
                                uint32_t *stats, StageTimer &tm) {
    int rc = MGX_OK;
    const int n = (int)w->robots.size();
    uint32_t created = 0, deleted = 0;
    // a robot's row of the search and its connection set are both ascending in order key (BTreeSet<Entity>): merges
    ConnSets &cs = w->sets;
    auto key = [&](int x) { return cs.keys[(size_t)x]; };

    // delete_interrobot_factors (robot.rs:1386-1439).  The pairs pass through a
    // HashMap<RobotId, RobotId> filled with `extend` (:1391,1400-1404): one entry per robot, the
    // LAST out-of-range id (largest key) wins; every out-of-range id leaves robots_connected_with
    // (:1406-1408) whether or not its factors get deleted.  The map's iteration order is
    // unspecified in the reference; ascending robot id here.
    // (one merge per robot does both halves of the pass — a robot's row of the search and its connection set are ascending in
    // order key: what is in the set and not in the row is out of range; what is in the row and not in the set is a new
    // neighbour, create_interrobot_factors' snapshot (robot.rs:1449-1461: within range \ connected, taken for every robot
    // before anything is created; the deletions in between leave the sets alone).)
    std::vector<int> victim((size_t)n, -1);
    std::vector<std::pair<int, int>> fresh;  // (robot, new neighbour), robots ascending, neighbours in row order
    for (int r = 0; r < n; r++) {
        int32_t *cw = cs.row((size_t)r);
        const int32_t n_cw = cs.cnt[(size_t)r];
        const bool gone = w->sets.removed[(size_t)r] != 0;  // not in the query any more: its set stays as it is
        int32_t kept = 0, q = 0;
        int32_t j = ptr[(size_t)r];
        const int32_t j1 = ptr[(size_t)r + 1];
        while (q < n_cw || j < j1) {
            if (j >= j1 || (q < n_cw && key(cw[q]) < key(idx[(size_t)j]))) {  // connected, not in range
                if (gone) cw[kept++] = cw[q];
                else victim[(size_t)r] = cw[q];
                q++;
            } else if (q >= n_cw || key(idx[(size_t)j]) < key(cw[q])) {  // in range, not connected
                fresh.emplace_back(r, idx[(size_t)j]);
                j++;
            } else {  // both
                cw[kept++] = cw[q];
                q++;
                j++;
            }
        }
        cs.cnt[(size_t)r] = kept;
    }
    tm.lap("range scan");
    {
        std::vector<std::pair<int, int>> pairs;
        for (int r = 0; r < n; r++)
            if (victim[(size_t)r] >= 0) pairs.emplace_back(r, victim[(size_t)r]);
        ir_disconnect_batch(w, pairs);
        deleted = (uint32_t)pairs.size();
        tm.lap("delete");
    }
    for (const auto &f : fresh) {
        const int r = f.first, o = f.second;
        rc = ir_connect(w, r, o, *robot_number_next);
        if (rc != MGX_OK) return rc;
        *robot_number_next += (uint64_t)(w->K - 1);
        cs.insert_sorted((size_t)r, o);  // :1546
        created++;
    }
    tm.lap("create");
    if (stats) { stats[0] = created; stats[1] = deleted; }
    return MGX_OK;
}

// ---- missions on the device ---------------------------------------------------------------------------
static std::vector<Launch> plan_launches(const uint8_t *steps, uint32_t n);
static int mission_download(mgx_world *w) {  // device -> host copies of what the device advances (before the arrays are laid out again)
    mgx_world::Mission &ms = w->mission;
    if (!ms.uploaded) return MGX_OK;
    std::vector<int32_t> tg;
    std::vector<float> tr;
    std::vector<long long> fin;
    HIP_TRY(ms.target_d.download(tg, w->stream));
    HIP_TRY(ms.translation_d.download(tr, w->stream));
    HIP_TRY(ms.finished_d.download(fin, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    for (size_t r = 0; r < tg.size() && r < ms.target.size(); r++) {
        ms.target[r] = tg[r];
        ms.finished_tick[r] = fin[r];
        for (int c = 0; c < 3; c++) ms.translation[3 * r + c] = tr[3 * r + c];
    }
    return MGX_OK;
}
static int mission_upload(mgx_world *w) {
    mgx_world::Mission &ms = w->mission;
    const size_t R = w->robots.size();
    ms.wp.resize(R);
    ms.target.resize(R, 0); ms.vars.resize(2 * R, 0); ms.dist2.resize(2 * R, 0.f); ms.translation.resize(3 * R, 0.f);
    ms.time_scale.resize(R, 0.0); ms.has.resize(R, 0); ms.finished_tick.resize(R, -1);
    std::vector<int32_t> ptr(R + 1, 0);
    std::vector<double> xy;
    for (size_t r = 0; r < R; r++) {
        xy.insert(xy.end(), ms.wp[r].begin(), ms.wp[r].end());
        ptr[r + 1] = (int32_t)(xy.size() / 2);
    }
    if (xy.empty()) xy.assign(2, 0.0);
    hipStream_t s = w->stream;
    HIP_TRY(ms.wp_ptr_d.upload(ptr, s));
    HIP_TRY(ms.wp_xy_d.upload(xy, s));
    HIP_TRY(ms.target_d.upload(ms.target, s));
    HIP_TRY(ms.vars_d.upload(ms.vars, s));
    HIP_TRY(ms.dist2_d.upload(ms.dist2, s));
    HIP_TRY(ms.translation_d.upload(ms.translation, s));
    HIP_TRY(ms.time_scale_d.upload(ms.time_scale, s));
    HIP_TRY(ms.has_d.upload(ms.has, s));
    HIP_TRY(ms.finished_d.upload(ms.finished_tick, s));
    HIP_TRY(ms.rec_d.reserve(4 * R));
    HIP_TRY(ms.robots_d.reserve(R));
    HIP_TRY(ms.waypoints_d.reserve(2 * R));
    HIP_TRY(ms.ts_list_d.reserve(R));
    HIP_TRY(ms.what_d.reserve(R));
    HIP_TRY(ms.moving_d.reserve(R));
    HIP_TRY(hipStreamSynchronize(s));
    if (ms.ev_cap < R + 1) {
        if (ms.ev_host) (void)hipHostFree(ms.ev_host);
        ms.ev_host = nullptr;
        HIP_TRY(hipHostMalloc((void **)&ms.ev_host, sizeof(unsigned int) * (R + 64), hipHostMallocMapped));
        ms.ev_cap = R + 64;
        ms.ev_host[0] = 0;
    }
    DevMission &d = ms.d;
    d.wp_ptr = ms.wp_ptr_d.p; d.wp_xy = ms.wp_xy_d.p; d.target = ms.target_d.p; d.vars = ms.vars_d.p; d.dist2 = ms.dist2_d.p;
    d.time_scale = ms.time_scale_d.p; d.has = ms.has_d.p; d.translation = ms.translation_d.p; d.finished_tick = ms.finished_d.p;
    ms.uploaded = true;
    ms.dirty = false;
    ms.alive_dirty = true;
    return MGX_OK;
}

int mgx_mission_set(mgx_world *w, int32_t robot, const mgx_mission_desc *desc) {
    MGX_ENTER(w);
    if (!w || !desc || robot < 0 || (size_t)robot >= w->robots.size()) return fail(MGX_ERR_INVALID, "bad argument");
    const Robot &rb = w->robots[(size_t)robot];
    if (rb.ghost || rb.removed) return fail(MGX_ERR_INVALID, "robot %d is not a live local robot", robot);
    if (desc->n_waypoints == 0 || !desc->waypoints_xy) return fail(MGX_ERR_INVALID, "a mission needs at least one waypoint");
    if ((int)desc->reach_var >= rb.K || (int)desc->finish_var >= rb.K) return fail(MGX_ERR_INVALID, "rule names a variable beyond the horizon");
    for (const Robot &q : w->robots)
        if (q.ghost) return fail(MGX_ERR_STATE, "missions run on unsharded worlds");
    mgx_world::Mission &ms = w->mission;
    if (ms.uploaded && !ms.dirty) {  // the device has advanced the missions it holds: fetch before the arrays are laid out again
        int rc = mission_download(w);
        if (rc != MGX_OK) return rc;
    }
    const size_t R = w->robots.size(), r = (size_t)robot;
    ms.wp.resize(R);
    ms.target.resize(R, 0); ms.vars.resize(2 * R, 0); ms.dist2.resize(2 * R, 0.f); ms.translation.resize(3 * R, 0.f);
    ms.time_scale.resize(R, 0.0); ms.has.resize(R, 0); ms.finished_tick.resize(R, -1);
    ms.wp[r].assign(desc->waypoints_xy, desc->waypoints_xy + 2 * (size_t)desc->n_waypoints);
    ms.target[r] = 0;
    ms.vars[2 * r] = desc->reach_var; ms.vars[2 * r + 1] = desc->finish_var;
    ms.dist2[2 * r] = desc->reach_dist2; ms.dist2[2 * r + 1] = desc->finish_dist2;
    for (int c = 0; c < 3; c++) ms.translation[3 * r + c] = desc->translation[c];
    ms.time_scale[r] = desc->time_scale;
    ms.has[r] = 1;
    ms.finished_tick[r] = -1;
    ms.any = true;
    ms.dirty = true;
    return MGX_OK;
}

int mgx_mission_tick(mgx_world *w, float comms_radius, uint32_t method, uint64_t *robot_number_next, int32_t despawn_finished,
                     const uint8_t *antennas, double max_speed, double delta_t, const uint8_t *steps, uint32_t n_steps, uint32_t *stats) {
    MGX_ENTER(w);
    int rc = mgx_mission_tick_begin(w, comms_radius, method, robot_number_next, despawn_finished, stats);
    return rc != MGX_OK ? rc : mgx_mission_tick_end(w, antennas, max_speed, delta_t, steps, n_steps);
}

int mgx_mission_tick_begin(mgx_world *w, float comms_radius, uint32_t method, uint64_t *robot_number_next, int32_t despawn_finished,
                           uint32_t *stats) {
    MGX_ENTER(w);
    if (!w || !robot_number_next) return fail(MGX_ERR_INVALID, "null argument");
    if (*robot_number_next == 0) return fail(MGX_ERR_INVALID, "robot_number is NonZeroUsize");
    if (method > MGX_NEIGHBOURS_GRID) return fail(MGX_ERR_INVALID, "bad method");
    mgx_world::Mission &ms = w->mission;
    if (!ms.any) return fail(MGX_ERR_STATE, "no robot has a mission (mgx_mission_set)");
    if (w->K < 3) return fail(MGX_ERR_INVALID, "needs K >= 3");
    int rc = commit(w);
    if (rc != MGX_OK) return rc;
    if (w->d.R_total != w->d.R_local) return fail(MGX_ERR_STATE, "missions run on unsharded worlds");
    if (ms.dirty || ms.has.size() != w->robots.size()) {
        if (ms.uploaded && !ms.dirty && (rc = mission_download(w)) != MGX_OK) return rc;
        if ((rc = mission_upload(w)) != MGX_OK) return rc;
        w->mission_search.valid = false;  // the host has touched the missions: Transforms may have moved
    }
    const int R = (int)w->robots.size();
    hipStream_t s = w->stream;
    StageTimer tm("mission_tick");
    // reached_waypoint (robot.rs:2080-2176), then update_robot_neighbours on the Transforms as they are (robot.rs:1362-1384);
    // both results come back at the tick's ONE synchronisation (inside neighbours())
    ms.ev_host[0] = 0;
    void *evd = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&evd, ms.ev_host, 0));
    HIP_TRY(launch_mission_reached(w->d, ms.d, R, ms.tick_no, (unsigned int *)evd, s));
    std::vector<int32_t> ptr, idx;
    // the search itself was enqueued by the last tick's end, in front of its GBP schedule, if nothing has changed since: the same
    // radius and method, the same robots alive (this tick's despawns are taken out of the rows below either way)
    mgx_world::PendingSearch &pf = w->mission_search;
    bool prefetched = pf.valid && pf.radius == comms_radius && pf.method == method && pf.n_all == R;
    if (prefetched) {
        size_t a = 0;
        for (int r = 0; r < R && prefetched; r++)
            if (!w->robots[(size_t)r].removed) {
                if (a >= pf.alive.size() || pf.alive[a] != r) prefetched = false;
                a++;
            }
        if (a != pf.alive.size()) prefetched = false;
    }
    rc = prefetched ? neighbours_collect(w, pf, ptr, idx) : neighbours(w, nullptr, comms_radius, method, ptr, idx);
    if (rc != MGX_OK) return rc;
    ms.search_radius = comms_radius;
    ms.search_method = method;
    ms.search_known = true;
    tm.lap(prefetched ? "reached + rows of the search enqueued last tick" : "reached + search");
    // robots that reached their last waypoint this tick: despawned before the topology systems see them (robot.rs:2172) —
    // the search still looked at them, so they are taken out of its rows here
    const unsigned n_fin = ms.ev_host[0];
    std::vector<uint8_t> gone;
    ms.last_finished.clear();
    if (n_fin) {
        std::vector<int> fin(ms.ev_host + 1, ms.ev_host + 1 + n_fin);
        std::sort(fin.begin(), fin.end());
        ms.last_finished.assign(fin.begin(), fin.end());
        for (int r : fin) ms.finished_tick[(size_t)r] = ms.tick_no;
        if (despawn_finished) {
            gone.assign((size_t)R, 0);
            for (int r : fin) {
                gone[(size_t)r] = 1;
                if ((rc = mgx_robot_remove(w, r)) != MGX_OK) return rc;
            }
            std::vector<int32_t> nptr((size_t)R + 1, 0), nidx;
            nidx.reserve(idx.size());
            for (int r = 0; r < R; r++) {
                if (!gone[(size_t)r])
                    for (int32_t q = ptr[(size_t)r]; q < ptr[(size_t)r + 1]; q++)
                        if (!gone[(size_t)idx[(size_t)q]]) nidx.push_back(idx[(size_t)q]);
                nptr[(size_t)r + 1] = (int32_t)nidx.size();
            }
            ptr.swap(nptr);
            idx.swap(nidx);
        }
    }
    uint32_t st[2] = {0, 0};
    rc = topology_bookkeeping(w, ptr, idx, robot_number_next, st, tm);
    if (rc != MGX_OK) return rc;
    if (stats) { stats[0] = st[0]; stats[1] = st[1]; stats[2] = n_fin; }
    ms.in_tick = true;
    return MGX_OK;
}

int mgx_mission_tick_end(mgx_world *w, const uint8_t *antennas, double max_speed, double delta_t, const uint8_t *steps, uint32_t n_steps) {
    MGX_ENTER(w);
    if (!w || (!steps && n_steps)) return fail(MGX_ERR_INVALID, "null argument");
    mgx_world::Mission &ms = w->mission;
    if (!ms.in_tick) return fail(MGX_ERR_STATE, "mgx_mission_tick_end without mgx_mission_tick_begin");
    ms.in_tick = false;
    int rc = MGX_OK;
    const int R = (int)w->robots.size();
    hipStream_t s = w->stream;
    StageTimer tm("mission_tick_end");
    if (ms.dirty || ms.has.size() != w->robots.size()) {  // robots joined between the two halves: lay the missions out again
        if (ms.uploaded && !ms.dirty && (rc = mission_download(w)) != MGX_OK) return rc;
        if ((rc = commit(w)) != MGX_OK) return rc;
        if ((rc = mission_upload(w)) != MGX_OK) return rc;
    }
    // update_failed_comms (robot.rs:1593-1601): the caller's draws for the robots still alive
    if (antennas) {
        bool changed = false;
        for (int r = 0; r < R; r++) {
            Robot &rb = w->robots[(size_t)r];
            if (rb.removed || rb.ghost) continue;
            const uint8_t a = antennas[r] ? 1 : 0;
            if (rb.antenna != a) { if (!changed) flush_counts(w); changed = true; rb.antenna = a; }
        }
        if (changed) w->flags_dirty = true;
    }
    if ((rc = commit(w)) != MGX_OK) return rc;  // edge tables / flags of what the pass changed
    tm.lap("antennas + commit");
    // the two prior updates + the Transform increment, from the device's own mission state; then iterate_gbp_v2
    {
        void *hp = nullptr, *dp = nullptr;
        int slot = 0;
        HIP_TRY(w->stage.acquire((size_t)R, &hp, &slot));
        uint8_t *mv = (uint8_t *)hp;
        for (int r = 0; r < R; r++) mv[r] = (!w->robots[(size_t)r].removed && !w->robots[(size_t)r].ghost) ? 1 : 0;
        HIP_TRY(hipHostGetDevicePointer(&dp, hp, 0));
        HIP_TRY(launch_copy_bytes(ms.moving_d.p, (const uint8_t *)dp, (size_t)R, s));
        HIP_TRY(w->stage.release(slot, s));
    }
    HIP_TRY(launch_mission_prepare(w->d, ms.d, R, ms.moving_d.p, ms.rec_d.p, ms.robots_d.p, ms.waypoints_d.p, ms.ts_list_d.p, ms.what_d.p, s));
    // the Transforms after this tick's move travel to the host behind the launch: complete at the next synchronisation
    // (the next tick's own one), read without one by mgx_mission_translations
    if (ms.tr_cap < (size_t)R) {
        if (ms.tr_host) (void)hipHostFree(ms.tr_host);
        ms.tr_host = nullptr;
        HIP_TRY(hipHostMalloc((void **)&ms.tr_host, sizeof(float) * 3 * ((size_t)R + 64), hipHostMallocDefault));
        ms.tr_cap = (size_t)R + 64;
    }
    HIP_TRY(hipMemcpyAsync(ms.tr_host, ms.translation_d.p, sizeof(float) * 3 * (size_t)R, hipMemcpyDeviceToHost, s));
    ms.tr_n = (size_t)R;
    // update_robot_neighbours of the COMING tick (robot.rs:1362-1384): the Transforms it looks at are final now, so the search goes
    // in front of this tick's GBP schedule and its rows reach the host while that runs
    tm.lap("prepare");
    if (ms.search_known && (rc = neighbours_enqueue(w, nullptr, ms.search_radius, ms.search_method, w->mission_search)) != MGX_OK) return rc;
    tm.lap("search of the coming tick enqueued");
    for (int r = 0; r < R; r++) {  // message counters: the prior changes of the robots that move (what the device decides too)
        const Robot &rb = w->robots[(size_t)r];
        if (rb.removed || rb.ghost || !ms.has[(size_t)r] || ms.finished_tick[(size_t)r] >= 0) continue;
        log_change_prior(w, r, w->K - 1);
        log_change_prior(w, r, 0);
    }
    w->stale_kinds |= ~w->p.enable_mask & 15u;
    tm.lap("counter log");
    const std::vector<Launch> plan = plan_launches(steps, n_steps);
    if (w->pending.active) { const int rcc = confirm_resident(w); if (rcc != MGX_OK) return rcc; }  // (a declined launch is run again here: not this call's launches)
    w->last_sweep_launches = 0;
    const bool fuse = !plan.empty() && plan[0].ext == 0 && plan[0].n_int > 0 && w->thaw_kinds == 0;
    if (!fuse) {
        HIP_TRY(launch_update_priors(w->d, R, ms.robots_d.p, ms.waypoints_d.p, ms.ts_list_d.p, ms.what_d.p, max_speed, delta_t, s));
        rc = iterate_now(w, steps, n_steps);
    } else {
        w->d.upd = ms.rec_d.p; w->d.upd_max_speed = max_speed; w->d.upd_delta_t = delta_t;
        const int resident = run_resident(w, plan);
        if (resident != 0) {
            w->d.upd = nullptr;
            rc = resident < 0 ? resident : MGX_OK;
        } else {
            bool first = true;
            rc = MGX_OK;
            for (const Launch &l : plan) {
                if (!first) w->d.upd = nullptr;
                rc = sweep(w, -1, l.ext, l.n_int ? (PH_INT_FACTOR | PH_INT_VARIABLE) : 0, l.n_int, l.hints);
                first = false;
                if (rc != MGX_OK) break;
            }
            w->d.upd = nullptr;
        }
    }
    ms.tick_no += 1;
    tm.lap("prior updates + schedule");
    return rc;
}

// Many ticks in one call (include/mgx.h): the host loop of a headless run — tick, comms draws, tick — without the caller's
// interpreter in it.  Nothing new happens on the device: the two halves above, once per tick.
int mgx_mission_run(mgx_world *w, mgx_mission_run_desc *d) {
    MGX_ENTER(w);
    if (!w || !d || !d->robot_number_next || (!d->steps && d->n_steps) || !d->created || !d->deleted || !d->n_finished ||
        (d->finished_capacity && !d->finished))
        return fail(MGX_ERR_INVALID, "null argument");
    if (!(d->failure_rate >= 0.0 && d->failure_rate <= 1.0)) return fail(MGX_ERR_INVALID, "failure_rate is outside [0, 1]");
    mgx_world::Mission &ms = w->mission;
    const size_t R = w->robots.size();
    d->ticks_done = 0;
    d->finished_total = 0;
    // rand 0.8.5 Bernoulli over wyrand 0.2.0 (restated in magics_amd/prng.py, which is the checker of this copy): one u64 per
    // draw unless p == 1
    const bool always = d->failure_rate == 1.0;
    const unsigned long long p_int = always ? 0ull : (unsigned long long)(d->failure_rate * 18446744073709551616.0);
    auto next_u64 = [&]() {
        unsigned long long &st = *reinterpret_cast<unsigned long long *>(d->wyrand_state);
        st += 0xA0761D6478BD642Full;
        const unsigned __int128 t = (unsigned __int128)st * (unsigned __int128)(st ^ 0xE7037ED1A0B428DBull);
        return (unsigned long long)(t >> 64) ^ (unsigned long long)t;
    };
    std::vector<uint8_t> ant(R, 1);
    int rc = MGX_OK;
    for (uint32_t t = 0; t < d->n_ticks; t++) {
        uint32_t st[3] = {0, 0, 0};
        if ((rc = mgx_mission_tick_begin(w, d->comms_radius, d->method, d->robot_number_next, d->despawn_finished, st)) != MGX_OK) return rc;
        if (w->robots.size() != R) return fail(MGX_ERR_STATE, "robots joined during mgx_mission_run");
        // (the tick's synchronisation lies behind: the Transforms the tick before sent to the host are complete)
        if (t > 0 && d->translations && ms.tr_host) memcpy(d->translations + (size_t)(t - 1) * R * 3, ms.tr_host, sizeof(float) * 3 * R);
        d->created[t] = st[0];
        d->deleted[t] = st[1];
        d->n_finished[t] = st[2];
        for (int32_t r : ms.last_finished) {
            if (d->finished_total < d->finished_capacity) d->finished[d->finished_total] = r;
            d->finished_total++;
        }
        // update_failed_comms (robot.rs:1593-1601): one draw per robot alive after this tick's despawns, id order
        bool any_off = false;
        if (d->wyrand_state)
            for (size_t r = 0; r < R; r++) {
                const Robot &rb = w->robots[r];
                if (rb.removed || rb.ghost) { ant[r] = 1; continue; }
                const bool fails = always ? true : next_u64() < p_int;
                ant[r] = fails ? 0 : 1;
                any_off = any_off || fails;
            }
        if (d->antennas) memcpy(d->antennas + (size_t)t * R, ant.data(), R);
        (void)any_off;
        if ((rc = mgx_mission_tick_end(w, (d->wyrand_state && d->failure_rate > 0.0) ? ant.data() : nullptr, d->max_speed, d->delta_t, d->steps,
                                       d->n_steps)) != MGX_OK)
            return rc;
        d->ticks_done = t + 1;
        if (d->stop_when_all_finished) {
            bool all = true;
            for (size_t r = 0; r < R && all; r++) all = !ms.has[r] || ms.finished_tick[r] >= 0;
            if (all) break;
        }
    }
    if (d->ticks_done && d->translations) {  // the last tick's Transforms: behind its launches
        if (w->linger.open && (rc = linger_close(w)) != MGX_OK) return rc;
        if (w->pending.active && (rc = confirm_resident(w)) != MGX_OK) return rc;
        HIP_TRY(hipStreamSynchronize(w->stream));
        if (ms.tr_host) memcpy(d->translations + (size_t)(d->ticks_done - 1) * R * 3, ms.tr_host, sizeof(float) * 3 * R);
        if ((rc = check_device_error(w)) != MGX_OK) return rc;
    }
    if (d->finished_total > d->finished_capacity) return fail(MGX_ERR_INVALID, "%u missions completed, room for %u", d->finished_total, d->finished_capacity);
    return MGX_OK;
}

int mgx_mission_finished(mgx_world *w, int32_t *robots, uint32_t capacity, uint32_t *n) {
    MGX_ENTER(w);
    if (!w || !n) return fail(MGX_ERR_INVALID, "null argument");
    const std::vector<int32_t> &f = w->mission.last_finished;
    *n = (uint32_t)f.size();
    if (robots)
        for (size_t i = 0; i < f.size() && i < capacity; i++) robots[i] = f[i];
    return MGX_OK;
}
int mgx_mission_translations(mgx_world *w, float *translations, uint32_t capacity_robots, uint32_t *n_robots) {
    MGX_ENTER(w);
    if (!w || !translations) return fail(MGX_ERR_INVALID, "null argument");
    const mgx_world::Mission &ms = w->mission;
    const size_t n = std::min<size_t>(ms.tr_n, capacity_robots);
    if (ms.tr_host && n) memcpy(translations, ms.tr_host, sizeof(float) * 3 * n);
    if (n_robots) *n_robots = (uint32_t)ms.tr_n;
    return MGX_OK;
}

int mgx_mission_read(mgx_world *w, float *translations, int32_t *targets, int64_t *finished_tick) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    mgx_world::Mission &ms = w->mission;
    if (!ms.any) return fail(MGX_ERR_STATE, "no robot has a mission");
    if (ms.uploaded && !ms.dirty) {
        int rc = mission_download(w);
        if (rc != MGX_OK) return rc;
    }
    const size_t R = std::min(w->robots.size(), ms.target.size());
    for (size_t r = 0; r < R; r++) {
        if (translations) for (int c = 0; c < 3; c++) translations[3 * r + c] = ms.translation[3 * r + c];
        if (targets) targets[r] = ms.has[r] ? ms.target[r] : -1;
        if (finished_tick) finished_tick[r] = ms.finished_tick[r];
    }
    return check_device_error(w);
}

int mgx_sweep(mgx_world *w, int32_t robot, uint32_t external_phases, uint32_t internal_phases, uint32_t n_internal,
              uint32_t hints) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    if ((external_phases & ~3u) || (internal_phases & ~3u)) return fail(MGX_ERR_INVALID, "bad phase mask");
    if (hints & ~MGX_HINT_NEXT_STARTS_EXTERNAL) return fail(MGX_ERR_INVALID, "bad hints");
    if (n_internal > 1 && internal_phases != 3u) return fail(MGX_ERR_INVALID, "fused iterations need both internal phases");
    const uint32_t h = ((hints & MGX_HINT_NEXT_STARTS_EXTERNAL) && (external_phases & 1u)) ? HINT_IR_DEAD : 0u;
    return sweep(w, robot, external_phases, internal_phases << 2, internal_phases ? (int)n_internal : 0, h);
}

// the launches of a schedule: phases I / E flattened (robot.rs:1787-1860: internal first, then external, per
// step) and grouped into launches of the form [E] I* (one workgroup-resident pass each)
static std::vector<Launch> plan_launches(const uint8_t *steps, uint32_t n) {
    std::vector<uint8_t> ph;
    for (uint32_t i = 0; i < n; i++) {
        if (steps[i] & MGX_STEP_INTERNAL) ph.push_back('I');
        if (steps[i] & MGX_STEP_EXTERNAL) ph.push_back('E');
    }
    std::vector<Launch> out;
    size_t i = 0;
    while (i < ph.size()) {
        Launch l{0u, 0, 0u};
        if (ph[i] == 'E') { l.ext = PH_EXT_FACTOR | PH_EXT_VARIABLE; i++; }
        while (i < ph.size() && ph[i] == 'I') { l.n_int++; i++; }
        // the next launch of this call (if any) starts with an external phase: the inter-robot messages
        // this launch computes are recomputed before anything reads their HBM copy
        l.hints = (l.ext && i < ph.size()) ? HINT_IR_DEAD : 0u;
        // later launches of this call that run a variable sweep rewrite the belief images (nothing reads them in between)
        for (size_t j = i; j < ph.size(); j++) l.hints |= (ph[j] == 'E') ? HINT_LATER_EXT_VARIABLE : HINT_LATER_INT_VARIABLE;
        out.push_back(l);
    }
    return out;
}

// ---- batches: several schedules, one submission ---------------------------------------------------------------------------
// A resident schedule launch pays for itself once: the robots' graphs go HBM -> LDS when it starts and back when it ends, some
// 12 us of a 94 us launch at 1000 x 16 (stamps: staging 17.6 k + write-back 14.7 k of 242 k clocks).  A caller that issues
// schedule after schedule with nothing in between (a planner that runs ahead, a benchmark loop) can bracket the loop: the
// schedules are recorded and submitted together, merged into as few launches as their segments fit (MAX_SEGS per launch) —
// the engine's form of capturing a launch-bound loop in a graph.  Nothing is reordered and nothing is skipped:
// iterate(a); iterate(b) computes exactly what iterate(a ++ b) computes (the phases are flattened either way), and every other
// call on the world first submits what was recorded (MGX_ENTER), so it finds the world as if each schedule had run when it was
// issued.
static int submit_batch(mgx_world *w) {
    std::vector<uint8_t> st;
    st.swap(w->batch.steps);  // (first: whatever runs below may pass MGX_ENTER again)
    if (st.empty()) return MGX_OK;
    const int rc = iterate_now(w, st.data(), (uint32_t)st.size());
    w->batch.submissions++;
    w->batch.launches += w->last_sweep_launches;
    return rc;
}

int mgx_batch_begin(mgx_world *w) {
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    if (w->batch.open) return fail(MGX_ERR_STATE, "a batch is open already");
    w->batch.open = true;
    w->batch.schedules = w->batch.submissions = w->batch.launches = 0;
    return MGX_OK;
}
int mgx_batch_end(mgx_world *w, uint32_t *n_schedules, uint32_t *n_launches) {
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    if (!w->batch.open) return fail(MGX_ERR_STATE, "no batch is open");
    w->batch.open = false;
    const int rc = submit_batch(w);
    if (n_schedules) *n_schedules = w->batch.schedules;
    if (n_launches) *n_launches = w->batch.launches;
    return rc;
}

int mgx_iterate(mgx_world *w, const uint8_t *steps, uint32_t n) {
    if (!w || (!steps && n)) return fail(MGX_ERR_INVALID, "null argument");
    // (the same checks inside and outside a batch: a call that would fail when it runs fails when it is issued)
    for (uint32_t i = 0; i < n; i++)
        if (steps[i] & ~(MGX_STEP_INTERNAL | MGX_STEP_EXTERNAL)) return fail(MGX_ERR_INVALID, "bad step %u", i);
    if (!w->batch.open) return iterate_now(w, steps, n);
    if (w->robots.empty()) return fail(MGX_ERR_STATE, "world has no robots");
    int rc0 = check_device_error(w);
    if (rc0 != MGX_OK) return rc0;
    mgx_world::Batch &b = w->batch;
    if (!b.steps.empty()) {  // does it still fit the launch the recorded ones make?
        std::vector<uint8_t> both(b.steps);
        both.insert(both.end(), steps, steps + n);
        if (plan_launches(both.data(), (uint32_t)both.size()).size() > (size_t)MAX_SEGS) {
            const int rc = submit_batch(w);
            if (rc != MGX_OK) return rc;
        }
    }
    b.steps.insert(b.steps.end(), steps, steps + n);
    b.schedules++;  // (recorded: a schedule whose predecessors' submission failed above is not counted)
    return MGX_OK;
}

static int iterate_now(mgx_world *w, const uint8_t *steps, uint32_t n) {
    const std::vector<Launch> plan = plan_launches(steps, n);
    if (w->pending.active) {  // (a declined launch is run again here: not this call's launches)
        w->linger.hold = true;
        const int rcc = confirm_resident(w);
        w->linger.hold = false;
        if (rcc != MGX_OK) return rcc;
    }
    w->last_sweep_launches = 0;
    w->linger.streak++;  // (every other entry point resets it: MGX_ENTER)
    const int resident = run_resident(w, plan);
    if (resident != 0) return resident < 0 ? resident : MGX_OK;
    for (const Launch &l : plan) {
        int rc = sweep(w, -1, l.ext, l.n_int ? (PH_INT_FACTOR | PH_INT_VARIABLE) : 0, l.n_int, l.hints);
        if (rc != MGX_OK) return rc;
    }
    return MGX_OK;
}

int mgx_internal_factor_iteration(mgx_world *w, int32_t robot) {
    MGX_ENTER(w);
    return w ? sweep(w, robot, 0, PH_INT_FACTOR, 1) : fail(MGX_ERR_INVALID, "null world");
}
int mgx_internal_variable_iteration(mgx_world *w, int32_t robot) {
    MGX_ENTER(w);
    return w ? sweep(w, robot, 0, PH_INT_VARIABLE, 1) : fail(MGX_ERR_INVALID, "null world");
}
int mgx_external_factor_iteration(mgx_world *w, int32_t robot) {
    MGX_ENTER(w);
    return w ? sweep(w, robot, PH_EXT_FACTOR, 0, 0) : fail(MGX_ERR_INVALID, "null world");
}
int mgx_external_variable_iteration(mgx_world *w, int32_t robot) {
    MGX_ENTER(w);
    return w ? sweep(w, robot, PH_EXT_VARIABLE, 0, 0) : fail(MGX_ERR_INVALID, "null world");
}

int mgx_change_priors(mgx_world *w, uint32_t n, const int32_t *robots, const uint32_t *var_ix, const double *means) {
    MGX_ENTER(w);
    if (!w || !robots || !var_ix || !means) return fail(MGX_ERR_INVALID, "null argument");
    if (n == 0) return MGX_OK;
    for (uint32_t i = 0; i < n; i++) {
        if (robots[i] < 0 || (size_t)robots[i] >= w->robots.size() || w->robots[(size_t)robots[i]].ghost || w->robots[(size_t)robots[i]].removed || (int)var_ix[i] >= w->K)
            return fail(MGX_ERR_INVALID, "bad (robot, variable) at %u", i);
    }
    int rc = commit(w);
    if (rc != MGX_OK) return rc;
    // packed arguments, f64 words: means[4n] | device robot (int32)[n] | variable (uint32)[n]
    const size_t words = 4 * (size_t)n + (n + 1) / 2 + (n + 1) / 2;
    void *hp = nullptr;
    int slot = 0;
    HIP_TRY(w->stage.acquire(words * sizeof(double), &hp, &slot));
    double *hm = (double *)hp;
    int32_t *hr = (int32_t *)(hm + 4 * (size_t)n);
    uint32_t *hv = (uint32_t *)(hm + 4 * (size_t)n + (n + 1) / 2);
    memcpy(hm, means, 4 * (size_t)n * sizeof(double));
    for (uint32_t i = 0; i < n; i++) { hr[i] = w->dev_of[(size_t)robots[i]]; hv[i] = var_ix[i]; log_change_prior(w, robots[i], (int)var_ix[i]); }
    void *dp = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dp, hp, 0));
    const double *dm = (const double *)dp;
    w->stale_kinds |= ~w->p.enable_mask & 15u;
    HIP_TRY(launch_change_prior(w->d, (int)n, (const int32_t *)(dm + 4 * (size_t)n), (const uint32_t *)(dm + 4 * (size_t)n + (n + 1) / 2), dm,
                                w->stream));
    HIP_TRY(w->stage.release(slot, w->stream));
    return MGX_OK;
}
int mgx_update_priors(mgx_world *w, uint32_t n, const int32_t *robots, const double *waypoints_xy, const double *time_scale,
                      const uint8_t *what, double max_speed, double delta_t) {
    MGX_ENTER(w);
    if (!w || !robots || !waypoints_xy || !time_scale || !what) return fail(MGX_ERR_INVALID, "null argument");
    if (n == 0) return MGX_OK;
    if (w->K < 3) return fail(MGX_ERR_INVALID, "needs K >= 3");
    for (uint32_t i = 0; i < n; i++)
        if (robots[i] < 0 || (size_t)robots[i] >= w->robots.size() || w->robots[(size_t)robots[i]].ghost || w->robots[(size_t)robots[i]].removed || (what[i] & ~3u))
            return fail(MGX_ERR_INVALID, "bad entry %u", i);
    int rc = commit(w);
    if (rc != MGX_OK) return rc;
    // packed arguments, f64 words: waypoints[2n] | time_scale[n] | device robot (int32)[n] | what (u8)[n]
    const size_t w_r = (n + 1) / 2, w_w = (n + 7) / 8, words = 3 * (size_t)n + w_r + w_w;
    void *hp = nullptr;
    int slot = 0;
    HIP_TRY(w->stage.acquire(words * sizeof(double), &hp, &slot));
    double *hw = (double *)hp;
    int32_t *hr = (int32_t *)(hw + 3 * (size_t)n);
    uint8_t *hh = (uint8_t *)(hw + 3 * (size_t)n + w_r);
    memcpy(hw, waypoints_xy, 2 * (size_t)n * sizeof(double));
    memcpy(hw + 2 * (size_t)n, time_scale, (size_t)n * sizeof(double));
    for (uint32_t i = 0; i < n; i++) {
        hr[i] = w->dev_of[(size_t)robots[i]];
        if (what[i] & 1u) log_change_prior(w, robots[i], w->K - 1);
        if (what[i] & 2u) log_change_prior(w, robots[i], 0);
    }
    memcpy(hh, what, n);
    void *dp = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dp, hp, 0));
    const double *dw = (const double *)dp;
    w->stale_kinds |= ~w->p.enable_mask & 15u;
    HIP_TRY(launch_update_priors(w->d, (int)n, (const int32_t *)(dw + 3 * (size_t)n), dw, dw + 2 * (size_t)n,
                                 (const uint8_t *)(dw + 3 * (size_t)n + w_r), max_speed, delta_t, w->stream));
    HIP_TRY(w->stage.release(slot, w->stream));
    return MGX_OK;
}

// One driver tick in one call: update_prior_of_horizon_state + update_prior_of_current_state_v3 for the listed
// robots, then iterate_gbp_v2 (robot.rs:86-103).  When the schedule opens with an internal iteration — every
// schedule of the reference does — the two prior updates ride in the launch that runs it: each robot's
// workgroup applies them to the image it has just staged, which saves the separate kernel and its trip
// through HBM.  Otherwise (or while factors are thawing) this is mgx_update_priors followed by mgx_iterate.
int mgx_tick(mgx_world *w, uint32_t n, const int32_t *robots, const double *waypoints_xy, const double *time_scale,
             const uint8_t *what, double max_speed, double delta_t, const uint8_t *steps, uint32_t n_steps) {
    MGX_ENTER_SCHEDULE(w);
    if (!w || (!steps && n_steps) || (n && (!robots || !waypoints_xy || !time_scale || !what))) return fail(MGX_ERR_INVALID, "null argument");
    const std::vector<Launch> plan = plan_launches(steps, n_steps);
    if (w->pending.active) {  // (a declined launch is run again here: not this call's launches)
        w->linger.hold = true;
        const int rcc = confirm_resident(w);
        w->linger.hold = false;
        if (rcc != MGX_OK) return rcc;
    }
    w->last_sweep_launches = 0;
    const bool fuse = n > 0 && !plan.empty() && plan[0].ext == 0 && plan[0].n_int > 0 && w->thaw_kinds == 0 && w->K >= 3;
    if (!fuse) {
        const int rc = n ? mgx_update_priors(w, n, robots, waypoints_xy, time_scale, what, max_speed, delta_t) : MGX_OK;
        return rc != MGX_OK ? rc : iterate_now(w, steps, n_steps);
    }
    for (uint32_t i = 0; i < n; i++)  // (the robots' ghost / removed flags from their compact copies)
        if (robots[i] < 0 || (size_t)robots[i] >= w->robots.size() || w->sets.ghost[(size_t)robots[i]] || w->sets.removed[(size_t)robots[i]] || (what[i] & ~3u))
            return fail(MGX_ERR_INVALID, "bad entry %u", i);
    StageTimer tmk("tick");
    w->linger.hold = true;  // (a lingering launch stays: this tick is posted into it if it qualifies)
    int rc = commit(w);
    w->linger.hold = false;
    tmk.lap("commit (confirm + table rebuild)");
    if (rc != MGX_OK) return rc;
    const size_t RL = (size_t)w->d.R_local;
    void *hp = nullptr, *dp = nullptr;
    int slot = 0;
    // a launch lingers and takes this tick: the records go straight into the coming number's slot of its box (the pinned ring's
    // slots are guarded by events, and an event behind a launch that lingers does not complete)
    int posting = 0;
    if (w->linger.open && (posting = linger_prepare_post(w, plan)) < 0) return posting;
    double *rec = nullptr;
    if (posting) {
        rec = linger_upd_slot(w, w->launch_seq + 1ull);
    } else {
        HIP_TRY(w->stage.acquire(4 * RL * sizeof(double), &hp, &slot));
        rec = (double *)hp;
    }
    std::fill(rec, rec + 4 * RL, 0.0);
    for (uint32_t i = 0; i < n; i++) {
        double *q = rec + 4 * (size_t)w->dev_of[(size_t)robots[i]];
        q[0] = waypoints_xy[2 * i]; q[1] = waypoints_xy[2 * i + 1]; q[2] = time_scale[i]; q[3] = (double)what[i];
        if (what[i] & 1u) log_change_prior(w, robots[i], w->K - 1);
        if (what[i] & 2u) log_change_prior(w, robots[i], 0);
    }
    w->linger.streak++;  // (every other entry point resets it: MGX_ENTER)
    if (posting) {
        (void)linger_post(w, plan, true, max_speed, delta_t);
        tmk.lap("update records + counter log + post");
        return MGX_OK;
    }
    HIP_TRY(hipHostGetDevicePointer(&dp, hp, 0));
    w->stale_kinds |= ~w->p.enable_mask & 15u;
    tmk.lap("update records + counter log");
    {   // the whole tick as one resident launch when the world qualifies: the prior updates ride in it all the same
        w->d.upd = (const double *)dp; w->d.upd_max_speed = max_speed; w->d.upd_delta_t = delta_t;
        w->upd_ring_slot = slot;
        w->upd_host = rec;
        const int resident = run_resident(w, plan);
        w->upd_host = nullptr;
        w->upd_ring_slot = -1;
        tmk.lap("resident launch enqueued");
        if (resident != 0) {
            w->d.upd = nullptr;
            hipError_t e = w->stage.release(slot, w->stream);
            if (resident < 0) return resident;
            return e == hipSuccess ? MGX_OK : fail(MGX_ERR_HIP, "event record: %s", hipGetErrorString(e));
        }
        w->d.upd = nullptr;
    }
    bool first = true;
    for (const Launch &l : plan) {
        if (first) { w->d.upd = (const double *)dp; w->d.upd_max_speed = max_speed; w->d.upd_delta_t = delta_t; }
        rc = sweep(w, -1, l.ext, l.n_int ? (PH_INT_FACTOR | PH_INT_VARIABLE) : 0, l.n_int, l.hints);
        if (first) {
            w->d.upd = nullptr;
            first = false;
            hipError_t e = w->stage.release(slot, w->stream);
            if (rc == MGX_OK && e != hipSuccess) rc = fail(MGX_ERR_HIP, "event record: %s", hipGetErrorString(e));
        }
        if (rc != MGX_OK) return rc;
    }
    return MGX_OK;
}

// FactorGraph::reset_variables (factorgraph.rs:1541-1564: VariableNode::reset, variable.rs:350-360, for every variable, then
// FactorNode::empty_inbox, factor/mod.rs:480-483, for every factor of the graph).  In the engine's terms:
//   belief mean / precision  <- the given mean / diag(sigma) (the reference uses the sigma AS the precision's diagonal, +inf
//                               included); information vector, covariance, validity and the prior stay;
//   variable inboxes emptied <- own factor -> variable messages and the messages of foreign inter-robot factors attached to
//                               these variables become the empty (= zero) message;
//   own factors' inboxes emptied <- the variables' delivery counts restart at zero (an own factor's inbox entry is "present"
//                               once its variable has delivered; a tracking factor reads the record's mean even before, so
//                               that becomes zero as well), and for the inter-robot factors this graph owns the entry from
//                               the other robot's variable (its last response mean) is emptied and their creation epoch
//                               restarts with the count.
// A rare call (the reference makes it when a global path has been found, robot.rs:700-790): the device state is pulled,
// edited on the host mirror and laid out again by the next launch.
int mgx_reset_variables(mgx_world *w, int32_t robot, const double *means, uint32_t n_means, double first_last_sigma, double inbetween_sigma) {
    MGX_ENTER(w);
    if (!w || !means || robot < 0 || (size_t)robot >= w->robots.size()) return fail(MGX_ERR_INVALID, "bad argument");
    if (w->robots[(size_t)robot].ghost || w->robots[(size_t)robot].removed) return fail(MGX_ERR_INVALID, "robot %d is not a live local robot", robot);
    if ((int)n_means != w->robots[(size_t)robot].K)  // factorgraph.rs:1548 asserts variable_indices.len() == means.len()
        return fail(MGX_ERR_INVALID, "%u means for a graph of %d variables", n_means, w->robots[(size_t)robot].K);
    int rc = pull(w);
    if (rc != MGX_OK) return rc;
    flush_counts(w);
    Robot &rb = w->robots[(size_t)robot];
    const int K = rb.K, E = 4 * K - 6;
    for (int i = 0; i < K; i++) {
        const double sigma = (i == 0 || i == K - 1) ? first_last_sigma : inbetween_sigma;
        for (int c = 0; c < 4; c++) rb.bel_mu[4 * i + c] = means[4 * i + c];
        for (int c = 0; c < 16; c++) rb.bel_lam[16 * i + c] = (c % 5 == 0) ? sigma : 0.0;
        rb.epoch[(size_t)i] = 0;
        for (int c = 0; c < 4; c++) rb.snap[24 * i + 20 + c] = 0.0;
    }
    std::fill(rb.fv_eta.begin(), rb.fv_eta.begin() + 4 * E, 0.0);
    std::fill(rb.fv_lam.begin(), rb.fv_lam.begin() + 16 * E, 0.0);
    for (IrConn &c : w->conns) {
        if (c.other == robot)  // foreign factors attached to these variables: their message to us is emptied
            for (IrEdge &ed : c.edges) { std::fill(ed.fv_eta, ed.fv_eta + 4, 0.0); std::fill(ed.fv_lam, ed.fv_lam + 16, 0.0); }
        if (c.owner == robot)  // own inter-robot factors: both inbox entries emptied
            for (IrEdge &ed : c.edges) { std::fill(ed.bmu, ed.bmu + 4, 0.0); ed.created = 0; ed.fresh = false; }
    }
    w->dirty = true;
    w->dev_valid = false;  // the host mirror is the truth now: the next launch lays it out again (and must not pull over it)
    return MGX_OK;
}
// FactorGraph::reset_tracking_factors (factorgraph.rs:1566-1590): set_timeout(10) on every tracking factor of the graph —
// its next ten updates are skipped (tracking.rs:362-371) and send the empty message.
int mgx_reset_tracking_factors(mgx_world *w, int32_t robot) {
    MGX_ENTER(w);
    if (!w || robot < 0 || (size_t)robot >= w->robots.size()) return fail(MGX_ERR_INVALID, "bad argument");
    if (w->robots[(size_t)robot].ghost || w->robots[(size_t)robot].removed) return fail(MGX_ERR_INVALID, "robot %d is not a live local robot", robot);
    int rc = pull(w);
    if (rc != MGX_OK) return rc;
    Robot &rb = w->robots[(size_t)robot];
    for (int32_t &rec : rb.trk_record) rec = (rec & 0xffff) | (11 << 16);  // Some(10), gbp_math.h tracking_timeout_skips
    w->dirty = true;
    w->dev_valid = false;
    return MGX_OK;
}

// Sharded worlds: a prior change applied on ANOTHER rank (to a robot that is a ghost here) still delivers a message to the
// inter-robot factors local robots own on that variable (variable.rs:210-221): the counters are told, nothing else happens.
int mgx_note_change_priors(mgx_world *w, uint32_t n, const int32_t *robots, const uint32_t *var_ix) {
    MGX_ENTER(w);
    if (!w || (n && (!robots || !var_ix))) return fail(MGX_ERR_INVALID, "null argument");
    for (uint32_t i = 0; i < n; i++)
        if (robots[i] < 0 || (size_t)robots[i] >= w->robots.size() || (int)var_ix[i] >= w->K) return fail(MGX_ERR_INVALID, "bad (robot, variable) at %u", i);
    for (uint32_t i = 0; i < n; i++)
        if (!w->robots[(size_t)robots[i]].removed) log_change_prior(w, robots[i], (int)var_ix[i]);
    return MGX_OK;
}

int mgx_change_prior(mgx_world *w, int32_t robot, uint32_t var_ix, const double mean[4]) {
    MGX_ENTER(w);
    return mgx_change_priors(w, 1, &robot, &var_ix, mean);
}

int mgx_set_resident_launches(mgx_world *w, int32_t enabled) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    w->resident_off = enabled == 0;
    w->resident_decline = enabled == 2;
    return MGX_OK;
}
int mgx_is_thawing(mgx_world *w, int32_t *thawing) {
    MGX_ENTER(w);
    if (!w || !thawing) return fail(MGX_ERR_INVALID, "null argument");
    *thawing = (w->thaw_kinds || w->ir_thaw_active || w->n_keyless > 0) ? 1 : 0;
    return MGX_OK;
}
int mgx_last_launch_count(mgx_world *w, uint32_t *n_launches) {
    MGX_ENTER_SCHEDULE(w);
    if (!w || !n_launches) return fail(MGX_ERR_INVALID, "null argument");
    if (w->pending.active) { const int rcc = confirm_resident(w); if (rcc != MGX_OK) return rcc; }
    *n_launches = w->last_sweep_launches;
    return MGX_OK;
}

int mgx_num_robots(mgx_world *w, uint32_t *n_robots, uint32_t *n_variables) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    uint32_t nr = 0;
    for (const Robot &r : w->robots) nr += r.ghost ? 0 : 1;
    if (n_robots) *n_robots = nr;
    if (n_variables) *n_variables = nr * (uint32_t)w->K;
    return MGX_OK;
}

// the mean of variable `var_ix` of every local robot, id order: what the driver reads every tick
// (nth_variable(0) / last_variable for reached_waypoint, robot.rs:2125-2136; variables 0 and 1 for the
// Transform increment, robot.rs:2309-2330) — gathered on the device straight into pinned memory
int mgx_read_variable_means(mgx_world *w, uint32_t var_ix, double *means) {
    MGX_ENTER(w);
    if (!w || !means) return fail(MGX_ERR_INVALID, "null argument");
    int rc = commit(w);
    if (rc != MGX_OK) return rc;
    if ((int)var_ix >= w->K) return fail(MGX_ERR_INVALID, "variable index out of range");
    const size_t bytes = (size_t)w->d.R_local * 4 * sizeof(double);
    void *hp = nullptr, *dp = nullptr;
    int slot = 0;
    HIP_TRY(w->stage.acquire(bytes, &hp, &slot));
    HIP_TRY(hipHostGetDevicePointer(&dp, hp, 0));
    HIP_TRY(launch_gather_variable_means(w->d, (int)var_ix, (double *)dp, w->stream));
    HIP_TRY(w->stage.release(slot, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    memcpy(means, hp, bytes);
    return check_device_error(w);
}

// FactorGraph::messages_sent / messages_received (factorgraph.rs:876-890) of one robot's graph
int mgx_message_counts(mgx_world *w, int32_t robot, uint64_t counts[4]) {
    MGX_ENTER(w);
    if (!w || !counts || robot < 0 || (size_t)robot >= w->robots.size()) return fail(MGX_ERR_INVALID, "bad argument");
    if (w->pending.active) { const int rcc = confirm_resident(w); if (rcc != MGX_OK) return rcc; }
    if (w->robots[(size_t)robot].ghost) return fail(MGX_ERR_INVALID, "robot %d is a ghost here: its graph is counted on the rank that owns it", robot);
    flush_counts(w);
    const Robot &rb = w->robots[(size_t)robot];
    for (int c = 0; c < 4; c++) counts[c] = rb.cnt[c];
    for (const IrConn &cn : w->conns) {
        if (cn.owner != robot) continue;
        for (int c = 0; c < 4; c++) counts[c] += cn.cnt[c];
    }
    return MGX_OK;
}

// bulk read of the belief means only (what reached_waypoint and the visualisers read, robot.rs:2125-2136)
int mgx_read_means(mgx_world *w, double *means) {
    MGX_ENTER(w);
    if (!w || !means) return fail(MGX_ERR_INVALID, "null argument");
    return mgx_read_beliefs(w, nullptr, nullptr, means);
}


int mgx_read_beliefs(mgx_world *w, double *eta, double *lam, double *means) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    int rc = commit(w);
    if (rc != MGX_OK) return rc;
    const int K = w->K;
    const BlobLayout L(K);
    const size_t BS = (size_t)w->d.BS, R = (size_t)w->d.R_local;
    // strided device -> host copies of the belief (eta, lam) and mean rows of every local blob
    std::vector<double> bel(R * 20 * K), mu(R * 4 * K);
    if (eta || lam)
        HIP_TRY(hipMemcpy2DAsync(bel.data(), 20 * K * sizeof(double), w->blob.p + L.bel(), BS * sizeof(double),
                                 20 * K * sizeof(double), R, hipMemcpyDeviceToHost, w->stream));
    if (means)
        HIP_TRY(hipMemcpy2DAsync(mu.data(), 4 * K * sizeof(double), w->blob.p + L.mu(), BS * sizeof(double),
                                 4 * K * sizeof(double), R, hipMemcpyDeviceToHost, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    if ((rc = check_device_error(w)) != MGX_OK) return rc;
    // device local order == id order of non-ghost robots
    for (size_t r = 0; r < R; r++)
        for (int i = 0; i < K; i++) {
            const size_t v = r * K + i;
            if (eta)
                for (int c = 0; c < 4; c++) eta[4 * v + c] = bel[r * 20 * K + c * K + i];
            if (lam)
                for (int c = 0; c < 16; c++) lam[16 * v + c] = bel[r * 20 * K + (4 + c) * K + i];
            if (means)
                for (int c = 0; c < 4; c++) means[4 * v + c] = mu[r * 4 * K + c * K + i];
        }
    return MGX_OK;
}

int mgx_get_belief(mgx_world *w, int32_t robot, uint32_t var_ix, double eta[4], double lam[16], double mean[4], double cov[16],
                   int32_t *valid) {
    MGX_ENTER(w);
    if (!w || robot < 0 || (size_t)robot >= w->robots.size() || (int)var_ix >= w->K) return fail(MGX_ERR_INVALID, "bad (robot, variable)");
    int rc = commit(w);
    if (rc != MGX_OK) return rc;
    const int K = w->K, i = (int)var_ix;
    const BlobLayout L(K);
    std::vector<double> b((size_t)w->d.BS);
    HIP_TRY(hipMemcpyAsync(b.data(), w->blob.p + (size_t)w->dev_of[(size_t)robot] * w->d.BS, sizeof(double) * b.size(),
                           hipMemcpyDeviceToHost, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    if ((rc = check_device_error(w)) != MGX_OK) return rc;
    if (eta)
        for (int c = 0; c < 4; c++) eta[c] = b[L.bel() + c * K + i];
    if (lam)
        for (int c = 0; c < 16; c++) lam[c] = b[L.bel() + (4 + c) * K + i];
    if (mean)
        for (int c = 0; c < 4; c++) mean[c] = b[L.mu() + c * K + i];
    if (cov)
        for (int c = 0; c < 16; c++) cov[c] = b[L.cov() + c * K + i];
    if (valid) *valid = reinterpret_cast<const int32_t *>(b.data() + L.valid())[i];
    return MGX_OK;
}

uint32_t mgx_halo_words(uint32_t K) { return (SNAP_W + 1) * K; }

int mgx_halo_plan(mgx_world *w, uint32_t n_send, const int32_t *send_robots, uint32_t n_recv, const int32_t *recv_ghosts) {
    MGX_ENTER(w);
    if (!w || (n_send && !send_robots) || (n_recv && !recv_ghosts)) return fail(MGX_ERR_INVALID, "null argument");
    for (uint32_t i = 0; i < n_send; i++)
        if (send_robots[i] < 0 || (size_t)send_robots[i] >= w->robots.size() || w->robots[(size_t)send_robots[i]].ghost)
            return fail(MGX_ERR_INVALID, "send list entry %u is not a local robot", i);
    for (uint32_t i = 0; i < n_recv; i++)
        if (recv_ghosts[i] < 0 || (size_t)recv_ghosts[i] >= w->robots.size() || !w->robots[(size_t)recv_ghosts[i]].ghost)
            return fail(MGX_ERR_INVALID, "receive list entry %u is not a ghost robot", i);
    w->halo_send.assign(send_robots, send_robots + n_send);
    w->halo_recv.assign(recv_ghosts, recv_ghosts + n_recv);
    w->halo_dirty = true;
    w->direct.aimed = false;
    return MGX_OK;
}

// The exchange lists a rank needs for the connections it holds: every connection A -> B is evaluated on
// the rank of B, which needs A's snapshot records.  With the replicated bookkeeping of a sharded world
// that follows its topology (all robots present everywhere) both ends derive the same lists.
int mgx_halo_plan_from_connections(mgx_world *w, const int32_t *rank_of, uint32_t n_robots, int32_t my_rank, uint32_t n_ranks,
                                   uint32_t *send_counts, uint32_t *recv_counts) {
    MGX_ENTER(w);
    if (!w || !rank_of || !send_counts || !recv_counts) return fail(MGX_ERR_INVALID, "null argument");
    if (n_robots != w->robots.size() || my_rank < 0 || (uint32_t)my_rank >= n_ranks) return fail(MGX_ERR_INVALID, "bad rank table");
    for (uint32_t r = 0; r < n_robots; r++) {
        if (rank_of[r] < 0 || (uint32_t)rank_of[r] >= n_ranks) return fail(MGX_ERR_INVALID, "robot %u: bad rank %d", r, rank_of[r]);
        if ((rank_of[r] != my_rank) != w->robots[r].ghost) return fail(MGX_ERR_INVALID, "robot %u: rank table and ghost flag disagree", r);
    }
    std::vector<std::vector<int32_t>> send(n_ranks), recv(n_ranks);
    for (const IrConn &c : w->conns) {
        const int ro = rank_of[(size_t)c.owner], rt = rank_of[(size_t)c.other];
        if (ro == rt) continue;
        if (rt == my_rank) recv[(size_t)ro].push_back(c.owner);
        else if (ro == my_rank) send[(size_t)rt].push_back(c.owner);
    }
    w->halo_send.clear();
    w->halo_recv.clear();
    for (uint32_t p = 0; p < n_ranks; p++) {
        for (std::vector<int32_t> *l : {&send[p], &recv[p]}) {
            std::sort(l->begin(), l->end());
            l->erase(std::unique(l->begin(), l->end()), l->end());
        }
        send_counts[p] = (uint32_t)send[p].size();
        recv_counts[p] = (uint32_t)recv[p].size();
        w->halo_send.insert(w->halo_send.end(), send[p].begin(), send[p].end());
        w->halo_recv.insert(w->halo_recv.end(), recv[p].begin(), recv[p].end());
    }
    w->halo_dirty = true;
    w->direct.aimed = false;
    return MGX_OK;
}

// the exchange lists as they stand (robot ids; send list by consumer rank, receive list by producer rank — the order of the counts
// mgx_halo_plan_from_connections returned): n_send / n_recv receive the lengths, the arrays are filled up to their capacities
int mgx_halo_get_lists(mgx_world *w, int32_t *send_robots, uint32_t send_capacity, int32_t *recv_robots, uint32_t recv_capacity,
                       uint32_t *n_send, uint32_t *n_recv) {
    MGX_ENTER(w);
    if (!w || !n_send || !n_recv) return fail(MGX_ERR_INVALID, "null argument");
    *n_send = (uint32_t)w->halo_send.size();
    *n_recv = (uint32_t)w->halo_recv.size();
    if (send_robots) for (size_t i = 0; i < w->halo_send.size() && i < send_capacity; i++) send_robots[i] = w->halo_send[i];
    if (recv_robots) for (size_t i = 0; i < w->halo_recv.size() && i < recv_capacity; i++) recv_robots[i] = w->halo_recv[i];
    return MGX_OK;
}

// ---- migration: a robot changes its owning rank (worlds that follow their topology) --------------------------------------
// Everything that exists on the owner's rank ONLY travels in one flat record: the graph's numeric state (priors, beliefs,
// factor -> variable messages, snapshot records and delivery counts, tracking records, iteration count, the path, the
// frozen inboxes of switched-off kinds), the totals of its own MessageCount, and the state of every inter-robot factor
// attached to its variables (kept at the TARGET's rank).  The replicated bookkeeping — connection sets, node slots, robot
// numbers, flags, the connections' counters — is the same on every rank already and stays where it is.
}  // extern "C" (the record's writer and reader are templates)
namespace {
struct MigWriter {
    std::vector<uint8_t> b;
    template <class T> void pod(const T &v) { const uint8_t *p = reinterpret_cast<const uint8_t *>(&v); b.insert(b.end(), p, p + sizeof(T)); }
    template <class T> void vec(const std::vector<T> &v) {
        pod((uint64_t)v.size());
        const uint8_t *p = reinterpret_cast<const uint8_t *>(v.data());
        b.insert(b.end(), p, p + sizeof(T) * v.size());
        while (b.size() & 7) b.push_back(0);
    }
};
struct MigReader {
    const uint8_t *p, *end;
    bool ok = true;
    template <class T> void pod(T &v) {
        if (!ok || (size_t)(end - p) < sizeof(T)) { ok = false; return; }
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
    }
    template <class T> void vec(std::vector<T> &v, size_t want = (size_t)-1) {
        uint64_t n = 0;
        pod(n);
        if (!ok || n > (uint64_t)(end - p) / sizeof(T) || (want != (size_t)-1 && n != want)) { ok = false; return; }
        v.resize((size_t)n);
        if (n) memcpy(v.data(), p, sizeof(T) * (size_t)n);
        p += ((sizeof(T) * (size_t)n + 7) & ~(size_t)7);
        if (p > end) ok = false;
    }
};
constexpr uint32_t MIG_MAGIC = 0x4d47584du;  // "MXGM"
constexpr uint32_t MIG_VERSION = 1;
}  // namespace
extern "C" {

// The record of a robot this rank owns (mgx_robot_import takes it on the rank that is to own it).  `bytes` receives the
// record's size; with buf == NULL or capacity too small nothing is copied (MGX_ERR_INVALID when a buffer was given).
// The world is brought up to date first: schedules issued so far have run, the tables of the last topology pass are laid
// out.  Call it BETWEEN ticks — after the sweeps that followed the last topology pass — on every rank at the same point.
int mgx_robot_export(mgx_world *w, int32_t robot, void *buf, uint64_t capacity, uint64_t *bytes) {
    MGX_ENTER(w);
    if (!w || !bytes || robot < 0 || (size_t)robot >= w->robots.size()) return fail(MGX_ERR_INVALID, "bad argument");
    if (w->robots[(size_t)robot].ghost) return fail(MGX_ERR_INVALID, "robot %d is a ghost here: its owner exports it", robot);
    if ((size_t)robot < w->mission.has.size() && w->mission.has[(size_t)robot])
        return fail(MGX_ERR_STATE, "robot %d has a mission on this device (mgx_mission_set): missions do not migrate — routes, next waypoints "
                                   "and Transforms are state of unsharded worlds (include/mgx.h)", robot);
    int rc = check_device_error(w);
    if (rc != MGX_OK) return rc;
    if (w->dev_valid && (w->dirty || w->conns_dirty || w->flags_dirty)) {  // (not valid: the host mirror is the truth already)
        rc = commit(w);
        if (rc != MGX_OK) return rc;
    }
    rc = pull(w);
    if (rc != MGX_OK) return rc;
    flush_counts(w);
    const Robot &rb = w->robots[(size_t)robot];
    MigWriter wr;
    wr.pod(MIG_MAGIC); wr.pod(MIG_VERSION);
    wr.pod((int32_t)rb.K); wr.pod((int32_t)robot);
    wr.pod((uint64_t)rb.order_key);
    wr.pod((int32_t)rb.iter_factor); wr.pod((int32_t)rb.thaw);
    for (int q = 0; q < 4; q++) wr.pod((uint64_t)rb.cnt[q]);
    wr.vec(rb.prior_eta); wr.vec(rb.prior_lam); wr.vec(rb.bel_eta); wr.vec(rb.bel_lam); wr.vec(rb.bel_mu); wr.vec(rb.bel_cov);
    wr.vec(rb.valid); wr.vec(rb.snap); wr.vec(rb.epoch); wr.vec(rb.fv_eta); wr.vec(rb.fv_lam);
    wr.vec(rb.trk_record); wr.vec(rb.trk_last_pos); wr.vec(rb.trk_last_val); wr.vec(rb.path);
    wr.vec(rb.frozen); wr.vec(rb.frozen_flag); wr.vec(rb.ir_frozen_snap); wr.vec(rb.ir_frozen_epoch); wr.vec(rb.ir_thaw_epoch);
    uint32_t n_conn = 0;
    for (const IrConn &c : w->conns) n_conn += c.other == robot ? 1u : 0u;
    wr.pod(n_conn); wr.pod((uint32_t)0);
    for (const IrConn &c : w->conns) {
        if (c.other != robot) continue;
        wr.pod((int32_t)c.owner); wr.pod((int32_t)c.node_first);
        wr.pod((uint64_t)c.first_number);
        for (const IrEdge &ed : c.edges) {
            for (double v : ed.fv_eta) wr.pod(v);
            for (double v : ed.fv_lam) wr.pod(v);
            for (double v : ed.bmu) wr.pod(v);
            wr.pod((uint32_t)ed.created); wr.pod((uint32_t)(ed.fresh ? 1 : 0));
        }
    }
    *bytes = (uint64_t)wr.b.size();
    if (!buf) return MGX_OK;
    if (capacity < wr.b.size()) return fail(MGX_ERR_INVALID, "the record of robot %d takes %zu bytes, %llu given", robot, wr.b.size(), (unsigned long long)capacity);
    memcpy(buf, wr.b.data(), wr.b.size());
    return MGX_OK;
}

// The robot — a ghost here so far — becomes this rank's: its graph and the factors attached to its variables take the
// state of the record.  The device state is pulled, the host mirror edited, and the next launch lays the world out again
// (locals first: device indices change, so every wiring that names them — exchange lists, direct / resident halo — is
// made again by the launcher, as after mgx_robot_add).  The replicated bookkeeping has to be in step with the exporting
// rank's: the record names its connections (owner, first robot number, first node slot) and a mismatch is refused.
int mgx_robot_import(mgx_world *w, int32_t robot, const void *buf, uint64_t bytes) {
    MGX_ENTER(w);
    if (!w || !buf || robot < 0 || (size_t)robot >= w->robots.size()) return fail(MGX_ERR_INVALID, "bad argument");
    if (!w->robots[(size_t)robot].ghost) return fail(MGX_ERR_STATE, "robot %d is owned here already", robot);
    if ((size_t)robot < w->mission.has.size() && w->mission.has[(size_t)robot])
        return fail(MGX_ERR_STATE, "robot %d has a mission on this device (mgx_mission_set): missions do not migrate — routes, next waypoints "
                                   "and Transforms are state of unsharded worlds (include/mgx.h)", robot);
    int rc = check_device_error(w);
    if (rc != MGX_OK) return rc;
    if (w->dev_valid && (w->dirty || w->conns_dirty || w->flags_dirty)) {
        rc = commit(w);
        if (rc != MGX_OK) return rc;
    }
    rc = pull(w);
    if (rc != MGX_OK) return rc;
    flush_counts(w);
    MigReader rd{(const uint8_t *)buf, (const uint8_t *)buf + bytes};
    uint32_t magic = 0, version = 0, n_conn = 0, pad = 0;
    int32_t K = 0, id = 0, itf = 0, thaw = 0;
    uint64_t key = 0, cnt[4] = {0, 0, 0, 0};
    rd.pod(magic); rd.pod(version); rd.pod(K); rd.pod(id); rd.pod(key); rd.pod(itf); rd.pod(thaw);
    for (uint64_t &c : cnt) rd.pod(c);
    if (!rd.ok || magic != MIG_MAGIC || version != MIG_VERSION) return fail(MGX_ERR_INVALID, "not a robot record of this library version");
    Robot tmp = w->robots[(size_t)robot];  // (edited aside: a record that turns out malformed leaves the world untouched)
    if (K != tmp.K || id != robot || key != tmp.order_key)
        return fail(MGX_ERR_STATE, "the record is robot %d (K = %d, order key %llu): not robot %d of this world", id, K, (unsigned long long)key, robot);
    const size_t Ks = (size_t)K, E = (size_t)(4 * K - 6);
    rd.vec(tmp.prior_eta, 4 * Ks); rd.vec(tmp.prior_lam, 16 * Ks); rd.vec(tmp.bel_eta, 4 * Ks); rd.vec(tmp.bel_lam, 16 * Ks);
    rd.vec(tmp.bel_mu, 4 * Ks); rd.vec(tmp.bel_cov, 16 * Ks);
    rd.vec(tmp.valid, Ks); rd.vec(tmp.snap, 24 * Ks); rd.vec(tmp.epoch, Ks); rd.vec(tmp.fv_eta, 4 * E); rd.vec(tmp.fv_lam, 16 * E);
    rd.vec(tmp.trk_record, Ks - 2); rd.vec(tmp.trk_last_pos, 2 * (Ks - 2)); rd.vec(tmp.trk_last_val, Ks - 2); rd.vec(tmp.path);
    rd.vec(tmp.frozen); rd.vec(tmp.frozen_flag); rd.vec(tmp.ir_frozen_snap); rd.vec(tmp.ir_frozen_epoch); rd.vec(tmp.ir_thaw_epoch);
    rd.pod(n_conn); rd.pod(pad);
    if (!rd.ok || (tmp.path.size() & 1)) return fail(MGX_ERR_INVALID, "malformed robot record");
    std::vector<size_t> mine;
    for (size_t ci = 0; ci < w->conns.size(); ci++)
        if (w->conns[ci].other == robot) mine.push_back(ci);
    if (mine.size() != n_conn)
        return fail(MGX_ERR_STATE, "the record holds %u connections into robot %d, this rank's bookkeeping %zu: the ranks' topology passes are out of step",
                    n_conn, robot, mine.size());
    std::vector<std::vector<IrEdge>> edges(mine.size());
    for (size_t m = 0; m < mine.size(); m++) {
        const IrConn &c = w->conns[mine[m]];
        int32_t owner = 0, node_first = 0;
        uint64_t first_number = 0;
        rd.pod(owner); rd.pod(node_first); rd.pod(first_number);
        if (!rd.ok || owner != c.owner || node_first != c.node_first || first_number != c.first_number)
            return fail(MGX_ERR_STATE, "connection %zu into robot %d differs between the ranks (owner %d / %d): the replicated bookkeeping diverged",
                        m, robot, owner, c.owner);
        edges[m].resize(c.edges.size());
        for (IrEdge &ed : edges[m]) {
            uint32_t created = 0, fresh = 0;
            for (double &v : ed.fv_eta) rd.pod(v);
            for (double &v : ed.fv_lam) rd.pod(v);
            for (double &v : ed.bmu) rd.pod(v);
            rd.pod(created); rd.pod(fresh);
            ed.created = created;
            ed.fresh = fresh != 0;
        }
    }
    if (!rd.ok || rd.p != rd.end) return fail(MGX_ERR_INVALID, "malformed robot record (length)");
    tmp.ghost = false;
    tmp.iter_factor = itf;
    tmp.thaw = (uint8_t)thaw;
    for (int q = 0; q < 4; q++) tmp.cnt[q] = cnt[q];
    w->robots[(size_t)robot] = std::move(tmp);
    for (size_t m = 0; m < mine.size(); m++) {
        bool any_fresh = false;
        for (const IrEdge &ed : edges[m]) any_fresh = any_fresh || ed.fresh;
        w->conns[mine[m]].edges = std::move(edges[m]);
        w->conn_hot[mine[m]].has_fresh = any_fresh ? 1 : 0;
    }
    w->sets.ghost[(size_t)robot] = 0;
    w->dirty = true;
    w->dev_valid = false;  // the host mirror is the truth now (mgx_reset_variables does the same)
    w->conns_dirty = true;
    w->flags_dirty = true;
    return MGX_OK;
}

// The other half on the rank that gave the robot away (after mgx_robot_export): it stays in this world as a ghost — its
// records arrive by the exchange from now on, the factors attached to its variables are its new owner's.
int mgx_robot_release(mgx_world *w, int32_t robot) {
    MGX_ENTER(w);
    if (!w || robot < 0 || (size_t)robot >= w->robots.size()) return fail(MGX_ERR_INVALID, "bad robot id");
    if (w->robots[(size_t)robot].ghost) return fail(MGX_ERR_STATE, "robot %d is a ghost here already", robot);
    if ((size_t)robot < w->mission.has.size() && w->mission.has[(size_t)robot])
        return fail(MGX_ERR_STATE, "robot %d has a mission on this device (mgx_mission_set): missions do not migrate — routes, next waypoints "
                                   "and Transforms are state of unsharded worlds (include/mgx.h)", robot);
    int rc = check_device_error(w);
    if (rc != MGX_OK) return rc;
    if (w->dev_valid && (w->dirty || w->conns_dirty || w->flags_dirty)) {
        rc = commit(w);
        if (rc != MGX_OK) return rc;
    }
    rc = pull(w);
    if (rc != MGX_OK) return rc;
    flush_counts(w);
    Robot &rb = w->robots[(size_t)robot];
    rb.ghost = true;
    rb.path.clear();
    w->sets.ghost[(size_t)robot] = 1;
    w->dirty = true;
    w->dev_valid = false;
    w->conns_dirty = true;
    w->flags_dirty = true;
    return MGX_OK;
}

static int halo_commit(mgx_world *w) {
    if (w->pending.active) { const int rcc = confirm_resident(w); if (rcc != MGX_OK) return rcc; }
    // Changed connections alone (conns_dirty) are left to the next sweep: an exchange does not read the
    // edge tables, and the edges created by a topology pass must find the ghosts' records of the exchange
    // that follows the pass (their creation epoch is the owner's delivery count at that moment).
    if (w->dirty || !w->dev_valid) {
        int rc = commit(w);
        if (rc != MGX_OK) return rc;
    } else if (!device_ok()) {
        return fail(MGX_ERR_NO_DEVICE, "no usable HIP device");
    }
    if (!w->halo_dirty) return MGX_OK;
    std::vector<int32_t> a(w->halo_send.size()), b(w->halo_recv.size());
    for (size_t i = 0; i < a.size(); i++) a[i] = w->dev_of[(size_t)w->halo_send[i]];
    for (size_t i = 0; i < b.size(); i++) b[i] = w->dev_of[(size_t)w->halo_recv[i]];
    HIP_TRY(w->halo_send_dev.upload(a, w->stream));
    HIP_TRY(w->halo_recv_dev.upload(b, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    w->halo_dirty = false;
    return MGX_OK;
}

}  // extern "C" (helpers below have C++ linkage)

// A slot-wired exchange (mgx_halo_direct_setup_slots) is only as good as its last aim: the push walks dst[] by the send list's
// length and the wait indexes the receive area by ghost number, so lists or a layout newer than the aim — or more ghosts than
// slots — are an error here, not a store to wherever the old tables point.
static int direct_slots_ok(mgx_world *w) {
    const mgx_world::DirectHalo &dh = w->direct;
    if (!dh.by_slot) return MGX_OK;
    const size_t NG = (size_t)(w->d.R_total - w->d.R_local);
    if (NG > dh.slot_cap)
        return fail(MGX_ERR_STATE, "%zu ghost robots but the direct exchange was wired with %zu slots: wire it again (mgx_halo_direct_setup_slots)", NG, dh.slot_cap);
    if (!dh.aimed)
        return fail(MGX_ERR_STATE, "the exchange lists or the device layout changed since the direct exchange was aimed: "
                                   "mgx_halo_direct_connect_slots again (robots joined: mgx_halo_direct_setup_slots on every rank first)");
    return MGX_OK;
}
// push: this rank's boundary records go to the consumers (exchange number push_seq + 1);
// wait: the ghosts are filled once every producer has published exchange seq + 1.
static int direct_push(mgx_world *w) {
    int rc = halo_commit(w);
    if (rc != MGX_OK) return rc;
    mgx_world::DirectHalo &dh = w->direct;
    if ((rc = direct_slots_ok(w)) != MGX_OK) return rc;
    if (dh.push_seq != dh.seq) return fail(MGX_ERR_STATE, "exchange %llu is already pushed and not yet waited for", dh.push_seq);
    dh.push_seq += 1;
    const int par = (int)(dh.push_seq & 1ull);
    HIP_TRY(launch_halo_push(w->d, (int)w->halo_send.size(), w->halo_send_dev.p, dh.dst[par].p, dh.n_peers, dh.peer_flags.p, dh.push_seq,
                             dh.done.p, w->stream, dh.by_slot));
    return MGX_OK;
}
static int direct_wait(mgx_world *w) {
    int rc = halo_commit(w);
    if (rc != MGX_OK) return rc;
    mgx_world::DirectHalo &dh = w->direct;
    if ((rc = direct_slots_ok(w)) != MGX_OK) return rc;
    if (dh.push_seq != dh.seq + 1) return fail(MGX_ERR_STATE, "nothing pushed for exchange %llu", dh.seq + 1);
    dh.seq += 1;
    const int par = (int)(dh.seq & 1ull);
    HIP_TRY(launch_halo_wait_unpack(w->d, (int)w->halo_recv.size(), w->halo_recv_dev.p, dh.recv + (size_t)par * dh.recv_words,
                                    dh.n_sources, dh.flags, dh.seq, dh.flags + dh.n_sources, dh.timeout_ticks, dh.ready.p, w->d.sweep_err,
                                    w->stream, dh.by_slot));
    return MGX_OK;
}
static int direct_exchange(mgx_world *w) {
    if (w->direct.push_seq == w->direct.seq) {  // not pushed ahead by the caller
        int rc = direct_push(w);
        if (rc != MGX_OK) return rc;
    }
    return direct_wait(w);
}

// pack -> grouped ncclSend / ncclRecv (the all-to-all-v of boundary snapshots, RCCL over xGMI) -> unpack,
// all enqueued on the world's stream
static int rccl_exchange(mgx_world *w) {
    int rc = halo_commit(w);
    if (rc != MGX_OK) return rc;
    mgx_world::RcclHalo &rh = w->rccl;
    const size_t words = (size_t)mgx_halo_words((uint32_t)w->K);
    HIP_TRY(launch_halo_pack(w->d, (int)w->halo_send.size(), w->halo_send_dev.p, rh.send_buf.p, w->stream));
    int e = g_rccl.group_start();
    for (size_t p = 0; p < rh.peer_rank.size() && e == 0; p++) {
        const size_t ns = (size_t)(rh.send_first[p + 1] - rh.send_first[p]) * words, nr = (size_t)(rh.recv_first[p + 1] - rh.recv_first[p]) * words;
        if (ns) e = g_rccl.send(rh.send_buf.p + (size_t)rh.send_first[p] * words, ns, NCCL_FLOAT64, rh.peer_rank[p], rh.comm, w->stream);
        if (nr && e == 0) e = g_rccl.recv(rh.recv_buf.p + (size_t)rh.recv_first[p] * words, nr, NCCL_FLOAT64, rh.peer_rank[p], rh.comm, w->stream);
    }
    const int e2 = g_rccl.group_end();
    if (e || e2) return fail(MGX_ERR_HIP, "RCCL: %s", g_rccl.error_string ? g_rccl.error_string(e ? e : e2) : "error");
    HIP_TRY(launch_halo_unpack(w->d, (int)w->halo_recv.size(), w->halo_recv_dev.p, rh.recv_buf.p, w->stream));
    return MGX_OK;
}

extern "C" {

// ---- halo exchange through RCCL inside the library -----------------------------------------------------
int mgx_rccl_unique_id(uint8_t id[128]) {
    if (!id) return fail(MGX_ERR_INVALID, "null argument");
    if (!g_rccl.load()) return fail(MGX_ERR_STATE, "RCCL is not available in this process");
    const int e = g_rccl.get_unique_id(id);
    if (e) return fail(MGX_ERR_HIP, "ncclGetUniqueId: %s", g_rccl.error_string ? g_rccl.error_string(e) : "error");
    return MGX_OK;
}
int mgx_halo_rccl_connect(mgx_world *w, const uint8_t id[128], uint32_t n_ranks, uint32_t rank, uint32_t n_peers, const uint32_t *peer_rank,
                          const uint32_t *send_first, const uint32_t *recv_first) {
    MGX_ENTER(w);
    if (!w || !id || (n_peers && (!peer_rank || !send_first || !recv_first))) return fail(MGX_ERR_INVALID, "null argument");
    if (rank >= n_ranks) return fail(MGX_ERR_INVALID, "rank out of range");
    if (!g_rccl.load()) return fail(MGX_ERR_STATE, "RCCL is not available in this process");
    int rc = halo_commit(w);
    if (rc != MGX_OK) return rc;
    mgx_world::RcclHalo &rh = w->rccl;
    if (n_peers && (send_first[0] != 0 || send_first[n_peers] != w->halo_send.size() || recv_first[0] != 0 || recv_first[n_peers] != w->halo_recv.size()))
        return fail(MGX_ERR_INVALID, "segments do not cover the send / receive lists");
    for (uint32_t p = 0; p < n_peers; p++)
        if (peer_rank[p] >= n_ranks || peer_rank[p] == rank) return fail(MGX_ERR_INVALID, "bad peer rank");
    if (!rh.comm) {
        RcclApi::Id128 uid;
        memcpy(uid.internal, id, 128);
        const int e = g_rccl.comm_init_rank(&rh.comm, (int)n_ranks, uid, (int)rank);  // collective: every rank calls it
        if (e) { rh.comm = nullptr; return fail(MGX_ERR_HIP, "ncclCommInitRank: %s", g_rccl.error_string ? g_rccl.error_string(e) : "error"); }
    }
    const size_t words = (size_t)mgx_halo_words((uint32_t)w->K);
    HIP_TRY(rh.send_buf.reserve(std::max<size_t>(w->halo_send.size() * words, 1)));
    HIP_TRY(rh.recv_buf.reserve(std::max<size_t>(w->halo_recv.size() * words, 1)));
    rh.peer_rank.assign(peer_rank, peer_rank + n_peers);
    rh.send_first.assign(send_first, send_first + n_peers + (n_peers ? 1 : 0));
    rh.recv_first.assign(recv_first, recv_first + n_peers + (n_peers ? 1 : 0));
    rh.connected = true;
    return MGX_OK;
}
int mgx_halo_rccl_disconnect(mgx_world *w) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    HIP_TRY(hipStreamSynchronize(w->stream));
    w->rccl.connected = false;
    return MGX_OK;
}

// ---- direct halo exchange (peer-mapped stores) ---------------------------------------------
int mgx_halo_direct_setup(mgx_world *w, uint32_t n_sources, void **recv_base, void **flag_base) {
    MGX_ENTER(w);
    if (!w || !recv_base || !flag_base) return fail(MGX_ERR_INVALID, "null argument");
    int rc = halo_commit(w);
    if (rc != MGX_OK) return rc;
    mgx_world::DirectHalo &dh = w->direct;
    HIP_TRY(hipStreamSynchronize(w->stream));
    dh.connected = false;
    dh.by_slot = false;
    dh.aimed = false;
    if (dh.recv) { (void)hipFree(dh.recv); dh.recv = nullptr; }
    if (dh.flags) { (void)hipFree(dh.flags); dh.flags = nullptr; }
    dh.recv_words = w->halo_recv.size() * (size_t)mgx_halo_words((uint32_t)w->K);
    dh.n_sources = (int)n_sources;
    const size_t rb = std::max<size_t>(2 * dh.recv_words, 1) * sizeof(double), fb = ((size_t)n_sources + 1) * sizeof(unsigned long long);
    // fine-grained: coherent with stores arriving from other GPUs / processes while kernels run
    HIP_TRY(hipExtMallocWithFlags((void **)&dh.recv, rb, hipDeviceMallocFinegrained));
    HIP_TRY(hipExtMallocWithFlags((void **)&dh.flags, fb, hipDeviceMallocFinegrained));
    HIP_TRY(hipMemsetAsync(dh.recv, 0, rb, w->stream));
    HIP_TRY(hipMemsetAsync(dh.flags, 0, fb, w->stream));
    {
        std::vector<unsigned long long> z(1, 0ull);
        HIP_TRY(dh.ready.upload(z, w->stream));
    }
    HIP_TRY(hipStreamSynchronize(w->stream));
    if (w->sweep_err_host) *w->sweep_err_host = 0ull;  // a freshly wired exchange starts clean
    dh.seq = dh.push_seq = 0;
    if (const char *ms = getenv("MGX_HALO_TIMEOUT_MS")) {
        const long long v = atoll(ms);
        if (v > 0) dh.timeout_ticks = v * 100000ll;
    }
    *recv_base = dh.recv;
    *flag_base = dh.flags;
    return MGX_OK;
}

int mgx_halo_direct_connect(mgx_world *w, uint32_t n_peers, const uint32_t *send_first, void *const *peer_recv_base,
                            const uint64_t *peer_recv_records, const uint64_t *peer_record_offset, void *const *peer_flag_slot) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    mgx_world::DirectHalo &dh = w->direct;
    if (!dh.flags) return fail(MGX_ERR_STATE, "mgx_halo_direct_setup first");
    if (n_peers && (!send_first || !peer_recv_base || !peer_recv_records || !peer_record_offset || !peer_flag_slot))
        return fail(MGX_ERR_INVALID, "null argument");
    if ((int)n_peers != dh.n_sources)
        return fail(MGX_ERR_INVALID, "%u consumers but %d producers: the exchange must be symmetric (every peer both sends and receives)",
                    n_peers, dh.n_sources);
    const size_t n_send = w->halo_send.size(), words = (size_t)mgx_halo_words((uint32_t)w->K);
    if (n_peers && (send_first[0] != 0 || send_first[n_peers] != n_send)) return fail(MGX_ERR_INVALID, "send_first does not cover the send list");
    std::vector<unsigned long long> d0(std::max<size_t>(n_send, 1), 0ull), d1(std::max<size_t>(n_send, 1), 0ull), pf(std::max<size_t>(n_peers, 1), 0ull);
    for (uint32_t p = 0; p < n_peers; p++) {
        if (send_first[p + 1] <= send_first[p]) return fail(MGX_ERR_INVALID, "peer %u receives nothing", p);
        if (!peer_recv_base[p] || !peer_flag_slot[p]) return fail(MGX_ERR_INVALID, "peer %u: null address", p);
        const uint64_t cnt = send_first[p + 1] - send_first[p];
        if (peer_record_offset[p] + cnt > peer_recv_records[p]) return fail(MGX_ERR_INVALID, "peer %u: segment exceeds its receive area", p);
        for (uint32_t i = send_first[p]; i < send_first[p + 1]; i++) {
            const unsigned long long base = (unsigned long long)(uintptr_t)peer_recv_base[p];
            const unsigned long long rec = peer_record_offset[p] + (i - send_first[p]);
            d0[i] = base + (0ull * peer_recv_records[p] + rec) * words * sizeof(double);
            d1[i] = base + (1ull * peer_recv_records[p] + rec) * words * sizeof(double);
        }
        pf[p] = (unsigned long long)(uintptr_t)peer_flag_slot[p];
    }
    std::vector<unsigned int> zero(1, 0u);
    HIP_TRY(dh.dst[0].upload(d0, w->stream));
    HIP_TRY(dh.dst[1].upload(d1, w->stream));
    HIP_TRY(dh.peer_flags.upload(pf, w->stream));
    HIP_TRY(dh.done.upload(zero, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    dh.n_peers = (int)n_peers;
    dh.seq = dh.push_seq = 0;
    dh.connected = true;
    return MGX_OK;
}

// The same exchange wired ONCE for a world whose exchange lists change (worlds that follow their topology, include/mgx.h): the
// receive area holds `slot_capacity` record slots per parity, slot g = the g-th ghost robot of this rank in device order
// (mgx_halo_ghost_slots), and EVERY other rank is a source — with or without records in a given exchange.
int mgx_halo_direct_setup_slots(mgx_world *w, uint32_t n_sources, uint32_t slot_capacity, void **recv_base, void **flag_base) {
    MGX_ENTER(w);
    if (!w || !recv_base || !flag_base) return fail(MGX_ERR_INVALID, "null argument");
    int rc = halo_commit(w);
    if (rc != MGX_OK) return rc;
    const size_t NG = (size_t)(w->d.R_total - w->d.R_local);
    if ((size_t)slot_capacity < NG) return fail(MGX_ERR_INVALID, "%u slots for %zu ghost robots", slot_capacity, NG);
    mgx_world::DirectHalo &dh = w->direct;
    HIP_TRY(hipStreamSynchronize(w->stream));
    dh.connected = false;
    if (dh.recv) { (void)hipFree(dh.recv); dh.recv = nullptr; }
    if (dh.flags) { (void)hipFree(dh.flags); dh.flags = nullptr; }
    dh.by_slot = true;
    dh.aimed = false;
    dh.slot_cap = slot_capacity;
    dh.recv_words = (size_t)slot_capacity * (size_t)mgx_halo_words((uint32_t)w->K);
    dh.n_sources = (int)n_sources;
    const size_t rb = std::max<size_t>(2 * dh.recv_words, 1) * sizeof(double), fb = ((size_t)n_sources + 1) * sizeof(unsigned long long);
    HIP_TRY(hipExtMallocWithFlags((void **)&dh.recv, rb, hipDeviceMallocFinegrained));
    HIP_TRY(hipExtMallocWithFlags((void **)&dh.flags, fb, hipDeviceMallocFinegrained));
    HIP_TRY(hipMemsetAsync(dh.recv, 0, rb, w->stream));
    HIP_TRY(hipMemsetAsync(dh.flags, 0, fb, w->stream));
    {
        std::vector<unsigned long long> z(1, 0ull);
        HIP_TRY(dh.ready.upload(z, w->stream));
    }
    HIP_TRY(hipStreamSynchronize(w->stream));
    if (w->sweep_err_host) *w->sweep_err_host = 0ull;
    dh.seq = dh.push_seq = 0;
    if (const char *ms = getenv("MGX_HALO_TIMEOUT_MS")) {
        const long long v = atoll(ms);
        if (v > 0) dh.timeout_ticks = v * 100000ll;
    }
    *recv_base = dh.recv;
    *flag_base = dh.flags;
    return MGX_OK;
}

// slot of every listed robot among this rank's ghosts (-1: not a ghost here) — what the robot's owner stores its record into
int mgx_halo_ghost_slots(mgx_world *w, uint32_t n, const int32_t *robots, int32_t *slots) {
    MGX_ENTER(w);
    if (!w || (n && (!robots || !slots))) return fail(MGX_ERR_INVALID, "null argument");
    int rc = halo_commit(w);
    if (rc != MGX_OK) return rc;
    for (uint32_t i = 0; i < n; i++) {
        const int32_t g = robots[i];
        if (g < 0 || (size_t)g >= w->robots.size()) return fail(MGX_ERR_INVALID, "bad robot id %d", g);
        slots[i] = w->robots[(size_t)g].ghost ? w->dev_of[(size_t)g] - w->d.R_local : -1;
    }
    return MGX_OK;
}

// (Re)aim the pushes after the exchange lists changed (mgx_halo_plan / mgx_halo_plan_from_connections): peers in the order of the
// send list's segments — every other rank, an empty segment for a rank that takes nothing now — and for every entry of the send
// list the slot of that robot in its consumer's area.  The exchange numbers go on: both ends keep counting.
int mgx_halo_direct_connect_slots(mgx_world *w, uint32_t n_peers, const uint32_t *send_first, void *const *peer_recv_base,
                                  const uint64_t *peer_slot_capacity, const uint32_t *entry_slot, void *const *peer_flag_slot) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    mgx_world::DirectHalo &dh = w->direct;
    if (!dh.flags || !dh.by_slot) return fail(MGX_ERR_STATE, "mgx_halo_direct_setup_slots first");
    if (n_peers && (!send_first || !peer_recv_base || !peer_slot_capacity || !peer_flag_slot)) return fail(MGX_ERR_INVALID, "null argument");
    if ((int)n_peers != dh.n_sources) return fail(MGX_ERR_INVALID, "%u consumers but %d producers: every other rank is both", n_peers, dh.n_sources);
    int rc = halo_commit(w);
    if (rc != MGX_OK) return rc;
    const size_t n_send = w->halo_send.size(), words = (size_t)mgx_halo_words((uint32_t)w->K);
    if ((size_t)(w->d.R_total - w->d.R_local) > dh.slot_cap)
        return fail(MGX_ERR_STATE, "%d ghost robots but the receive area has %zu slots: mgx_halo_direct_setup_slots again (on every rank)",
                    w->d.R_total - w->d.R_local, dh.slot_cap);
    if (n_peers && (send_first[0] != 0 || send_first[n_peers] != n_send)) return fail(MGX_ERR_INVALID, "send_first does not cover the send list");
    if (n_send && !entry_slot) return fail(MGX_ERR_INVALID, "null argument");
    std::vector<unsigned long long> d0(std::max<size_t>(n_send, 1), 0ull), d1(std::max<size_t>(n_send, 1), 0ull), pf(std::max<size_t>(n_peers, 1), 0ull);
    for (uint32_t p = 0; p < n_peers; p++) {
        if (send_first[p + 1] < send_first[p]) return fail(MGX_ERR_INVALID, "send_first is not ascending");
        if (!peer_recv_base[p] || !peer_flag_slot[p]) return fail(MGX_ERR_INVALID, "peer %u: null address", p);
        const unsigned long long base = (unsigned long long)(uintptr_t)peer_recv_base[p];
        for (uint32_t i = send_first[p]; i < send_first[p + 1]; i++) {
            if ((uint64_t)entry_slot[i] >= peer_slot_capacity[p]) return fail(MGX_ERR_INVALID, "entry %u: slot %u beyond the peer's %llu", i, entry_slot[i], (unsigned long long)peer_slot_capacity[p]);
            d0[i] = base + (0ull * peer_slot_capacity[p] + entry_slot[i]) * words * sizeof(double);
            d1[i] = base + (1ull * peer_slot_capacity[p] + entry_slot[i]) * words * sizeof(double);
        }
        pf[p] = (unsigned long long)(uintptr_t)peer_flag_slot[p];
    }
    HIP_TRY(dh.dst[0].upload(d0, w->stream));
    HIP_TRY(dh.dst[1].upload(d1, w->stream));
    HIP_TRY(dh.peer_flags.upload(pf, w->stream));
    if (!dh.connected) {
        std::vector<unsigned int> zero(1, 0u);
        HIP_TRY(dh.done.upload(zero, w->stream));
    }
    HIP_TRY(hipStreamSynchronize(w->stream));
    dh.n_peers = (int)n_peers;
    dh.connected = true;
    dh.aimed = true;
    return MGX_OK;
}

int mgx_halo_direct_exchange(mgx_world *w, uint32_t what) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    if (!w->direct.connected) return fail(MGX_ERR_STATE, "direct halo exchange is not connected");
    if (what == MGX_HALO_PUSH) return direct_push(w);
    if (what == MGX_HALO_WAIT) return direct_wait(w);
    if (what == (MGX_HALO_PUSH | MGX_HALO_WAIT)) return direct_exchange(w);
    return fail(MGX_ERR_INVALID, "bad phase mask");
}

int mgx_halo_direct_status(mgx_world *w, uint64_t *exchanges, uint64_t *failed_exchange) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    mgx_world::DirectHalo &dh = w->direct;
    if (!dh.flags) return fail(MGX_ERR_STATE, "direct halo exchange is not set up");
    unsigned long long err = 0;
    HIP_TRY(hipStreamSynchronize(w->stream));
    HIP_TRY(hipMemcpy(&err, dh.flags + dh.n_sources, sizeof err, hipMemcpyDeviceToHost));
    if (exchanges) *exchanges = dh.seq;
    if (failed_exchange) *failed_exchange = err;
    if (err) return fail(MGX_ERR_STATE, "halo exchange %llu timed out waiting for a peer", err);
    return MGX_OK;
}

int mgx_halo_direct_disconnect(mgx_world *w) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    HIP_TRY(hipStreamSynchronize(w->stream));
    w->direct.connected = false;
    return MGX_OK;
}

// ---- resident schedule launches on sharded worlds (include/mgx.h) -----------------------------------------
// Layout of a ghost area for NG ghost slots of K variables: the agreement word's line, exchange records of parity 0, of parity 1
// (NG * K * XREC_BYTES each: mgx_dev.h), NG progress words.  Both ends compute it.
namespace {
struct GhostAreaLayout {
    static constexpr size_t agree = 0, HEAD = 64;  // the ranks' agreement word (used in rank 0's area only) has the first line to itself
    size_t xrec[2], flag, bytes;
    GhostAreaLayout(size_t ng, size_t K) {
        const size_t S = ng * K * (size_t)XREC_BYTES;
        xrec[0] = HEAD; xrec[1] = HEAD + S;
        flag = HEAD + 2 * S;
        bytes = flag + std::max<size_t>(ng, 1) * sizeof(unsigned long long);
    }
};
}  // namespace

int mgx_halo_resident_setup(mgx_world *w, void **area_base, uint32_t *n_ghost_slots, uint32_t *parity, uint64_t *segment_count,
                            int32_t *recv_slots, int32_t *eligible) {
    MGX_ENTER(w);
    if (!w || !area_base || !n_ghost_slots || !parity || !segment_count || !eligible) return fail(MGX_ERR_INVALID, "null argument");
    int rc = halo_commit(w);
    if (rc != MGX_OK) return rc;
    if (w->conns_dirty || w->flags_dirty) { rc = commit(w); if (rc != MGX_OK) return rc; }
    if (!w->halo_recv.empty() && !recv_slots) return fail(MGX_ERR_INVALID, "null argument");
    mgx_world::ResidentHalo &xr = w->xres;
    HIP_TRY(hipStreamSynchronize(w->stream));
    // The area (rank 0's also holds the agreement word) may still be mapped by the peers, whose launches store into it: a layout
    // change only switches THIS rank's launches off (commit).  It is given back only after a disconnect — which every rank runs
    // behind a barrier, before any of them sets up again (include/mgx.h; sharded.rewire_resident).
    if (xr.area && xr.wired)
        return fail(MGX_ERR_STATE, "mgx_halo_resident_setup: the ghost area of the previous wiring is still connected; call "
                    "mgx_halo_resident_disconnect on every rank (behind a barrier) first");
    xr.connected = false;
    xr.agree = nullptr;
    w->d.agree = nullptr;
    if (xr.area) { (void)hipFree(xr.area); xr.area = nullptr; }
    const DevWorld &d = w->d;
    const int NG = d.R_total - d.R_local;
    // can this rank run its schedules as resident launches at all?
    const bool has_factors = d.ir_max_edges > 0 && !w->conns.empty();
    bool ok = resident_enabled() && (w->p.enable_mask & 2u) && sweep_lds_bytes(w->K, d.ir_max_edges, true) <= sweep_resident_lds_max();
    if (ok && has_factors) {
        if (w->resident_cap_sharded < 0) w->resident_cap_sharded = sweep_resident_capacity(d, true);
        ok = d.R_local + 1 <= w->resident_cap_sharded;  // (+ the launch's decider workgroup)
    }
    rc = ensure_resident_tables(w);  // settles this rank's segment count (progress words are created here)
    if (rc != MGX_OK) return rc;
    const GhostAreaLayout L((size_t)NG, (size_t)w->K);
    // fine-grained: coherent with stores arriving from other GPUs / processes while kernels run
    HIP_TRY(hipExtMallocWithFlags(&xr.area, L.bytes, hipDeviceMallocFinegrained));
    xr.bytes = L.bytes;
    xr.n_ghosts = NG;
    HIP_TRY(hipMemsetAsync(xr.area, 0, L.bytes, w->stream));
    {   // every ghost "has completed" what this rank's segment count says: nothing of an earlier launch is still being read
        std::vector<unsigned long long> f((size_t)std::max(NG, 1), w->flag_base);
        HIP_TRY(hipMemcpyAsync((char *)xr.area + L.flag, f.data(), sizeof(unsigned long long) * f.size(), hipMemcpyHostToDevice, w->stream));
        HIP_TRY(hipStreamSynchronize(w->stream));
    }
    for (size_t i = 0; i < w->halo_recv.size(); i++) recv_slots[i] = w->dev_of[(size_t)w->halo_recv[i]] - d.R_local;
    *area_base = xr.area;
    *n_ghost_slots = (uint32_t)NG;
    *parity = (uint32_t)d.cur;
    *segment_count = w->flag_base;
    // 2: everything but inter-robot factors is there — a world that follows its topology may get them later (every schedule is then
    // decided where the ranks agree: a rank that cannot run one as a resident launch says so there)
    *eligible = ok ? (has_factors ? 1 : 2) : 0;
    return MGX_OK;
}

int mgx_halo_resident_connect(mgx_world *w, uint32_t n_targets, const int32_t *robots, void *const *peer_area_base,
                              const uint32_t *peer_ghost_slots, const uint32_t *peer_slot, const uint32_t *peer_parity,
                              const uint64_t *peer_segment_count, void *coordinator_area, uint32_t n_ranks) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    if (n_ranks > 0x3fffu) return fail(MGX_ERR_INVALID, "at most %u ranks", 0x3fffu);  // (the agreement word counts them in 14 bits)
    if (n_targets && (!robots || !peer_area_base || !peer_ghost_slots || !peer_slot || !peer_parity || !peer_segment_count))
        return fail(MGX_ERR_INVALID, "null argument");
    mgx_world::ResidentHalo &xr = w->xres;
    if (!xr.area) return fail(MGX_ERR_STATE, "mgx_halo_resident_setup first");
    if (!w->dev_valid || w->dirty) return fail(MGX_ERR_STATE, "the world's layout changed since mgx_halo_resident_setup");
    DevWorld &d = w->d;
    const size_t R = (size_t)d.R_local, K = (size_t)w->K;
    std::vector<std::pair<int, XPushRec>> recs;
    recs.reserve(n_targets);
    for (uint32_t t = 0; t < n_targets; t++) {
        if (robots[t] < 0 || (size_t)robots[t] >= w->robots.size() || w->robots[(size_t)robots[t]].ghost)
            return fail(MGX_ERR_INVALID, "target %u: robot %d is not a local robot", t, robots[t]);
        if (!peer_area_base[t] || peer_slot[t] >= peer_ghost_slots[t] || peer_parity[t] > 1u)
            return fail(MGX_ERR_INVALID, "target %u: bad area / slot / parity", t);
        const GhostAreaLayout L((size_t)peer_ghost_slots[t], K);
        const unsigned long long base = (unsigned long long)(uintptr_t)peer_area_base[t];
        const unsigned x = ((unsigned)d.cur ^ peer_parity[t]) & 1u;  // this rank's parity p is the consumer's p ^ x (both flip together)
        XPushRec r;
        for (unsigned p = 0; p < 2; p++) r.xrec[p] = base + L.xrec[p ^ x] + (size_t)peer_slot[t] * K * (size_t)XREC_BYTES;
        r.flag = base + L.flag + (size_t)peer_slot[t] * sizeof(unsigned long long);
        r.flag_delta = peer_segment_count[t] - w->flag_base;  // modulo 2^64
        recs.emplace_back(w->dev_of[(size_t)robots[t]], r);
    }
    std::stable_sort(recs.begin(), recs.end(), [](const std::pair<int, XPushRec> &a, const std::pair<int, XPushRec> &b) { return a.first < b.first; });
    std::vector<int32_t> ptr(R + 1, 0);
    std::vector<XPushRec> flat(std::max<size_t>(recs.size(), 1));
    for (size_t i = 0; i < recs.size(); i++) { ptr[(size_t)recs[i].first + 1]++; flat[i] = recs[i].second; }
    for (size_t r = 0; r < R; r++) ptr[r + 1] += ptr[r];
    HIP_TRY(xr.xp_ptr.upload(ptr, w->stream));
    HIP_TRY(xr.xp_rec.upload(flat, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    const GhostAreaLayout Lm((size_t)xr.n_ghosts, K);
    for (int p = 0; p < 2; p++) d.gxrec[p] = (const unsigned char *)xr.area + Lm.xrec[p];
    d.gflag = (const unsigned long long *)((const char *)xr.area + Lm.flag);
    d.xp_ptr = xr.xp_ptr.p;
    d.xp_rec = xr.xp_rec.p;
    // the ranks' agreement word: first word of rank 0's area (zeroed by its setup; schedules are numbered from 1 on every rank)
    xr.agree = coordinator_area && n_ranks >= 2 ? (unsigned long long *)((char *)coordinator_area + GhostAreaLayout::agree) : nullptr;
    xr.n_ranks = xr.agree ? (int)n_ranks : 0;
    xr.agree_seq = 0;
    d.agree = xr.agree;
    d.n_ranks = xr.n_ranks;
    xr.connected = true;
    xr.wired = true;
    return MGX_OK;
}

// The same wiring for a world whose exchange lists change (one that follows its topology): the peers ONCE — every other rank's
// ghost area, its number of ghost slots, its buffer parity and segment count as ITS setup returned them — and, after every change of
// the lists, which local robot's records go into which slot of which peer (mgx_halo_resident_aim).
int mgx_halo_resident_connect_peers(mgx_world *w, uint32_t n_peers, void *const *peer_area_base, const uint32_t *peer_ghost_slots,
                                    const uint32_t *peer_parity, const uint64_t *peer_segment_count, void *coordinator_area, uint32_t n_ranks) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    if (n_ranks > 0x3fffu) return fail(MGX_ERR_INVALID, "at most %u ranks", 0x3fffu);
    if (n_peers && (!peer_area_base || !peer_ghost_slots || !peer_parity || !peer_segment_count)) return fail(MGX_ERR_INVALID, "null argument");
    mgx_world::ResidentHalo &xr = w->xres;
    if (!xr.area) return fail(MGX_ERR_STATE, "mgx_halo_resident_setup first");
    if (!w->dev_valid || w->dirty) return fail(MGX_ERR_STATE, "the world's layout changed since mgx_halo_resident_setup");
    DevWorld &d = w->d;
    xr.peers.assign(n_peers, mgx_world::ResidentHalo::Peer());
    for (uint32_t p = 0; p < n_peers; p++) {
        if (!peer_area_base[p] || peer_parity[p] > 1u) return fail(MGX_ERR_INVALID, "peer %u: bad area / parity", p);
        xr.peers[p].base = (unsigned long long)(uintptr_t)peer_area_base[p];
        xr.peers[p].n_slots = peer_ghost_slots[p];
        xr.peers[p].x = ((unsigned)d.cur ^ peer_parity[p]) & 1u;
        xr.peers[p].flag_delta = peer_segment_count[p] - w->flag_base;  // modulo 2^64
    }
    const GhostAreaLayout Lm((size_t)xr.n_ghosts, (size_t)w->K);
    for (int p = 0; p < 2; p++) d.gxrec[p] = (const unsigned char *)xr.area + Lm.xrec[p];
    d.gflag = (const unsigned long long *)((const char *)xr.area + Lm.flag);
    xr.agree = coordinator_area && n_ranks >= 2 ? (unsigned long long *)((char *)coordinator_area + GhostAreaLayout::agree) : nullptr;
    xr.n_ranks = xr.agree ? (int)n_ranks : 0;
    xr.agree_seq = 0;
    d.agree = xr.agree;
    d.n_ranks = xr.n_ranks;
    xr.connected = true;
    xr.wired = true;
    return mgx_halo_resident_aim(w, 0, nullptr, nullptr, nullptr);
}

// n_targets entries (local robot, index of the peer in mgx_halo_resident_connect_peers' order, the robot's ghost slot there).  Call
// on every rank after a change of the lists, with every rank's launches through (the callers synchronise and meet at a barrier):
// the progress words of this rank's ghosts start over at "through with everything so far" — a robot that becomes somebody's
// neighbour across ranks has never stored one here.
int mgx_halo_resident_aim(mgx_world *w, uint32_t n_targets, const int32_t *robots, const uint32_t *peer_index, const uint32_t *peer_slot) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    if (n_targets && (!robots || !peer_index || !peer_slot)) return fail(MGX_ERR_INVALID, "null argument");
    mgx_world::ResidentHalo &xr = w->xres;
    if (!xr.area || !xr.connected) return fail(MGX_ERR_STATE, "mgx_halo_resident_connect_peers first");
    if (w->pending.active) { const int rcc = confirm_resident(w); if (rcc != MGX_OK) return rcc; }
    if (!w->dev_valid || w->dirty) return fail(MGX_ERR_STATE, "the world's layout changed since mgx_halo_resident_setup");
    const DevWorld &d = w->d;
    const size_t R = (size_t)d.R_local, K = (size_t)w->K;
    std::vector<std::pair<int, XPushRec>> recs;
    recs.reserve(n_targets);
    for (uint32_t t = 0; t < n_targets; t++) {
        if (robots[t] < 0 || (size_t)robots[t] >= w->robots.size() || w->robots[(size_t)robots[t]].ghost)
            return fail(MGX_ERR_INVALID, "target %u: robot %d is not a local robot", t, robots[t]);
        if ((size_t)peer_index[t] >= xr.peers.size()) return fail(MGX_ERR_INVALID, "target %u: peer %u of %zu", t, peer_index[t], xr.peers.size());
        const mgx_world::ResidentHalo::Peer &pr = xr.peers[(size_t)peer_index[t]];
        if ((size_t)peer_slot[t] >= pr.n_slots) return fail(MGX_ERR_INVALID, "target %u: slot %u of %zu", t, peer_slot[t], pr.n_slots);
        const GhostAreaLayout L(pr.n_slots, K);
        XPushRec r;
        for (unsigned p = 0; p < 2; p++) r.xrec[p] = pr.base + L.xrec[p ^ pr.x] + (size_t)peer_slot[t] * K * (size_t)XREC_BYTES;
        r.flag = pr.base + L.flag + (size_t)peer_slot[t] * sizeof(unsigned long long);
        r.flag_delta = pr.flag_delta;
        recs.emplace_back(w->dev_of[(size_t)robots[t]], r);
    }
    std::stable_sort(recs.begin(), recs.end(), [](const std::pair<int, XPushRec> &a, const std::pair<int, XPushRec> &b) { return a.first < b.first; });
    std::vector<int32_t> ptr(R + 1, 0);
    std::vector<XPushRec> flat(std::max<size_t>(recs.size(), 1));
    for (size_t i = 0; i < recs.size(); i++) { ptr[(size_t)recs[i].first + 1]++; flat[i] = recs[i].second; }
    for (size_t r = 0; r < R; r++) ptr[r + 1] += ptr[r];
    HIP_TRY(hipStreamSynchronize(w->stream));
    HIP_TRY(xr.xp_ptr.upload(ptr, w->stream));
    HIP_TRY(xr.xp_rec.upload(flat, w->stream));
    {
        const GhostAreaLayout Lm((size_t)xr.n_ghosts, K);
        std::vector<unsigned long long> f((size_t)std::max(xr.n_ghosts, 1), w->flag_base);
        HIP_TRY(hipMemcpyAsync((char *)xr.area + Lm.flag, f.data(), sizeof(unsigned long long) * f.size(), hipMemcpyHostToDevice, w->stream));
    }
    HIP_TRY(hipStreamSynchronize(w->stream));
    w->d.xp_ptr = xr.xp_ptr.p;
    w->d.xp_rec = xr.xp_rec.p;
    return MGX_OK;
}

int mgx_halo_resident_disconnect(mgx_world *w) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    if (w->pending.active) { const int rcc = confirm_resident(w); if (rcc != MGX_OK) return rcc; }
    HIP_TRY(hipStreamSynchronize(w->stream));
    w->xres.connected = false;
    w->xres.wired = false;
    w->xres.peers.clear();
    w->xres.agree = nullptr;
    w->d.agree = nullptr;
    return MGX_OK;
}

int mgx_resident_outcome(mgx_world *w, int32_t *outcome) {
    MGX_ENTER_SCHEDULE(w);  // (a query: a launch that lingers stays — and its verdict is still there to be asked for)
    if (!w || !outcome) return fail(MGX_ERR_INVALID, "null argument");
    return confirm_resident(w, false, outcome);
}

int mgx_resident_ready(mgx_world *w, const uint8_t *steps, uint32_t n, int32_t *ready) {
    MGX_ENTER(w);
    if (!w || !ready || (!steps && n)) return fail(MGX_ERR_INVALID, "null argument");
    int rc = commit(w);
    if (rc != MGX_OK) return rc;
    *ready = resident_gate(w, plan_launches(steps, n)) ? 1 : 0;
    if (*ready && !(w->xres.connected && w->xres.agree)) {  // nobody to agree with: what this world's own launch needs
        const DevWorld &d = w->d;
        const bool sharded = w->xres.connected;
        int &cap = sharded ? w->resident_cap_sharded : w->resident_cap;
        bool can = d.ir_max_edges > 0 && !w->conns.empty() && !w->thaw_kinds && !w->ir_thaw_active && w->n_keyless == 0 && !w->resident_decline &&
                   sweep_lds_bytes(w->K, d.ir_max_edges, true) <= sweep_resident_lds_max();
        if (can && cap < 0) cap = sweep_resident_capacity(d, sharded);
        *ready = can && d.R_local + 1 <= cap ? 1 : 0;
    }
    return MGX_OK;
}

int mgx_resident_stats(mgx_world *w, uint64_t *launches, uint64_t *declined, uint32_t *backoff) {
    MGX_ENTER_SCHEDULE(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    if (w->pending.active) { const int rcc = confirm_resident(w); if (rcc != MGX_OK) return rcc; }
    if (launches) *launches = w->resident_launches;
    if (declined) *declined = w->resident_aborts;
    if (backoff) *backoff = (uint32_t)w->resident_backoff;
    return MGX_OK;
}

// hipIpc* wrappers so that a host language needs no HIP binding of its own to share the areas
int mgx_ipc_export(const void *dev_ptr, uint8_t handle[64]) {
    if (!dev_ptr || !handle) return fail(MGX_ERR_INVALID, "null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, const_cast<void *>(dev_ptr)));
    memcpy(handle, &h, 64);
    return MGX_OK;
}
int mgx_ipc_open(const uint8_t handle[64], void **dev_ptr) {
    if (!handle || !dev_ptr) return fail(MGX_ERR_INVALID, "null argument");
    if (!device_ok()) return fail(MGX_ERR_NO_DEVICE, "no HIP device");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, 64);
    HIP_TRY(hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return MGX_OK;
}
int mgx_ipc_close(void *dev_ptr) {
    if (!dev_ptr) return MGX_OK;
    HIP_TRY(hipIpcCloseMemHandle(dev_ptr));
    return MGX_OK;
}

int mgx_halo_pack(mgx_world *w, void *dev_buf) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    int rc = halo_commit(w);
    if (rc != MGX_OK) return rc;
    if (w->halo_send.empty()) return MGX_OK;
    if (!dev_buf) return fail(MGX_ERR_INVALID, "null buffer");
    HIP_TRY(launch_halo_pack(w->d, (int)w->halo_send.size(), w->halo_send_dev.p, (double *)dev_buf, w->stream));
    return MGX_OK;
}
int mgx_halo_unpack(mgx_world *w, const void *dev_buf) {
    MGX_ENTER(w);
    if (!w) return fail(MGX_ERR_INVALID, "null world");
    int rc = halo_commit(w);
    if (rc != MGX_OK) return rc;
    if (w->halo_recv.empty()) return MGX_OK;
    if (!dev_buf) return fail(MGX_ERR_INVALID, "null buffer");
    HIP_TRY(launch_halo_unpack(w->d, (int)w->halo_recv.size(), w->halo_recv_dev.p, (const double *)dev_buf, w->stream));
    return MGX_OK;
}

#ifdef MGX_STAMPS
// diagnostic build only: copy the per-wave phase cycle sums to the host
int mgx_debug_read_stamps(mgx_world *w, unsigned long long *out, uint32_t n) {
    MGX_ENTER(w);
    if (!w || !out) return fail(MGX_ERR_INVALID, "null argument");
    std::vector<unsigned long long> h;
    HIP_TRY(w->dbg.download(h, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    for (uint32_t i = 0; i < n && i < h.size(); i++) out[i] = h[i];
    return (int)h.size();
}
#endif

}  // extern "C"
