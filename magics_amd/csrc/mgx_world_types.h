// mgx_world_types.h — what the host side is made of: launcher prototypes of the kernel files, error text, device buffers, the pinned argument ring, the host mirror's records (Robot, IrConn), connection sets and index, the world itself, the entry hooks of the C ABI (MGX_ENTER).
// Part of ONE translation unit (mgx_world.hip includes its parts in order); not a stand-alone header.
#pragma once
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
hipError_t launch_retopo_robots(const DevWorld &w, const RetopoBlock &b, const int32_t *in_old, const IrSlotRec *slots_old, const int32_t *peers_old,
                                int32_t *in_dst, IrSlotRec *slots_new, int32_t *peers_dst, int32_t *var_ptr, int32_t *var_mid, int stride_new,
                                IrEdgeRec *recs, double *fv_eta, double *fv_lam, double *bmu, uint8_t *gate, hipStream_t stream);
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
hipError_t neighbours_rows(const float *pos, int n, float radius, int32_t cap, int32_t *cnt, int32_t *rows, hipStream_t s, float *stage,
                           int32_t *prev = nullptr, int prev_valid = 0, bool *flagged = nullptr);
int neighbours_prev_stride();
int32_t neighbours_changed_bit();
}  // namespace mgx

using namespace mgx;

static thread_local std::string g_err;
// MGX_TIMING=1: host-side stage times of the topology pass and the table rebuild on stderr (diagnostic); MGX_TIMING=2: the same
// times kept in memory and summed up per stage when the process ends (printing every lap shifts the very phases being looked at)
struct StageTimer {
    struct Sum { const char *what, *stage; double total = 0.0; long n = 0; };
    struct Book {
        std::vector<Sum> rows;
        ~Book() { for (const Sum &r : rows) fprintf(stderr, "[mgx timing] %s: %s n=%ld mean %.1f us\n", r.what, r.stage, r.n, r.total / (double)std::max(r.n, 1L)); }
        void add(const char *what, const char *stage, double dt) {
            for (Sum &r : rows) if (r.what == what && r.stage == stage) { if (++r.n > 0) r.total += dt; return; }
            rows.push_back(Sum{what, stage, 0.0, -7});  // (the first eight of every stage size tables and warm caches: not counted)
        }
    };
    int on;
    double t0;
    const char *what;
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
    explicit StageTimer(const char *w) : what(w) { static const int e = [] { const char *v = getenv("MGX_TIMING"); return v ? (v[0] == '2' ? 2 : 1) : 0; }(); on = e; t0 = on ? now() : 0.0; }
    void lap(const char *stage) {
        if (!on) return;
        const double t = now();
        if (on == 2) { static Book book; book.add(what, stage, t - t0); }
        else fprintf(stderr, "[mgx timing] %s: %s %.1f us\n", what, stage, t - t0);
        t0 = on == 2 ? now() : t;
    }
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
    // ids ascending == order keys ascending?  (the rule: robots get their keys in the order they are added.)  Looked at once per
    // robot count: the per-tick merges then compare ids instead of looking two keys up per comparison.
    size_t mono_n = 0;
    bool mono = true;
    bool monotone() {
        if (mono_n != keys.size()) {
            mono = true;
            for (size_t r = 1; r < keys.size() && mono; r++) mono = keys[r - 1] < keys[r];
            mono_n = keys.size();
        }
        return mono;
    }
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
    DevBuf<IrSlotRec> slot_recs, slot_recs_b;  // slot records of the layout on the device (current / being built): what a robot whose incoming list did not change keeps
    std::vector<int32_t> dev_in_ptr;  // [R_local + 1] incoming-slot ranges of the tables now on the device
    DevBuf<int32_t> trk_record, path_ptr, iter_factor, ir_var_ptr, ir_var_mid;
    DevBuf<uint32_t> epoch0, epoch1;
    DevBuf<float> trk_last_pos, path_xy;
    DevBuf<uint8_t> ir_gate, antenna, idle, sdf;
    StageRing stage;  // packed per-tick arguments
    // resident schedule launches (SegPlan, mgx_dev.h): progress words, peer lists, the abort / error words
    DevBuf<unsigned long long> sweep_flag_buf, sweep_abort_buf;
    DevBuf<unsigned char> xrec_buf;  // exchange records of the local robots' variables, two parities (mgx_dev.h)
    DevBuf<int32_t> peer_ptr_dev, peer_ptr_dev_b;  // [R + 1 row pointers | entries] (current / being built by a topology change)
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
        int32_t dev_q;      // position of this connection in its target's incoming list ON THE DEVICE (-1: not there; its slot is dev_in_ptr[target] + dev_q) — kept HERE only
        uint8_t has_fresh;  // some edge still carries `fresh` (created since the device tables were last laid out) — kept HERE only
    };
    std::vector<ConnHot> conn_hot;
    ConnIndex cidx;
    // per robot, cumulative since the world began: internal variable sweeps run, external variable sweeps, external factor sweeps,
    // prior changes of variables that carry inter-robot factors — what the connections' counters are settled against (IrConn::base)
    struct Cum { std::vector<uint64_t> nIv, nEv, nEf, on_ir; } cum;
    bool conns_unsettled = false;  // some flush since the last full one left the connections' counters behind (lazy)
    std::vector<uint8_t> scratch_dead;    // ir_disconnect_batch's scratch
    std::vector<int32_t> scratch_dead_list;
    std::vector<int> scratch_gone;
    std::vector<int> scratch_victim;      // topology_bookkeeping's scratch
    std::vector<uint8_t> scratch_chg;     // neighbours_collect: the rows-changed flags taken out of the counts
    // storage of deleted connections' edge and node lists, handed to the connections created next (a topology pass deletes and
    // creates dozens per tick: 3 KB from the allocator and back for each was a third of the pass's bookkeeping)
    std::vector<std::vector<IrEdge>> pool_edges;
    std::vector<std::vector<int>> pool_nodes;
    std::vector<std::pair<int, int>> scratch_fresh;
    Incoming retopo_tables;               // scratch of the full table builds (MGX_CHECK_INDEX, ensure_resident_tables)
    // retopo as DIFFERENCES (a world that follows its topology changes a few dozen of its thousands of connections per tick): which
    // robots' incoming lists / peer lists changed since the device tables were laid out (marked where the connection index is
    // edited), and what stays from tick to tick on the host — the lower / higher key split of every local robot.  `all`: every
    // robot counts as changed (after commit(), after the index was rebuilt from the list, before the first retopo).
    struct RetopoInc {
        bool all = true;
        std::vector<uint8_t> in_mark, peer_mark;     // [robot ids]
        std::vector<int32_t> in_changed, peer_changed;  // the ids marked
        std::vector<int32_t> mid;                    // [R_local] the lower / higher key split of every robot's list on the device
        std::vector<int32_t> in_ptr;                 // per-retopo scratch (storage kept)
        std::vector<uint8_t> mark;                   // [R_local] this pass sends the robot's piece
        void mark_in(int32_t id) {
            if ((size_t)id >= in_mark.size()) in_mark.resize((size_t)id + 1, 0);
            if (!in_mark[(size_t)id]) { in_mark[(size_t)id] = 1; in_changed.push_back(id); }
        }
        void mark_peer(int32_t id) {
            if ((size_t)id >= peer_mark.size()) peer_mark.resize((size_t)id + 1, 0);
            if (!peer_mark[(size_t)id]) { peer_mark[(size_t)id] = 1; peer_changed.push_back(id); }
        }
        void clear_marks() {
            for (int32_t id : in_changed) in_mark[(size_t)id] = 0;
            for (int32_t id : peer_changed) peer_mark[(size_t)id] = 0;
            in_changed.clear();
            peer_changed.clear();
        }
    } rinc;
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
        bool has_chg = false;  // rows mode: the counts carry "this robot's row changed" (see nb_prev)
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
    // The rows of the last topology pass's search, kept on the device: after a pass a robot's connection set IS its row, so the
    // next search can say which robots' rows changed (k_grid_rows) and the host's pass looks at those only.  nb_prev_valid: the
    // kept rows are the sets — false whenever the sets were touched by anything but a pass that compared against them.
    DevBuf<int32_t> nb_prev;
    bool nb_prev_valid = false;
    size_t nb_prev_n = 0;
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
// slot of connection ci in the tables ON THE DEVICE (-1: not there — created since they were laid out, or its target is another rank's)
static inline int32_t dev_slot_of(const mgx_world *w, size_t ci) {
    const mgx_world::ConnHot &h = w->conn_hot[ci];
    return h.dev_q < 0 ? -1 : w->dev_in_ptr[(size_t)w->dev_of[(size_t)h.other]] + h.dev_q;
}
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
