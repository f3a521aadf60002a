// mgx_dev.h — device-side view of a world (plain pointers, passed to kernels by value).
//
// HBM layout: every per-item quantity is stored component-major ("SoA"):
//   arr[c * stride + item], so that lanes working on consecutive items issue coalesced
//   8-byte loads for each component c.  Items:
//     variables       v = robot * K + i                      (stride V, ghosts at the end)
//     internal edges  e = robot * E + slot, E = 4K - 6       (stride EI)
//         slot [0, K-1)        dynamic factor i   -> variable i       (factor slot 0)
//         slot [K-1, 2K-2)     dynamic factor i   -> variable i + 1   (factor slot 1)
//         slot [2K-2, 3K-4)    obstacle factor j  -> variable j + 1
//         slot [3K-4, 4K-6)    tracking factor j  -> variable j + 1
//     inter-robot edges, stored at their TARGET variable (CSR by variable, ir_var_ptr)
#pragma once
#include <cstdint>

namespace mgx {

constexpr int SNAP_W = 24;  // eta(4) lam(16) mu(4)

// Blob of one robot (f64 words; K variables, E = 4K-6 internal edges, component-major rows):
//   prior  [20][K]    prior eta(4), lam(16)                          read-only for the sweep kernel
//   bel    [20][K]    belief eta, lam of the last variable sweep     output
//   cov    [16][K]    belief covariance                              in/out   \
//   mu     [4][K]     belief mean                                    in/out    | one contiguous
//   fv     [20][E+1]  factor -> variable messages (eta 4, lam 16);   in/out    | in/out region
//                     column E is all zeros (absent edges)                     |
//   valid  [K] int32  (occupies K f64 slots)                         in/out   /
struct BlobLayout {
    int K, E1;  // E1 = E + 1
    __host__ __device__ constexpr BlobLayout(int k) : K(k), E1(4 * k - 6 + 1) {}
    __host__ __device__ constexpr int prior() const { return 0; }
    __host__ __device__ constexpr int bel() const { return 20 * K; }
    __host__ __device__ constexpr int cov() const { return 40 * K; }
    __host__ __device__ constexpr int mu() const { return 56 * K; }
    __host__ __device__ constexpr int fv() const { return 60 * K; }
    __host__ __device__ constexpr int valid() const { return 60 * K + 20 * E1; }
    __host__ __device__ constexpr int inout_words() const { return 16 * K + 4 * K + 20 * E1 + K; }  // cov..valid
    __host__ __device__ constexpr int words() const { return 61 * K + 20 * E1; }
};

// One inter-robot factor F_AB as seen from its target variable B.var_i.
struct IrEdgeRec {
    int32_t src_var;    // snapshot index of the owner's variable A.var_i
    int32_t src_robot;  // device index of A
    uint32_t created;   // epoch of A.var_i when the factor was created (its inbox slot is empty until then)
    int32_t dst;        // local index i of the target variable | (1 << 16 if B has the higher order key)
    double d_safe;      // safety distance of the OWNER (interrobot.rs:64)
    double offset;      // tiny offset 1e-6f * robot_number (interrobot.rs:52,75)
};

// One incoming connection of a local robot (K-1 factors, one per variable 1..K-1): what the device
// needs to lay out that robot's edges when the topology changes (k_retopo_robots).
struct IrSlotRec {
    int32_t tgt_robot;      // local device index of the target robot
    int32_t src_robot;      // device index of the owner
    int32_t old_slot;       // slot of this connection in the tables being replaced, -1 = created now
    int32_t flags;          // bit 0: the target has the higher order key (slot order of the factor)
    double d_safe;          // interrobot.rs:64
    uint64_t first_number;  // robot_number of the factor on variable 1 (robot.rs:1527)
};
// What the host sends for a topology change (k_retopo_robots): ONE pinned block, read over the host link in as few transactions
// as its layout allows — a 16-byte header per robot (robot r reads its own and its successor's: one 32-byte read), and, for the
// robots whose lists changed, their slot records with their peer row right behind (one contiguous piece each).
struct RetopoHeader {
    int32_t in0;   // first slot of the robot in the new layout (the successor's in0 ends its range; entry R closes the last one)
    int32_t pp0;   // first entry of its row in the new peer table
    int32_t mid;   // first slot whose owner has the HIGHER graph key, relative to the robot's range
    int32_t off;   // where its piece starts in `data`, in 16-byte units; -1: neither its incoming list nor its peers changed
};
struct RetopoBlock {
    const RetopoHeader *hdr;  // [R + 1]
    const uint4 *data;        // per changed robot: n_in slot records (IrSlotRec, old_slot: position in the robot's OLD list, -1 = created now), then its peer row (int32, padded to 16 bytes)
};

// EXCHANGE RECORDS of resident schedule launches — the hand-off between neighbouring robots' workgroups (and, on sharded worlds,
// between GPUs).  What an inter-robot factor F_AB reads of its owner A is the message A's variable sends it: (eta, lam, mu) — of
// which the factor's arithmetic touches eta, lam and the two position means (interrobot.rs:149-159: the Jacobian has no velocity
// columns) — and whether that variable has delivered at all (its delivery count): 22 f64 and one count, 45 dwords.  At the end of
// a segment every robot publishes them as FIFTEEN 16-byte chunks per variable; chunk c = { D[3c], D[3c+1], D[3c+2], sequence word }
// where D[0..43] are the 22 f64 as (low, high) dword pairs and D[44] is the delivery count.  A 16-byte aligned store is one
// transaction, so a chunk whose sequence word is the expected one carries that publication's payload: consumers POLL THE CHUNKS
// THEMSELVES — no drain of the producer's stores, no progress word, no dependent second trip (experiments/handoff/records.hip,
// form B: half the hand-off time of "records, acknowledgements, progress word, then the loads").
// Layout of a robot's records: CHUNK-MAJOR, [XREC_CHUNKS][K variables][16 bytes].  A connection hangs one factor on each of the
// variables 1 .. K-1 of its target, so the edge lanes of a consuming workgroup are ordered (owner, variable): consecutive lanes
// read chunk c of CONSECUTIVE variables of one owner — contiguous 16-byte pieces, each lane its own record, no transposition
// across lanes — and the publishing wave's stores are one contiguous kilobyte per instruction.
// Sequence word: bit 31 set (memory that was only ever zeroed never validates) | the low 31 bits of the consumer's count of the
// segment the record is for; compared for equality.  Two parities of records per robot, written alternately (a producer is
// never two publications ahead of a consumer: it needs that consumer's record of the segment in between — see wait_for_peers in
// mgx_sweep.h for the one-sided case).
#ifdef MGX_XREC_CHECKSUM
constexpr int XREC_CHUNKS = 16;  // diagnostic build: one more chunk per variable, { xor of the 45 payload dwords, producer's identity, 0 }
#else
constexpr int XREC_CHUNKS = 15;
#endif
constexpr int XREC_PAYLOAD_DWORDS = 45, XREC_BYTES = 16 * XREC_CHUNKS, XREC_EPOCH_DWORD = 44;  // XREC_BYTES: per variable
__host__ __device__ constexpr uint32_t xrec_seq(unsigned long long count) { return 0x80000000u | (uint32_t)(count & 0x7fffffffull); }
// Records that cross GPUs (a boundary robot's chunks stored into another rank's ghost area over xGMI) do not lean on the 16 bytes
// arriving as one write — that is measured inside one GPU (experiments/handoff/atomicity.hip) and unobserved across the fabric: their
// sequence word is stored XORed with a mix of the chunk's three payload dwords, and the consumer undoes the mix with the payload
// it READ.  A chunk seen half-written (new word over old payload, or the reverse) does not validate — it is simply asked for again,
// like a chunk that has not arrived — except when the stale dwords mix to the same value as the new ones (2^-32 for unrelated
// data; identical data is no error).  Memory that was only ever zeroed: mix(0, 0, 0) = 0, no top bit, never valid.
__host__ __device__ constexpr uint32_t xrec_mix(uint32_t a, uint32_t b, uint32_t c) {
    return a ^ ((b << 11) | (b >> 21)) ^ ((c << 22) | (c >> 10));
}

// Resident schedule launches of a SHARDED world: where the exchange records of a local robot that another rank holds as a ghost
// go at the end of a segment — addresses inside that rank's ghost area (peer-mapped, fine-grained), indexed by THIS rank's
// buffer parity; the consumer's progress word for the ghost, and what has to be added to this rank's segment count to speak
// the consumer's (each rank counts the segments of its own resident launches).
struct XPushRec {
    unsigned long long xrec[2];  // [XREC_CHUNKS][K][16 bytes]
    unsigned long long flag;     // one u64
    unsigned long long flag_delta;
};

struct DevWorld {
    int R_local, R_total, K, E;
    int V, EI, ND, NT, NI;
    int cur;  // snapshot buffer read by this launch; the other one is written
    uint32_t enable;

    // Per-robot private state: one contiguous blob per robot, laid out exactly like the workgroup's
    // LDS image so that staging / write-back are straight wide copies (see blob_layout below).
    double *blob;
    int BS;  // blob stride (f64 words per robot)
    // Snapshot exchange buffers — the only state OTHER robots' workgroups read: one 192-byte record
    // (eta, lam, mu) per variable, double buffered (a launch reads `cur`, writes `1 - cur`).
    double *snap[2];          // [V][24]
    uint32_t *snap_epoch[2];  // [V] number of deliveries (internal sweeps + prior changes)
    double *dyn_m;            // [16][ND] compact J^T Q J of each dynamic factor
    // tracking factor state
    int32_t *trk_record;      // [NT]
    float *trk_last_pos;      // [2][NT]
    double *trk_last_val;     // [NT]
    const int32_t *path_ptr;  // [R_local + 1]
    const float *path_xy;
    int32_t *iter_factor;     // [R_local] FactorGraph.iteration_count.factor

    // inter-robot edges (at the target variable)
    const int32_t *ir_var_ptr;    // [R_local * K + 1]
    const int32_t *ir_var_mid;    // [R_local * K] first edge whose owner has a HIGHER graph key
    const IrEdgeRec *ir_rec;      // [NI] constants of each edge (one 32-byte load)
    const uint8_t *ir_gate;       // [NI] non-zero iff the OWNER robot is on air (antenna on, not idle): refreshed with
                                  //      the flags so that the edge lane needs no dependent flag loads; 1: the edge
                                  //      lane evaluates the factor, 2: k_thaw_ir already has (see ir_thaw_epoch)
    double *ir_fv_eta, *ir_fv_lam;  // [4][NI],[16][NI] factor -> target variable
    // only rows eta[0..2) and lam[0][0..2), lam[1][0..2) are live: an inter-robot factor constrains
    // positions only, the rest of its message is structurally zero (mgx_kernels.hip, compact messages)
    double *ir_bmu;                 // [4][NI] mean of the target variable -> factor message (the
                                    // only part of that inbox entry the kept output depends on)

    // Factor kinds switched back ON at run time (mgx_set_enabled).  A disabled factor receives nothing
    // (factor/mod.rs:307-310), so when its kind is enabled again it resumes from the inbox it had when it
    // was switched off.  That inbox is kept per internal edge, in edge-slot order: 2(K-1) dynamic lanes
    // x 20 (eta, lam of the OTHER variable's message), then K-2 obstacle and K-2 tracking factors x 4 (the
    // mean they linearise at); frozen_flag says whether the entry is present (a message, not empty).
    double *frozen;        // [R_local][frozen_words(K)] or null: no kind has ever been switched at run time
    uint8_t *frozen_flag;  // [R_local][E]
    uint8_t *thaw;         // [R_local] kinds (MGX_FACTOR_* bits) whose factors take their next update from `frozen`
    uint8_t *skip0;        // [R_local] kinds whose first factor sweep of the coming launch k_thaw has computed; null
                           //           whenever no robot is thawing (the sweep kernel then reads nothing)

    // Inter-robot factors switched back on: a factor F_AB resumes from what A's variable had sent it when the
    // kind was switched off (ir_frozen_*), until A's variable delivers again (its epoch leaves ir_thaw_epoch).
    // k_thaw_ir evaluates such edges in front of the sweep launch and marks them ir_gate == 2 ("on air, already
    // evaluated"); all three are null unless the world is thawing inter-robot factors.
    const double *ir_frozen_snap;     // [V][24]
    const uint32_t *ir_frozen_epoch;  // [V]
    const uint32_t *ir_thaw_epoch;    // [V] delivery counts when the kind came back

    // mgx_tick: the driver's two per-tick prior updates folded into the launch that opens the tick.  One
    // record per local robot in device order: waypoint x, y, time scale, what (0..3 as a double); null otherwise
    const double *upd;
    double upd_max_speed, upd_delta_t;

    const uint8_t *antenna, *idle;  // [R_total]

    // obstacle image
    const uint8_t *sdf;
    uint32_t sdf_w, sdf_h;
    double world_w, world_h, obs_delta;
    int ir_max_edges;  // largest number of inter-robot edges attached to one robot (LDS staging size)
    int trk_cols;      // 0: tracking factors have never been enabled in this world (their message columns are all zero)

    double inv_s2_obs, inv_s2_ir, inv_s2_trk, trk_pad, trk_attr;

    // Resident schedule launches (k_robot_sweep<.., PERSIST>, SegPlan below): the exchange records of the local robots' variables
    // (above; two parities), one progress word per local robot ("segments of resident launches completed", monotonic over the
    // life of the device arrays — polled inside a launch only by a robot that READS this one's records without being read by it,
    // the reference's one-sided connections), the robots each one exchanges records with (owners of its incoming edges and
    // targets of its outgoing ones, CSR), and two words for a wait that gave up: [0] device-side "stop waiting", [1] a
    // host-mapped copy the host reads
    unsigned char *xrec[2];           // [R_local][XREC_CHUNKS][K][16 bytes]
    unsigned long long *sweep_flag;   // [R_local]
    const int32_t *peer_ptr;          // [R_local + 1]
    const int32_t *peer_idx;          // device robot indices
    unsigned long long *sweep_abort;  // device memory
    unsigned long long *sweep_err;    // host-mapped
    // ... and the launch's residency census (SegPlan): one word per workgroup of a resident launch, "I have started launch
    // number ..." (monotonic, never reset, nobody shares a word: plain write-through stores); the launch's go / abort decision
    // (launch number * 4 + state) in device memory and its host-mapped copy, which the host looks at before it enqueues
    // anything behind the launch
    unsigned long long *census;          // [R_local + 1]
    unsigned long long *decision;        // device memory
    unsigned long long *decision_host;   // host-mapped
    // Resident schedule launches of a sharded world (k_robot_sweep<.., PERSIST, SHARD>).  This rank's GHOST AREA is fine-grained
    // device memory that the ghosts' owner ranks store into from inside their launches (peer-mapped: hipIpc across processes):
    // the ghosts' exchange records for the two buffer parities and one progress word per ghost (ghost g = device robot
    // R_local + g), all read here with system-scope loads.  xp_*: the local robots other ranks hold as ghosts.
    const unsigned char *gxrec[2];    // [NG][XREC_CHUNKS][K][16 bytes]
    const unsigned long long *gflag;  // [NG]
    const int32_t *xp_ptr;            // [R_local + 1] push targets of each local robot (most have none)
    const XPushRec *xp_rec;
    // ... and the word on which the RANKS agree whether their launches of one schedule go ahead (SegPlan: agree_seq): it lives
    // in the ghost area of rank 0, every rank reaches it through its peer mapping, system-scope compare-and-swap only
    unsigned long long *agree;
    int32_t n_ranks;
    // diagnostic builds only (-DMGX_STAMPS, tools/stamps.py): per-workgroup phase cycle sums
    unsigned long long *dbg;
};

// An inter-robot factor created WHILE its kind was switched off dropped the two messages that would have filled its inbox
// (factor/mod.rs:307-310): once enabled it has no inbox KEYS until its variables deliver again, and FactorNode::update answers
// keys (factor/mod.rs:336-349,412-449).  The host knows which keys are there (they fill structurally); k_keyless_ir evaluates
// such a factor in front of the sweep launch: key bit 0 = the owner's variable has delivered, bit 1 = the target's has.
struct KeylessRec {
    int32_t edge;       // index into the edge arrays
    int32_t tgt_robot;  // device index of the target robot (whose workgroup would evaluate the edge)
    uint32_t keys;
    uint32_t pad;
};

// Missions on the device (the driver's reached_waypoint and Transform increment, robot.rs:2080-2176,2309-2335):
// per robot (device index == robot id: unsharded worlds) its route, the next waypoint, the reached-when rules of
// formation.yaml and the Bevy Transform it moves.
struct DevMission {
    const int32_t *wp_ptr;     // [R + 1] first waypoint of each robot in wp_xy
    const double *wp_xy;       // [.][2] waypoint positions (compared as f32 Vec2s by reached_waypoint, used as f64 by the prior update)
    int32_t *target;           // [R] index of the next waypoint within the robot's route; == count: mission complete
    const uint32_t *vars;      // [R][2] variable whose mean is tested against an intermediate / the final waypoint
    const float *dist2;        // [R][2] squared distance limits (f32)
    const double *time_scale;  // [R] fixed_dt / t0 (f32 quotient widened, robot.rs:2309)
    const uint8_t *has;        // [R] the robot has a mission (and is not despawned)
    float *translation;        // [R][3] Transform::translation (x, height, y)
    long long *finished_tick;  // [R] tick at which the last waypoint was reached, -1 before
};

__host__ __device__ constexpr int frozen_words(int K) { return 40 * (K - 1) + 8 * (K - 2); }

// phases of one launch
constexpr uint32_t PH_EXT_FACTOR = 1u;    // external_factor_iteration (+ routing)
constexpr uint32_t PH_EXT_VARIABLE = 2u;  // external_variable_iteration (+ routing)
constexpr uint32_t PH_INT_FACTOR = 4u;    // internal_factor_iteration
constexpr uint32_t PH_INT_VARIABLE = 8u;  // internal_variable_iteration

// A whole schedule (mgx_iterate / mgx_tick) in ONE launch: the sequence of [external iteration] internal*
// segments that the launch-per-segment path runs as separate launches.  Every robot's workgroup stays resident,
// keeps its graph in LDS across the segments, publishes its snapshot records at the end of each one (write-through
// stores + one progress word) and, in front of an external iteration, waits only for the robots it shares
// inter-robot factors with.  Needs every workgroup of the launch co-resident (checked on the host).
constexpr int MAX_SEGS = 32;

// LINGERING resident launches (unsharded worlds).  A resident launch costs its iterations plus a fixed part — the graphs HBM -> LDS
// when it starts and back when it ends, the dispatch, the census — that the NEXT launch of a driver's tick loop pays straight
// again (robot.rs:85-108 runs iterate_gbp_v2 tick after tick with nothing in between but the two prior updates, which mgx_tick
// folds into the launch).  So a launch does not end with its plan: the robots' workgroups keep their graphs in LDS and wait — a
// bounded time — for the host to POST the next plan, or for the word to leave.
//   * The host writes a post (plan + the tick's prior-update records) into one of two slots of a host-mapped box and raises
//     `posted`; numbers are the world's launch numbers (monotonic, every post and every fresh launch takes one).
//   * The launch's extra workgroup (the census' decider, kept alive) is the only poller of the box — and its only reader: it copies
//     a post into the launch's device-side slot (not before every robot has picked up the post two before it, whose slot that was),
//     turns `posted` into the launch's GO WORD (2 S: plan S may be run; 2 S + 1: the launch ends behind plan S) by compare-and-swap,
//     says what was `taken` (the host may write that slot of the box again), and turns the word odd on the host's request
//     (`close_req`: every other entry point of the C ABI asks for it first).
//   * A robot's workgroup that has finished plan S polls the go word (device memory); 2 S + 2 or more: it reads plan S + 1 and its
//     prior updates from the device-side slot and goes on where it stands — parities and sequence numbers of the exchange records run on over
//     the plans (the first segment of a posted plan continues the last segment of the plan before: posts open with an internal
//     iteration); exactly 2 S + 1: it writes back and returns, as every launch does.  A workgroup that has waited `linger_ticks`
//     raises the word to 2 S + 1 itself (atomic max: whoever moves first decides for all) — a host that died or went elsewhere
//     leaves no spinning GPU.
// A post the launch never took (the word went odd first) is run by the host as a fresh launch: nothing is lost, nothing twice.
struct LingerPlan {  // one posted plan as the device reads it (dwords; host-mapped memory)
    uint32_t n, has_upd;                            // segments; 1: the tick's prior updates ride along (upd records in the slot's block)
    uint32_t ext[MAX_SEGS / 4], n_int[MAX_SEGS / 4];  // SegPlan's bytes
    double upd_max_speed, upd_delta_t;
    unsigned long long number;                      // the post's number (checked: a slot that holds another post is reported)
};
constexpr int LINGER_PLAN_DWORDS = (int)(sizeof(LingerPlan) / 4);
// The launch's DEVICE-side slots (the postman's copies of the posts) are made of self-validating 16-byte chunks like the exchange
// records: three payload dwords + xrec_seq(number of the post).  Head: the plan's 24 dwords in 8 chunks; then per robot its
// prior-update record (4 f64 = 8 dwords) in three chunks.
constexpr unsigned LINGER_SLOT_HEAD = 128u, LINGER_UPD_BYTES = 48u;
static_assert(LINGER_PLAN_DWORDS == 24, "eight chunks of three dwords");
struct LingerBox {  // host-mapped
    unsigned long long posted;     // host -> device: number of the newest post
    unsigned long long close_req;  // host -> device: the launch numbered at most this is asked to end
    unsigned long long taken;      // device -> host: newest post the go word covers
    unsigned long long reserved;
    unsigned long long closed;     // device -> host: the go word as it stands once it is odd (2 S + 1: the launch ended behind plan S)
    unsigned long long pad[3];
    LingerPlan plan[2];            // slot = number & 1; the prior-update records follow the box: double upd[2][4 * R_cap]
};

struct SegPlan {
    int32_t n;                     // segments in this launch
    uint8_t ext[MAX_SEGS];         // 1: the segment opens with an external iteration (factor + variable sweep)
    uint8_t n_int[MAX_SEGS];       // internal iterations that follow
    unsigned long long flag_base;  // every progress word holds at least this much when the launch starts
    long long timeout_ticks;       // 100 MHz wall-clock ticks a wait may take before it gives up (reported, never a hang)
    // Residency census.  Every workgroup of the launch waits for neighbours inside it, and a plain (or cooperative) launch
    // promises co-residency to nobody — another tenant of the GPU can hold the CUs the tail of the grid needs, for as long as
    // the head spins.  So every workgroup signs in when it starts, and before anything that cannot be taken back is written
    // the launch's extra workgroup decides for all: everybody signed in within census_ticks -> go; otherwise abort — every
    // workgroup (also those that start later) returns at once, the world is as it was, and the host runs the schedule launch
    // by launch instead.
    unsigned long long launch_seq;        // number of this resident launch (census and decision words are monotonic in it); 0: no census
    long long census_ticks;
    // Sharded worlds: the ranks' launches of one schedule wait for each other's ghost records, so they go ahead together or
    // not at all.  agree_seq numbers the schedule (the same on every rank; 0: this world has no other ranks).  A rank whose
    // census is complete signs in on the agreement word (DevWorld::agree, agree_on_launch below); the last one to sign in
    // turns the word to go, a rank that has waited census_ticks for the others — or whose own workgroups are not all there —
    // turns it to abort, both by compare-and-swap, so that every rank reads the same answer.
    unsigned long long agree_seq;
    // lingering (above): wall-clock ticks a robot's workgroup waits for the next post before it ends the launch (0: the launch
    // ends with its plan), the box, its prior-update records and the go word.  launch_seq is the number of the launch's own plan.
    long long linger_ticks;
    const LingerBox *linger_box;       // host-mapped
    const double *linger_upd;          // host-mapped, behind the box: slot s of prior-update records at linger_upd + s * linger_upd_stride (f64 words)
    unsigned long long linger_upd_stride;
    unsigned char *linger_dev;         // device memory: slot s at linger_dev + s * linger_dev_stride — the plan (128 bytes), then its records
    unsigned long long linger_dev_stride;
    unsigned long long *linger_go;     // device memory
};
constexpr unsigned RESIDENT_GO = 1u, RESIDENT_ABORT = 2u;

// One look at / one move on the agreement word of a sharded world's resident launches (SegPlan::agree_seq).
// action 0: look; 1: sign in (the caller's census is complete — once per launch); 2: vote abort.
// Returns the word's state for schedule L: 0 while undecided, RESIDENT_GO or RESIDENT_ABORT — or AGREE_LOST when the word has
// moved on by more schedules than it remembers (the caller reports it: the ranks' parities and segment counts have parted).
// The word: schedule number << 28 | outcomes of the AGREE_HISTORY schedules before, two bits each, most recent lowest << 16 |
// state << 14 | ranks signed in.  The outcomes before ride along because ranks that share no robots do not wait for each other:
// some of them can be through schedule L, and through later ones declined for this rank's absence, before a slow rank's
// decider has looked at L's outcome — which it then still finds, as it was decided.  (One remembered outcome was not enough:
// a rank that had signed in for L without being the last and then stalled two schedules read "abort" for a schedule the
// others ran.)  A schedule nobody moved on (every rank skipped the resident form: back-off) reads as declined.
constexpr int AGREE_LOOK = 0, AGREE_SIGN_IN = 1, AGREE_ABORT = 2;
constexpr int AGREE_HISTORY = 6;
constexpr unsigned AGREE_LOST = 3u;
__device__ inline unsigned agree_on_launch(unsigned long long *word, unsigned long long L, unsigned n_ranks, int action) {
    constexpr unsigned long long HMASK = (1ull << (2 * AGREE_HISTORY)) - 1ull;
    unsigned long long declined = 0ull;  // "declined" in every remembered place
    for (int i = 0; i < AGREE_HISTORY; i++) declined |= (unsigned long long)RESIDENT_ABORT << (2 * i);
    unsigned long long cur = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (;;) {
        const unsigned long long cl = cur >> 28, hist = (cur >> 16) & HMASK;
        const unsigned st = (unsigned)(cur >> 14) & 3u, cnt = (unsigned)cur & 0x3fffu;
        if (cl == L && st) return st;
        if (cl > L) {  // the word has moved on: what was decided for L is in its history, if L is not too long ago
            const unsigned long long back = cl - L;
            if (back > (unsigned long long)AGREE_HISTORY) return AGREE_LOST;
            const unsigned was = (unsigned)(hist >> (2 * (back - 1ull))) & 3u;
            return was ? was : AGREE_LOST;
        }
        if (action == AGREE_LOOK) return 0u;
        // (a word of an earlier schedule: the first move of schedule L takes that one's outcome, and "declined" for every
        // schedule in between that nobody moved on, into the history)
        unsigned long long h = hist;
        if (cl < L) {
            const unsigned long long gap = L - cl;
            h = gap > (unsigned long long)AGREE_HISTORY ? declined
                : (((hist << 2) | (unsigned long long)(st ? st : RESIDENT_ABORT)) << (2 * (gap - 1ull)) | (declined & ((1ull << (2 * (gap - 1ull))) - 1ull))) & HMASK;
        }
        unsigned long long nw;
        if (action == AGREE_SIGN_IN) {
            const unsigned c = (cl == L ? cnt : 0u) + 1u;
            nw = (L << 28) | (h << 16) | ((unsigned long long)(c >= n_ranks ? RESIDENT_GO : 0u) << 14) | c;
        } else {
            nw = (L << 28) | (h << 16) | ((unsigned long long)RESIDENT_ABORT << 14) | (cl == L ? cnt : 0u);
        }
        if (__hip_atomic_compare_exchange_strong(word, &cur, nw, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))
            return (unsigned)(nw >> 14) & 3u;
    }
}

// launch hints
constexpr uint32_t HINT_IR_DEAD = 1u;  // the next sweep recomputes every inter-robot message this one computes
// A later launch of the same mgx_iterate call runs an external / internal variable sweep: a robot that takes part
// in it (flags do not change inside a call) rewrites its belief (eta, lam) image there, so this launch need not store it
constexpr uint32_t HINT_LATER_EXT_VARIABLE = 2u;
constexpr uint32_t HINT_LATER_INT_VARIABLE = 4u;

}  // namespace mgx
