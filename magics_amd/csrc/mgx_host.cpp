// mgx_host.cpp — host-only helpers of the C ABI (no device needed): the GBP iteration schedules
// (crates/gbp_schedule/src/schedules/*.rs) and the variable-timestep rule
// (crates/magics/src/utils.rs:35-75) the driver uses to build a robot's graph.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/mgx.h"

namespace {

void reverse(uint8_t *s, int n) {
    for (int i = 0, j = n - 1; i < j; i++, j--) {
        uint8_t t = s[i];
        s[i] = s[j];
        s[j] = t;
    }
}
void fill_every(uint8_t *s, int len, int period) {
    for (int i = 0; i < len; i++) s[i] = (i % period) == 0;
}

// InterleaveEvenlyIter::recurse (interleave_evenly.rs:41-104): spread n trues over slice s
void interleave(uint8_t *s, int len, int n) {
    const int half = len / 2;
    if (n == len) { memset(s, 1, (size_t)len); return; }
    if (n == 0) { memset(s, 0, (size_t)len); return; }
    const bool n_odd = n & 1, len_odd = len & 1;
    if (n_odd && len_odd) {
        if (len % n == 0) { fill_every(s, len, len / n); return; }
        interleave(s, half, n / 2);
        s[half] = 1;
        interleave(s + half + 1, len - half - 1, n / 2);
        reverse(s + half + 1, len - half - 1);
    } else if (!n_odd && len_odd) {
        interleave(s, half, n / 2);
        reverse(s, half);
        s[half] = 0;
        interleave(s + half + 1, len - half - 1, n / 2);
    } else if (!n_odd && !len_odd) {
        if (len % n == 0) { fill_every(s, len, len / n); return; }
        interleave(s, half, n / 2);
        interleave(s + half, len - half, n / 2);
    } else {
        interleave(s, half, n / 2 + 1);
        reverse(s, half);
        interleave(s + half, len - half, n / 2);
    }
}

// one boolean stream of length `max` with exactly n trues
void stream(int kind, int n, int max, uint8_t *s) {
    switch (kind) {
    case MGX_SCHEDULE_CENTERED: {  // centered.rs:19-48
        const int mid = max / 2, hn = n / 2;
        const int start = mid >= hn ? mid - hn : 0;
        const int end = (start + n <= max) ? start + n - 1 : max - 1;
        for (int i = 0; i < max; i++) s[i] = (i >= start && i <= end);
        break;
    }
    case MGX_SCHEDULE_SOON_AS_POSSIBLE:  // soon_as_possible.rs:27-51
        for (int i = 0; i < max; i++) s[i] = i < n;
        break;
    case MGX_SCHEDULE_LATE_AS_POSSIBLE:  // late_as_possible.rs:30-48
        for (int i = 0; i < max; i++) s[i] = i >= max - n;
        break;
    case MGX_SCHEDULE_INTERLEAVE_EVENLY: interleave(s, max, n); break;
    case MGX_SCHEDULE_HALF_BEGINNING_HALF_END: {  // half_beginning_half_end.rs:19-43
        const int hn = n / 2, rem = n % 2;
        for (int i = 0; i < max; i++) s[i] = (i < hn || i >= max - hn - rem);
        break;
    }
    }
}

}  // namespace

extern "C" {

int mgx_schedule(int32_t kind, uint8_t n_internal, uint8_t n_external, uint8_t *steps, uint32_t capacity) {
    const int max = n_internal > n_external ? n_internal : n_external;
    if (kind < 0 || kind > 4 || !steps || (int)capacity < max) return MGX_ERR_INVALID;
    std::vector<uint8_t> a((size_t)max + 1), b((size_t)max + 1);
    stream(kind, n_internal, max, a.data());
    stream(kind, n_external, max, b.data());
    for (int i = 0; i < max; i++) steps[i] = (uint8_t)((a[i] ? MGX_STEP_INTERNAL : 0) | (b[i] ? MGX_STEP_EXTERNAL : 0));
    return max;
}

int mgx_variable_timesteps(uint32_t lookahead_horizon, uint32_t lookahead_multiple, uint32_t *timesteps, uint32_t capacity) {
    if (!timesteps || lookahead_multiple == 0) return MGX_ERR_INVALID;
    const float h = (float)lookahead_horizon, m = (float)lookahead_multiple;
    const uint32_t n = 1u + (uint32_t)(0.5f * (-1.0f + sqrtf(1.0f + 8.0f * h / m)));
    uint32_t cnt = 0;
    for (uint32_t i = 0; i < lookahead_multiple * (n + 1u); i++) {
        const float section = (float)(i / lookahead_multiple);
        const float f = fmaf(m / 2.0f, section, fmaf(section, -m, (float)i)) * (section + 1.0f);
        if (cnt >= capacity) return MGX_ERR_INVALID;
        if (f >= h) {
            timesteps[cnt++] = lookahead_horizon;
            break;
        }
        timesteps[cnt++] = (uint32_t)f;
    }
    return (int)cnt;
}

}  // extern "C"

// ---- sharding plan: who owns which robot, which robots a rank needs ghost copies of, what travels
// where in the one exchange per external iteration.  Pure host logic, identical on every rank, no communication.
struct mgx_shard_plan {
    uint32_t n_ranks = 0;
    int32_t rank = 0;
    std::vector<int32_t> local, ghosts;
    std::vector<uint32_t> conns;  // indices of the connections evaluated on this rank (their TARGET robot is local)
    std::vector<uint32_t> send_first, recv_first;
    std::vector<int32_t> send_robots, recv_robots;
};

extern "C" {

int mgx_set_error_(int code, const char *text);  // mgx_world.hip: the thread-local error text

// Owner rank of every robot: contiguous strips in (y, x) order with equal robot counts — spatial blocks keep the
// cross-rank neighbour pairs few.  Ties are broken by robot index (a stable sort), so every rank computes the same map.
int mgx_shard_partition(const double *positions_xy, uint32_t n_robots, uint32_t n_ranks, int32_t *owner) {
    if ((n_robots && (!positions_xy || !owner)) || n_ranks == 0) return mgx_set_error_(MGX_ERR_INVALID, "bad argument");
    std::vector<uint32_t> order(n_robots);
    for (uint32_t i = 0; i < n_robots; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        const double ya = positions_xy[2 * a + 1], yb = positions_xy[2 * b + 1];
        if (ya != yb) return ya < yb;
        return positions_xy[2 * a] < positions_xy[2 * b];
    });
    for (uint32_t k = 0; k < n_ranks; k++) {
        const uint64_t lo = (uint64_t)k * n_robots / n_ranks, hi = (uint64_t)(k + 1) * n_robots / n_ranks;
        for (uint64_t q = lo; q < hi; q++) owner[order[q]] = (int32_t)k;
    }
    return MGX_OK;
}

// The plan of one rank.  Connections are the directed (owner robot a, other robot b) pairs of create_interrobot_factors
// (robot.rs:1490-1541): the K-1 factors of a connection are evaluated on the rank that owns their only consumer b
// (factorgraph.rs:745-754 keeps only the message to b), which needs a's snapshot records: a is a ghost there when it
// lives elsewhere, and a's rank sends them.  Lists are ascending by robot id inside each peer's segment.
int mgx_shard_plan_create(const int32_t *owner, uint32_t n_robots, const int32_t *conn_owner, const int32_t *conn_other, uint32_t n_conns,
                          int32_t rank, uint32_t n_ranks, mgx_shard_plan **out) {
    if (!out || (n_robots && !owner) || (n_conns && (!conn_owner || !conn_other)) || n_ranks == 0 || rank < 0 || (uint32_t)rank >= n_ranks)
        return mgx_set_error_(MGX_ERR_INVALID, "bad argument");
    for (uint32_t i = 0; i < n_robots; i++)
        if (owner[i] < 0 || (uint32_t)owner[i] >= n_ranks) return mgx_set_error_(MGX_ERR_INVALID, "owner rank out of range");
    for (uint32_t c = 0; c < n_conns; c++)
        if (conn_owner[c] < 0 || (uint32_t)conn_owner[c] >= n_robots || conn_other[c] < 0 || (uint32_t)conn_other[c] >= n_robots)
            return mgx_set_error_(MGX_ERR_INVALID, "connection names a robot that does not exist");
    mgx_shard_plan *p = new (std::nothrow) mgx_shard_plan();
    if (!p) return mgx_set_error_(MGX_ERR_NOMEM, "out of memory");
    p->n_ranks = n_ranks;
    p->rank = rank;
    for (uint32_t i = 0; i < n_robots; i++)
        if (owner[i] == rank) p->local.push_back((int32_t)i);
    std::vector<uint8_t> is_ghost(n_robots, 0);
    std::vector<std::vector<int32_t>> send(n_ranks);
    for (uint32_t c = 0; c < n_conns; c++) {
        const int32_t a = conn_owner[c], b = conn_other[c];
        if (owner[b] == rank) {
            p->conns.push_back(c);
            if (owner[a] != rank) is_ghost[(size_t)a] = 1;
        } else if (owner[a] == rank) {
            send[(size_t)owner[b]].push_back(a);
        }
    }
    for (uint32_t i = 0; i < n_robots; i++)
        if (is_ghost[i]) p->ghosts.push_back((int32_t)i);
    p->send_first.assign(n_ranks + 1, 0);
    p->recv_first.assign(n_ranks + 1, 0);
    for (uint32_t q = 0; q < n_ranks; q++) {
        std::vector<int32_t> &s = send[q];
        std::sort(s.begin(), s.end());
        s.erase(std::unique(s.begin(), s.end()), s.end());
        p->send_robots.insert(p->send_robots.end(), s.begin(), s.end());
        p->send_first[q + 1] = (uint32_t)p->send_robots.size();
        for (int32_t g : p->ghosts)
            if ((uint32_t)owner[g] == q) p->recv_robots.push_back(g);
        p->recv_first[q + 1] = (uint32_t)p->recv_robots.size();
    }
    *out = p;
    return MGX_OK;
}
void mgx_shard_plan_destroy(mgx_shard_plan *p) { delete p; }
int mgx_shard_plan_counts(const mgx_shard_plan *p, uint32_t *n_local, uint32_t *n_ghosts, uint32_t *n_connections, uint32_t *n_send, uint32_t *n_recv) {
    if (!p) return mgx_set_error_(MGX_ERR_INVALID, "null plan");
    if (n_local) *n_local = (uint32_t)p->local.size();
    if (n_ghosts) *n_ghosts = (uint32_t)p->ghosts.size();
    if (n_connections) *n_connections = (uint32_t)p->conns.size();
    if (n_send) *n_send = (uint32_t)p->send_robots.size();
    if (n_recv) *n_recv = (uint32_t)p->recv_robots.size();
    return MGX_OK;
}
int mgx_shard_plan_get(const mgx_shard_plan *p, int32_t *local, int32_t *ghosts, uint32_t *connections, uint32_t *send_first, int32_t *send_robots,
                       uint32_t *recv_first, int32_t *recv_robots) {
    if (!p) return mgx_set_error_(MGX_ERR_INVALID, "null plan");
    if (local) std::copy(p->local.begin(), p->local.end(), local);
    if (ghosts) std::copy(p->ghosts.begin(), p->ghosts.end(), ghosts);
    if (connections) std::copy(p->conns.begin(), p->conns.end(), connections);
    if (send_first) std::copy(p->send_first.begin(), p->send_first.end(), send_first);
    if (send_robots) std::copy(p->send_robots.begin(), p->send_robots.end(), send_robots);
    if (recv_first) std::copy(p->recv_first.begin(), p->recv_first.end(), recv_first);
    if (recv_robots) std::copy(p->recv_robots.begin(), p->recv_robots.end(), recv_robots);
    return MGX_OK;
}

}  // extern "C"
