// mgx_topology.hip — comms-range neighbour search on the device (update_robot_neighbours,
// crates/magics/src/planner/robot.rs:1362-1384): for every robot the set of other robots j with
// NOT (radius < |p_i - p_j|), f32 distance as glam's Vec3::distance.  The reference scans all
// pairs ("TODO: use kdtree", robot.rs:1368); here a uniform hash grid of cell size ~radius
// bounds the candidates to 3x3 cells, and an all-pairs kernel covers small worlds and the
// degenerate inputs (non-finite positions, radius <= 0 or NaN) with the same predicate.
//
// Output is CSR (ptr[n+1], idx[]) with every row ascending in robot index.  Integer result:
// identical to the all-pairs scan by construction (the grid only prunes pairs whose x or z
// separation alone exceeds the radius).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <limits>

#include "gbp_math.h"

namespace mgx {

// |a-b| as f32 Vec3::distance: sqrt((dx*dx + dy*dy) + dz*dz), each operation rounded to f32.
// Written through f64 (53 >= 2*24+2 bits: every double rounding is innocuous) so that no
// contraction setting can fuse the products into the sums.
__device__ __forceinline__ bool in_comms_range(float ax, float ay, float az, float bx, float by, float bz, float radius) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    const float px = (float)((double)dx * (double)dx), py = (float)((double)dy * (double)dy), pz = (float)((double)dz * (double)dz);
    const float s = (float)((double)(float)((double)px + (double)py) + (double)pz);
    const float d = (float)sqrt((double)s);
    return !(radius < d);
}

// the same predicate with the f32 operations spelled as round-to-nearest intrinsics (never contracted either); squared form:
// sqrt is monotone and correctly rounded, so !(radius < sqrt(s)) == !(r2_hi < s) for the largest f32 r2_hi whose root does
// not exceed radius — the caller passes that threshold (threshold_of), and a NaN s (or radius) still answers "in range"
// exactly as the comparison above does
__device__ __forceinline__ bool in_comms_range_sq(float ax, float ay, float az, float bx, float by, float bz, float s_max) {
    const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    const float s = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    return !(s_max < s);
}

__device__ __forceinline__ bool finite3(float x, float y, float z) { return isfinite(x) && isfinite(y) && isfinite(z); }

// cell coordinate of one axis: floor(x / cell) in f64, clamped (monotone, so robots within one
// radius of each other stay within one cell of each other)
__device__ __forceinline__ int cell_of(float x, double inv_cell) {
    double c = floor((double)x * inv_cell);
    c = fmin(fmax(c, -1073741824.0), 1073741824.0);
    return (int)c;
}
__device__ __forceinline__ uint32_t bucket_of(int cx, int cz, uint32_t mask) {
    return (((uint32_t)cx * 73856093u) ^ ((uint32_t)cz * 19349663u)) & mask;
}

// ---- all pairs ------------------------------------------------------------------------------------
// one thread per robot i, positions of j staged through LDS in tiles; FILL = second pass
template <bool FILL>
__global__ void __launch_bounds__(256) k_pairs(const float *__restrict__ pos, int n, float radius, int32_t *__restrict__ cnt,
                                                const int32_t *__restrict__ ptr, int32_t *__restrict__ idx, int32_t cap) {
    __shared__ float tile[256 * 3];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (FILL && ptr[n] > cap) return;  // speculative second pass: the rows do not fit the buffer the host guessed
    const bool live = i < n;
    const float ax = live ? pos[3 * i] : 0.f, ay = live ? pos[3 * i + 1] : 0.f, az = live ? pos[3 * i + 2] : 0.f;
    int m = 0;
    int32_t out = (FILL && live) ? ptr[i] : 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int nj = min(256, n - j0);
        __syncthreads();
        for (int q = threadIdx.x; q < 3 * nj; q += 256) tile[q] = pos[3 * j0 + q];
        __syncthreads();
        if (live)
            for (int q = 0; q < nj; q++) {
                const int j = j0 + q;
                if (j != i && in_comms_range(ax, ay, az, tile[3 * q], tile[3 * q + 1], tile[3 * q + 2], radius)) {
                    if (FILL) idx[out + m] = j;
                    m++;
                }
            }
    }
    if (!FILL && live) cnt[i] = m;
}

// ---- all pairs, ONE pass, rows of a fixed capacity (small worlds) ------------------------------------
// A world of a few thousand robots is searched faster by one small kernel than by the six launches and three clears
// of the grid.  The positions are the CALLER's, in a pinned block the kernel reads over the host link (a copy would be a
// launch of its own), so what the kernel costs is link round trips, not arithmetic: every workgroup fetches ALL positions
// in a few goes — 16-byte loads, six per lane in flight together — into LDS (structure of arrays), and only then compares
// (the first version pulled them tile by tile, three dependent trips per tile of 64: 90 us for 1000 robots on an idle
// device, most of a tick).  One wave = 16 robots x 4 lanes per workgroup.  Lane l of a robot's quad tests candidates
// 4 k + l (neighbouring LDS words: no bank conflicts), 64 of them per pass; the quad then exchanges its four hit masks, and
// every lane writes its own hits at their rank among all four masks — row i (rows[i * cap + m]) comes out ascending in j.
// A row that outgrows `cap` keeps counting, and the host repeats the search with a larger capacity (it remembers the
// largest row).
// L lanes per robot, ROWS_BLOCK threads per workgroup.  Two shapes.  <4, 64>: one wave, 16 robots — the default.  <2, 128>: two waves,
// 64 robots, SIXTEEN workgroups for up to 1024 robots — the worlds that fill the device with their resident launch: 1000 robots
// take 125 of an XCD's 128 workgroup slots, and what such a launch leaves is two or three HOLES per XCD, each the shape of one of its
// own workgroups (two waves on two SIMDs, 40 KB of LDS) and each good for ONE workgroup of another kernel, whatever its size
// (measured with the one-wave kernel cut down to a part of the robots: 4, 8 and 16 workgroups ran beside the launch in 37 us, 24
// and more — also 32 of two lanes per robot, and 16 of FOUR waves — finished when the launch did, 110 us later).  So the search
// of such a world is sixteen workgroups in the shape of the holes.
template <int L, int ROWS_BLOCK>
__global__ void __launch_bounds__(ROWS_BLOCK) k_pairs_rows(const float *__restrict__ pos, int n, float s_max, int32_t cap,
                                                           int32_t *__restrict__ cnt, int32_t *__restrict__ rows) {
    constexpr int ROBOTS = ROWS_BLOCK / L;
    extern __shared__ float lds_pos[];
    const int npad = (n + 3) & ~3;
    float *X = lds_pos, *Y = lds_pos + npad, *Z = lds_pos + 2 * npad;
    {   // 3 n floats, 16 bytes at a time (the block is 16-byte aligned; the last, partial group goes float by float)
        const int n3 = 3 * n, n4 = n3 >> 2;
        const float4 *p4 = reinterpret_cast<const float4 *>(pos);
        constexpr int INFLIGHT = 6;
        for (int g0 = 0; g0 < n4; g0 += ROWS_BLOCK * INFLIGHT) {
            float4 v[INFLIGHT];
#pragma unroll
            for (int u = 0; u < INFLIGHT; u++) {
                const int g = g0 + u * ROWS_BLOCK + (int)threadIdx.x;
                v[u] = g < n4 ? p4[g] : float4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < INFLIGHT; u++) {
                const int g = g0 + u * ROWS_BLOCK + (int)threadIdx.x;
                if (g < n4) {
                    const float c[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int e = 4 * g + q, j = e / 3, a = e - 3 * j;
                        (a == 0 ? X : a == 1 ? Y : Z)[j] = c[q];
                    }
                }
            }
        }
        for (int e = 4 * n4 + (int)threadIdx.x; e < n3; e += ROWS_BLOCK) {
            const int j = e / 3, a = e - 3 * j;
            (a == 0 ? X : a == 1 ? Y : Z)[j] = pos[e];
        }
    }
    __syncthreads();
    const int i = blockIdx.x * ROBOTS + (int)threadIdx.x / L, l = (int)threadIdx.x % L;
    const bool live = i < n;
    const int ii = live ? i : 0;
    const float ax = X[ii], ay = Y[ii], az = Z[ii];
    int m = 0;  // hits of robot i so far (the same in all its lanes)
    int32_t *row = rows + (size_t)ii * (size_t)cap;
    const int group0 = ((int)threadIdx.x & 63) - l;  // (lane of the group's first member within its wave)
    for (int j0 = 0; j0 < n; j0 += 64 * L) {
        unsigned long long hits = 0ull;
#pragma unroll 4
        for (int k = 0; k < 64; k++) {
            const int j = j0 + L * k + l;
            if (j < n && j != i && in_comms_range_sq(ax, ay, az, X[j], Y[j], Z[j], s_max)) hits |= 1ull << k;
        }
        if (!live) hits = 0ull;
        unsigned long long mk[L];
#pragma unroll
        for (int o = 0; o < L; o++) {
            const unsigned lo = (unsigned)__shfl((int)(unsigned)hits, group0 + o, 64), hi = (unsigned)__shfl((int)(unsigned)(hits >> 32), group0 + o, 64);
            mk[o] = ((unsigned long long)hi << 32) | lo;
        }
        unsigned long long mine = hits;
        while (mine) {
            const int k = __ffsll((long long)mine) - 1;
            mine &= mine - 1ull;
            const unsigned long long below = (1ull << k) - 1ull;
            int at = m;
#pragma unroll
            for (int o = 0; o < L; o++) at += __popcll(mk[o] & below) + (o < l ? (int)((mk[o] >> k) & 1ull) : 0);
            if (at < cap) row[at] = j0 + L * k + l;
        }
#pragma unroll
        for (int o = 0; o < L; o++) m += __popcll(mk[o]);
    }
    if (live && l == 0) cnt[i] = m;
}
// ---- hash grid, ONE pass, rows of a fixed capacity (worlds of up to 1024 robots, a usable radius) ----------------------------
// The all-pairs kernel above tests a million pairs for a thousand robots; beside a resident launch, sharing its SIMDs, that is
// 70 us.  Here every workgroup (128 threads, two lanes per robot: sixteen workgroups for 1000 robots — they fit the holes such a
// launch leaves, see above) builds the SAME hash grid of all robots in its LDS — cells of radius x 1.001 over x and z, 1024
// buckets, counting sort — and a robot then looks at the 3 x 3 cells around its own: a candidate counts if its TRUE cell is the
// one being looked at (two of the nine cells may share a bucket: no candidate is seen twice, none of another cell is taken for
// one of these), it is not the robot itself, and the predicate — the squared one of the all-pairs kernel — holds.  Robots with
// a non-finite coordinate are in no cell: every robot looks at all of them as well, and they look at everybody (a NaN distance
// is "in range", as in the reference).  Same rows as the all-pairs scan by construction: the grid only leaves out pairs whose
// x or z separation alone exceeds the radius.  The two lanes of a robot share the nine cells and note their hits in LDS as they come
// (bucket order); the first lane then takes both lists into registers, sorts them (odd-even transposition, a dozen hits per
// robot) and writes the row; more than `cap`
// hits are counted only (the host then repeats the search with more room — beyond 32 per row with the all-pairs kernel).
constexpr int GRID_BLOCK = 128, GRID_ROBOTS = GRID_BLOCK / 2, GRID_M = 1024;
constexpr int NEIGHBOURS_PREV_STRIDE = 33;  // words per robot of the kept rows: count, then up to 32 entries
constexpr int32_t NEIGHBOURS_CHANGED = 1 << 30;  // in a robot's count: its row is not the one of the search before
template <int REG>
__global__ void __launch_bounds__(GRID_BLOCK) k_grid_rows(const float *__restrict__ pos, int n, float s_max, double inv_cell, int32_t cap,
                                                          int32_t *__restrict__ cnt, int32_t *__restrict__ rows, int32_t *__restrict__ prev,
                                                          int prev_valid) {
    extern __shared__ float lds_pos[];
    const int npad = (n + 3) & ~3;
    float *X = lds_pos, *Y = lds_pos + npad, *Z = lds_pos + 2 * npad;
    // Small on purpose (26.8 KB for 1000 robots): a workgroup of this kernel has to fit the LDS a resident schedule launch leaves
    // in one of its holes — 160 KB less three workgroups of up to 40 KB, in 1280-byte granules: a 38424-byte workgroup beside three
    // others left no room for the 44904 bytes this kernel once took, and the search ended with the launch instead of beside it —
    // so robot numbers (n <= 1024) are 16-bit words, and a robot's cell is computed again from its position where it is asked for.
    int32_t *fill = reinterpret_cast<int32_t *>(lds_pos + 3 * npad);    // [GRID_M] counts, then cursors
    int32_t *part = fill + GRID_M;                                      // [GRID_BLOCK]
    uint16_t *start = reinterpret_cast<uint16_t *>(part + GRID_BLOCK);  // [GRID_M + 2] (the last word pads)
    uint16_t *members = start + GRID_M + 2;                             // [n]
    uint16_t *special = members + npad;                                 // [n]
    uint16_t *hit = special + npad;                                     // [GRID_BLOCK][REG] the hits of this workgroup's robots as they are found
    __shared__ int32_t n_special;
    const int tid = (int)threadIdx.x;
    {   // all positions, 16 bytes at a time (see k_pairs_rows)
        const int n3 = 3 * n, n4 = n3 >> 2;
        const float4 *p4 = reinterpret_cast<const float4 *>(pos);
        constexpr int INFLIGHT = 6;
        for (int g0 = 0; g0 < n4; g0 += GRID_BLOCK * INFLIGHT) {
            float4 v[INFLIGHT];
#pragma unroll
            for (int u = 0; u < INFLIGHT; u++) {
                const int g = g0 + u * GRID_BLOCK + tid;
                v[u] = g < n4 ? p4[g] : float4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < INFLIGHT; u++) {
                const int g = g0 + u * GRID_BLOCK + tid;
                if (g < n4) {
                    const float c[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int e = 4 * g + q, j = e / 3, a = e - 3 * j;
                        (a == 0 ? X : a == 1 ? Y : Z)[j] = c[q];
                    }
                }
            }
        }
        for (int e = 4 * n4 + tid; e < n3; e += GRID_BLOCK) {
            const int j = e / 3, a = e - 3 * j;
            (a == 0 ? X : a == 1 ? Y : Z)[j] = pos[e];
        }
    }
    for (int b = tid; b < GRID_M; b += GRID_BLOCK) fill[b] = 0;
    if (tid == 0) n_special = 0;
    __syncthreads();
    // counting sort by bucket
    for (int j = tid; j < n; j += GRID_BLOCK) {
        if (finite3(X[j], Y[j], Z[j])) {
            atomicAdd(&fill[bucket_of(cell_of(X[j], inv_cell), cell_of(Z[j], inv_cell), GRID_M - 1)], 1);
        } else {
            special[atomicAdd(&n_special, 1)] = (uint16_t)j;
        }
    }
    __syncthreads();
    {
        constexpr int PER = GRID_M / GRID_BLOCK;
        int32_t sum = 0;
#pragma unroll
        for (int q = 0; q < PER; q++) sum += fill[tid * PER + q];
        part[tid] = sum;
        __syncthreads();
        for (int d = 1; d < GRID_BLOCK; d <<= 1) {
            const int32_t v = tid >= d ? part[tid - d] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        int32_t run = part[tid] - sum;
#pragma unroll
        for (int q = 0; q < PER; q++) {
            start[tid * PER + q] = (uint16_t)run;
            run += fill[tid * PER + q];
            fill[tid * PER + q] = 0;
        }
        if (tid == GRID_BLOCK - 1) start[GRID_M] = (uint16_t)run;
    }
    __syncthreads();
    for (int j = tid; j < n; j += GRID_BLOCK)
        if (finite3(X[j], Y[j], Z[j])) {
            const uint32_t b = bucket_of(cell_of(X[j], inv_cell), cell_of(Z[j], inv_cell), GRID_M - 1);
            members[(int)start[b] + atomicAdd(&fill[b], 1)] = (uint16_t)j;
        }
    __syncthreads();
    // two lanes per robot: lane h looks at cells h, h + 2, ... of the nine (and lane 1 at the non-finite robots)
    const int i = blockIdx.x * GRID_ROBOTS + (tid >> 1), h = tid & 1;
    const bool live = i < n;
    const int ii = live ? i : 0;
    const float ax = X[ii], ay = Y[ii], az = Z[ii];
    int32_t *row = rows + (size_t)ii * (size_t)cap;
    const bool wild = !finite3(ax, ay, az);
    // prev (may be null): the row this robot got in the search before — its connection set, when that search's pass went through
    // (the host says so: prev_valid).  Asked for here, compared at the end: a robot whose row did not change needs nothing from
    // the host's pass (no NEIGHBOURS_CHANGED in its count), and a world that follows its topology changes a fifth of its rows per tick.
    int32_t *pr = prev ? prev + (size_t)ii * NEIGHBOURS_PREV_STRIDE : nullptr;
    int32_t pv[REG];
    int pc = -1;
    if (pr && live && !wild && h == 0) {
        pc = pr[0];
#pragma unroll
        for (int p = 0; p < REG; p++) pv[p] = pr[1 + p];
    }
    int m = 0;
    if (live && wild && h == 0) {  // compared with everybody, ascending by construction
        for (int j = 0; j < n; j++)
            if (j != i && in_comms_range_sq(ax, ay, az, X[j], Y[j], Z[j], s_max)) {
                if (m < cap) row[m] = j;
                m++;
            }
        cnt[i] = pr ? (m | NEIGHBOURS_CHANGED) : m;
        if (pr) pr[0] = -1;  // (no row kept for it: changed, this time and the next)
    }
    uint16_t *mine = hit + tid * REG;  // (a lane's own REG words)
    if (live && !wild) {
        const int cx = cell_of(ax, inv_cell), cz = cell_of(az, inv_cell);
        for (int c = h; c < 9; c += 2) {
            const int tx = cx + c / 3 - 1, tz = cz + c % 3 - 1;
            const uint32_t b = bucket_of(tx, tz, GRID_M - 1);
            const int q1 = (int)start[b + 1];
            for (int q = (int)start[b]; q < q1; q++) {
                const int j = (int)members[q];
                const float bx = X[j], bz = Z[j];
                // it is in range, it is not the robot itself, and its TRUE cell is the one being looked at (members are finite)
                if (j != i && in_comms_range_sq(ax, ay, az, bx, Y[j], bz, s_max) && cell_of(bx, inv_cell) == tx && cell_of(bz, inv_cell) == tz) {
                    if (m < REG) mine[m] = (uint16_t)j;
                    m++;
                }
            }
        }
        if (h == 1) {
            const int ns = n_special;
            for (int q = 0; q < ns; q++) {
                const int j = (int)special[q];
                if (in_comms_range_sq(ax, ay, az, X[j], Y[j], Z[j], s_max)) {  // (j != i: this robot is finite)
                    if (m < REG) mine[m] = (uint16_t)j;
                    m++;
                }
            }
        }
    }
    __syncthreads();
    const int m_other = __shfl_xor(m, 1, 64);
    if (!live || wild || h == 1) return;
    const int m0 = m, m1 = m_other;
    m = m0 + m1;
    if (m > cap || m > REG) {  // the host repeats the search with more room: nobody reads this row
        cnt[i] = pr ? (m | NEIGHBOURS_CHANGED) : m;
        if (pr) pr[0] = -1;
        return;
    }
    // the row in ascending order: the hits came in bucket order — into registers, an odd-even transposition sort, out
    const uint16_t *theirs = mine + REG;
    int32_t keep[REG];
#pragma unroll
    for (int p = 0; p < REG; p++) keep[p] = p < m0 ? (int32_t)mine[p] : (p < m ? (int32_t)theirs[p - m0] : 0x7fffffff);
#pragma unroll
    for (int round = 0; round < REG; round++) {
#pragma unroll
        for (int p = round & 1; p + 1 < REG; p += 2) {
            const int32_t lo = min(keep[p], keep[p + 1]), hi = max(keep[p], keep[p + 1]);
            keep[p] = lo;
            keep[p + 1] = hi;
        }
        if (round >= m) break;  // (m entries are in order after m rounds; the padding sits behind them from the start)
    }
#pragma unroll
    for (int p = 0; p < REG; p++)
        if (p < m) row[p] = keep[p];
    if (pr) {
        bool same = prev_valid != 0 && pc == m;
#pragma unroll
        for (int p = 0; p < REG; p++)
            if (p < m) { same = same && pv[p] == keep[p]; pr[1 + p] = keep[p]; }
        pr[0] = m;
        cnt[i] = same ? m : (m | NEIGHBOURS_CHANGED);  // (the flag rides in the count: no write of its own over the host link)
    } else {
        cnt[i] = m;
    }
}

// the largest f32 s with RN_f32(sqrt(s)) <= radius (see in_comms_range_sq); NaN radius -> NaN (everybody in range), radius < 0 ->
// negative (nobody but NaN distances)
static float squared_threshold(float radius) {
    if (radius != radius) return radius;
    if (radius < 0.f) return -1.f;
    if (std::isinf(radius)) return radius;
    auto root = [](float s) { return (float)std::sqrt((double)s); };
    float s = (float)((double)radius * (double)radius);
    if (std::isinf(s)) s = std::numeric_limits<float>::max();
    while (root(s) > radius) s = std::nextafter(s, -1.f);
    for (;;) {
        const float up = std::nextafter(s, std::numeric_limits<float>::infinity());
        if (std::isinf(up) || root(up) > radius) break;
        s = up;
    }
    return s;
}
// positions from the callers' pinned block into device memory, by a kernel as small as the search itself (every workgroup of
// the search reads all of them: over the host link that would be the search's whole time)
__global__ void __launch_bounds__(64) k_stage_positions(const float *__restrict__ src, float *__restrict__ dst, int n3) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t < n3) dst[t] = src[t];
}
// LDS of one workgroup of the grid search (k_grid_rows) for n robots and rows of up to `cap` entries; its workgroups: one per 64 robots
static size_t neighbours_rows_lds(int n, int32_t cap) {
    const size_t npad = (size_t)((n + 3) & ~3);
    return sizeof(float) * 3 * npad + sizeof(int32_t) * ((size_t)GRID_M + GRID_BLOCK) +
           sizeof(uint16_t) * ((size_t)GRID_M + 2 + 2 * npad + (size_t)GRID_BLOCK * (cap <= 16 ? 16 : 32));
}
int neighbours_prev_stride() { return NEIGHBOURS_PREV_STRIDE; }
int32_t neighbours_changed_bit() { return NEIGHBOURS_CHANGED; }
// prev / prev_valid (may be null / 0): see k_grid_rows; *flagged says whether the kernel that ran marks the counts of changed rows
hipError_t neighbours_rows(const float *pos, int n, float radius, int32_t cap, int32_t *cnt, int32_t *rows, hipStream_t s, float *stage,
                           int32_t *prev, int prev_valid, bool *flagged) {
    if (flagged) *flagged = false;
    if (n <= 0) return hipSuccess;
    if (stage) {
        hipLaunchKernelGGL(k_stage_positions, dim3((unsigned)((3 * n + 63) / 64)), dim3(64), 0, s, pos, stage, 3 * n);
        pos = stage;
    }
    const size_t npad = (size_t)((n + 3) & ~3);
    const size_t lds = sizeof(float) * 3 * npad;  // <= 48 KB: the host takes this kernel for n <= 4096
    const float s_max = squared_threshold(radius);
    if (n <= GRID_M && cap <= 32 && std::isfinite(radius) && radius > 0.f) {  // the grid in LDS (a usable radius, room for the rows in registers)
        const double inv_cell = 1.0 / ((double)radius * 1.001);
        const dim3 grid((unsigned)((n + GRID_ROBOTS - 1) / GRID_ROBOTS));
        if (!flagged) prev = nullptr;  // (a caller that does not ask cannot read flagged counts)
        if (cap <= 16) hipLaunchKernelGGL(k_grid_rows<16>, grid, dim3(GRID_BLOCK), neighbours_rows_lds(n, 16), s, pos, n, s_max, inv_cell, cap, cnt, rows, prev, prev_valid);
        else hipLaunchKernelGGL(k_grid_rows<32>, grid, dim3(GRID_BLOCK), neighbours_rows_lds(n, 32), s, pos, n, s_max, inv_cell, cap, cnt, rows, prev, prev_valid);
        if (flagged) *flagged = prev != nullptr;
        return hipGetLastError();
    }
    if (n > 512 && n <= 1024)
        hipLaunchKernelGGL((k_pairs_rows<2, 128>), dim3((unsigned)((n + 63) / 64)), dim3(128), lds, s, pos, n, s_max, cap, cnt, rows);
    else
        hipLaunchKernelGGL((k_pairs_rows<4, 64>), dim3((unsigned)((n + 15) / 16)), dim3(64), lds, s, pos, n, s_max, cap, cnt, rows);
    return hipGetLastError();
}

// ---- hash grid ------------------------------------------------------------------------------------
// bucket histogram; robots with a non-finite coordinate go to the `special` list instead
__global__ void k_grid_hist(const float *__restrict__ pos, int n, double inv_cell, uint32_t mask, int32_t *__restrict__ bucket_cnt,
                            int32_t *__restrict__ special, int32_t *__restrict__ n_special) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
    if (!finite3(x, y, z)) {
        special[atomicAdd(n_special, 1)] = i;
        return;
    }
    atomicAdd(&bucket_cnt[bucket_of(cell_of(x, inv_cell), cell_of(z, inv_cell), mask)], 1);
}
__global__ void k_grid_scatter(const float *__restrict__ pos, int n, double inv_cell, uint32_t mask,
                               const int32_t *__restrict__ bucket_ptr, int32_t *__restrict__ cursor, int32_t *__restrict__ members) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
    if (!finite3(x, y, z)) return;
    const uint32_t b = bucket_of(cell_of(x, inv_cell), cell_of(z, inv_cell), mask);
    members[bucket_ptr[b] + atomicAdd(&cursor[b], 1)] = i;
}

// exclusive scan of a[0..n) into out[0..n], out[n] = total; one workgroup of 1024 threads
__global__ void __launch_bounds__(1024) k_scan(const int32_t *__restrict__ a, int n, int32_t *__restrict__ out) {
    __shared__ int32_t part[1024];
    const int t = threadIdx.x;
    const int chunk = (n + 1023) / 1024;
    const int lo = min(n, t * chunk), hi = min(n, lo + chunk);
    int32_t s = 0;
    for (int q = lo; q < hi; q++) s += a[q];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int32_t v = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int32_t run = part[t] - s;
    for (int q = lo; q < hi; q++) {
        out[q] = run;
        run += a[q];
    }
    if (t == 1023) out[n] = part[1023];
}

// Sixteen lanes per robot: lanes 0..8 each walk the bucket of one of the 3x3 cells (a cell whose bucket an
// earlier lane already has is skipped: two of the nine can hash alike), lane 9 walks the list of
// robots with non-finite coordinates; a robot that is itself non-finite is compared with everybody
// (a NaN distance counts as "in range"), all sixteen lanes striding.  The dependent loads of the nine
// buckets run side by side instead of one after the other; counts are combined with a scan inside
// the group.  FILL = second pass: lanes write their hits behind each other, lane 0 then orders the row.
constexpr int QG = 16;  // lanes per robot
template <bool FILL>
__global__ void __launch_bounds__(256) k_grid_query(const float *__restrict__ pos, int n, float radius, double inv_cell, uint32_t mask,
                                                     const int32_t *__restrict__ bucket_ptr, const int32_t *__restrict__ members,
                                                     const int32_t *__restrict__ special, const int32_t *__restrict__ n_special,
                                                     int32_t *__restrict__ cnt, const int32_t *__restrict__ ptr,
                                                     int32_t *__restrict__ idx, int32_t cap) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = t / QG, l = t % QG;
    const bool live = i < n;
    const int ii = live ? i : 0;  // every lane runs the shuffles
    const float ax = pos[3 * ii], ay = pos[3 * ii + 1], az = pos[3 * ii + 2];
    const bool wild = !finite3(ax, ay, az);
    // this lane's candidate list: [q0, q1) of `list`, or every l-th robot
    const int32_t *list = members;
    int q0 = 0, q1 = 0, stride = 1;
    uint32_t bucket = 0xffffffffu - (uint32_t)l;  // distinct dummies for the lanes without a bucket
    if (live && wild) {
        list = nullptr;
        q0 = l; q1 = n; stride = QG;
    } else if (live && l < 9) {
        const int cx = cell_of(ax, inv_cell), cz = cell_of(az, inv_cell);
        bucket = bucket_of(cx + (l % 3) - 1, cz + (l / 3) - 1, mask);
    } else if (live && l == 9) {
        list = special;
        q0 = 0; q1 = *n_special;
    }
    bool dup = false;
    for (int c = 0; c < 8; c++) {  // has an earlier lane of the group the same bucket?
        const uint32_t other = __shfl(bucket, (threadIdx.x & 63 & ~(QG - 1)) + c, 64);
        dup |= (c < l) && other == bucket;
    }
    if (live && !wild && l < 9 && !dup) {
        q0 = bucket_ptr[bucket];
        q1 = bucket_ptr[bucket + 1];
    }
    int m = 0;
    auto hit = [&](int q) {
        const int j = list ? list[q] : q;
        return j != ii && in_comms_range(ax, ay, az, pos[3 * j], pos[3 * j + 1], pos[3 * j + 2], radius) ? j : -1;
    };
    for (int q = q0; q < q1; q += stride) m += hit(q) >= 0;
    // exclusive scan of the counts over the group
    int incl = m;
    for (int d = 1; d < QG; d <<= 1) {
        const int v = __shfl_up(incl, d, QG);
        if (l >= d) incl += v;
    }
    const int total = __shfl(incl, (threadIdx.x & 63 & ~(QG - 1)) + QG - 1, 64);
    if (!live) return;
    if (!FILL) {
        if (l == 0) cnt[i] = total;
        return;
    }
    if (ptr[n] > cap) return;  // speculative second pass: the rows do not fit the buffer the host guessed
    const int32_t base = ptr[i];
    int w = base + incl - m;
    for (int q = q0; q < q1; q += stride) {
        const int j = hit(q);
        if (j >= 0) idx[w++] = j;
    }
    __threadfence_block();  // the group's stores before lane 0 reads them back
    if (l == 0)              // rows ascending in robot index (candidates arrive in bucket order)
        for (int a = 1; a < total; a++) {
            const int32_t v = idx[base + a];
            int b = a;
            while (b > 0 && idx[base + b - 1] > v) {
                idx[base + b] = idx[base + b - 1];
                b--;
            }
            idx[base + b] = v;
        }
}

// ---- host-side sequencing ---------------------------------------------------------------------------
// Scratch (all device): cnt[n], bucket_cnt[M], bucket_ptr[M+1], cursor[M], members[n], special[n],
// n_special[1].  `ptr` is [n+1].  Phase 1 leaves the row counts scanned in `ptr`; the caller
// reads ptr[n], sizes `idx` and runs phase 2.
hipError_t neighbours_count(const float *pos, int n, float radius, bool grid, uint32_t M, int32_t *cnt, int32_t *bucket_cnt,
                            int32_t *bucket_ptr, int32_t *cursor, int32_t *members, int32_t *special, int32_t *n_special,
                            int32_t *ptr, hipStream_t s) {
    if (n <= 0) return hipMemsetAsync(ptr, 0, sizeof(int32_t), s);
    const dim3 g256((unsigned)((n + 255) / 256));
    if (grid) {
        const double inv_cell = 1.0 / ((double)radius * 1.001);
        hipError_t e = hipMemsetAsync(bucket_cnt, 0, sizeof(int32_t) * M, s);
        if (e != hipSuccess) return e;
        if ((e = hipMemsetAsync(cursor, 0, sizeof(int32_t) * M, s)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(n_special, 0, sizeof(int32_t), s)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_grid_hist, g256, dim3(256), 0, s, pos, n, inv_cell, M - 1, bucket_cnt, special, n_special);
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, bucket_cnt, (int)M, bucket_ptr);
        hipLaunchKernelGGL(k_grid_scatter, g256, dim3(256), 0, s, pos, n, inv_cell, M - 1, bucket_ptr, cursor, members);
        hipLaunchKernelGGL(k_grid_query<false>, dim3((unsigned)(((size_t)n * QG + 255) / 256)), dim3(256), 0, s, pos, n, radius, inv_cell, M - 1, bucket_ptr, members, special,
                           n_special, cnt, (const int32_t *)nullptr, (int32_t *)nullptr, 0);
    } else {
        hipLaunchKernelGGL(k_pairs<false>, g256, dim3(256), 0, s, pos, n, radius, cnt, (const int32_t *)nullptr, (int32_t *)nullptr, 0);
    }
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, cnt, n, ptr);
    return hipGetLastError();
}
hipError_t neighbours_fill(const float *pos, int n, float radius, bool grid, uint32_t M, const int32_t *bucket_ptr,
                           const int32_t *members, const int32_t *special, const int32_t *n_special, const int32_t *ptr,
                           int32_t *idx, int32_t cap, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const dim3 g256((unsigned)((n + 255) / 256));
    if (grid) {
        const double inv_cell = 1.0 / ((double)radius * 1.001);
        hipLaunchKernelGGL(k_grid_query<true>, dim3((unsigned)(((size_t)n * QG + 255) / 256)), dim3(256), 0, s, pos, n, radius, inv_cell, M - 1, bucket_ptr, members, special,
                           n_special, (int32_t *)nullptr, ptr, idx, cap);
    } else {
        hipLaunchKernelGGL(k_pairs<true>, g256, dim3(256), 0, s, pos, n, radius, (int32_t *)nullptr, ptr, idx, cap);
    }
    return hipGetLastError();
}

}  // namespace mgx
