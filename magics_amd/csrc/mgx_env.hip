// mgx_env.hip — the environment rasteriser on the device: env_to_png::env_to_image /
// env_to_sdf_image (crates/env_to_png/src/lib.rs:149-479) with the placeable shapes of
// gbp_environment (crates/gbp_environment/src/lib.rs:114-530), and image 0.25.1's
// imageops::blur (Cargo.lock:4052; vertical_sample + horizontal_sample, Gaussian of support
// 2 sigma) — what simulation_loader.rs:154-162 runs for every scenario before the first robot
// spawns.  Output: the u8 image ObstacleFactor::measure samples (factor/obstacle.rs:141-188).
//
// Division of labour.  Everything that is per ENVIRONMENT is computed on the host with the C
// library the reference's std calls (sinf / cosf / expf / sin / cos): tile thresholds, shape
// vertices, the rotation quaternion of every obstacle, the normalised blur weights of every output
// row / column.  Everything that is per PIXEL runs on the device and uses only + - * / floor ceil
// and comparisons in the reference's types (f32 for the coordinate maps, the rotation, circles and
// triangles; f64 for polygons and rectangles), which are correctly rounded on both sides, so the
// image is bit-identical to the CPU restatement (oracle/env.py).  FMA contraction is off for this
// file whatever the build (a*b+c fused would move pixels on shape borders).
//
// Byte work, HBM-bound: one thread per pixel, rows coalesced; the red plane only (the reference
// writes R = G = B and the factor reads pixel[0]).  The blur's vertical pass re-reads each source
// row once per tap through L2.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/mgx.h"

#pragma clang fp contract(off)

extern "C" int mgx_set_error_(int code, const char *text);  // mgx_world.hip: thread-local last error

namespace mgx {

static int env_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return mgx_set_error_(code, buf);
}
#define ENV_HIP(expr)                                                                               \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) return env_fail(MGX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// one placeable obstacle, constants only (device record)
struct EnvObstacle {
    int32_t kind, row, col;
    int32_t n, p0;     // regular polygon / polygon: vertex count and first vertex in `pts`
    float tx, ty;      // translation as f32 (Vec2::from(RelativePoint))
    float s, c, k;     // Quat::from_rotation_z: sin, cos of the half angle; k = c c - s s
    float f[6];        // circle: f[0] = (r r) as f32; triangle: ax ay bx by cx cy
    double hw, hh;     // rectangle: width / 4, height / 4
};

struct EnvRaster {
    uint32_t W, H, n_rows, n_cols, resolution;
    float res_f, tile_size;
    float ow, owp, lo, hi;  // is_tile_obstacle thresholds (lib.rs:346-351,368,376,384,392)
    int n_obstacles;
    const uint32_t *tiles;
    const EnvObstacle *obstacles;
    const double *pts;  // (x, y) pairs
    int *error;         // set when a Percentage::new assertion of the reference would fire
};

// tile_units_to_percentage(image_to_tile_units(..)) (lib.rs:208-243)
__device__ __forceinline__ float pixel_percentage(uint32_t p, float res, float tile_size) {
    const float u = ((float)p + 0.5f) / res * tile_size;
    return -(ceilf(u / tile_size) * tile_size - u) / tile_size + 1.0f;
}

__device__ bool tile_obstacle(uint32_t tile, float px, float py, const EnvRaster &e) {  // lib.rs:338-479
    const bool xl = px < e.ow, xh = px > e.owp, yl = py < e.ow, yh = py > e.owp;
    switch (tile) {
    case 0x2500: return yl || yh;                    // ─
    case 0x2502: return xl || xh;                    // │
    case 0x2574: return yl || yh || px > e.lo;       // ╴
    case 0x2576: return yl || yh || px < e.hi;       // ╶
    case 0x2577: return xl || xh || py < e.hi;       // ╷
    case 0x2575: return xl || xh || py > e.lo;       // ╵
    case 0x250C: return xl || yl || (xh && yh);      // ┌
    case 0x2510: return xh || yl || (xl && yh);      // ┐
    case 0x2514: return xl || yh || (xh && yl);      // └
    case 0x2518: return xh || yh || (xl && yl);      // ┘
    case 0x252C: return yl || (yh && (xl || xh));    // ┬
    case 0x2534: return yh || (yl && (xl || xh));    // ┴
    case 0x251C: return xl || (xh && (yl || yh));    // ├
    case 0x2524: return xh || (xl && (yl || yh));    // ┤
    case 0x253C: return (xl || xh) && (yl || yh);    // ┼
    case 0x20: return true;                          // ' '
    default: return false;
    }
}

__device__ bool shape_inside(const EnvObstacle &o, const double *pts, float x, float y) {
    switch (o.kind) {
    case MGX_SHAPE_CIRCLE:  // Circle::inside (gbp_environment lib.rs:139-143)
        return x * x + y * y <= o.f[0];
    case MGX_SHAPE_TRIANGLE: {  // Triangle::inside (lib.rs:206-223)
        const float ax = o.f[0], ay = o.f[1], bx = o.f[2], by = o.f[3], cx = o.f[4], cy = o.f[5];
        const float d1 = (x - bx) * (ay - by) - (ax - bx) * (y - by);
        const float d2 = (x - cx) * (by - cy) - (bx - cx) * (y - cy);
        const float d3 = (x - ax) * (cy - ay) - (cx - ax) * (y - ay);
        const bool has_neg = d1 < 0.0f || d2 < 0.0f || d3 < 0.0f, has_pos = d1 > 0.0f || d2 > 0.0f || d3 > 0.0f;
        return !(has_neg && has_pos);
    }
    case MGX_SHAPE_REGULAR_POLYGON: {  // RegularPolygon::inside (lib.rs:283-300)
        const double X = (double)x * 2.0, Y = (double)y * 2.0;
        bool inside = false;
        int j = o.n - 1;
        for (int i = 0; i < o.n; i++) {
            const double xi = pts[2 * (o.p0 + i)], yi = pts[2 * (o.p0 + i) + 1];
            const double xj = pts[2 * (o.p0 + j)], yj = pts[2 * (o.p0 + j) + 1];
            if ((yi < Y && yj >= Y) || (yj < Y && yi >= Y))
                if (xi + (Y - yi) / (yj - yi) * (xj - xi) < X) inside = !inside;
            j = i;
        }
        return inside;
    }
    case MGX_SHAPE_RECTANGLE: {  // Rectangle::inside (lib.rs:328-340)
        const double X = (double)x, Y = (double)y;
        return X >= -o.hh && X <= o.hh && Y >= -o.hw && Y <= o.hw;
    }
    case MGX_SHAPE_POLYGON: {  // is_point_in_polygon (lib.rs:398-415)
        const double X = (double)x, Y = (double)y;
        bool inside = false;
        int j = o.n - 1;
        for (int i = 0; i < o.n; i++) {
            const double ix = pts[2 * (o.p0 + i)], iy = pts[2 * (o.p0 + i) + 1];
            const double jx = pts[2 * (o.p0 + j)], jy = pts[2 * (o.p0 + j) + 1];
            if ((iy > Y) != (jy > Y) && X < (jx - ix) * (Y - iy) / (jy - iy) + ix) inside = !inside;
            j = i;
        }
        return inside;
    }
    default: return false;
    }
}

// four horizontally adjacent bytes per thread go out as one 32-bit store when the rows allow it
__device__ __forceinline__ void store4(uint8_t *__restrict__ out, uint32_t W, uint32_t x, uint32_t y, const uint8_t (&v)[4]) {
    uint8_t *p = out + (size_t)y * W + x;
    if ((W & 3u) == 0 && x + 3 < W) {
        *reinterpret_cast<uint32_t *>(p) = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
    } else {
        for (uint32_t i = 0; i < 4 && x + i < W; i++) p[i] = v[i];
    }
}

// env_to_image (lib.rs:165-206): one thread per four pixels of a row of the red plane
__global__ void __launch_bounds__(256) k_env_raster(EnvRaster e, uint8_t *__restrict__ out) {
    const uint32_t x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y;
    if (x0 >= e.W || y >= e.H) return;
    const uint32_t tcy = (uint32_t)floorf((float)y / e.res_f);  // lib.rs:248-257
    const float py = pixel_percentage(y, e.res_f, e.tile_size);
    uint8_t v[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < 4 && x0 + i < e.W; i++) {
        const uint32_t x = x0 + i;
        const uint32_t tcx = (uint32_t)floorf((float)x / e.res_f);
        const float px = pixel_percentage(x, e.res_f, e.tile_size);
        if (!(px >= 0.0f && px <= 1.0f && py >= 0.0f && py <= 1.0f) || tcx >= e.n_cols || tcy >= e.n_rows) {
            *e.error = 1;  // Percentage::new assert / "Tile not found"
            continue;
        }
        bool obstacle = tile_obstacle(e.tiles[tcy * e.n_cols + tcx], px, py, e);
        for (int q = 0; q < e.n_obstacles && !obstacle; q++) {  // is_placeable_obstacle (lib.rs:279-336)
            const EnvObstacle &o = e.obstacles[q];
            if ((uint32_t)o.col != tcx || (uint32_t)o.row != tcy) continue;
            const float tx = px - o.tx, ty = py - o.ty;
            // glam Quat::from_rotation_z(a).mul_vec3((tx, ty, 0)): v (w w - b.b) + b (2 v.b) + (w (b x v)) 2, b = (0, 0, s)
            const float rx = (tx * o.k + 0.0f) + (o.c * (0.0f - ty * o.s)) * 2.0f;
            const float ry = (ty * o.k + 0.0f) + (o.c * (o.s * tx)) * 2.0f;
            obstacle = shape_inside(o, e.pts, rx, ry);
        }
        v[i] = obstacle ? 0 : 255;
    }
    store4(out, e.W, x0, y, v);
}

// imageops::blur, first pass (vertical_sample): u8 plane -> f32 plane.  Row `o` of the output is the
// weighted sum of rows left[o] .. left[o] + cnt[o] - 1, accumulated in f32 in tap order.  One thread
// per four pixels of a row: a 32-bit load per tap, a 16-byte store.
__global__ void __launch_bounds__(256) k_blur_vertical(const uint8_t *__restrict__ src, float *__restrict__ tmp, uint32_t W, uint32_t H,
                                                       const int32_t *__restrict__ left, const int32_t *__restrict__ cnt,
                                                       const float *__restrict__ wgt, int T) {
    const uint32_t x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, o = blockIdx.y;
    if (x0 >= W || o >= H) return;
    const int l = left[o], n = cnt[o];
    const float *w = wgt + (size_t)o * T;
    float t[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if ((W & 3u) == 0) {
        for (int i = 0; i < n; i++) {
            const uint32_t q = *reinterpret_cast<const uint32_t *>(src + (size_t)(l + i) * W + x0);
            const float wi = w[i];
#pragma unroll
            for (int c = 0; c < 4; c++) t[c] += (float)((q >> (8 * c)) & 0xffu) * wi;
        }
        *reinterpret_cast<float4 *>(tmp + (size_t)o * W + x0) = make_float4(t[0], t[1], t[2], t[3]);
    } else {
        for (uint32_t c = 0; c < 4 && x0 + c < W; c++) {
            for (int i = 0; i < n; i++) t[c] += (float)src[(size_t)(l + i) * W + x0 + c] * w[i];
            tmp[(size_t)o * W + x0 + c] = t[c];
        }
    }
}
// second pass (horizontal_sample): f32 plane -> u8, clamp to [0, 255] and round half away from zero.
// One output per thread; the 256 outputs of a workgroup read overlapping windows of one row, so the
// row segment they cover (256 + taps floats) is staged in LDS once instead of being fetched once per
// tap through L1 / L2; four lanes then merge their bytes so that every fourth lane issues one
// 32-bit store.  The weight table is tap-major ([tap][column]) so that a wave's weight loads coalesce.
constexpr int BLUR_TAPS_REG = 16;
__global__ void __launch_bounds__(256) k_blur_horizontal(const float *__restrict__ tmp, uint8_t *__restrict__ out, uint32_t W, uint32_t H,
                                                         const int32_t *__restrict__ left, const int32_t *__restrict__ cnt,
                                                         const float *__restrict__ wgt, int T, uint32_t rows) {
    // `rows` rows per workgroup: the column's weights are fetched once and reused (large images); small
    // images keep one row per workgroup so that the grid still fills the chip
    extern __shared__ float s_row[];
    const uint32_t o_first = blockIdx.x * blockDim.x, o = o_first + threadIdx.x, y0 = blockIdx.y * rows;
    const uint32_t o_last = (o_first + blockDim.x - 1 < W) ? o_first + blockDim.x - 1 : W - 1;
    const int lo = left[o_first], hi = left[o_last] + cnt[o_last];  // windows move right with the output
    const int l = (o < W) ? left[o] - lo : 0, n = (o < W) ? cnt[o] : 0;
    float wreg[BLUR_TAPS_REG];  // this column's weights, when they fit (sigma up to ~3 pixels)
    const bool in_regs = T <= BLUR_TAPS_REG;
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < BLUR_TAPS_REG; i++) wreg[i] = (i < n) ? wgt[(size_t)i * W + o] : 0.0f;
    }
    for (uint32_t y = y0; y < y0 + rows && y < H; y++) {
        const float *row = tmp + (size_t)y * W;
        __syncthreads();  // the previous row's readers are done
        for (int j = threadIdx.x; j < hi - lo; j += blockDim.x) s_row[j] = row[lo + j];
        __syncthreads();
        uint32_t v = 0;
        if (o < W) {
            float t = 0.0f;
            if (in_regs) {
#pragma unroll
                for (int i = 0; i < BLUR_TAPS_REG; i++)
                    if (i < n) t += s_row[l + i] * wreg[i];
            } else {
                for (int i = 0; i < n; i++) t += s_row[l + i] * wgt[(size_t)i * W + o];
            }
            t = (t < 0.0f) ? 0.0f : (t > 255.0f ? 255.0f : t);
            v = (uint32_t)roundf(t);
        }
        const uint32_t v1 = __shfl_down(v, 1, 64), v2 = __shfl_down(v, 2, 64), v3 = __shfl_down(v, 3, 64);
        if (o < W) {
            if ((W & 3u) == 0) {
                if ((o & 3u) == 0) *reinterpret_cast<uint32_t *>(out + (size_t)y * W + o) = v | (v1 << 8) | (v2 << 16) | (v3 << 24);
            } else {
                out[(size_t)y * W + o] = (uint8_t)v;
            }
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------
namespace {

bool percentage_ok(float v) { return v >= 0.0f && v <= 1.0f; }  // Percentage::new (lib.rs:52-57)
bool spf(double v) { return v > 0.0 && std::isfinite(v); }      // StrictlyPositiveFinite

// image::imageops::sample::gaussian
float gaussian(float x, float r) { return 1.0f / (sqrtf(2.0f * 3.14159265358979323846f) * r) * expf(-(x * x) / (2.0f * (r * r))); }

// taps of {vertical,horizontal}_sample when the new size equals the old one (ratio = sratio = 1)
void blur_taps(uint32_t n, float sigma, std::vector<int32_t> &left, std::vector<int32_t> &cnt, std::vector<float> &wgt, int &T) {
    const float support = 2.0f * sigma;
    T = (int)std::ceil(2.0 * (double)support) + 3;
    left.assign(n, 0);
    cnt.assign(n, 0);
    wgt.assign((size_t)n * T, 0.0f);
    for (uint32_t o = 0; o < n; o++) {
        float inp = ((float)o + 0.5f) * 1.0f;
        long long l = (long long)floorf(inp - support);
        l = l < 0 ? 0 : (l > (long long)n - 1 ? (long long)n - 1 : l);
        long long r = (long long)ceilf(inp + support);
        r = r < l + 1 ? l + 1 : (r > (long long)n ? (long long)n : r);
        inp = inp - 0.5f;
        float sum = 0.0f;
        float *w = wgt.data() + (size_t)o * T;
        int c = 0;
        for (long long i = l; i < r && c < T; i++, c++) {
            w[c] = gaussian(((float)i - inp) / 1.0f, sigma);
            sum += w[c];
        }
        for (int i = 0; i < c; i++) w[i] /= sum;
        left[o] = (int32_t)l;
        cnt[o] = c;
    }
}

template <class T>
struct Dev {
    T *p = nullptr;
    ~Dev() { if (p) (void)hipFree(p); }
    hipError_t put(const std::vector<T> &h, hipStream_t s) {
        hipError_t e = hipMalloc((void **)&p, sizeof(T) * (h.empty() ? 1 : h.size()));
        if (e != hipSuccess || h.empty()) return e;
        return hipMemcpyAsync(p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice, s);
    }
    hipError_t alloc(size_t n) { return hipMalloc((void **)&p, sizeof(T) * (n ? n : 1)); }
};

}  // namespace

// Rasterises (and blurs when blur_percent * resolution >= 1) on stream `s`; returns the red plane.
int env_red_plane(const mgx_env_desc *d, uint32_t resolution, float expansion, float blur_percent, bool with_blur, hipStream_t s,
                  std::vector<uint8_t> &red, uint32_t &W, uint32_t &H) {
    if (!d || !d->tiles || !d->n_rows || !d->n_cols) return env_fail(MGX_ERR_INVALID, "EmptyGrid: environment matrix representation is empty");
    if (d->n_obstacles && !d->obstacles) return env_fail(MGX_ERR_INVALID, "null obstacle list");
    if (!resolution) return env_fail(MGX_ERR_INVALID, "Pixels per tile must be non-zero");
    if (!percentage_ok(expansion) || !percentage_ok(d->path_width) || (with_blur && !percentage_ok(blur_percent)))
        return env_fail(MGX_ERR_INVALID, "percentage outside [0, 1]");
    if ((uint64_t)d->n_cols * resolution > (1u << 24) || (uint64_t)d->n_rows * resolution > 65535u)
        return env_fail(MGX_ERR_INVALID, "image too large");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return env_fail(MGX_ERR_NO_DEVICE, "no usable HIP device (the rasteriser has no CPU path)");

    EnvRaster e{};
    e.W = W = d->n_cols * resolution;
    e.H = H = d->n_rows * resolution;
    e.n_rows = d->n_rows; e.n_cols = d->n_cols; e.resolution = resolution;
    e.res_f = (float)resolution; e.tile_size = d->tile_size;
    {  // is_tile_obstacle's constants (lib.rs:346-351); each is a Percentage::new
        const float pw = d->path_width - expansion;
        const float almost_full = 1.0f - pw;
        e.ow = almost_full / 2.0f;
        e.owp = e.ow + pw;
        e.lo = 0.5f - expansion / 2.0f;
        e.hi = 0.5f + expansion / 2.0f;
        if (!percentage_ok(pw) || !percentage_ok(almost_full) || !percentage_ok(e.ow) || !percentage_ok(e.owp) || !percentage_ok(e.lo))
            return env_fail(MGX_ERR_INVALID, "percentage outside [0, 1] (path-width %g, expansion %g)", (double)d->path_width, (double)expansion);
        bool uses_hi = false;
        for (uint32_t i = 0; i < d->n_rows * d->n_cols; i++) uses_hi |= d->tiles[i] == 0x2576 || d->tiles[i] == 0x2577;
        if (uses_hi && !percentage_ok(e.hi)) return env_fail(MGX_ERR_INVALID, "percentage outside [0, 1]");
    }
    std::vector<EnvObstacle> obs;
    std::vector<double> pts;
    const double ex = (double)expansion;  // PlaceableShape::expanded(expansion.0 as f64), lib.rs:295
    const float PI32 = 3.14159265358979323846f, HALF_PI32 = 1.57079632679489661923f;
    for (uint32_t q = 0; q < d->n_obstacles; q++) {
        const mgx_env_obstacle &in = d->obstacles[q];
        EnvObstacle o{};
        o.kind = in.shape; o.row = in.tile_row; o.col = in.tile_col;
        if (!(in.translation_x >= 0.0 && in.translation_x <= 1.0 && in.translation_y >= 0.0 && in.translation_y <= 1.0))
            return env_fail(MGX_ERR_INVALID, "obstacle %u: Invalid relative point", q);
        if (!(in.rotation >= 0.0 && in.rotation <= 2.0 * 3.14159265358979323846))
            return env_fail(MGX_ERR_INVALID, "obstacle %u: Angle value %g is not inside [0,2pi]", q, in.rotation);
        o.tx = (float)in.translation_x; o.ty = (float)in.translation_y;
        float offset = HALF_PI32;  // lib.rs:299-312
        switch (in.shape) {
        case MGX_SHAPE_CIRCLE: {
            if (!spf(in.radius) || !spf(in.radius + ex)) return env_fail(MGX_ERR_INVALID, "obstacle %u: radius must be strictly positive and finite", q);
            const double r = in.radius + ex;
            o.f[0] = (float)(r * r);
            break;
        }
        case MGX_SHAPE_TRIANGLE: {  // Triangle::points (gbp_environment lib.rs:186-204)
            if (!spf(in.radius) || !spf(in.radius + ex)) return env_fail(MGX_ERR_INVALID, "obstacle %u: radius must be strictly positive and finite", q);
            const float a = (float)in.angle_a, b = (float)in.angle_b, c = PI32 - (a + b), r = (float)(in.radius + ex);
            const float hyp[3] = {r / sinf(a), r / sinf(b), r / sinf(c)};
            const float ang[3] = {PI32 + a / 2.0f, -b / 2.0f, PI32 - b - c / 2.0f};
            for (int v = 0; v < 3; v++) {
                o.f[2 * v] = cosf(ang[v]) * hyp[v];
                o.f[2 * v + 1] = sinf(ang[v]) * hyp[v];
            }
            break;
        }
        case MGX_SHAPE_REGULAR_POLYGON: {  // RegularPolygon::expanded / point_at (lib.rs:246-279)
            if (!spf(in.radius) || !spf(in.radius + ex * 2.0) || in.sides < 1) return env_fail(MGX_ERR_INVALID, "obstacle %u: bad regular polygon", q);
            const double r = in.radius + ex * 2.0;
            o.n = (int32_t)in.sides; o.p0 = (int32_t)(pts.size() / 2);
            for (uint32_t i = 0; i < in.sides; i++) {
                const double angle = 2.0 * 3.14159265358979323846 / (double)in.sides * (double)i + 0.78539816339744830962;
                pts.push_back(cos(angle) * r);
                pts.push_back(sin(angle) * r);
            }
            offset = HALF_PI32 + HALF_PI32 + ((in.sides % 2 != 0) ? PI32 / (float)in.sides : 0.0f);
            break;
        }
        case MGX_SHAPE_RECTANGLE: {  // Rectangle::expanded / inside (lib.rs:318-340)
            if (!spf(in.width) || !spf(in.height) || !spf(in.width + ex * 2.0) || !spf(in.height + ex * 2.0))
                return env_fail(MGX_ERR_INVALID, "obstacle %u: width / height must be strictly positive and finite", q);
            o.hw = (in.width + ex * 2.0) / 4.0;
            o.hh = (in.height + ex * 2.0) / 4.0;
            break;
        }
        case MGX_SHAPE_POLYGON: {  // Polygon::expanded (lib.rs:352-380)
            if (!in.n_points || !in.points_xy) return env_fail(MGX_ERR_INVALID, "obstacle %u: polygon without points", q);
            double ax = 0.0, ay = 0.0;
            for (uint32_t i = 0; i < in.n_points; i++) { ax = ax + in.points_xy[2 * i]; ay = ay + in.points_xy[2 * i + 1]; }
            const double cx = ax / (double)in.n_points, cy = ay / (double)in.n_points;
            o.n = (int32_t)in.n_points; o.p0 = (int32_t)(pts.size() / 2);
            for (uint32_t i = 0; i < in.n_points; i++) {
                const double x = in.points_xy[2 * i], y = in.points_xy[2 * i + 1];
                pts.push_back(x + (x - cx) * 4.0 * ex);
                pts.push_back(y + (y - cy) * 4.0 * ex);
            }
            offset = 0.0f;
            break;
        }
        default: return env_fail(MGX_ERR_INVALID, "obstacle %u: unknown shape %d", q, in.shape);
        }
        const float half = ((float)in.rotation + offset) * 0.5f;  // Quat::from_rotation_z
        o.s = sinf(half); o.c = cosf(half);
        o.k = o.c * o.c - o.s * o.s;
        obs.push_back(o);
    }

    Dev<uint32_t> tiles_d;
    Dev<EnvObstacle> obs_d;
    Dev<double> pts_d;
    Dev<int> err_d;
    Dev<uint8_t> plane, blurred;
    std::vector<uint32_t> tiles(d->tiles, d->tiles + (size_t)d->n_rows * d->n_cols);
    ENV_HIP(tiles_d.put(tiles, s));
    ENV_HIP(obs_d.put(obs, s));
    ENV_HIP(pts_d.put(pts, s));
    ENV_HIP(err_d.alloc(1));
    ENV_HIP(hipMemsetAsync(err_d.p, 0, sizeof(int), s));
    ENV_HIP(plane.alloc((size_t)W * H));
    e.n_obstacles = (int)obs.size();
    e.tiles = tiles_d.p; e.obstacles = obs_d.p; e.pts = pts_d.p; e.error = err_d.p;
    const dim3 grid((W + 1023) / 1024, H), block(256);  // four pixels of a row per thread
    hipLaunchKernelGGL(k_env_raster, grid, block, 0, s, e, plane.p);
    ENV_HIP(hipGetLastError());
    const uint8_t *result = plane.p;

    // env_to_sdf_image (lib.rs:149-163)
    std::vector<int32_t> vl, vc, hl, hc;
    std::vector<float> vw, hw;
    Dev<int32_t> vl_d, vc_d, hl_d, hc_d;
    Dev<float> vw_d, hw_d, tmp;
    const float blur_pixels = blur_percent * (float)resolution;
    if (with_blur && !(blur_pixels < 1.0f)) {
        const float sigma = blur_pixels <= 0.0f ? 1.0f : blur_pixels;
        int TV = 0, TH = 0;
        blur_taps(H, sigma, vl, vc, vw, TV);
        blur_taps(W, sigma, hl, hc, hw, TH);
        {  // the horizontal pass reads its weights tap-major ([tap][column]) so that a wave's loads coalesce
            std::vector<float> t((size_t)W * TH);
            for (uint32_t o = 0; o < W; o++)
                for (int i = 0; i < TH; i++) t[(size_t)i * W + o] = hw[(size_t)o * TH + i];
            hw.swap(t);
        }
        ENV_HIP(vl_d.put(vl, s)); ENV_HIP(vc_d.put(vc, s)); ENV_HIP(vw_d.put(vw, s));
        ENV_HIP(hl_d.put(hl, s)); ENV_HIP(hc_d.put(hc, s)); ENV_HIP(hw_d.put(hw, s));
        const uint32_t hrows = ((uint64_t)W * H >= (1u << 22)) ? 16u : ((uint64_t)W * H >= (1u << 19) ? 4u : 1u);
        ENV_HIP(tmp.alloc((size_t)W * H));
        ENV_HIP(blurred.alloc((size_t)W * H));
        hipLaunchKernelGGL(k_blur_vertical, grid, block, 0, s, plane.p, tmp.p, W, H, vl_d.p, vc_d.p, vw_d.p, TV);
        hipLaunchKernelGGL(k_blur_horizontal, dim3((W + 255) / 256, (H + hrows - 1) / hrows), block, sizeof(float) * (size_t)(256 + TH + 2), s, tmp.p, blurred.p, W, H,
                           hl_d.p, hc_d.p, hw_d.p, TH, hrows);
        ENV_HIP(hipGetLastError());
        result = blurred.p;
    }
    red.resize((size_t)W * H);
    int err = 0;
    ENV_HIP(hipMemcpyAsync(red.data(), result, red.size(), hipMemcpyDeviceToHost, s));
    ENV_HIP(hipMemcpyAsync(&err, err_d.p, sizeof(int), hipMemcpyDeviceToHost, s));
    ENV_HIP(hipStreamSynchronize(s));
    if (err) return env_fail(MGX_ERR_INVALID, "percentage outside [0, 1] or tile not found while rasterising");
    return MGX_OK;
}

}  // namespace mgx

extern "C" {

int mgx_env_image_size(const mgx_env_desc *env, uint32_t resolution, uint32_t *width, uint32_t *height) {
    if (!env || !width || !height || !resolution) return mgx::env_fail(MGX_ERR_INVALID, "bad arguments");
    *width = env->n_cols * resolution;
    *height = env->n_rows * resolution;
    return MGX_OK;
}

static int to_rgb(const mgx_env_desc *env, uint32_t resolution, float expansion, float blur, bool with_blur, uint8_t *rgb) {
    if (!rgb) return mgx::env_fail(MGX_ERR_INVALID, "null image buffer");
    std::vector<uint8_t> red;
    uint32_t W = 0, H = 0;
    const int rc = mgx::env_red_plane(env, resolution, expansion, blur, with_blur, nullptr, red, W, H);
    if (rc != MGX_OK) return rc;
    for (size_t i = 0; i < red.size(); i++) rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = red[i];
    return MGX_OK;
}

int mgx_env_to_image(const mgx_env_desc *env, uint32_t resolution, float expansion, uint8_t *rgb) {
    return to_rgb(env, resolution, expansion, 0.0f, false, rgb);
}

int mgx_env_to_sdf_image(const mgx_env_desc *env, uint32_t resolution, float expansion, float blur_percent, uint8_t *rgb) {
    return to_rgb(env, resolution, expansion, blur_percent, true, rgb);
}

}  // extern "C"
