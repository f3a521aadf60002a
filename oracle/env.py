"""TEST INFRASTRUCTURE — CPU restatement (numpy, f32/f64 exactly as the reference types them) of the
environment rasteriser: `env_to_png::env_to_sdf_image` (crates/env_to_png/src/lib.rs:149-479) with
the shapes of `gbp_environment` (crates/gbp_environment/src/lib.rs:114-530).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline may import this; the product rasterises on the
GPU (magics_amd/csrc/mgx_env.hip).

Third-party arithmetic restated from the published algorithms (absent from /root/reference):
  * glam 0.25.0 (Cargo.lock:3713)  Quat::from_rotation_z, Quat::mul_vec3, Vec2::from_angle
  * image 0.25.1 (Cargo.lock:4052) imageops::blur = vertical_sample + horizontal_sample with a
    Gaussian kernel of support 2 sigma, f32 accumulation, round-half-away on the way back to u8
The reference holds no numeric vectors for either => parity UNPINNED by the reference for the blur
and the rotated shapes; the four unit tests of env_to_png (lib.rs:482-533) pin the coordinate maps
and the tile predicate (tests/test_env_oracle.py).

Transcendentals go through the C library (sinf, cosf, expf, sin, cos of libm.so.6 — what Rust's
std calls on linux-gnu), never numpy's own SIMD kernels, so that both sides of a parity test see
the same constants.
"""
import ctypes
import ctypes.util
import math

import numpy as np

F = np.float32
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _n in ("sinf", "cosf", "expf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
for _n in ("sin", "cos"):
    getattr(_libm, _n).restype = ctypes.c_double
    getattr(_libm, _n).argtypes = [ctypes.c_double]


def sinf(x):
    return F(_libm.sinf(float(F(x))))


def cosf(x):
    return F(_libm.cosf(float(F(x))))


def expf(x):
    return F(_libm.expf(float(F(x))))


PI32, FRAC_PI_2_32 = F(math.pi), F(math.pi / 2)


class EnvError(ValueError):
    """The reference panics (assert! in Percentage::new, lib.rs:52-57) or returns Err."""


def _percentage(v):  # Percentage::new (lib.rs:52-57)
    v = F(v)
    if not (v >= 0.0 and v <= 1.0):
        raise EnvError(f"percentage {v} outside [0, 1]")
    return v


def image_to_tile_units(px, resolution, tile_size):  # lib.rs:208-222
    return (np.asarray(px, dtype=F) + F(0.5)) / F(resolution) * F(tile_size)


def offset_modulus(value, modulus):  # lib.rs:239-243
    value, modulus = np.asarray(value, dtype=F), F(modulus)
    return -(np.ceil(value / modulus) * modulus - value) / modulus + F(1.0)


def tile_units_to_percentage(units, tile_size):  # lib.rs:224-237
    p = offset_modulus(units, tile_size)
    if not ((p >= 0.0) & (p <= 1.0)).all():
        raise EnvError("percentage outside [0, 1]")
    return p


def image_to_tile_coords(px, resolution):  # lib.rs:245-257
    return np.floor(np.asarray(px, dtype=F) / F(resolution)).astype(np.int64)


def tile_thresholds(path_width, expansion):
    """is_tile_obstacle's constants (lib.rs:346-351)."""
    pw = _percentage(_percentage(path_width) - _percentage(expansion))
    almost_full = _percentage(F(1.0) - pw)
    ow = _percentage(almost_full / F(2.0))
    owp = _percentage(ow + pw)
    lo = _percentage(F(0.5) - F(expansion) / F(2.0))
    hi = F(0.5) + F(expansion) / F(2.0)  # constructed only inside the arms that use it
    return ow, owp, lo, hi


def is_tile_obstacle(tile, path_width, px, py, expansion):  # lib.rs:338-479
    """px, py: broadcastable f32 percentages; returns a bool array."""
    ow, owp, lo, hi = tile_thresholds(path_width, expansion)
    xl, xh, yl, yh = px < ow, px > owp, py < ow, py > owp
    false = np.zeros(np.broadcast(px, py).shape, dtype=bool)
    if tile == "─":
        return (yl | yh) | false
    if tile == "│":
        return (xl | xh) | false
    if tile == "╴":
        return yl | yh | (px > lo)
    if tile == "╶":
        return yl | yh | (px < _percentage(hi))
    if tile == "╷":
        return xl | xh | (py < _percentage(hi))
    if tile == "╵":
        return xl | xh | (py > lo)
    if tile == "┌":
        return xl | yl | (xh & yh)
    if tile == "┐":
        return xh | yl | (xl & yh)
    if tile == "└":
        return xl | yh | (xh & yl)
    if tile == "┘":
        return xh | yh | (xl & yl)
    if tile == "┬":
        return yl | (yh & (xl | xh))
    if tile == "┴":
        return yh | (yl & (xl | xh))
    if tile == "├":
        return xl | (xh & (yl | yh))
    if tile == "┤":
        return xh | (xl & (yl | yh))
    if tile == "┼":
        return (xl | xh) & (yl | yh)
    if tile == " ":
        return ~false
    return false


# ---- placeable shapes (gbp_environment/src/lib.rs) --------------------------------------------
def _spf(v):  # StrictlyPositiveFinite::new(..).unwrap()
    v = float(v)
    if not (v > 0.0 and math.isfinite(v)):
        raise EnvError(f"{v} is not strictly positive and finite")
    return v


def triangle_points(angle_a, angle_b, radius):  # Triangle::points (lib.rs:186-204)
    a, b = F(angle_a), F(angle_b)
    c = PI32 - (a + b)
    r = F(radius)
    hyp = [r / sinf(a), r / sinf(b), r / sinf(c)]
    ang = [PI32 + a / F(2.0), -b / F(2.0), PI32 - b - c / F(2.0)]
    return [(cosf(t) * h, sinf(t) * h) for t, h in zip(ang, hyp)]  # Vec2::from_angle = (cos, sin)


def regular_polygon_points(sides, radius):  # RegularPolygon::point_at (lib.rs:263-279)
    pts = []
    for i in range(sides):
        angle = 2.0 * math.pi / float(sides) * float(i) + math.pi / 4
        pts.append((_libm.cos(angle) * radius, _libm.sin(angle) * radius))
    return pts


def polygon_expanded(points, expansion):  # Polygon::expanded (lib.rs:352-380)
    ax = ay = 0.0
    for x, y in points:
        ax, ay = ax + x, ay + y
    cx, cy = ax / float(len(points)), ay / float(len(points))
    return [(x + (x - cx) * 4.0 * expansion, y + (y - cy) * 4.0 * expansion) for x, y in points]


def rotation_offset(shape):  # lib.rs:299-312
    if shape["kind"] == "regular-polygon":
        extra = PI32 / F(shape["sides"]) if shape["sides"] % 2 != 0 else F(0.0)
        return FRAC_PI_2_32 + FRAC_PI_2_32 + extra
    if shape["kind"] == "polygon":
        return F(0.0)
    return FRAC_PI_2_32


def rotate_z(angle, x, y):
    """glam Quat::from_rotation_z(angle).mul_vec3((x, y, 0)).xy(): q = (0, 0, s, c),
    v' = v (w^2 - b.b) + b (2 v.b) + (w (b x v)) 2 with b = (0, 0, s)."""
    half = F(angle) * F(0.5)
    s, c = sinf(half), cosf(half)
    k = c * c - s * s
    zero = F(0.0)
    rx = (x * k + zero) + (c * (zero - y * s)) * F(2.0)
    ry = (y * k + zero) + (c * (s * x)) * F(2.0)
    return rx, ry


def shape_inside(shape, expansion32, x, y):
    """PlaceableShape::expanded(expansion as f64).inside(point) (lib.rs:506-530); x, y f32 arrays."""
    e = float(F(expansion32))
    kind = shape["kind"]
    if kind == "circle":  # lib.rs:127-143
        r = _spf(_spf(shape["radius"]) + e)
        return x * x + y * y <= F(r * r)
    if kind == "triangle":  # lib.rs:165-223
        r = _spf(_spf(shape["radius"]) + e)
        (ax, ay), (bx, by), (cx, cy) = triangle_points(shape["angles"][0], shape["angles"][1], r)

        def sign(p2x, p2y, p3x, p3y):
            return (x - p3x) * (p2y - p3y) - (p2x - p3x) * (y - p3y)
        d1, d2, d3 = sign(ax, ay, bx, by), sign(bx, by, cx, cy), sign(cx, cy, ax, ay)
        has_neg = (d1 < 0.0) | (d2 < 0.0) | (d3 < 0.0)
        has_pos = (d1 > 0.0) | (d2 > 0.0) | (d3 > 0.0)
        return ~(has_neg & has_pos)
    if kind == "regular-polygon":  # lib.rs:244-300
        r = _spf(_spf(shape["radius"]) + e * 2.0)
        n = int(shape["sides"])
        pts = regular_polygon_points(n, r)
        X, Y = x.astype(np.float64) * 2.0, y.astype(np.float64) * 2.0
        inside = np.zeros(X.shape, dtype=bool)
        j = n - 1
        with np.errstate(divide="ignore", invalid="ignore"):
            for i in range(n):
                (xi, yi), (xj, yj) = pts[i], pts[j]
                cond = ((yi < Y) & (yj >= Y)) | ((yj < Y) & (yi >= Y))
                hit = cond & (xi + (Y - yi) / (yj - yi) * (xj - xi) < X)
                inside ^= hit
                j = i
        return inside
    if kind == "rectangle":  # lib.rs:318-340
        w, h = _spf(_spf(shape["width"]) + e * 2.0), _spf(_spf(shape["height"]) + e * 2.0)
        X, Y = x.astype(np.float64), y.astype(np.float64)
        hw, hh = w / 4.0, h / 4.0
        return (X >= -hh) & (X <= hh) & (Y >= -hw) & (Y <= hw)
    if kind == "polygon":  # lib.rs:382-415
        pts = polygon_expanded([tuple(map(float, p)) for p in shape["points"]], e)
        X, Y = x.astype(np.float64), y.astype(np.float64)
        inside = np.zeros(X.shape, dtype=bool)
        j = len(pts) - 1
        with np.errstate(divide="ignore", invalid="ignore"):
            for i in range(len(pts)):
                (ix, iy), (jx, jy) = pts[i], pts[j]
                hit = ((iy > Y) != (jy > Y)) & (X < (jx - ix) * (Y - iy) / (jy - iy) + ix)
                inside ^= hit
                j = i
        return inside
    raise EnvError(f"unknown shape {kind!r}")


def env_to_image(env, resolution, expansion):
    """env_to_image (lib.rs:165-206) -> HxW u8 red plane (the reference writes R = G = B)."""
    grid = env["tiles"]["grid"]
    st = env["tiles"]["settings"]
    nrows, ncols = len(grid), len(grid[0])
    if resolution <= 0:
        raise EnvError("Pixels per tile must be non-zero")
    expansion = _percentage(expansion)
    ts = F(st["tile-size"])
    W, H = ncols * resolution, nrows * resolution
    xs, ys = np.arange(W), np.arange(H)
    tcx, tcy = image_to_tile_coords(xs, resolution), image_to_tile_coords(ys, resolution)
    px = tile_units_to_percentage(image_to_tile_units(xs, resolution, ts), ts)
    py = tile_units_to_percentage(image_to_tile_units(ys, resolution, ts), ts)
    out = np.empty((H, W), dtype=np.uint8)
    for row in range(nrows):
        for col in range(ncols):
            xm, ym = np.nonzero(tcx == col)[0], np.nonzero(tcy == row)[0]
            if not len(xm) or not len(ym):
                continue
            PX, PY = px[xm][None, :], py[ym][:, None]
            obst = is_tile_obstacle(grid[row][col], st["path-width"], PX, PY, expansion)
            for ob in env.get("obstacles") or []:  # is_placeable_obstacle (lib.rs:279-336)
                tc = ob["tile-coordinates"]
                if (tc["col"], tc["row"]) != (col, row):
                    continue
                tx, ty = F(ob["translation"]["x"]), F(ob["translation"]["y"])
                X, Y = np.broadcast_arrays(PX - tx, PY - ty)
                angle = F(ob["rotation"]) + rotation_offset(ob["shape"])
                rx, ry = rotate_z(angle, X, Y)
                obst = obst | shape_inside(ob["shape"], expansion, rx, ry)
            out[np.ix_(ym, xm)] = np.where(obst, 0, 255).astype(np.uint8)
    if (tcx >= ncols).any() or (tcy >= nrows).any():
        raise EnvError("Tile not found")
    return out


# ---- image 0.25.1 imageops::blur ------------------------------------------------------------------
def gaussian(x, r):  # image::imageops::sample::gaussian
    x, r = F(x), F(r)
    return F(1.0) / (np.sqrt(F(2.0) * PI32) * r) * expf(-(x * x) / (F(2.0) * (r * r)))


def blur_taps(n, sigma):
    """Per output index: (left, normalised f32 weights) of {vertical,horizontal}_sample with
    new size == old size (ratio = sratio = 1)."""
    sigma = F(sigma)
    support = F(2.0) * sigma
    taps = []
    for o in range(n):
        inp = (F(o) + F(0.5)) * F(1.0)
        left = int(np.floor(inp - support))
        left = min(max(left, 0), n - 1)
        right = int(np.ceil(inp + support))
        right = min(max(right, left + 1), n)
        inp = inp - F(0.5)
        ws, total = [], F(0.0)
        for i in range(left, right):
            wgt = gaussian((F(i) - inp) / F(1.0), sigma)
            ws.append(wgt)
            total = total + wgt
        taps.append((left, np.array([wgt / total for wgt in ws], dtype=F)))
    return taps


def blur(plane, sigma):
    """imageops::blur on one u8 channel (all three channels of the reference image are equal and
    go through identical arithmetic)."""
    sigma = F(sigma)
    if sigma <= 0.0:
        sigma = F(1.0)
    H, W = plane.shape
    src = plane.astype(F)
    tmp = np.empty((H, W), dtype=F)
    for o, (left, ws) in enumerate(blur_taps(H, sigma)):  # vertical_sample
        t = np.zeros(W, dtype=F)
        for i, wgt in enumerate(ws):
            t = t + src[left + i] * wgt
        tmp[o] = t
    out = np.empty((H, W), dtype=np.uint8)
    for o, (left, ws) in enumerate(blur_taps(W, sigma)):  # horizontal_sample
        t = np.zeros(H, dtype=F)
        for i, wgt in enumerate(ws):
            t = t + tmp[:, left + i] * wgt
        t = np.clip(t, F(0.0), F(255.0))
        out[:, o] = np.floor(t.astype(np.float64) + 0.5).astype(np.uint8)  # f32::round of a value in [0, 255], exact in f64
    return out


def env_to_sdf_image(env, resolution=None, expansion=None, blur_percent=None):
    """env_to_sdf_image (lib.rs:149-163) with the arguments simulation_loader.rs:154-162 passes."""
    sdf = env["tiles"]["settings"].get("sdf") or {"resolution": 200, "expansion": 0.1, "blur": 0.05}
    resolution = sdf["resolution"] if resolution is None else resolution
    expansion = sdf["expansion"] if expansion is None else expansion
    blur_percent = sdf["blur"] if blur_percent is None else blur_percent
    image = env_to_image(env, int(resolution), expansion)
    blur_pixels = _percentage(blur_percent) * F(int(resolution))
    if blur_pixels < 1.0:
        return image
    return blur(image, blur_pixels)
