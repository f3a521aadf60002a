"""CPU checks of the environment rasteriser's oracle (oracle/env.py): the reference's own unit
tests (crates/env_to_png/src/lib.rs:482-533) restated as data, and analytic checks of what the
reference leaves unpinned (tile glyphs, shape areas, blur)."""
import json
import math
import os

import numpy as np
import pytest

from oracle import env as E

F = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(grid, path_width=0.2, tile_size=10.0, obstacles=(), sdf=None):
    return {"tiles": {"grid": grid, "settings": {"tile-size": tile_size, "path-width": path_width, "obstacle-height": 1.0,
                                                 "sdf": sdf or {"resolution": 50, "expansion": 0.0, "blur": 0.0}}},
            "obstacles": list(obstacles)}


# ---- the reference's unit tests -----------------------------------------------------------------
# Two of the four (lib.rs:487-504) predate the code they sit next to: they expect pixel 23 of 100 on
# a 10-unit tile at 2.3 units and "0.3" of the tile, but the shipped image_to_tile_units samples the
# pixel CENTRE (+0.5, lib.rs:214) and offset_modulus(2.3, 10) is 0.23, so `cargo test` fails on
# them.  The shipped code is what the simulator runs, so it is the behaviour restated here; the
# stale expectations are kept as the half-pixel / factor they are off by.
def test_image_to_tile_units():  # lib.rs:487-495 (expects 2.3, 5.6: the pixel's left edge)
    assert E.image_to_tile_units(23, 100, 10.0) == F(23.5) / F(100) * F(10)
    assert E.image_to_tile_units(56, 100, 10.0) == F(56.5) / F(100) * F(10)
    assert abs(float(E.image_to_tile_units(23, 100, 10.0)) - 2.3 - 0.05) < 1e-6


def test_tile_units_to_percentage():  # lib.rs:497-504 (expects 0.3, 0.6)
    assert float(E.tile_units_to_percentage(F(2.3), 10.0)) == pytest.approx(0.23, abs=1e-6)
    assert float(E.tile_units_to_percentage(F(5.6), 10.0)) == pytest.approx(0.56, abs=1e-6)
    assert E.tile_units_to_percentage(F(10.0), 10.0) == F(1.0)   # a tile's far edge is 1, not 0
    assert float(E.tile_units_to_percentage(F(12.5), 10.0)) == pytest.approx(0.25, abs=1e-6)


def test_image_to_tile_coords():  # lib.rs:506-513
    assert E.image_to_tile_coords(134, 100) == 1
    assert E.image_to_tile_coords(240, 100) == 2


def test_is_obstacle():  # lib.rs:515-533
    assert not E.is_tile_obstacle("─", 0.5, F(0.3), F(0.6), 0.0)
    assert E.is_tile_obstacle("─", 0.5, F(0.1), F(0.6), 0.0) == (F(0.6) > F(0.75))  # the reference asserts `true` here ...
    # ... with percentage (0.1, 0.6): a horizontal corridor of width 0.5 spans y in [0.25, 0.75], so
    # 0.6 is inside it whatever x is.  The reference's second assertion can therefore not hold for
    # the code it ships (the test module is not run by its CI); the first one pins the predicate.


# ---- tile glyphs: which of the four edge midpoints and the centre are free ------------------------
ARMS = {"─": "EW", "│": "NS", "╴": "W", "╶": "E", "╷": "S", "╵": "N", "┌": "ES", "┐": "SW", "└": "NE", "┘": "NW",
        "┬": "ESW", "┴": "NEW", "├": "NES", "┤": "NSW", "┼": "NESW"}


@pytest.mark.parametrize("glyph", sorted(ARMS))
def test_glyph_openings(glyph):
    img = E.env_to_image(_env([glyph]), 50, 0.0)
    n = img.shape[0]
    free = {"N": img[1, n // 2], "S": img[n - 2, n // 2], "W": img[n // 2, 1], "E": img[n // 2, n - 2]}
    for side, value in free.items():
        assert (value == 255) == (side in ARMS[glyph]), (glyph, side)
    assert img[2, 2] == 0 and img[n - 3, n - 3] == 0  # corners are always obstacle


def test_blank_and_full_tiles():
    assert (E.env_to_image(_env([" "]), 20, 0.0) == 0).all()
    assert (E.env_to_image(_env(["█"]), 20, 0.0) == 255).all()


def test_expansion_narrows_the_corridor():
    a = E.env_to_image(_env(["┼"], path_width=0.3), 100, 0.0)
    b = E.env_to_image(_env(["┼"], path_width=0.3), 100, 0.1)
    assert (b <= a).all() and (a == 255).sum() > (b == 255).sum()
    assert (a[50] == 255).all() and (b[50] == 255).all()


def test_percentage_assertions():
    with pytest.raises(E.EnvError):
        E.env_to_image(_env(["┼"], path_width=0.1), 50, 0.2)  # path_width - expansion < 0: Percentage::new panics
    with pytest.raises(E.EnvError):
        E.env_to_image(_env(["┼"]), 50, 1.5)


# ---- shapes: rasterised area against the analytic area of what the code describes ---------------------
def _ob(shape, rotation=0.0, at=(0.5, 0.5)):
    return {"shape": shape, "rotation": rotation, "translation": {"x": at[0], "y": at[1]}, "tile-coordinates": {"row": 0, "col": 0}}


def _area(shape, rotation=0.0, res=400):
    img = E.env_to_image(_env(["█"], obstacles=[_ob(shape, rotation)]), res, 0.0)
    return (img == 0).mean()


@pytest.mark.parametrize("rotation", [0.0, 0.7, 3.0])
def test_shape_areas(rotation):
    r = 0.1
    assert _area({"kind": "circle", "radius": r}, rotation) == pytest.approx(math.pi * r * r, rel=0.02)
    # Rectangle::inside halves twice: extents are height / 4 and width / 4 either side of the centre
    assert _area({"kind": "rectangle", "width": 0.4, "height": 0.2}, rotation) == pytest.approx(0.4 * 0.2 / 4, rel=0.03)
    # RegularPolygon::inside doubles the test point: circumradius r / 2
    for n in (3, 4, 5, 8):
        want = n / 2 * (0.3 / 2) ** 2 * math.sin(2 * math.pi / n)
        assert _area({"kind": "regular-polygon", "sides": n, "radius": 0.3}, rotation) == pytest.approx(want, rel=0.03)
    a, b = math.radians(50), math.radians(70)
    # Triangle::points puts the vertices at radius / sin(angle) from the centre (not the incircle
    # construction the doc comment names): the area checked is that of those vertices
    tri = [(float(x), float(y)) for x, y in E.triangle_points(a, b, 0.05)]
    want = abs(sum(tri[i][0] * tri[(i + 1) % 3][1] - tri[(i + 1) % 3][0] * tri[i][1] for i in range(3))) / 2
    assert _area({"kind": "triangle", "angles": (a, b), "radius": 0.05}, rotation) == pytest.approx(want, rel=0.03)
    pts = [(-0.1, -0.1), (0.2, -0.05), (0.1, 0.15)]
    want = abs(sum(pts[i][0] * pts[(i + 1) % 3][1] - pts[(i + 1) % 3][0] * pts[i][1] for i in range(3))) / 2
    assert _area({"kind": "polygon", "points": pts}, rotation) == pytest.approx(want, rel=0.03)


def test_rotation_turns_the_shape():
    rect = {"kind": "rectangle", "width": 0.6, "height": 0.1}
    a = E.env_to_image(_env(["█"], obstacles=[_ob(rect, 0.0)]), 200, 0.0) == 0
    b = E.env_to_image(_env(["█"], obstacles=[_ob(rect, math.pi / 2)]), 200, 0.0) == 0
    assert np.ptp(np.nonzero(a.any(axis=0))[0]) != np.ptp(np.nonzero(a.any(axis=1))[0])
    assert abs(np.ptp(np.nonzero(a.any(axis=0))[0]) - np.ptp(np.nonzero(b.any(axis=1))[0])) <= 2  # a quarter turn swaps the extents


def test_obstacles_only_in_their_tile():
    img = E.env_to_image(_env(["██"], obstacles=[_ob({"kind": "circle", "radius": 0.2})]), 50, 0.0)
    assert (img[:, 50:] == 255).all() and (img[:, :50] == 0).any()


# ---- blur ---------------------------------------------------------------------------------------------
def test_blur_keeps_constants_and_is_symmetric():
    assert (E.blur(np.full((40, 30), 255, np.uint8), 2.0) == 255).all()
    assert (E.blur(np.zeros((40, 30), np.uint8), 3.5) == 0).all()
    step = np.zeros((64, 64), np.uint8)
    step[:, 32:] = 255
    out = E.blur(step, 2.0).astype(int)
    assert (np.diff(out, axis=1) >= 0).all() and (out == out[0]).all()
    assert (out[:, :32 - 5] == 0).all() and (out[:, 32 + 5:] == 255).all()  # support is 2 sigma + the pixel
    assert np.abs(out[0, :32][::-1] + out[0, 32:] - 255).max() <= 1


def test_blur_taps_are_normalised_windows():
    for n, sigma in ((50, 2.0), (7, 4.0), (200, 10.0)):
        for left, ws in E.blur_taps(n, sigma):
            assert 0 <= left and left + len(ws) <= n and len(ws) >= 1
            assert float(ws.sum(dtype=np.float64)) == pytest.approx(1.0, abs=1e-5)


def test_sdf_image_skips_blur_below_one_pixel():
    env = _env(["┼"], sdf={"resolution": 50, "expansion": 0.0, "blur": 0.01})
    assert np.array_equal(E.env_to_sdf_image(env), E.env_to_image(env, 50, 0.0))
    env["tiles"]["settings"]["sdf"]["blur"] = 0.04
    assert len(np.unique(E.env_to_sdf_image(env))) > 2


def test_reference_scenarios_rasterise():
    with open(os.path.join(ROOT, "tests", "golden", "scenarios.json"), encoding="utf-8") as f:
        scenarios = json.load(f)
    for name in ("Junction Twoway", "Environment Obstacles Experiment", "Circle Experiment"):
        env = scenarios[name]["environment"]
        img = E.env_to_sdf_image(env)
        res = env["tiles"]["settings"]["sdf"]["resolution"]
        assert img.shape == (len(env["tiles"]["grid"]) * res, len(env["tiles"]["grid"][0]) * res)
    junction = E.env_to_sdf_image(scenarios["Junction Twoway"]["environment"])
    assert junction[100, 100] == 255 and junction[5, 5] == 0 and junction[100, 5] == 255 and junction[5, 100] == 255
