"""GPU parity of the environment rasteriser: the device image (mgx_env_to_image /
mgx_env_to_sdf_image / mgx_world_set_environment through the C ABI) against the CPU restatement
(oracle/env.py), byte for byte."""
import json
import math
import os

import numpy as np
import pytest

import oracle
from magics_amd import MgxError, World, environment as ENV, scenarios as S
from oracle import env as E

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(env, resolution=None, expansion=None, blur=None):
    dev = ENV.env_to_sdf_image(env, resolution, expansion, blur)
    ref = E.env_to_sdf_image(env, resolution, expansion, blur)
    assert dev.shape == ref.shape + (3,)
    assert (dev[:, :, 0] == dev[:, :, 1]).all() and (dev[:, :, 0] == dev[:, :, 2]).all()
    bad = np.argwhere(dev[:, :, 0] != ref)
    assert not len(bad), f"{len(bad)} pixels differ, first at {bad[0]}: {dev[tuple(bad[0])][0]} vs {ref[tuple(bad[0])]}"
    return ref


@pytest.mark.parametrize("name", ["intersection", "intermediate", "complex_", "maze", "test", "circle"])
def test_builtin_environments(name):
    env = getattr(ENV, name)()
    for res, exp, blur in ((40, 0.0, 0.0), (64, 0.05, 0.05), (33, 0.1, 0.2)):
        if name == "maze" and res > 40:
            continue
        if name == "circle" and exp > 0.0:
            # Environment::circle() has path-width 0: `path_width - expansion` fails Percentage::new
            # in the reference (a panic), here an error on both sides
            with pytest.raises(MgxError):
                ENV.env_to_sdf_image(env, res, exp, blur)
            with pytest.raises(E.EnvError):
                E.env_to_sdf_image(env, res, exp, blur)
            exp = 0.0
        _same(env, res, exp, blur)


def test_reference_scenarios():
    with open(os.path.join(ROOT, "tests", "golden", "scenarios.json"), encoding="utf-8") as f:
        scenarios = json.load(f)
    seen = set()
    for name, sc in scenarios.items():
        key = json.dumps(sc["environment"], sort_keys=True)
        if key in seen:
            continue
        seen.add(key)
        img = _same(sc["environment"])
        assert img.size > 0, name
    assert len(seen) >= 6


def _random_env(rng):
    glyphs = "─│╴╶╷╵┌┐└┘┬┴├┤┼ █"
    rows, cols = rng.integers(1, 4), rng.integers(1, 4)
    grid = ["".join(glyphs[i] for i in rng.integers(0, len(glyphs), cols)) for _ in range(rows)]
    obstacles = []
    for _ in range(rng.integers(0, 8)):
        kind = ["circle", "triangle", "regular-polygon", "polygon", "rectangle"][rng.integers(0, 5)]
        if kind == "circle":
            shape = {"kind": kind, "radius": float(rng.uniform(0.01, 0.3))}
        elif kind == "triangle":
            a = float(rng.uniform(0.3, 1.6))
            shape = {"kind": kind, "angles": (a, float(rng.uniform(0.3, math.pi - a - 0.3))), "radius": float(rng.uniform(0.01, 0.1))}
        elif kind == "regular-polygon":
            shape = {"kind": kind, "sides": int(rng.integers(3, 9)), "radius": float(rng.uniform(0.02, 0.4))}
        elif kind == "polygon":
            shape = {"kind": kind, "points": [(float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.3, 0.3))) for _ in range(rng.integers(3, 7))]}
        else:
            shape = {"kind": kind, "width": float(rng.uniform(0.05, 0.8)), "height": float(rng.uniform(0.05, 0.8))}
        obstacles.append({"shape": shape, "rotation": float(rng.uniform(0, 2 * math.pi)),
                          "translation": {"x": float(rng.uniform(0, 1)), "y": float(rng.uniform(0, 1))},
                          "tile-coordinates": {"row": int(rng.integers(0, rows)), "col": int(rng.integers(0, cols))}})
    return ENV.new(grid, float(rng.uniform(0.2, 0.8)), 1.0, float(rng.choice([1.0, 10.0, 37.5, 100.0])), obstacles=obstacles)


def test_fuzz_environments():
    rng = np.random.default_rng(805)
    for _ in range(40):
        env = _random_env(rng)
        _same(env, int(rng.choice([17, 50, 64, 101])), float(rng.uniform(0.0, 0.15)), float(rng.choice([0.0, 0.01, 0.05, 0.1, 0.3])))


def test_pixels_on_shape_borders():
    # lattice-aligned shapes put many pixel centres exactly on an edge: the comparisons must agree there too
    for res in (10, 20, 40, 100):
        env = ENV.new(["█"], 0.5, 1.0, 1.0, obstacles=[
            {"shape": {"kind": "rectangle", "width": 0.4, "height": 0.8}, "rotation": 0.0, "translation": {"x": 0.5, "y": 0.5},
             "tile-coordinates": {"row": 0, "col": 0}},
            {"shape": {"kind": "circle", "radius": 0.25}, "rotation": 0.0, "translation": {"x": 0.25, "y": 0.25},
             "tile-coordinates": {"row": 0, "col": 0}},
            {"shape": {"kind": "polygon", "points": [(-0.25, -0.25), (0.25, -0.25), (0.25, 0.25), (-0.25, 0.25)]}, "rotation": 0.0,
             "translation": {"x": 0.75, "y": 0.75}, "tile-coordinates": {"row": 0, "col": 0}}])
        _same(env, res, 0.0, 0.0)
        _same(ENV.new(["┼─", "│┘"], 0.5, 1.0, 8.0), res, 0.25, 0.0)


def test_blur_widths():
    env = ENV.new(["┼"], 0.3, 1.0, 100.0)
    for res, blur in ((100, 0.009), (100, 0.01), (100, 0.025), (64, 0.2), (30, 1.0)):
        _same(env, res, 0.02, blur)


def test_errors_match_the_reference_panics():
    env = ENV.new(["┼"], 0.1, 1.0, 100.0)
    with pytest.raises(MgxError):
        ENV.env_to_image(env, 50, 0.2)     # path-width - expansion < 0 (Percentage::new)
    with pytest.raises(E.EnvError):
        E.env_to_image(env, 50, 0.2)
    with pytest.raises(MgxError):
        ENV.env_to_image(env, 0, 0.0)      # PixelsPerTile::new(0)
    bad = ENV.new(["█"], 0.1, 1.0, 100.0, obstacles=[{"shape": {"kind": "circle", "radius": -1.0}, "rotation": 0.0,
                                                     "translation": {"x": 0.5, "y": 0.5}, "tile-coordinates": {"row": 0, "col": 0}}])
    with pytest.raises(MgxError):
        ENV.env_to_image(bad, 50, 0.0)     # StrictlyPositiveFinite


def test_world_set_environment_feeds_the_obstacle_factors():
    """Same robots, image installed through mgx_world_set_environment on one side and through the
    CPU rasteriser on the other: beliefs stay bit-identical, and differ from a world without obstacles."""
    sc = S.grid_scenario(16, 10, interrobot=False)
    env = ENV.new(["┼"], 0.3, 1.0, float(sc["sdf"]["world_w"]), sdf={"resolution": 120, "expansion": 0.02, "blur": 0.03})
    eng, ref, bare = World(sc["params"]), oracle.OracleWorld(sc["params"]), oracle.OracleWorld(sc["params"])
    for w in (eng, ref):
        w.set_environment(env)
    for w in (eng, ref, bare):
        for rb in sc["robots"]:
            w.add_robot(rb["mean0"], rb["prior_diag"], rb["dt"], rb["radius"], order_key=rb["order_key"])
        w.iterate([1] * 12)
    for a, b in zip(eng.read_beliefs(), ref.read_beliefs()):
        assert np.array_equal(a, b)
    assert not np.array_equal(ref.read_beliefs()[2], bare.read_beliefs()[2])


def test_junction_scenario_on_device_rasterised_tiles():
    """BASELINE configs[4] in small: crossroads rasterised on the device, robots on the lanes with
    tracking paths; beliefs after whole ticks equal the oracle's, whose image comes from the CPU rasteriser."""
    sc = S.junction_scenario(40, 12, tiles=2)
    eng, ref = World(sc["params"]), oracle.OracleWorld(sc["params"])
    S.populate(eng, sc)
    S.populate(ref, sc)
    tick = S.tick_inputs(sc)
    for _ in range(4):
        for w in (eng, ref):
            w.update_priors(**tick)
            w.iterate(sc["steps"])
    for a, b in zip(eng.read_beliefs(), ref.read_beliefs()):
        assert np.isfinite(a).all() and np.array_equal(a, b)


def test_full_size_environment_through_properties():
    """BASELINE configs[4] size (20 x 20 tiles, 4000 x 4000 pixels) is beyond what the numpy oracle
    finishes in seconds; the image is checked through what the domain guarantees instead: every tile
    is the same crossroads, so away from the image border the picture is periodic with the tile, and
    its interior equals the oracle's rasterisation of a 3 x 3 neighbourhood's centre tile."""
    env = S.junction_environment(20)
    img = ENV.env_to_sdf_image(env)[:, :, 0]
    assert img.shape == (4000, 4000)
    res = 200
    centre = E.env_to_sdf_image(S.junction_environment(3))[res:2 * res, res:2 * res]
    tiles = img.reshape(20, res, 20, res).transpose(0, 2, 1, 3)
    assert (tiles[1:-1, 1:-1] == centre).all()            # 18 x 18 interior tiles, byte for byte
    edge = E.env_to_sdf_image(S.junction_environment(3))
    assert (tiles[0, 0] == edge[:res, :res]).all() and (tiles[-1, -1] == edge[-res:, -res:]).all()
    assert (tiles[0, 5] == edge[:res, res:2 * res]).all() and (tiles[7, -1] == edge[res:2 * res, -res:]).all()
    # the raw (unblurred) image: strictly black / white, and symmetric under the crossroads' symmetries
    raw = ENV.env_to_image(env, res, 0.01)[:, :, 0]
    assert set(np.unique(raw)) == {0, 255}
    assert (raw == raw[::-1, ::-1]).all() and (raw == raw.T).all()
