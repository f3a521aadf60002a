"""`gbp_environment::Environment` (crates/gbp_environment/src/lib.rs:729-971) as plain data, its YAML
reader (`Environment::from_file / parse / validate`, lib.rs:770-826) and the device rasteriser
behind the C ABI (`mgx_env_to_image`, `mgx_env_to_sdf_image`: env_to_png::env_to_image /
env_to_sdf_image, crates/env_to_png/src/lib.rs:149-206).

An environment is a dict with the reference's kebab-case keys::

    {"tiles": {"grid": ["┼"], "settings": {"tile-size": 100.0, "path-width": 0.16,
                                             "obstacle-height": 2.0,
                                             "sdf": {"resolution": 200, "expansion": 0.01, "blur": 0.01}}},
     "obstacles": [{"shape": {"kind": "circle", "radius": 0.05}, "rotation": 0.0,
                    "translation": {"x": 0.5, "y": 0.5}, "tile-coordinates": {"row": 0, "col": 0}}]}

serde_yaml's enum tags (`shape: !regular-polygon {sides: 4, radius: 0.05}`) become
``{"kind": "regular-polygon", ...}``.
"""
import ctypes as C
import math

import numpy as np
import yaml

from . import hostlib
from .hostlib import EnvDesc, EnvObstacle

DEFAULT_SDF = {"resolution": 200, "expansion": 0.1, "blur": 0.05}  # SdfSettings::default (lib.rs:609-617)
_KINDS = {"circle": hostlib.SHAPE_CIRCLE, "triangle": hostlib.SHAPE_TRIANGLE, "regular-polygon": hostlib.SHAPE_REGULAR_POLYGON,
          "polygon": hostlib.SHAPE_POLYGON, "rectangle": hostlib.SHAPE_RECTANGLE}


class EnvironmentError(ValueError):
    """gbp_environment::EnvironmentError / ParseError (lib.rs:743-767)."""


class _Loader(yaml.SafeLoader):
    pass


def _tagged(loader, suffix, node):
    if isinstance(node, yaml.MappingNode):
        value = loader.construct_mapping(node, deep=True)
    elif isinstance(node, yaml.SequenceNode):
        value = loader.construct_sequence(node, deep=True)
    else:
        value = loader.construct_scalar(node)
    return {"__tag__": suffix, "value": value}


_Loader.add_multi_constructor("!", _tagged)


def load_yaml(text):
    """YAML with serde_yaml's `!variant` enum tags -> plain Python ({"__tag__", "value"} for tags)."""
    return yaml.load(text, Loader=_Loader)


def _shape(node):
    if not (isinstance(node, dict) and "__tag__" in node):
        raise EnvironmentError(f"YAML error: shape must be a tagged variant, got {node!r}")
    kind, v = node["__tag__"], node["value"]
    if kind not in _KINDS:
        raise EnvironmentError(f"YAML error: unknown variant `{kind}`")
    out = {"kind": kind}
    if kind == "circle":
        out["radius"] = float(v["radius"])
    elif kind == "triangle":
        out["angles"] = (float(v["angles"]["A"]), float(v["angles"]["B"]))
        out["radius"] = float(v["radius"])
    elif kind == "regular-polygon":
        out["sides"], out["radius"] = int(v["sides"]), float(v["radius"])
    elif kind == "rectangle":
        out["width"], out["height"] = float(v["width"]), float(v["height"])
    else:
        out["points"] = [(float(p["x"]), float(p["y"])) for p in v["points"]]
    return out


def parse(text):
    """Environment::parse (lib.rs:792-803) + validate (lib.rs:805-826)."""
    try:
        raw = load_yaml(text)
        st = raw["tiles"]["settings"]
        settings = {"tile-size": float(st["tile-size"]), "path-width": float(st["path-width"]),
                    "obstacle-height": float(st["obstacle-height"])}
        sdf = st.get("sdf")
        settings["sdf"] = dict(DEFAULT_SDF) if sdf is None else {"resolution": int(sdf["resolution"]),
                                                                   "expansion": float(sdf["expansion"]), "blur": float(sdf["blur"])}
        env = {"tiles": {"grid": [str(r) for r in raw["tiles"]["grid"]], "settings": settings}, "obstacles": []}
        for ob in raw.get("obstacles") or []:
            env["obstacles"].append({
                "shape": _shape(ob["shape"]), "rotation": float(ob["rotation"]),
                "translation": {"x": float(ob["translation"]["x"]), "y": float(ob["translation"]["y"])},
                "tile-coordinates": {"row": int(ob["tile-coordinates"]["row"]), "col": int(ob["tile-coordinates"]["col"])}})
    except (KeyError, TypeError, yaml.YAMLError) as exc:
        raise EnvironmentError(f"YAML error: {exc!r}") from exc
    return validate(env)


def from_file(path):
    with open(path, encoding="utf-8") as f:
        return parse(f.read())


def validate(env):
    grid = env["tiles"]["grid"]
    if not grid:
        raise EnvironmentError("Environment matrix representation is empty")
    if any(len(r) != len(grid[0]) for r in grid):
        raise EnvironmentError("Environment matrix representation has rows of different lengths")
    for ob in env.get("obstacles") or []:
        if not 0.0 <= ob["rotation"] <= 2.0 * math.pi:
            raise EnvironmentError(f"Angle value {ob['rotation']} is not inside [0,2π]")
        if not (0.0 <= ob["translation"]["x"] <= 1.0 and 0.0 <= ob["translation"]["y"] <= 1.0):
            raise EnvironmentError("Invalid relative point")
    return env


def new(grid, path_width, obstacle_height, tile_size, sdf=None, obstacles=()):
    """Environment::new (lib.rs:828-849)."""
    return validate({"tiles": {"grid": list(grid), "settings": {"tile-size": tile_size, "path-width": path_width,
                                                                "obstacle-height": obstacle_height, "sdf": dict(sdf or DEFAULT_SDF)}},
                     "obstacles": list(obstacles)})


def shape(env):
    """TileGrid::shape (lib.rs:60-63): (nrows, ncols)."""
    grid = env["tiles"]["grid"]
    return len(grid), len(grid[0])


def world_size(env):
    """WorldSize / WorldDimensions of an environment (robot.rs:1259-1264, spawner.rs:437-442)."""
    nrows, ncols = shape(env)
    ts = float(np.float32(env["tiles"]["settings"]["tile-size"]))
    return ts * ncols, ts * nrows


# ---- the reference's built-in environments (lib.rs:851-958) ---------------------------------------
def intersection():
    return new(["┼"], 0.1325, 1.0, 100.0)


def intermediate():
    return new(["┌┬┐ ", "┘└┼┬", "  └┘"], 0.1325, 1.0, 50.0)


def complex_():
    return new(["┌─┼─┬─┐┌", "┼─┘┌┼┬┼┘", "┴┬─┴┼┘│ ", "┌┴┐┌┼─┴┬", "├─┴┘└──┘"], 0.4, 1.0, 25.0)


def maze():
    return new(["               ", " ╶─┬─┐┌─────┬┐ ", " ┌─┤┌┤│╷╶──┬┘│ ", " │╷│╵├┤├─┬┬┴┬┤ ", " └┤├─┘││╷╵├─┘│ ", " ╷│╵╷╶┤│├┐└╴┌┘ ",
                " │├─┴╴│╵│└──┤╷ ", " └┤┌─┐└┬┘┌─┐└┘ ", " ┌┴┤╷├╴│┌┤╷└─┐ ", " │┌┤├┘┌┘││└──┤ ", " ╵│╵├┬┘┌┘└──┐╵ ", " ┌┘╶┘├─┴─┐╷╷└┐ ",
                " └─┬─┴──┐├┘├─┘ ", " ┌┐│╷┌─╴││╶┘╶┐ ", " │└┼┘├──┘├──┬┤ ", " ╵╶┴─┘╶──┴──┴┘ ", "               "], 0.75, 1.0, 10.0)


def test():
    return new(["┌┬┐├", "└┴┘┤", "│─ ┼", "╴╵╶╷"], 0.1325, 1.0, 50.0)


def circle():
    def ob(shape_, rotation, translation):
        return {"shape": shape_, "rotation": rotation, "translation": {"x": translation[0], "y": translation[1]},
                "tile-coordinates": {"row": 0, "col": 0}}

    def poly4(radius):
        return {"kind": "regular-polygon", "sides": 4, "radius": radius}
    rad = math.radians
    return new(["█"], 0.0, 1.0, 100.0, obstacles=[
        ob(poly4(0.0525), 0.0, (0.625, 0.60125)), ob(poly4(0.035), 0.0, (0.44125, 0.57125)), ob(poly4(0.0225), 0.0, (0.4835, 0.428)),
        ob({"kind": "rectangle", "width": 0.0875, "height": 0.035}, 0.0, (0.589, 0.3965)),
        ob({"kind": "triangle", "angles": (rad(30.0), rad(30.0)), "radius": 0.05}, 0.0, (0.5575, 0.5145)),
        ob({"kind": "triangle", "angles": (rad(110.0), rad(40.0)), "radius": 0.03}, 5.225, (0.38, 0.432))])


# ---- C ABI ---------------------------------------------------------------------------------------
class _Desc:
    """An mgx_env_desc together with the arrays it points into."""

    def __init__(self, env):
        validate(env)
        grid, st = env["tiles"]["grid"], env["tiles"]["settings"]
        sdf = st.get("sdf") or DEFAULT_SDF
        self.tiles = (C.c_uint32 * (len(grid) * len(grid[0])))(*[ord(ch) for row in grid for ch in row])
        obstacles = env.get("obstacles") or []
        self.obstacles = (EnvObstacle * max(1, len(obstacles)))()
        self._points = []
        for o, ob in zip(self.obstacles, obstacles):
            sh = ob["shape"]
            o.shape = _KINDS[sh["kind"]]
            o.tile_row, o.tile_col = ob["tile-coordinates"]["row"], ob["tile-coordinates"]["col"]
            o.rotation = ob["rotation"]
            o.translation_x, o.translation_y = ob["translation"]["x"], ob["translation"]["y"]
            o.radius = sh.get("radius", 0.0)
            o.sides = sh.get("sides", 0)
            if "angles" in sh:
                o.angle_a, o.angle_b = sh["angles"]
            o.width, o.height = sh.get("width", 0.0), sh.get("height", 0.0)
            if "points" in sh:
                pts = np.ascontiguousarray(sh["points"], dtype=np.float64).reshape(-1, 2)
                self._points.append(pts)
                o.n_points, o.points_xy = len(pts), pts.ctypes.data_as(hostlib.c_double_p)
        self.desc = EnvDesc(len(grid), len(grid[0]), self.tiles, st["tile-size"], st["path-width"], int(sdf["resolution"]),
                            sdf["expansion"], sdf["blur"], len(obstacles), self.obstacles)


def env_to_image(env, resolution, expansion):
    """env_to_png::env_to_image (lib.rs:165-206) on the device -> HxWx3 u8."""
    d = _Desc(env)
    nrows, ncols = shape(env)
    rgb = np.empty((nrows * int(resolution), ncols * int(resolution), 3), dtype=np.uint8)
    hostlib.check(hostlib.lib().mgx_env_to_image(C.byref(d.desc), int(resolution), float(expansion), rgb.ctypes.data))
    return rgb


def env_to_sdf_image(env, resolution=None, expansion=None, blur_percent=None):
    """env_to_png::env_to_sdf_image (lib.rs:149-163) on the device; the defaults are the
    environment's own sdf settings (simulation_loader.rs:154-162)."""
    d = _Desc(env)
    sdf = env["tiles"]["settings"].get("sdf") or DEFAULT_SDF
    resolution = int(sdf["resolution"] if resolution is None else resolution)
    expansion = sdf["expansion"] if expansion is None else expansion
    blur_percent = sdf["blur"] if blur_percent is None else blur_percent
    nrows, ncols = shape(env)
    rgb = np.empty((nrows * resolution, ncols * resolution, 3), dtype=np.uint8)
    hostlib.check(hostlib.lib().mgx_env_to_sdf_image(C.byref(d.desc), resolution, float(expansion), float(blur_percent), rgb.ctypes.data))
    return rgb
