"""include/mgx.hpp — the C++ mirror of the reference's FactorGraph API above the C ABI.  The client
(tests/cpp_api/cpp_client.cpp) is compiled with g++, linked against libmgx.so and run: host-only
checks everywhere; on a GPU it replays a scenario handed over as a binary file and its beliefs and
message counts must equal the oracle's bit for bit."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle
from magics_amd import scenarios as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_client(tmp_path):
    exe = str(tmp_path / "cpp_client")
    libdir = os.path.join(ROOT, "magics_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp_api", "cpp_client.cpp"), "-o", exe, "-L", libdir, "-lmgx", f"-Wl,-rpath,{libdir}"],
                   check=True)
    return exe


def test_cpp_client_host_side(tmp_path):
    exe = build_client(tmp_path)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode == 0 and b"host ok" in r.stdout, (r.returncode, r.stdout.decode())


@pytest.mark.gpu
def test_cpp_client_scenario_equals_oracle(tmp_path):
    exe = build_client(tmp_path)
    sc = S.grid_scenario(6, 10, interrobot=True, pitch=2.0, comm_radius=5.0)
    n, K, ticks = len(sc["robots"]), sc["K"], 3
    tk = S.tick_inputs(sc)
    p = sc["params"]
    rgb = np.ascontiguousarray(sc["sdf"]["rgb"], dtype=np.uint8)
    path = str(tmp_path / "scenario.bin")
    pairs = [(a, b) for a, b, _ in sc["ir"]]
    with open(path, "wb") as f:
        f.write(struct.pack("12d", p["sigma_dynamics"], p["sigma_interrobot"], p["sigma_obstacle"], p["sigma_tracking"],
                            p["safety_multiplier"], p.get("enable_mask", 7), n, K, len(pairs), rgb.shape[1], rgb.shape[0],
                            sc["sdf"]["world_w"]))
        f.write(rgb.tobytes())
        for r, rb in enumerate(sc["robots"]):
            f.write(np.ascontiguousarray(rb["mean0"], dtype=np.float64).tobytes())
            f.write(np.ascontiguousarray(rb["prior_diag"], dtype=np.float64).tobytes())
            f.write(np.ascontiguousarray(rb["dt"], dtype=np.float64).tobytes())
            f.write(struct.pack("4d", rb["radius"], tk["waypoints_xy"][r][0], tk["waypoints_xy"][r][1], tk["time_scale"][r]))
        f.write(np.array(pairs, dtype=np.float64).tobytes())
        f.write(struct.pack("3d", tk["max_speed"], tk["delta_t"], ticks))
    assert sc["sdf"]["world_w"] == sc["sdf"]["world_h"]
    # the connections must come in the order the generator numbers them (a asc, b asc)
    assert [n0 for _, _, n0 in sc["ir"]] == [1 + (K - 1) * i for i in range(len(pairs))]
    r = subprocess.run([exe, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr.decode()[-2000:])
    lines = r.stdout.decode().split("\n")
    assert lines[0] == "host ok"
    got = np.array([[float.fromhex(x) for x in ln.split()] for ln in lines[1:1 + n * K]])
    ref = oracle.OracleWorld(sc["params"])
    S.populate(ref, sc)
    for _ in range(ticks):
        ref.update_priors(**tk)
        ref.iterate(sc["steps"])
    ref.set_antenna(0, False)
    ref.iterate(sc["steps"])
    assert np.array_equal(got, ref.read_means())
    counts = tuple(int(x) for x in lines[1 + n * K].split()[1:])
    assert counts == ref.message_counts(1)
