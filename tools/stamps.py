"""Diagnostic: builds libmgx_stamps.so (-DMGX_STAMPS, never shipped) and prints where the sweep
kernel's waves spend their cycles per internal iteration (factor phase, barrier, variable phase,
barrier).  Run on the GPU box:  python tools/stamps.py [--rpb-note]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magics_amd import hostlib, scenarios as S  # noqa: E402
from magics_amd.world import World  # noqa: E402

out = os.path.join(ROOT, "gpurun_out", "libmgx_stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
import __graft_entry__ as _ge  # noqa: E402
if os.environ.get("MGX_STAMPS_LIB"):  # a prebuilt -DMGX_STAMPS library (experiments)
    out = os.environ["MGX_STAMPS_LIB"]
else:
    _ge.build_library(out, ["-ffp-contract=off"], extra_defines=["-DMGX_STAMPS"] + os.environ.get("MGX_STAMPS_DEFINES", "").split())
hostlib.LIB_PATH = out
hostlib._libs.clear()
n_iter = 10
N = int(os.environ.get("MGX_STAMPS_N", "1000"))  # robots
for name, kw in (("config2", dict(interrobot=False)), ("config3", dict(interrobot=True))):
    sc = S.grid_scenario(N, 16, **kw)
    w = World(sc["params"])
    S.populate(w, sc)
    steps = [1] * n_iter if name == "config2" else sc["steps"]
    for _ in range(3):
        w.iterate(steps)
    w.synchronize()
    L = hostlib.lib()
    L.mgx_debug_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_uint32]
    buf = (C.c_ulonglong * ((N + 4) * 48))()
    n = L.mgx_debug_read_stamps(w._w, buf, len(buf))
    allw = np.array(buf[:n], dtype=np.uint64)
    raw = allw[:(N + 4) * 16].reshape(-1, 2, 8)
    sub = allw[(N + 4) * 16:(N + 4) * 48].reshape(-1, 2, 16).astype(np.float64)
    if name == "config3" and os.environ.get("MGX_PERSISTENT", "1") != "0":  # resident launch: cycles per stage over the 10 iterations
        a = raw.astype(np.float64)
        a = a[a[:, 0, 7] > 0]
        for role, rn in ((0, "DYN"), (1, "UV ")):
            wt, ef, ev, it_, pb, rt, stg, whole = (a[:, role, k].mean() for k in range(8))
            print(f"config3 resident {rn}: per launch (10 iterations) cycles: wait {wt:.0f}  ext factor {ef:.0f}  ext variable {ev:.0f}  internal {it_:.0f}  "
                  f"finish+publish {pb:.0f}  staging {stg:.0f}  whole kernel {whole:.0f}  (wait max {a[:, role, 0].max():.0f} min {a[:, role, 0].min():.0f})")
            names = ["poll", "poll barrier", "ext factor edges", "its barrier", "ext var sums", "barrier", "ext finish | adopt", "barrier", "response means",
                     "(internal)", "int finish", "publish stores", "drain", "early factor sweep", "records there (from sweep start)", "-"]
            m = sub[:N, role, :].mean(axis=0) / 10.0
            print("   per iteration: " + "  ".join(f"{nm} {v:.0f}" for nm, v in zip(names, m) if nm != "-"))
        continue
    ext_f, ext_v = (raw[:, :, 0] >> np.uint64(32)).astype(np.float64), (raw[:, :, 1] >> np.uint64(32)).astype(np.float64)
    raw[:, :, 0] &= np.uint64(0xffffffff)
    raw[:, :, 1] &= np.uint64(0xffffffff)
    a = raw.astype(np.float64)
    print(f"{name}: external factor sweep {ext_f[ext_f > 0].mean() if (ext_f > 0).any() else 0:.0f} cycles, external variable sweep {ext_v[ext_v > 0].mean() if (ext_v > 0).any() else 0:.0f} cycles")
    a = a[a[:, 0, 7] > 0]
    it = n_iter if name == "config2" else 1
    for role, rn in ((0, "DYN"), (1, "UV ")):
        f, fb, v, vb, tot = (a[:, role, k].mean() / it for k in range(5))
        clk = a[:, role, 4].mean() / (a[:, role, 5].mean() / 100e6) / 1e9
        print(f"{name} {rn}: staging {a[:, role, 6].mean():.0f} cycles, whole kernel {a[:, role, 7].mean():.0f} cycles, loop total {a[:, role, 4].mean():.0f}")
        print(f"{name} {rn}: clock {clk:.2f} GHz; per iteration cycles  factor {f:8.0f}  barrier {fb:8.0f}  variable {v:8.0f}  barrier {vb:8.0f}  loop {tot:8.0f}   ({len(a)} workgroups)")
