"""Diagnostics: where the hand-off of a resident schedule launch spends its time — LATENCY (a neighbour's records are stored, and
how long until this robot holds them) or SKEW (the neighbours do not publish at the same moment, and everybody waits for its last
one).  The -DMGX_STAMPS build (never shipped) stamps, per robot and segment, the 100 MHz wall clock at: publication begins /
stores issued / the next segment's gather begins / every record of that gather is there.
usage (GPU box): python tools/handoff_timeline.py [n_robots] [K]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magics_amd import hostlib, scenarios as S  # noqa: E402
from magics_amd.world import World  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

out = os.environ.get("MGX_STAMPS_LIB") or os.path.join(ROOT, "gpurun_out", "libmgx_stamps.so")
if not os.environ.get("MGX_STAMPS_LIB"):
    os.makedirs(os.path.dirname(out), exist_ok=True)
    ge.build_library(out, ["-ffp-contract=off"], extra_defines=["-DMGX_STAMPS"] + os.environ.get("MGX_STAMPS_DEFINES", "").split())
hostlib.LIB_PATH = out
hostlib._libs.clear()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sc = S.grid_scenario(n, K, interrobot=True)
w = World(sc["params"])
S.populate(w, sc)
for _ in range(5):
    w.iterate(sc["steps"])
w.synchronize()
L = hostlib.lib()
L.mgx_debug_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_uint32]
words = (n + 4) * 48 + n * 64
buf = (C.c_ulonglong * words)()
got = L.mgx_debug_read_stamps(w._w, buf, words)
assert got >= words, (got, words)
tl = np.array(buf[(n + 4) * 48:words], dtype=np.uint64).reshape(n, 16, 4).astype(np.int64)
nb = [[] for _ in range(n)]
for a, b, _ in sc["ir"]:  # factor owned by a, consumed at b: b gathers a's records
    nb[b].append(a)
nseg = len(sc["steps"])  # 10 steps = 10 segments; publications k = 0 .. nseg - 2
T = 10.0  # ns per tick
rows = []
for k in range(1, nseg - 1):
    pub0, pub1, g0, arr = (tl[:, k, c] for c in range(4))
    ok = (pub1 > 0) & (arr > 0)
    last_nb = np.array([max(pub1[a] for a in nb[b]) if nb[b] else pub1[b] for b in range(n)])
    first_nb = np.array([min(pub1[a] for a in nb[b]) if nb[b] else pub1[b] for b in range(n)])
    rows.append(dict(
        k=k,
        publish_ns=np.mean((pub1 - pub0)[ok]) * T,
        own_pub_to_gather_ns=np.mean((g0 - pub1)[ok]) * T,
        wait_in_gather_ns=np.mean((arr - g0)[ok]) * T,
        latency_after_last_neighbour_ns=np.mean((arr - last_nb)[ok]) * T,
        lat_p10=np.percentile((arr - last_nb)[ok], 10) * T, lat_p90=np.percentile((arr - last_nb)[ok], 90) * T,
        neighbour_spread_ns=np.mean((last_nb - first_nb)[ok]) * T,
        last_neighbour_after_own_pub_ns=np.mean((last_nb - pub1)[ok]) * T,
        chip_spread_ns=(np.percentile(pub1[ok], 95) - np.percentile(pub1[ok], 5)) * T,
        period_ns=np.mean((tl[:, k, 1] - tl[:, k - 1, 1])[ok]) * T))
keys = [k for k in rows[0] if k != "k"]
print(f"{n} robots x {K}: hand-off timeline of one resident launch, ns (100 MHz clock: +-10), mean over robots, per segment")
print("seg  " + "  ".join(f"{k[:-3] if k.endswith('_ns') else k:>32s}" for k in keys))
for r in rows:
    print(f"{r['k']:3d}  " + "  ".join(f"{r[k]:32.0f}" for k in keys))
print("mean " + "  ".join(f"{np.mean([r[k] for r in rows]):32.0f}" for k in keys))

# ---- who paces the lattice: the robots that never wait.  Per robot: mean wait inside the gather, against where it runs and what it computes
wait = np.array([np.mean([(tl[b, k, 3] - tl[b, k, 2]) for k in range(1, nseg - 1) if tl[b, k, 3] > 0]) * T for b in range(n)])
own = np.array([np.mean([(tl[b, k, 1] - tl[b, k - 1, 1]) for k in range(2, nseg - 1) if tl[b, k, 1] > 0]) * T for b in range(n)])
live = tl[:, 15, 0] + tl[:, 15, 1]
hw = tl[:, 15, 2]
xcc = (hw >> 32) & 0xf
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
cu_key = xcc * 1000 + se * 100 + sh * 10 + cu
per_cu = {k: int((cu_key == k).sum()) for k in np.unique(cu_key)}
load = np.array([per_cu[k] for k in cu_key])
deg = np.array([len(x) for x in nb])
print(f"\nper robot: wait inside the gather {wait.mean():.0f} ns mean, p10 {np.percentile(wait, 10):.0f}, p50 {np.percentile(wait, 50):.0f}, p90 {np.percentile(wait, 90):.0f}; "
      f"robots that wait < 300 ns: {(wait < 300).sum()}")
for name, v in (("workgroups on the robot's CU", load), ("incoming connections", deg), ("factors inside the safety distance (both waves, segment 5)", live)):
    print(f"  by {name}:")
    for x in np.unique(v):
        sel = v == x
        print(f"    {int(x):4d}: {int(sel.sum()):5d} robots, wait {wait[sel].mean():6.0f} ns (p10 {np.percentile(wait[sel], 10):6.0f})")
print("  CUs by workgroups held:", {c: list(per_cu.values()).count(c) for c in sorted(set(per_cu.values()))}, "XCDs:", {int(x): int((xcc == x).sum()) for x in np.unique(xcc)})
slow = np.argsort(wait)[:40]
print("  the 40 robots that wait least: connections", np.bincount(deg[slow]).tolist(), "live factors mean", live[slow].mean(), "vs all", live.mean(), "CU load", np.bincount(load[slow]).tolist())
