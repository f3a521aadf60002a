"""Diagnostics: the hand-off of resident schedule launches with a CHECKSUM in every exchange record (build flag
-DMGX_XREC_CHECKSUM, never shipped: the spare dwords of a record carry the xor of its payload and the producer's identity; a
consumer whose validated record does not add up reports through the world's error word).  This is what found the gfx950
store hazard behind an SGPR soffset (experiments/README.md).  Run on the GPU box: python tools/xrec_checksum.py [robots] [ticks]"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import __graft_entry__ as ge
from magics_amd import hostlib, scenarios as S
from magics_amd.world import World

out = os.path.join(ROOT, "gpurun_out", "libmgx_checksum.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
ge.build_library(out, ["-ffp-contract=off"], extra_defines=["-DMGX_XREC_CHECKSUM"])
hostlib.LIB_PATH = out
hostlib._libs.clear()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 50
K = 16
sc = S.grid_scenario(n, K, interrobot=True)
bad = 0
for trial in range(6):
    w = World(sc["params"]); S.populate(w, sc)
    try:
        for _ in range(ticks):
            w.iterate(sc["steps"])
        w.synchronize(); w.read_beliefs()
        print(f"trial {trial}: {ticks} ticks, {w.last_launch_count()} launch per tick, every record added up")
    except Exception as e:  # noqa
        bad += 1
        m = re.search(r"(\d{15,})", str(e))
        v = int(m.group(1)) if m else 0
        print(f"trial {trial}: CHECKSUM MISMATCH producer robot {((v >> 28) & 0xfffff) // 64} variable {((v >> 28) & 0xfffff) % 64}, "
              f"consumer wanted variable index {(v >> 8) & 0xfffff}, sequence & 255 = {v & 255}  ({str(e)[:80]})")
print("MISMATCHES" if bad else "OK", bad)
