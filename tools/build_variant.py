"""Diagnostics: build the engine from ANOTHER state of the kernel sources into experiments/ab/libmgx_<name>.so, for A/B runs
inside one gpurun call (box-to-box differences are as large as most kernel changes: compare only within a call).
    python tools/build_variant.py NAME [GIT_REV] [-DFLAG ...]      # GIT_REV default: the working tree
then on the GPU box:  MGX_LIB=experiments/ab/libmgx_NAME.so python tools/quick_ir_bench.py 1000"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

name = sys.argv[1]
rev = next((a for a in sys.argv[2:] if not a.startswith("-")), None)
defs = [a for a in sys.argv[2:] if a.startswith("-")]
out = os.path.join(ROOT, "experiments", "ab", f"libmgx_{name}.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
if rev:
    tmp = tempfile.mkdtemp(prefix=f"mgx_{name}_")
    tar = subprocess.run(["git", "-C", ROOT, "archive", rev, "magics_amd/csrc", "include"], check=True, capture_output=True).stdout
    subprocess.run(["tar", "-x", "-C", tmp], input=tar, check=True)
    g.CSRC = os.path.join(tmp, "magics_amd", "csrc")
    units = [u for u in g.UNITS if os.path.exists(os.path.join(g.CSRC, u[0]))]
    if len(units) != len(g.UNITS):  # a revision from before the sweep kernel was split into several objects
        units = [(s, []) for s in ("mgx_kernels.hip", "mgx_topology.hip", "mgx_env.hip", "mgx_world.hip", "mgx_host.cpp", "mgx_linalg.cpp")]
        g.HEADERS = [h for h in g.HEADERS if os.path.exists(os.path.join(g.CSRC, h))]
    g.UNITS = units
    g.SOURCES = sorted({u[0] for u in units})
g.build_library(out, ["-ffp-contract=off"], extra_defines=defs)
print(out)
