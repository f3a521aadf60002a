"""Guards the checker against its own compiler: the optimised oracle build must agree bit for
bit with an -O0 build on a scenario that exercises every factor kind (GCC 11's SLP vectoriser
was caught dropping an f32 rounding in the tracking factor; see oracle/Makefile)."""
import os
import subprocess
import tempfile

import numpy as np

import oracle
from magics_amd import scenarios as S

HERE = os.path.dirname(os.path.abspath(__file__))


def test_optimised_oracle_equals_O0_build():
    src = os.path.join(HERE, "..", "oracle", "gbp_oracle.c")
    with tempfile.TemporaryDirectory() as d:
        o0 = os.path.join(d, "liborc_O0.so")
        subprocess.run(["gcc", "-O0", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-o", o0, src, "-lm"], check=True)
        sc = S.grid_scenario(9, 10, interrobot=True, tracking=True, seed=5, pitch=2.0, comm_radius=5.0)
        out = []
        for path in (None, o0):
            w = oracle.OracleWorld(sc["params"], lib_path=path)
            S.populate(w, sc)
            for _ in range(3):
                w.iterate(sc["steps"])
            out.append(w.read_beliefs())
        for a, b in zip(*out):
            assert np.array_equal(a, b)
