"""CPU checks of the engine's per-lane arithmetic (magics_amd/csrc/gbp_math.h, the functions the
gfx950 kernels call) through a test-only g++ build: against numpy dense algebra and against the
oracle's generic (reference-shaped) factor update.  The kernels themselves are checked on the
GPU in test_gpu_parity.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle
from magics_amd import scenarios as S

HERE = os.path.dirname(os.path.abspath(__file__))
dp = oracle.binding._dp


@pytest.fixture(scope="module")
def H():
    src = os.path.join(HERE, "cpu_math", "math_harness.cpp")
    out = os.path.join(HERE, "cpu_math", "libmath_harness.so")
    hdr = os.path.join(HERE, "..", "magics_amd", "csrc", "gbp_math.h")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", out, src], check=True)
    return C.CDLL(out)


def _dyn_full(dt, sigma):
    I2, Z2 = np.eye(2), np.zeros((2, 2))
    qc = 1.0 / sigma ** 2
    Q = np.block([[12 / dt ** 3 * qc * I2, -6 / dt ** 2 * qc * I2], [-6 / dt ** 2 * qc * I2, 4 / dt * qc * I2]])
    J = np.block([[I2, dt * I2, -I2, Z2], [Z2, I2, Z2, -I2]])
    return J.T @ Q @ J


def _schur(lam, eta, a):
    b = slice(4, 8) if a.start == 0 else slice(0, 4)
    w = np.linalg.inv(lam[b, b])
    return eta[a] - lam[a, b] @ w @ eta[b], lam[a, a] - lam[a, b] @ w @ lam[b, a]


def test_dynamic_message_vs_dense(H):
    rng = np.random.default_rng(0)
    for trial in range(50):
        dt, sigma = rng.uniform(0.1, 0.5), rng.uniform(0.05, 1.0)
        full = _dyn_full(dt, sigma)
        M = np.ascontiguousarray(full[::2, ::2])
        # structure claim: lam_p = M (x) I2
        np.testing.assert_allclose(np.kron(M, np.eye(2)), full, rtol=0, atol=0)
        a = rng.normal(size=(4, 4))
        lo = a @ a.T * rng.uniform(0.1, 100)
        eo = rng.normal(size=4) * 10
        for slot in (0, 1):
            lam, eta = full.copy(), np.zeros(8)
            o = slice(4, 8) if slot == 0 else slice(0, 4)
            lam[o, o] += lo
            eta[o] += eo
            we, wl = _schur(lam, eta, slice(4 * slot, 4 * slot + 4))
            oe, ol = np.zeros(4), np.zeros((4, 4))
            assert H.h_dynamic_message(dp(M), slot, dp(eo), dp(lo), dp(oe), dp(ol)) == 1
            scale = np.abs(full).max()
            np.testing.assert_allclose(ol, wl, rtol=1e-9, atol=1e-9 * scale)
            np.testing.assert_allclose(oe, we, rtol=1e-9, atol=1e-9 * scale)


def test_four_lane_dynamic_message_equals_single_lane_bitwise(H):
    # the DYN waves split one message over four lanes; every element must be the single-lane value
    rng = np.random.default_rng(7)
    for trial in range(200):
        dt, sigma = rng.uniform(0.1, 0.5), rng.uniform(0.05, 1.0)
        M = np.ascontiguousarray(_dyn_full(dt, sigma)[::2, ::2])
        a = rng.normal(size=(4, 4))
        lo = a @ a.T * 10 ** rng.uniform(-3, 6) if trial % 5 else np.zeros((4, 4))
        eo = rng.normal(size=4) * 10 ** rng.uniform(-2, 4)
        for slot in (0, 1):
            e1, l1, e4, l4 = np.zeros(4), np.zeros((4, 4)), np.zeros(4), np.zeros((4, 4))
            ok1 = H.h_dynamic_message(dp(M), slot, dp(eo), dp(lo), dp(e1), dp(l1))
            ok4 = H.h_dynamic_message_4lane(dp(M), slot, dp(eo), dp(lo), dp(e4), dp(l4))
            assert ok1 == ok4
            if ok1:
                assert np.array_equal(e1, e4) and np.array_equal(l1, l4)


def test_interrobot_message_vs_dense(H):
    rng = np.random.default_rng(1)
    H.h_interrobot_message.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for trial in range(100):
        dsafe, off, sig = 2.5, 1e-6 * rng.integers(1, 1000), 0.01
        xlo = rng.normal(size=4)
        xhi = xlo + np.array([*rng.normal(size=2) * 1.2, 0.3, 0.1])
        a = rng.normal(size=(4, 4))
        lo = a @ a.T * 50 + np.eye(4)
        eo = rng.normal(size=4) * 10
        d = xlo[:2] - xhi[:2]
        skip = (d @ d) >= dsafe ** 2
        dd = d + off
        r = np.sqrt(dd @ dd)
        J = np.zeros((4, 8))
        h = np.zeros(4)
        if r <= dsafe:
            J[0, 0:2] = -dd / (dsafe * r)
            J[0, 4:6] = dd / (dsafe * r)
            h[0] = 1 - r / dsafe
        x0 = np.concatenate([xlo, xhi])
        lam_p = J.T @ J / sig ** 2
        eta_p = J.T @ (J @ x0 + (0 - h)) / sig ** 2
        for dst in (0, 1):
            o = slice(4, 8) if dst == 0 else slice(0, 4)
            lam, eta = lam_p.copy(), eta_p.copy()
            lam[o, o] += lo
            eta[o] += eo
            oe, ol = np.zeros(4), np.zeros((4, 4))
            ok = H.h_interrobot_message(xlo.ctypes.data, xhi.ctypes.data, dsafe, off, 1 / sig ** 2, dst,
                                        eo.ctypes.data, lo.ctypes.data, oe.ctypes.data, ol.ctypes.data)
            if skip:
                assert ok == 0
                continue
            assert ok == 1
            we, wl = _schur(lam, eta, slice(4 * dst, 4 * dst + 4))
            sc = max(1.0, np.abs(wl).max())
            np.testing.assert_allclose(ol, wl, rtol=1e-8, atol=1e-8 * sc)
            np.testing.assert_allclose(oe, we, rtol=1e-8, atol=1e-8 * max(1.0, np.abs(we).max()))
    # other inbox entry empty => singular => empty message (det == 0 exactly)
    z4, z16 = np.zeros(4), np.zeros((4, 4))
    oe, ol = np.zeros(4), np.zeros((4, 4))
    xlo, xhi = np.array([0.0, 0, 1, 1]), np.array([1.0, 0.5, 0, 0])
    assert H.h_interrobot_message(xlo.ctypes.data, xhi.ctypes.data, 2.5, 1e-6, 1e4, 1, z4.ctypes.data,
                                  z16.ctypes.data, oe.ctypes.data, ol.ctypes.data) == 0


def test_obstacle_message_vs_oracle_world(H):
    # the oracle's generic first_order_jacobian + update on a one-robot world == the lane function
    rng = np.random.default_rng(2)
    sc = S.grid_scenario(1, 10, obstacles=True, seed=3)
    rgb = sc["sdf"]["rgb"]
    red = np.ascontiguousarray(rgb[:, :, 0])
    hgt, wid = red.shape
    ww, wh = sc["sdf"]["world_w"], sc["sdf"]["world_h"]
    L = oracle.lib()
    ow = oracle.OracleWorld(sc["params"])
    ow.set_sdf(rgb, ww, wh)
    delta = (ww / wid + wh / hgt) / 2
    sig = sc["params"]["sigma_obstacle"]
    H.h_obstacle.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_double, C.c_double, C.c_double, C.c_double,
                             C.c_void_p, C.c_void_p, C.c_void_p]
    nonzero = 0
    for _ in range(300):
        x0 = np.array([rng.uniform(-ww / 2 - 2, ww / 2 + 2), rng.uniform(-wh / 2 - 2, wh / 2 + 2), rng.normal(), rng.normal()])
        oe, ol = np.zeros(4), np.zeros((4, 4))
        H.h_obstacle(red.ctypes.data, wid, hgt, ww, wh, delta, 1 / sig ** 2, x0.ctypes.data, oe.ctypes.data, ol.ctypes.data)
        # reference-shaped evaluation through the oracle's measure
        f = lambda p: L.orc_obstacle_measure(ow._w, dp(np.array(p)))
        xx = list(x0)
        h0 = f(xx)
        J = np.zeros(4)
        for i in range(4):
            xx[i] += delta
            J[i] = (f(xx) - h0) / delta
            xx[i] -= delta
        lam = np.outer(J / sig ** 2, J)
        eta = (J / sig ** 2) * (J @ x0 + (0 - h0))
        np.testing.assert_allclose(ol, lam, rtol=1e-13, atol=0)
        np.testing.assert_allclose(oe, eta, rtol=1e-12, atol=1e-9)
        nonzero += bool(np.any(J))
    assert nonzero > 5  # the blurred discs were actually sampled


def test_belief_update_rules(H):
    rng = np.random.default_rng(4)
    a = rng.normal(size=(4, 4))
    lam = a @ a.T + np.eye(4)
    eta = rng.normal(size=4)
    mu, cov, valid = np.full(4, 7.0), np.full((4, 4), 7.0), C.c_int(0)
    H.h_belief(dp(eta), dp(lam), dp(mu), dp(cov), C.byref(valid))
    np.testing.assert_allclose(cov, np.linalg.inv(lam), rtol=1e-10)
    np.testing.assert_allclose(mu, np.linalg.solve(lam, eta), rtol=1e-10)
    assert valid.value == 1
    # "zero" precision (no element > 1e-6): nothing changes (variable.rs:276)
    mu2, cov2, v2 = np.full(4, 7.0), np.full((4, 4), 7.0), C.c_int(1)
    H.h_belief(dp(eta), dp(1e-7 * np.eye(4)), dp(mu2), dp(cov2), C.byref(v2))
    assert (mu2 == 7).all() and (cov2 == 7).all() and v2.value == 1
    # singular but not "zero": inv() is None, nothing changes (variable.rs:278)
    sing = np.zeros((4, 4)); sing[0, 0] = 5.0
    H.h_belief(dp(eta), dp(sing), dp(mu2), dp(cov2), C.byref(v2))
    assert (mu2 == 7).all() and (cov2 == 7).all()


def test_tracking_message_vs_oracle_world(H):
    # drive the oracle's tracking factor through a one-robot world and the lane function with the
    # same linearisation points; state (record, last measurement) must evolve identically
    sc = S.grid_scenario(1, 10, tracking=True, obstacles=False, seed=11)
    ow = oracle.OracleWorld(sc["params"])
    S.populate(ow, sc)
    path = np.ascontiguousarray(sc["robots"][0]["path"], dtype=np.float32)
    sig = sc["params"]["sigma_tracking"]
    H.h_tracking.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    K = 10
    rec = [C.c_int(0) for _ in range(K)]
    lp = [np.array(sc["robots"][0]["mean0"][i, :2], dtype=np.float32) for i in range(K)]
    lv = [C.c_double(0.0) for _ in range(K)]
    ow.iterate([1] * 10)
    for it in range(6):
        _, _, mu_before = ow.read_beliefs()
        ow.internal_factor_iteration(0)
        for i in range(1, K - 1):
            x0 = np.ascontiguousarray(mu_before[i])
            oe, ol = np.zeros(4), np.zeros((4, 4))
            ok = H.h_tracking(path.ctypes.data, len(path), sc["params"]["tracking_switch_padding"],
                              sc["params"]["tracking_attraction_distance"], 1 / sig ** 2, x0.ctypes.data,
                              C.addressof(rec[i]), lp[i].ctypes.data, C.addressof(lv[i]), oe.ctypes.data, ol.ctypes.data)
            assert ok == 1
            box = [b for b in ow.variable_inbox(0, i) if b[1] >= 10 + 9 + 8]
            assert box[0][2]
            np.testing.assert_allclose(ol, box[0][4], rtol=1e-12, atol=1e-9)
            np.testing.assert_allclose(oe, box[0][3], rtol=1e-12, atol=1e-6)
        ow.internal_variable_iteration(0)
