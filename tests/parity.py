"""Shared helpers of the GPU parity tests: drive the HIP engine (through the C ABI) and the CPU
oracle with the same script and compare beliefs."""
import numpy as np

import oracle
from magics_amd import World, scenarios as S

# north-star tolerance (BASELINE.json): 1e-5 relative on belief means / precisions
TOL = 1e-5


def make_pair(sc, fma=False):
    eng, ref = World(sc["params"], fma=fma), oracle.OracleWorld(sc["params"])
    ids_e = S.populate(eng, sc)
    ids_r = S.populate(ref, sc)
    assert ids_e == ids_r
    return eng, ref


def errors(eng, ref):
    """(mean error relative to the largest |mean|; worst per-variable precision error relative to
    that variable's largest |lam| entry; same for eta).  Variables whose reference precision is
    "zero" by the reference's own criterion (no element > 1e-6, variable.rs:276 — their lam is
    the rounding residue of a rank-deficient Schur complement) are excluded from the precision /
    eta figures: a relative error of rounding noise is meaningless."""
    eta_e, lam_e, mu_e = eng.read_beliefs()
    eta_r, lam_r, mu_r = ref.read_beliefs()
    assert np.isfinite(mu_e).all()
    e_mu = np.abs(mu_e - mu_r).max() / max(1.0, np.abs(mu_r).max())
    flat_r = lam_r.reshape(len(lam_r), -1)
    informed = (flat_r - 1e-6 > 0).any(axis=1)
    if not informed.any():
        return e_mu, 0.0, 0.0
    scale = np.abs(flat_r).max(axis=1)[informed]
    e_lam = (np.abs(lam_e - lam_r).reshape(len(lam_r), -1).max(axis=1)[informed] / scale).max()
    sc_eta = scale * np.maximum(np.abs(mu_r).max(axis=1)[informed], 1.0)
    e_eta = (np.abs(eta_e - eta_r).max(axis=1)[informed] / sc_eta).max()
    return e_mu, e_lam, e_eta


def max_abs_diff(eng, ref):
    return max(np.abs(a - b).max() for a, b in zip(eng.read_beliefs(), ref.read_beliefs()))


def assert_identical(eng, ref, what=""):
    """The product build keeps the reference's scalar operation order without FMA contraction:
    its beliefs must equal the oracle's bit for bit (NaNs in the same places)."""
    for name, a, b in zip(("eta", "lam", "mean"), eng.read_beliefs(), ref.read_beliefs()):
        same = np.array_equal(a, b, equal_nan=True)
        if not same:
            bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
            raise AssertionError(f"{what}: {name} differs in {bad.sum()} elements, max |diff| "
                                 f"{np.nanmax(np.abs(a - b)):.3e}")
    print(f"[parity {what}] bit-identical")


def assert_parity(eng, ref, tol=TOL, what=""):
    e_mu, e_lam, e_eta = errors(eng, ref)
    print(f"[parity {what}] mean {e_mu:.2e} precision {e_lam:.2e} eta {e_eta:.2e}")
    assert e_mu < tol and e_lam < tol and e_eta < tol, (what, e_mu, e_lam, e_eta)
    return e_mu, e_lam


def both(eng, ref, fn):
    fn(eng)
    fn(ref)


def assert_identical_where_finite(eng, ref, what="", max_nan_only_mismatch=1e-3):
    """Bit-identity for worlds the reference's own arithmetic drives out of the finite range.  BASELINE configs[4]
    (inter-robot AND tracking factors, Junction sigmas) does that on every synthetic input tried — the first
    inter-robot messages are Schur complements of rank-1 blocks (rounding residue instead of zero), a variable
    whose only other information is the residue of the dynamics factors inverts that noise into a mean thousands
    of metres off, and the tracking factor's un-normalised Jacobian (tracking.rs:171-194: (x - m) / h with h
    clamped to 1) squares the distance into its precision: 1e+100 and inf / NaN within one tick, in the oracle
    and in the engine alike.  There the two must agree bit for bit wherever the oracle is finite,
    be non-finite in the same places, and may differ only where the oracle holds a NaN and the engine — which
    never multiplies by the structural zeros of a Jacobian, DESIGN.md §10 — a number; that set must be tiny."""
    total = bad_finite = nan_only = nonfinite = 0
    for name, a, b in zip(("eta", "lam", "mean"), eng.read_beliefs(), ref.read_beliefs()):
        same = (a == b) | (np.isnan(a) & np.isnan(b))
        bad = ~same
        total += a.size
        nonfinite += int((~np.isfinite(b)).sum())
        nan_only += int((bad & np.isnan(b)).sum())
        bad_finite += int((bad & ~np.isnan(b)).sum())
        if (bad & ~np.isnan(b)).any():
            i = np.argwhere(bad & ~np.isnan(b))[0]
            raise AssertionError(f"{what}: {name} differs where the oracle holds a number, first at {tuple(i)}: "
                                 f"engine {a[tuple(i)]!r} oracle {b[tuple(i)]!r} ({bad_finite} such elements)")
    frac = nan_only / max(total, 1)
    print(f"[parity {what}] bit-identical wherever the oracle holds a number; oracle non-finite in {nonfinite} of {total} "
          f"entries ({100.0 * nonfinite / total:.2f} %), engine finite where the oracle is NaN in {nan_only} ({100.0 * frac:.4f} %)")
    assert frac <= max_nan_only_mismatch, (what, nan_only, total)
    return nonfinite / total
