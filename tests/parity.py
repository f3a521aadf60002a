"""Shared helpers of the GPU parity tests: drive the HIP engine (through the C ABI) and the CPU
oracle with the same script and compare beliefs."""
import numpy as np

import oracle
from magics_amd import World, scenarios as S

# north-star tolerance (BASELINE.json): 1e-5 relative on belief means / precisions
TOL = 1e-5


def make_pair(sc):
    eng, ref = World(sc["params"]), oracle.OracleWorld(sc["params"])
    ids_e = S.populate(eng, sc)
    ids_r = S.populate(ref, sc)
    assert ids_e == ids_r
    return eng, ref


def errors(eng, ref):
    """(mean error relative to the largest |mean|, worst per-variable precision error relative to
    that variable's largest |lam| entry, same for eta)."""
    eta_e, lam_e, mu_e = eng.read_beliefs()
    eta_r, lam_r, mu_r = ref.read_beliefs()
    assert np.isfinite(mu_e).all() and np.isfinite(lam_e).all()
    e_mu = np.abs(mu_e - mu_r).max() / max(1.0, np.abs(mu_r).max())
    scale = np.maximum(np.abs(lam_r).reshape(len(lam_r), -1).max(axis=1), 1e-300)
    e_lam = (np.abs(lam_e - lam_r).reshape(len(lam_r), -1).max(axis=1) / scale).max()
    # eta = lam mu is compared relative to |lam| |mu| of the variable
    sc_eta = np.maximum(scale * np.maximum(np.abs(mu_r).max(axis=1), 1.0), 1e-300)
    e_eta = (np.abs(eta_e - eta_r).max(axis=1) / sc_eta).max()
    return e_mu, e_lam, e_eta


def assert_parity(eng, ref, tol=TOL, what=""):
    e_mu, e_lam, e_eta = errors(eng, ref)
    print(f"[parity {what}] mean {e_mu:.2e} precision {e_lam:.2e} eta {e_eta:.2e}")
    assert e_mu < tol and e_lam < tol and e_eta < tol, (what, e_mu, e_lam, e_eta)
    return e_mu, e_lam


def both(eng, ref, fn):
    fn(eng)
    fn(ref)
