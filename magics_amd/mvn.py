"""`gbp_linalg` / `gbp_multivariate_normal` mirrors over the C ABI (include/mgx.h): the vector norms
(crates/gbp_linalg/src/lib.rs:47-128) and `MultivariateNormal` in information form
(crates/gbp_multivariate_normal/src/lib.rs:38-410) with the reference's method names, operators
and error variants.  Host-only value types; no device involved."""
import ctypes as C

import numpy as np

from . import hostlib
from .hostlib import c_double_p

ERR_NON_SQUARE, ERR_LENGTH, ERR_SINGULAR_PRECISION, ERR_SINGULAR_COVARIANCE = -16, -17, -18, -19
ADD, SUB, MUL = 0, 1, 2


class MultivariateNormalError(Exception):
    """MultivariateNormalError (lib.rs:9-30); `variant` is the Rust variant's name."""

    def __init__(self, variant, text):
        super().__init__(text)
        self.variant = variant


_VARIANTS = {ERR_NON_SQUARE: "NonSquarePrecisionMatrix", ERR_LENGTH: "VectorLengthNotEqualMatrixShape",
             ERR_SINGULAR_PRECISION: "NonInvertiblePrecisionMatrix", ERR_SINGULAR_COVARIANCE: "NonInvertibleCovarianceMatrix"}


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _vec(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _chk(rc):
    if rc in _VARIANTS:
        raise MultivariateNormalError(_VARIANTS[rc], hostlib.lib().mgx_last_error().decode())
    return hostlib.check(rc)


def euclidean_norm(x):
    x = _vec(x)
    return hostlib.lib().mgx_euclidean_norm(_dp(x), x.size)


def l1_norm(x):
    x = _vec(x)
    return hostlib.lib().mgx_l1_norm(_dp(x), x.size)


def normalized(x):
    x = _vec(x).copy()
    hostlib.lib().mgx_normalize(_dp(x), x.size)
    return x


def det(a):
    a = _vec(a)
    assert a.ndim == 2 and a.shape[0] == a.shape[1]
    return hostlib.lib().mgx_det(_dp(a), a.shape[0])


def inv(a):
    """`Inverse::inv`: None when the determinant is exactly 0."""
    a = _vec(a)
    assert a.ndim == 2 and a.shape[0] == a.shape[1]
    out = np.zeros_like(a)
    return out if _chk(hostlib.lib().mgx_inverse(_dp(a), a.shape[0], _dp(out))) == 1 else None


class MultivariateNormal:
    def __init__(self, handle):
        self._L, self._h = hostlib.lib(), handle

    def __del__(self):
        try:
            if self._h:
                self._L.mgx_mvn_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @staticmethod
    def _shape(m):
        m = _vec(m)
        if m.ndim != 2:
            raise ValueError("matrix expected")
        return m, m.shape[0], m.shape[1]

    @classmethod
    def from_information_and_precision(cls, information, precision):
        v = _vec(information)
        m, r, c = cls._shape(precision)
        h = C.c_void_p()
        _chk(hostlib.lib().mgx_mvn_from_information_and_precision(_dp(v), v.size, _dp(m), r, c, C.byref(h)))
        return cls(h)

    @classmethod
    def from_mean_and_covariance(cls, mean, covariance):
        v = _vec(mean)
        m, r, c = cls._shape(covariance)
        h = C.c_void_p()
        _chk(hostlib.lib().mgx_mvn_from_mean_and_covariance(_dp(v), v.size, _dp(m), r, c, C.byref(h)))
        return cls(h)

    def __len__(self):
        return self._L.mgx_mvn_len(self._h)

    def _get(self):
        n = len(self)
        i, p, m = np.zeros(n), np.zeros((n, n)), np.zeros(n)
        _chk(self._L.mgx_mvn_get(self._h, _dp(i), _dp(p), _dp(m)))
        return i, p, m

    def information_vector(self):
        return self._get()[0]

    def precision_matrix(self):
        return self._get()[1]

    def mean(self):
        return self._get()[2]

    def covariance(self):
        n = len(self)
        out = np.zeros((n, n))
        _chk(self._L.mgx_mvn_covariance(self._h, _dp(out)))
        return out

    def update_information_vector(self, value):
        _chk(self._L.mgx_mvn_update_information_vector(self._h, _dp(_vec(value))))

    def update_precision_matrix(self, value):
        _chk(self._L.mgx_mvn_update_precision_matrix(self._h, _dp(_vec(value))))

    def set_information_vector(self, value):
        _chk(self._L.mgx_mvn_set_information_vector(self._h, _dp(_vec(value))))

    def set_precision_matrix(self, value):
        _chk(self._L.mgx_mvn_set_precision_matrix(self._h, _dp(_vec(value))))

    def add_assign_information_vector(self, value):
        _chk(self._L.mgx_mvn_add_assign_information_vector(self._h, _dp(_vec(value))))

    def add_assign_precision_matrix(self, value):
        _chk(self._L.mgx_mvn_add_assign_precision_matrix(self._h, _dp(_vec(value))))

    def update(self):
        return bool(_chk(self._L.mgx_mvn_update(self._h)))

    def _combine(self, other, op):
        h = C.c_void_p()
        _chk(self._L.mgx_mvn_combine(self._h, other._h, op, C.byref(h)))
        return MultivariateNormal(h)

    def _combine_assign(self, other, op):
        _chk(self._L.mgx_mvn_combine_assign(self._h, other._h, op))
        return self

    def __add__(self, o):
        return self._combine(o, ADD)

    def __sub__(self, o):
        return self._combine(o, SUB)

    def __mul__(self, o):
        return self._combine(o, MUL)

    def __iadd__(self, o):
        return self._combine_assign(o, ADD)

    def __isub__(self, o):
        return self._combine_assign(o, SUB)

    def __imul__(self, o):
        return self._combine_assign(o, MUL)
