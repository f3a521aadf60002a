"""TEST INFRASTRUCTURE ONLY — numpy restatement of crates/gbp_multivariate_normal/src/lib.rs:38-410
(the checker of magics_amd.mvn; nothing in the product imports it).  Pinned by the reference's own
unit tests (lib.rs:419-743), restated as data in tests/test_mvn.py.  `dot` is written as explicit
loops (k ascending) so that means can be compared bit for bit; det / inverse come from LAPACK and
are compared within a tolerance, except the exact "det == 0" contract."""
import numpy as np


class MvnError(Exception):
    def __init__(self, variant, *args):
        super().__init__(f"{variant}{args if args else ''}")
        self.variant, self.args_ = variant, args


def dot(m, v):
    n = len(v)
    out = np.zeros(n)
    for i in range(n):
        s = 0.0
        for k in range(n):
            s += m[i, k] * v[k]
        out[i] = s
    return out


def _exact_det_is_zero(m):
    """det() == 0.0 exactly: by cofactor expansion in Python floats for the small cases the tests use."""
    m = np.asarray(m, dtype=np.float64)
    n = m.shape[0]
    if n == 1:
        return m[0, 0] == 0.0
    if n == 2:
        return m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0] == 0.0
    if n <= 4:
        def det(a):
            k = a.shape[0]
            if k == 2:
                return a[0, 0] * a[1, 1] - a[0, 1] * a[1, 0]
            acc = 0.0
            for j in range(k):
                minor = np.delete(np.delete(a, 0, axis=0), j, axis=1)
                term = a[0, j] * (-det(minor) if j & 1 else det(minor))
                acc = term if j == 0 else acc + term
            return acc
        return det(m) == 0.0
    return np.linalg.matrix_rank(m) < n


class MultivariateNormal:
    def __init__(self, information, precision, mean):
        self.information, self.precision, self._mean, self.dirty = information, precision, mean, False

    @staticmethod
    def _check(v, m):
        if m.shape[0] != m.shape[1]:
            raise MvnError("NonSquarePrecisionMatrix", m.shape[0], m.shape[1])      # :67-71
        if len(v) != m.shape[0] or len(v) != m.shape[1]:
            raise MvnError("VectorLengthNotEqualMatrixShape", len(v), m.shape[0], m.shape[1])  # :72-79

    @classmethod
    def from_information_and_precision(cls, information, precision):
        v, m = np.array(information, dtype=np.float64), np.array(precision, dtype=np.float64)
        cls._check(v, m)
        if _exact_det_is_zero(m):
            raise MvnError("NonInvertiblePrecisionMatrix")                       # :82-84
        return cls(v, m, dot(m, v))                                              # :85

    @classmethod
    def from_mean_and_covariance(cls, mean, covariance):
        v, m = np.array(mean, dtype=np.float64), np.array(covariance, dtype=np.float64)
        cls._check(v, m)
        if _exact_det_is_zero(m):
            raise MvnError("NonInvertibleCovarianceMatrix")                      # :130-132
        p = np.linalg.inv(m)
        return cls(dot(p, v), p, v)                                              # :133

    def __len__(self):
        return len(self.information)

    def information_vector(self):
        return self.information

    def precision_matrix(self):
        return self.precision

    def mean(self):
        return self._mean

    def covariance(self):
        return np.linalg.inv(self.precision)

    def update(self):                                                            # :271-279
        if self.dirty:
            self._mean = dot(self.precision, self.information)
            self.dirty = False
            return True
        return False

    def update_information_vector(self, value):                                  # :158-161 (no dirty flag: mean stays)
        self.information = np.array(value, dtype=np.float64)
        self.update()

    def update_precision_matrix(self, value):                                    # :169-178
        value = np.array(value, dtype=np.float64)
        if _exact_det_is_zero(value):
            raise MvnError("NonInvertiblePrecisionMatrix")
        self.precision = value
        self.update()

    def set_information_vector(self, value):
        self.information = np.array(value, dtype=np.float64)
        self.dirty = True

    def set_precision_matrix(self, value):
        self.precision = np.array(value, dtype=np.float64)
        self.dirty = True

    def add_assign_information_vector(self, value):
        self.information = self.information + np.asarray(value, dtype=np.float64)
        self.dirty = True

    def add_assign_precision_matrix(self, value):
        self.precision = self.precision + np.asarray(value, dtype=np.float64)
        self.dirty = True

    def _combined(self, o, sign):
        i = self.information + sign * o.information
        p = self.precision + sign * o.precision
        return MultivariateNormal(i, p, dot(p, i))                               # :300-410

    def __add__(self, o):
        return self._combined(o, 1.0)

    def __sub__(self, o):
        return self._combined(o, -1.0)

    def __mul__(self, o):
        return self._combined(o, 1.0)

    def _assign(self, o, sign):
        r = self._combined(o, sign)
        self.information, self.precision, self._mean, self.dirty = r.information, r.precision, r._mean, False
        return self

    def __iadd__(self, o):
        return self._assign(o, 1.0)

    def __isub__(self, o):
        return self._assign(o, -1.0)

    def __imul__(self, o):
        return self._assign(o, 1.0)
