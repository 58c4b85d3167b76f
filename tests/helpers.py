"""Conversions between the oracle's Python integers/points and gnark memory images (uint64 limb arrays)."""
import numpy as np

import pyref
from pyref import BLS12_381, BN254, CURVES  # noqa: F401


def fr_to_arr(c, vals, mont=True):
    f = (lambda v: pyref.to_mont_limbs(v, c.r, 4)) if mont else (lambda v: pyref.to_limbs(v, 4))
    return np.array([f(v) for v in vals], dtype=np.uint64).reshape(-1, 4)


def arr_to_fr(c, arr, mont=True):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    if mont:
        return [pyref.from_mont_limbs(row, c.r) for row in arr]
    return [pyref.from_limbs(row) for row in arr]


def fp_limbs(c, v):
    return pyref.to_mont_limbs(v, c.p, c.fp_limbs)


def g1_to_arr(c, pts):
    rows = []
    for P in pts:
        if P is None:
            rows.append([0] * (2 * c.fp_limbs))
        else:
            rows.append(fp_limbs(c, P[0]) + fp_limbs(c, P[1]))
    return np.array(rows, dtype=np.uint64).reshape(-1, 2 * c.fp_limbs)


def g2_to_arr(c, pts):
    rows = []
    for P in pts:
        if P is None:
            rows.append([0] * (4 * c.fp_limbs))
        else:
            (x0, x1), (y0, y1) = P
            rows.append(fp_limbs(c, x0) + fp_limbs(c, x1) + fp_limbs(c, y0) + fp_limbs(c, y1))
    return np.array(rows, dtype=np.uint64).reshape(-1, 4 * c.fp_limbs)


def pts_to_arr(c, group, pts):
    return g1_to_arr(c, pts) if group == 0 else g2_to_arr(c, pts)


def _fp(c, limbs):
    return pyref.from_mont_limbs(limbs, c.p)


def arr_to_g1_affine(c, a):
    n = c.fp_limbs
    a = [int(x) for x in a]
    x, y = _fp(c, a[:n]), _fp(c, a[n:2 * n])
    return None if x == 0 and y == 0 else (x, y)


def arr_to_g2_affine(c, a):
    n = c.fp_limbs
    a = [int(x) for x in a]
    v = [_fp(c, a[i * n:(i + 1) * n]) for i in range(4)]
    return None if not any(v) else ((v[0], v[1]), (v[2], v[3]))


def jac_to_affine_py(c, group, a):
    """Jacobian image {X,Y,Z} -> python affine point (x = X/Z^2, y = Y/Z^3)."""
    n = c.fp_limbs
    a = [int(x) for x in a]
    if group == 0:
        X, Y, Z = (_fp(c, a[i * n:(i + 1) * n]) for i in range(3))
        if Z == 0:
            return None
        zi = pow(Z, -1, c.p)
        return (X * zi * zi % c.p, Y * zi * zi * zi % c.p)
    F = pyref.Fp2Ops(c.p)
    v = [_fp(c, a[i * n:(i + 1) * n]) for i in range(6)]
    X, Y, Z = (v[0], v[1]), (v[2], v[3]), (v[4], v[5])
    if Z == (0, 0):
        return None
    zi = F.inv(Z)
    zi2 = F.mul(zi, zi)
    return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))


def group_of(c, group):
    return pyref.g1_group(c) if group == 0 else pyref.g2_group(c)


def gen_of(c, group):
    return c.g1 if group == 0 else c.g2
