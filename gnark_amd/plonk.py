"""Mirror of the FFT-heavy core of gnark's PLONK prover (backend/plonk/bn254/prove.go): the quotient polynomial
(`computeNumerator` :841-1123 + `divideByZH` :1287-1350) and the grand-product polynomial (`iop.BuildRatioCopyConstraint`,
call site :645-655) computed on the GPU.  The round logic (Fiat-Shamir, KZG openings) stays on the host side; KZG commitments
are `ecc.MultiExp` over the SRS."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .device import _ptr, as_u64
from .fft import Domain

IDS = ("L", "R", "O", "Z", "Ql", "Qr", "Qm", "Qo", "Qk", "S1", "S2", "S3")   # prove.go:44-59 (ZS is Z shifted by one)


def Rho(n: int) -> int:
    """|domain1| / |domain0| (prove.go:247-251)"""
    return 8 if n < 6 else 4


FIXED_IDS = ("Ql", "Qr", "Qm", "Qo", "S1", "S2", "S3")     # circuit constants (with every Qcp<i>): pinned by ProvingKey
PROOF_IDS = ("L", "R", "O", "Z", "Qk")                      # per proof (with every Pi2<i>)


def _fill(a, names, polys, lagrange, n, keep):
    low = {"L": "l", "R": "r", "O": "o", "Z": "z", "Ql": "ql", "Qr": "qr", "Qm": "qm", "Qo": "qo", "Qk": "qk", "S1": "s1", "S2": "s2", "S3": "s3"}
    for k in names:
        arr = as_u64(np.asarray(polys[k]).reshape(-1, 4), 4)
        if arr.shape[0] != n:
            raise ValueError(f"every polynomial must have {n} coefficients")
        keep.append(arr)
        setattr(a, low[k], arr.ctypes.data)
    for nm in lagrange:
        if nm in IDS:
            a.lagrange_mask |= 1 << IDS.index(nm)
        else:   # "Qcp<i>" / "Pi2<i>"
            i = int(nm[3:])
            a.lagrange_mask |= 1 << (len(IDS) + 2 * i + (0 if nm.startswith("Qcp") else 1))


def _ptr_array(arrs, n, keep):
    out = [as_u64(np.asarray(x).reshape(-1, 4), 4) for x in arrs]
    for x in out:
        if x.shape[0] != n:
            raise ValueError(f"every polynomial must have {n} coefficients")
    keep.extend(out)
    pa = (C.c_void_p * len(out))(*[x.ctypes.data for x in out])
    keep.append(pa)
    return pa


class ProvingKey:
    """The circuit-constant half of the quotient pinned on the device (ga_plonk_pk_create): Ql, Qr, Qm, Qo, S1, S2, S3, Qcp_i
    evaluated on every coset once -- the precomputation prove.go:1030-1034 rules out on a CPU for its memory footprint."""

    def __init__(self, domain0: Domain, domain1: Domain, fixed: dict, qcp=(), lagrange=()):
        self.d0, self.d1, self.nb_bsb = domain0, domain1, len(qcp)
        lib = domain0.ctx.lib
        keep = []
        a = _lib.PlonkQuotientIn()
        a.nb_bsb = len(qcp)
        _fill(a, FIXED_IDS, fixed, lagrange, domain0.Cardinality, keep)
        if qcp:
            a.qcp = _ptr_array(qcp, domain0.Cardinality, keep)
        h = C.c_void_p()
        lib.check(lib.ga_plonk_pk_create(domain0.handle, domain1.handle, C.byref(a), C.byref(h)))
        self.handle = h

    def ComputeQuotient(self, polys: dict, pi2=(), *, bp: dict, alpha, beta, gamma, lagrange=()):
        """s.h from the per-proof polynomials L, R, O, Z, Qk (and the BSB22 commitment polynomials): ga_plonk_quotient_pinned"""
        lib = self.d0.ctx.lib
        if len(pi2) != self.nb_bsb:
            raise ValueError("one committed polynomial per Qcp")
        keep = []
        a = _lib.PlonkQuotientIn()
        a.nb_bsb = self.nb_bsb
        _fill(a, PROOF_IDS, polys, lagrange, self.d0.Cardinality, keep)
        if pi2:
            a.pi2 = _ptr_array(pi2, self.d0.Cardinality, keep)
        for k, cnt in (("Bl", 2), ("Br", 2), ("Bo", 2), ("Bz", 3)):
            v = as_u64(np.asarray(bp[k]).reshape(cnt, 4), 4)
            keep.append(v)
            setattr(a, k.lower(), v.ctypes.data)
        for k, v in (("alpha", alpha), ("beta", beta), ("gamma", gamma)):
            v = as_u64(np.asarray(v).reshape(1, 4), 4)
            keep.append(v)
            setattr(a, k, v.ctypes.data)
        out = np.zeros((self.d1.Cardinality, 4), dtype=np.uint64)
        lib.check(lib.ga_plonk_quotient_pinned(self.handle, C.byref(a), _ptr(out)))
        return out

    def close(self):
        if self.handle:
            self.d0.ctx.lib.ga_plonk_pk_destroy(self.handle)
            self.handle = None


def ComputeQuotient(domain0: Domain, domain1: Domain, polys: dict, qcp=(), pi2=(), *, bp: dict, alpha, beta, gamma, lagrange=()):
    """s.h = divideByZH(computeNumerator()).  polys: {id: (n, 4) fr images} for IDS, canonical coefficients unless the id is
    listed in `lagrange` (then evaluations on domain0, regular order; "Qcp<i>"/"Pi2<i>" name the BSB22 pairs);
    bp: {"Bl","Br","Bo": (2,4), "Bz": (3,4)} blinding polynomials.  Returns (rho*n, 4) canonical coefficients of h."""
    lib = domain0.ctx.lib
    n = domain0.Cardinality
    arrs = [as_u64(np.asarray(polys[k]).reshape(-1, 4), 4) for k in IDS]
    qc = [as_u64(np.asarray(a).reshape(-1, 4), 4) for a in qcp]
    pi = [as_u64(np.asarray(a).reshape(-1, 4), 4) for a in pi2]
    if len(qc) != len(pi):
        raise ValueError("qcp and pi2 must have the same length")
    for a in arrs + qc + pi:
        if a.shape[0] != n:
            raise ValueError(f"every polynomial must have {n} coefficients")
    names = list(IDS) + [x for i in range(len(qc)) for x in (f"Qcp{i}", f"Pi2{i}")]
    mask = 0
    for nm in lagrange:
        mask |= 1 << names.index(nm)
    a = _lib.PlonkQuotientIn()
    a.nb_bsb = len(qc)
    for k, arr in zip(("l", "r", "o", "z", "ql", "qr", "qm", "qo", "qk", "s1", "s2", "s3"), arrs):
        setattr(a, k, arr.ctypes.data)
    if qc:
        a.qcp = (C.c_void_p * len(qc))(*[x.ctypes.data for x in qc])
        a.pi2 = (C.c_void_p * len(pi))(*[x.ctypes.data for x in pi])
    a.lagrange_mask = mask
    keep = []
    for k, cnt in (("Bl", 2), ("Br", 2), ("Bo", 2), ("Bz", 3)):
        v = as_u64(np.asarray(bp[k]).reshape(cnt, 4), 4)
        keep.append(v)
        setattr(a, k.lower(), v.ctypes.data)
    for k, v in (("alpha", alpha), ("beta", beta), ("gamma", gamma)):
        v = as_u64(np.asarray(v).reshape(1, 4), 4)
        keep.append(v)
        setattr(a, k, v.ctypes.data)
    out = np.zeros((domain1.Cardinality, 4), dtype=np.uint64)
    lib.check(lib.ga_plonk_quotient(domain0.handle, domain1.handle, C.byref(a), _ptr(out)))
    return out


def BuildRatioCopyConstraint(domain0: Domain, L, R, O, permutation, beta, gamma) -> np.ndarray:
    """iop.BuildRatioCopyConstraint([L, R, O], s.trace.S, beta, gamma, Lagrange/Regular, domain0) -> Z evaluations (n, 4)."""
    lib = domain0.ctx.lib
    n = domain0.Cardinality
    L, R, O = (as_u64(np.asarray(x).reshape(-1, 4), 4) for x in (L, R, O))
    perm = np.ascontiguousarray(permutation, dtype=np.int64)
    if perm.shape != (3 * n,) or L.shape[0] != n or R.shape[0] != n or O.shape[0] != n:
        raise ValueError("L, R, O need n evaluations each and the permutation 3n entries")
    b, g = as_u64(np.asarray(beta).reshape(1, 4), 4), as_u64(np.asarray(gamma).reshape(1, 4), 4)
    out = np.zeros((n, 4), dtype=np.uint64)
    lib.check(lib.ga_plonk_build_z(domain0.handle, _ptr(L), _ptr(R), _ptr(O), _ptr(perm), _ptr(b), _ptr(g), 0, _ptr(out)))
    return out


def BatchInvert(ctx, curve, v) -> np.ndarray:
    """fr.BatchInvert (zeros stay zero)"""
    from .device import curve_id
    a = as_u64(np.asarray(v).reshape(-1, 4), 4).copy()
    ctx.lib.check(ctx.lib.ga_fr_batch_invert(ctx.handle, curve_id(curve), _ptr(a), a.shape[0], 0))
    return a


def LinearCombination(ctx, curve, scalars, vectors) -> np.ndarray:
    """sum_j scalars[j] * vectors[j] (polynomial folding / linearised polynomial): ga_fr_linear_combination, <= 16 terms"""
    from .device import curve_id
    vs = [as_u64(np.asarray(v).reshape(-1, 4), 4) for v in vectors]
    sc = as_u64(np.asarray(scalars).reshape(-1, 4), 4)
    if sc.shape[0] != len(vs) or any(v.shape != vs[0].shape for v in vs):
        raise ValueError("one scalar per vector, vectors of equal length")
    ptrs = (C.c_void_p * len(vs))(*[v.ctypes.data for v in vs])
    out = np.zeros_like(vs[0])
    ctx.lib.check(ctx.lib.ga_fr_linear_combination(ctx.handle, curve_id(curve), vs[0].shape[0], len(vs), ptrs, _ptr(sc), _ptr(out), 0))
    return out


def Evaluate(ctx, curve, poly, point) -> np.ndarray:
    """p(point) for canonical coefficients (iop.Polynomial.Evaluate): ga_fr_poly_evaluate"""
    from .device import curve_id
    a = as_u64(np.asarray(poly).reshape(-1, 4), 4)
    z = as_u64(np.asarray(point).reshape(1, 4), 4)
    out = np.zeros(4, dtype=np.uint64)
    ctx.lib.check(ctx.lib.ga_fr_poly_evaluate(ctx.handle, curve_id(curve), _ptr(a), a.shape[0], _ptr(z), _ptr(out), 0))
    return out
