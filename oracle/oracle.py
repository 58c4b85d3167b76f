"""ctypes wrapper of oracle/liboracle.so (oracle.c).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by gnark_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_P = C.c_void_p
FP_LIMBS = {0: 4, 1: 6}


class OraclePk(C.Structure):
    _fields_ = [("curve", C.c_int), ("n", C.c_uint64),
                ("alpha1", _P), ("beta1", _P), ("delta1", _P),
                ("A", _P), ("len_a", C.c_uint64), ("B", _P), ("len_b", C.c_uint64),
                ("Z", _P), ("len_z", C.c_uint64), ("K", _P), ("len_k", C.c_uint64),
                ("beta2", _P), ("delta2", _P), ("B2", _P), ("len_b2", C.c_uint64),
                ("inf_a", _P), ("inf_b", _P), ("nb_wires", C.c_uint64)]


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("oracle.c", "msm_fast.c", "msm_fast_body.inc", "oracle_constants.h")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call([os.path.join(_HERE, "build.sh")], stdout=subprocess.DEVNULL)
    return _SO


_dll = None


def dll():
    global _dll
    if _dll is None:
        _dll = C.CDLL(build())
    return _dll


def _p(a):
    return a.ctypes.data_as(_P)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def jac_words(curve, group):
    return FP_LIMBS[curve] * (3 if group == 0 else 6)


def aff_words(curve, group):
    return FP_LIMBS[curve] * (2 if group == 0 else 4)


def msm(curve, group, points, scalars, mont=True, nthreads=1, naive=False):
    points, scalars = _u64(points), _u64(scalars)
    n = scalars.reshape(-1, 4).shape[0]
    out = np.zeros(jac_words(curve, group), dtype=np.uint64)
    if naive:
        dll().oracle_msm_naive(curve, group, _p(points), _p(scalars), C.c_size_t(n), int(mont), _p(out))
    else:
        dll().oracle_msm(curve, group, _p(points), _p(scalars), C.c_size_t(n), int(mont), _p(out), nthreads)
    return out


def msm_fast(curve, points, scalars, mont=True, nthreads=1, force_c=0, force_splits=0):
    """msm_fast.c: G1 MSM with signed digits, batch-affine buckets and (window x point-range) tasks on `nthreads` threads -- the
    CPU baseline bench.py quotes (kind "port-batch-affine"); same result contract as msm(curve, 0, ...)"""
    points, scalars = _u64(points), _u64(scalars)
    n = scalars.reshape(-1, 4).shape[0]
    out = np.zeros(jac_words(curve, 0), dtype=np.uint64)
    d = dll()
    d.oracle_msm_fast.restype = C.c_int
    rc = d.oracle_msm_fast(curve, _p(points), _p(scalars), C.c_size_t(n), int(mont), _p(out), int(nthreads), int(force_c), int(force_splits))
    if rc != 0:
        raise ValueError("oracle_msm_fast: rc %d" % rc)
    return out


def msm_fast_plan(curve, n, nthreads):
    """(window bits, windows, range splits) msm_fast uses for n points on nthreads threads"""
    o = (C.c_int * 3)()
    dll().oracle_msm_fast_plan(curve, C.c_size_t(n), int(nthreads), o)
    return tuple(o)


def msm_windows(curve, n) -> int:
    return dll().oracle_msm_windows(curve, C.c_size_t(n))


def jac_to_affine(curve, group, jac):
    jac = _u64(jac)
    out = np.zeros(aff_words(curve, group), dtype=np.uint64)
    dll().oracle_jac_to_affine(curve, group, _p(jac), _p(out))
    return out


def jac_add(curve, group, a, b):
    a, b = _u64(a), _u64(b)
    out = np.zeros(jac_words(curve, group), dtype=np.uint64)
    dll().oracle_jac_add(curve, group, _p(a), _p(b), _p(out))
    return out


def generator_mul(curve, group, k: int):
    kk = np.array([(k >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    out = np.zeros(jac_words(curve, group), dtype=np.uint64)
    dll().oracle_generator_mul(curve, group, _p(kk), _p(out))
    return out


def gen_bases(curve, group, ks):
    ks = _u64(ks)
    out = np.zeros((ks.shape[0], aff_words(curve, group)), dtype=np.uint64)
    dll().oracle_gen_bases(curve, group, _p(ks), C.c_size_t(ks.shape[0]), _p(out))
    return out


def fr_dot(curve, a_mont, b_canon) -> int:
    a, b = _u64(a_mont), _u64(b_canon)
    out = np.zeros(4, dtype=np.uint64)
    dll().oracle_fr_dot(curve, _p(a), _p(b), C.c_size_t(a.reshape(-1, 4).shape[0]), _p(out))
    return sum(int(v) << (64 * i) for i, v in enumerate(out))


def fr_from_mont(curve, a):
    a = _u64(a).reshape(-1, 4)
    out = np.zeros_like(a)
    dll().oracle_fr_from_mont(curve, _p(a), C.c_size_t(a.shape[0]), _p(out))
    return out


def fr_mul(curve, a, b):
    a, b = _u64(a).reshape(-1, 4), _u64(b).reshape(-1, 4)
    out = np.zeros_like(a)
    dll().oracle_fr_mul(curve, _p(a), _p(b), C.c_size_t(a.shape[0]), _p(out))
    return out


def fr_add(curve, a, b):
    a, b = _u64(a).reshape(-1, 4), _u64(b).reshape(-1, 4)
    out = np.zeros_like(a)
    dll().oracle_fr_add(curve, _p(a), _p(b), C.c_size_t(a.shape[0]), _p(out))
    return out


def fr_sub(curve, a, b):
    a, b = _u64(a).reshape(-1, 4), _u64(b).reshape(-1, 4)
    out = np.zeros_like(a)
    dll().oracle_fr_sub(curve, _p(a), _p(b), C.c_size_t(a.shape[0]), _p(out))
    return out


def fr_powers(curve, base_mont, first_mont, n):
    """[first * base^i for i < n] (Montgomery)"""
    b, f = _u64(base_mont).reshape(4), _u64(first_mont).reshape(4)
    out = np.zeros((n, 4), dtype=np.uint64)
    dll().oracle_fr_powers(curve, _p(b), _p(f), C.c_size_t(n), _p(out))
    return out


def fr_horner(curve, coeffs_mont, x_mont):
    c, x = _u64(coeffs_mont).reshape(-1, 4), _u64(x_mont).reshape(4)
    out = np.zeros(4, dtype=np.uint64)
    dll().oracle_fr_horner(curve, _p(c), C.c_size_t(c.shape[0]), _p(x), _p(out))
    return out


def fft(curve, a, direction, decimation, on_coset, nthreads=1):
    out = _u64(a).reshape(-1, 4).copy()
    rc = dll().oracle_fft_mt(curve, _p(out), C.c_uint64(out.shape[0]), direction, decimation, int(on_coset), int(nthreads))
    assert rc == 0
    return out


def fr_eval_lagrange(curve, vecs, x_mont, nthreads=1):
    """[P_v(x)] for polynomials given by their values on the size-n domain (barycentric formula, O(n), x outside the domain)"""
    arrs = [_u64(v).reshape(-1, 4) for v in vecs]
    n = arrs[0].shape[0]
    assert all(a.shape[0] == n for a in arrs)
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    x = _u64(x_mont).reshape(4)
    out = np.zeros((len(arrs), 4), dtype=np.uint64)
    rc = dll().oracle_fr_eval_lagrange(curve, ptrs, len(arrs), C.c_uint64(n), _p(x), _p(out), int(nthreads))
    assert rc == 0
    return out


def fr_eval_bitrev(curve, coeffs_bitrev, x_mont):
    """P(x) for coefficients stored in bit-reversed order (computeH's output order)"""
    c, x = _u64(coeffs_bitrev).reshape(-1, 4), _u64(x_mont).reshape(4)
    out = np.zeros(4, dtype=np.uint64)
    rc = dll().oracle_fr_eval_bitrev(curve, _p(c), C.c_uint64(c.shape[0]), _p(x), _p(out))
    assert rc == 0
    return out


def plonk_quotient(curve, n, polys, bl, br, bo, bz, alpha, beta, gamma, nb_bsb=0):
    """computeNumerator + divideByZH on canonical coefficient vectors (order L R O Z Ql Qr Qm Qo Qk S1 S2 S3, then Qcp_i, Pi2_i)"""
    arrs = [_u64(x).reshape(-1, 4) for x in polys]
    assert len(arrs) == 12 + 2 * nb_bsb and all(a.shape[0] == n for a in arrs)
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    rho = 8 if n < 6 else 4
    out = np.zeros((rho * n, 4), dtype=np.uint64)
    small = [_u64(x).reshape(-1, 4) for x in (bl, br, bo, bz, alpha, beta, gamma)]
    rc = dll().oracle_plonk_quotient(curve, C.c_uint64(n), nb_bsb, ptrs, *[_p(x) for x in small], _p(out))
    assert rc == 0
    return out


def compute_h(curve, a, b, c, n, nthreads=1):
    a, b, c = (_u64(x).reshape(-1, 4) for x in (a, b, c))
    out = np.zeros((n, 4), dtype=np.uint64)
    dll().oracle_compute_h_mt(curve, _p(a), _p(b), _p(c), C.c_uint64(a.shape[0]), C.c_uint64(n), _p(out), int(nthreads))
    return out


def groth16_prove(curve, key: dict, W, A, B, Cc, nb_public, r_mont, s_mont, nthreads=1):
    """key: dict of uint64 arrays (alpha1,beta1,delta1,A,B,Z,K,beta2,delta2,B2) + infinityA/B (uint8) + n."""
    k = {name: _u64(key[name]) for name in ("alpha1", "beta1", "delta1", "A", "B", "Z", "K", "beta2", "delta2", "B2")}
    ia = np.ascontiguousarray(key["infinityA"], dtype=np.uint8)
    ib = np.ascontiguousarray(key["infinityB"], dtype=np.uint8)
    fp = FP_LIMBS[curve]
    pk = OraclePk()
    pk.curve, pk.n = curve, int(key["n"])
    pk.alpha1, pk.beta1, pk.delta1 = (k[x].ctypes.data for x in ("alpha1", "beta1", "delta1"))
    pk.A, pk.len_a = k["A"].ctypes.data, k["A"].size // (2 * fp)
    pk.B, pk.len_b = k["B"].ctypes.data, k["B"].size // (2 * fp)
    pk.Z, pk.len_z = k["Z"].ctypes.data, k["Z"].size // (2 * fp)
    pk.K, pk.len_k = k["K"].ctypes.data, k["K"].size // (2 * fp)
    pk.beta2, pk.delta2 = k["beta2"].ctypes.data, k["delta2"].ctypes.data
    pk.B2, pk.len_b2 = k["B2"].ctypes.data, k["B2"].size // (4 * fp)
    pk.inf_a, pk.inf_b, pk.nb_wires = ia.ctypes.data, ib.ctypes.data, ia.shape[0]
    W, A, B, Cc, r, s = (_u64(x) for x in (W, A, B, Cc, r_mont, s_mont))
    out = np.zeros(8 * fp, dtype=np.uint64)
    dll().oracle_groth16_prove(C.byref(pk), _p(W), _p(A), _p(B), _p(Cc), C.c_uint64(A.reshape(-1, 4).shape[0]),
                               C.c_uint64(nb_public), _p(r), _p(s), _p(out), nthreads)
    return out[:2 * fp], out[2 * fp:6 * fp], out[6 * fp:]
