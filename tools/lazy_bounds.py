#!/usr/bin/env python3
"""Interval analysis behind the lazy (unreduced) arithmetic of gnark_amd/csrc/field29.hip.h + msm.hip.h::madd29.

Every value is tracked by an upper bound.  The representation needs (NL limbs of L bits, R' = 2^(NL*L)):
  * values < R' (the top limb must stay below 2^L so that column sums of 2*NL products fit 64 bits);
  * for every  a - b + K*p :  b < K*p  (with a margin of one top-limb unit, 2^(L*(NL-1)));
  * Fp2 operands below FP2Z_K*p (the negation constant of the schoolbook-on-columns product).
`check(...)` iterates the mixed-addition formulas to a fixed point of the accumulator bounds and asserts all of the
above with exactly the constants the kernels use.  tests/test_lazy_bounds.py runs it for both curves."""
from math import log2

BN254_P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
BLS12_381_P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
CURVES = {"bn254": (BN254_P, 29, 9, 254), "bls12-381": (BLS12_381_P, 28, 14, 381)}

# constants used by msm.hip.h::madd29 (G1) and its Fp2 overload (G2), and by field29.hip.h's Fp2 product / square
G1 = dict(Kx=8, Ky=8, K3=4, Kq=8, Ky3=2)
G2 = dict(Kx=4, Ky=4, K3=4, Kq=8, Ky3=8, KQ=8, partial_reduce=("X",))


def check(curve: str, fp2: bool, verbose=False, init=None):
    """init: optional upper bounds (X, Y, ZZ, ZZZ) the accumulator may START from besides an affine point (the output of mdbl29)"""
    p, L, NL, bits = CURVES[curve]
    R = 1 << (L * NL)
    unit = 1 << (L * (NL - 1))
    k = G2 if fp2 else G1

    def lim(v):
        assert v < R, ("value exceeds R'", log2(v))
        return v

    def need(K, b, what):
        assert K * p - b > unit, (what, K, log2(b), log2(K * p))

    def mul1(a, b):
        lim(a), lim(b)
        return a * b // R + p

    def pr(v):   # f29_partial_reduce
        q = v >> bits
        return (1 << bits) + q * ((1 << bits) - p)

    if fp2:
        # Fp2 product (field29.hip.h f29_mul on F29x2): schoolbook on unreduced columns, real part a0*b0 + (K*p - a1)*b1 with
        # K = P::FP2Z_K (gen_constants.FP2_LAZY_K), imaginary part a0*b1 + a1*b0; two reductions
        import os
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from gen_constants import FP2_LAZY_K

        def mul(a, b):
            assert a < FP2_LAZY_K * p and b < FP2_LAZY_K * p, ("Fp2 operand above FP2Z_K*p", log2(a), log2(b))
            lim(a), lim(b)
            assert 2 * NL * (1 << (2 * L)) + NL * (1 << (2 * L)) < 1 << 64
            return max((a * b + (FP2_LAZY_K * p + unit) * b) // R + p, 2 * a * b // R + p)

    if fp2:
        def sqr(a):
            need(k["KQ"], a, "KQ")
            return max(mul1(lim(2 * a), a + k["KQ"] * p), 2 * mul1(a, a))
        red = set(k["partial_reduce"])
    else:
        mul, red = mul1, set()

        def sqr(a):
            return mul1(a, a)

    bx = by = bzz = bzzz = p                 # accumulator starts as an affine point (canonical limbs)
    by = 2 * p                               # (a negated y is 2p - y)
    if init is not None:
        bx, by, bzz, bzzz = (max(a, b) for a, b in zip((bx, by, bzz, bzzz), init))
    for _ in range(1000):
        qx, qy = p, 2 * p                    # table points are canonical; a negated y is 2p - y
        U2, S2 = mul(qx, bzz), mul(qy, bzzz)
        need(k["Kx"], bx, "Kx")
        need(k["Ky"], by, "Ky")
        Pp, Rr = U2 + k["Kx"] * p, S2 + k["Ky"] * p
        if "P" in red:
            Pp = pr(Pp)
        if "R" in red:
            Rr = pr(Rr)
        PP = sqr(Pp)
        PPP, Q = mul(Pp, PP), mul(bx, PP)
        if "PPP" in red:
            PPP = pr(PPP)
        need(k["K3"], PPP + 2 * Q, "K3")
        X3 = sqr(Rr) + k["K3"] * p
        if "X" in red:
            X3 = pr(X3)
        need(k["Kq"], X3, "Kq")
        t = Q + k["Kq"] * p
        # Y3 = R*t - Y1*PPP as ONE column accumulation R*t + (K*p - Y1)*PPP (f29_mul_sub): K = 8 (G1), FP2Z_K = 16 (G2)
        Kms = 16 if fp2 else 8
        assert by + unit < Kms * p, ("Y1 above the negation constant of f29_mul_sub", log2(by))
        lim(Rr), lim(t), lim(PPP)
        if fp2:
            Y3 = max(Rr * t + Kms * p * t + Kms * p * PPP + by * PPP, 2 * Rr * t + 2 * Kms * p * PPP) // R + p
            assert max(Rr, t, by, PPP) < 16 * p
        else:
            Y3 = (Rr * t + Kms * p * PPP) // R + p
        ZZ3, ZZZ3 = mul(bzz, PP), mul(bzzz, PPP)
        for v in (Pp, Rr, PP, PPP, Q, X3, t, Y3, ZZ3, ZZZ3):
            lim(v)
        nb = (max(bx, X3), max(by, Y3), max(bzz, ZZ3), max(bzzz, ZZZ3))
        if nb == (bx, by, bzz, bzzz):
            break
        bx, by, bzz, bzzz = nb
    else:
        raise AssertionError("accumulator bounds do not converge")
    out = {"X": log2(bx), "Y": log2(by), "ZZ": log2(bzz), "ZZZ": log2(bzzz), "P": log2(Pp), "R": log2(Rr), "limit": L * NL}
    if verbose:
        print(curve, "G2" if fp2 else "G1", {a: round(b, 2) for a, b in out.items()})
    return out


# ---- general XYZZ + XYZZ addition in the lazy representation (msm.hip.h::add29, window reduction) -------------------------
ADD_G1 = dict(KP=4, KR=4, K3=4, Kq=8, Kms=8, partial_reduce=())
ADD_G2 = dict(KP=4, KR=4, K3=4, Kq=8, Kms=16, partial_reduce=("X",))


def check_add(curve: str, fp2: bool, verbose=False):
    """Fixed point of the coordinate bounds under a = add29(a, b) when both operands are earlier results (running sums added
    into running sums); every subtraction constant and every Fp2 operand bound of msm.hip.h::add29 is asserted."""
    p, L, NL, bits = CURVES[curve]
    R = 1 << (L * NL)
    unit = 1 << (L * (NL - 1))
    k = ADD_G2 if fp2 else ADD_G1
    red = set(k["partial_reduce"])

    def lim(v):
        assert v < R // 4, ("value exceeds R'/4", log2(v))
        return v

    def need(K, b, what):
        assert K * p - b > unit, (what, K, log2(b), log2(K * p))

    def mul1(a, b):
        lim(a), lim(b)
        return a * b // R + p

    def pr(v):
        q = v >> bits
        return (1 << bits) + q * ((1 << bits) - p)

    if fp2:
        from gen_constants import FP2_LAZY_K as FK

        def mul(a, b):   # f29_mul(F29x2): real a0*b0 + (FK*p - a1)*b1, imaginary a0*b1 + a1*b0
            assert a + unit < FK * p and b < FK * p, ("Fp2 operand above FP2Z_K*p", log2(a), log2(b))
            lim(a), lim(b)
            return max((a * b + (FK * p + unit) * b) // R + p, 2 * a * b // R + p)

        def sqr(a):
            need(G2["KQ"], a, "KQ")
            return max(mul1(lim(2 * a), a + G2["KQ"] * p), 2 * mul1(a, a))

        def mulsub(K, a, b, c, d):
            assert max(a, c) + unit < K * p and max(a, b, c, d) < FK * p
            return max(a * b + K * p * b + K * p * d + c * d, 2 * a * b + 2 * K * p * d) // R + p
    else:
        mul = mul1

        def sqr(a):
            return mul1(a, a)

        def mulsub(K, a, b, c, d):
            assert c + unit < K * p
            lim(a), lim(b), lim(d)
            return (a * b + K * p * d) // R + p

    bx = by = bzz = bzzz = p
    for _ in range(1000):
        U = mul(bx, bzz)           # U1 = X1*ZZ2, U2 = X2*ZZ1: same bound
        S = mul(by, bzzz)
        need(k["KP"], U, "KP")
        need(k["KR"], S, "KR")
        Pp, Rr = U + k["KP"] * p, S + k["KR"] * p
        PP = sqr(Pp)
        PPP, Q = mul(Pp, PP), mul(U, PP)
        need(k["K3"], PPP + 2 * Q, "K3")
        X3 = sqr(Rr) + k["K3"] * p
        if "X" in red:
            X3 = pr(X3)
        need(k["Kq"], X3, "Kq")
        t = Q + k["Kq"] * p
        Y3 = mulsub(k["Kms"], Rr, t, S, PPP)
        ZZ3, ZZZ3 = mul(mul(bzz, bzz), PP), mul(mul(bzzz, bzzz), PPP)
        for v in (Pp, Rr, PP, PPP, Q, X3, t, Y3, ZZ3, ZZZ3):
            lim(v)
        nb = (max(bx, X3), max(by, Y3), max(bzz, ZZ3), max(bzzz, ZZZ3))
        if nb == (bx, by, bzz, bzzz):
            break
        bx, by, bzz, bzzz = nb
    else:
        raise AssertionError("bounds of the general addition do not converge")
    out = {"X": log2(bx), "Y": log2(by), "ZZ": log2(bzz), "ZZZ": log2(bzzz), "P": log2(Pp), "R": log2(Rr), "limit": L * NL}
    if verbose:
        print(curve, "add G2" if fp2 else "add G1", {a: round(b, 2) for a, b in out.items()})
    return out


# ---- affine doubling into the lazy accumulator (msm.hip.h::mdbl29, the complete variant of the bucket loop) ---------------------
def check_mdbl(curve: str, fp2: bool, verbose=False):
    """acc = 2*(qx, qy) for a table point (qx canonical, qy canonical or negated = 2p - y): every subtraction constant and Fp2
    operand bound of mdbl29, and the outputs must not exceed the accumulator bounds madd29 was analysed with (check())."""
    p, L, NL, bits = CURVES[curve]
    R = 1 << (L * NL)
    unit = 1 << (L * (NL - 1))

    def lim(v):
        assert v < R, ("value exceeds R'", log2(v))
        return v

    def need(K, b, what):
        assert K * p - b > unit, (what, K, log2(b), log2(K * p))

    def mul1(a, b):
        lim(a), lim(b)
        return a * b // R + p

    def pr(v):
        q = v >> bits
        return (1 << bits) + q * ((1 << bits) - p)

    if fp2:
        import os
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from gen_constants import FP2_LAZY_K as FK

        def mul(a, b):
            assert a + unit < FK * p and b < FK * p, ("Fp2 operand above FP2Z_K*p", log2(a), log2(b))
            return max((a * b + (FK * p + unit) * b) // R + p, 2 * a * b // R + p)

        def sqr(a):
            need(G2["KQ"], a, "KQ")
            return max(mul1(lim(2 * a), a + G2["KQ"] * p), 2 * mul1(a, a))

        def mulsub(K, a, b, c, d):
            assert max(a, c) + unit < K * p and max(a, b, c, d) < FK * p
            return max(a * b + K * p * b + K * p * d + c * d, 2 * a * b + 2 * K * p * d) // R + p
        Kms = FK
    else:
        mul = mul1

        def sqr(a):
            return mul1(a, a)

        def mulsub(K, a, b, c, d):
            assert c + unit < K * p
            lim(a), lim(b), lim(d)
            return (a * b + K * p * d) // R + p
        Kms = 8
    qx, qy = p, 2 * p
    U = 2 * qy
    V = sqr(U)
    W = mul(U, V)
    S = mul(qx, V)
    xx = sqr(qx)
    M = 3 * xx
    need(4, 2 * S, "X3: 2S below 4p")
    X3 = sqr(M) + 4 * p
    if fp2:
        X3 = pr(X3)
    need(8, X3, "t: X3 below 8p")
    t = S + 8 * p
    Y3 = mulsub(Kms, M, t, W, qy)
    out = {"X": log2(X3), "Y": log2(Y3), "ZZ": log2(V), "ZZZ": log2(W)}
    # the mixed additions that follow start from these values: every assertion of check() must hold from there too
    acc = check(curve, fp2, init=(X3, Y3, V, W))
    if verbose:
        print(curve, "mdbl G2" if fp2 else "mdbl G1", {a: round(b, 2) for a, b in out.items()}, "accumulator bounds from there",
              {k: round(acc[k], 2) for k in out})
    return out


# ---- repeated doubling of a general XYZZ point (msm.hip.h::dbl29, the window-table build) ----------------------------------------
def check_dbl(curve: str, fp2: bool, verbose=False):
    """fixed point of the coordinate bounds under a = dbl29(a), starting from an affine point with canonical coordinates"""
    p, L, NL, bits = CURVES[curve]
    R = 1 << (L * NL)
    unit = 1 << (L * (NL - 1))

    def lim(v):
        assert v < R // 4, ("value exceeds R'/4", log2(v))
        return v

    def need(K, b, what):
        assert K * p - b > unit, (what, K, log2(b), log2(K * p))

    def mul1(a, b):
        lim(a), lim(b)
        return a * b // R + p

    def pr(v):
        q = v >> bits
        return (1 << bits) + q * ((1 << bits) - p)

    if fp2:
        import os
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from gen_constants import FP2_LAZY_K as FK

        def mul(a, b):
            assert a + unit < FK * p and b < FK * p, ("Fp2 operand above FP2Z_K*p", log2(a), log2(b))
            return max((a * b + (FK * p + unit) * b) // R + p, 2 * a * b // R + p)

        def sqr(a):
            need(G2["KQ"], a, "KQ")
            return max(mul1(lim(2 * a), a + G2["KQ"] * p), 2 * mul1(a, a))

        def mulsub(K, a, b, c, d):
            assert max(a, c) + unit < K * p and max(a, b, c, d) < FK * p
            return max(a * b + K * p * b + K * p * d + c * d, 2 * a * b + 2 * K * p * d) // R + p
        Kms = FK
    else:
        mul = mul1

        def sqr(a):
            return mul1(a, a)

        def mulsub(K, a, b, c, d):
            assert c + unit < K * p
            lim(a), lim(b), lim(d)
            return (a * b + K * p * d) // R + p
        Kms = 8
    bx = by = bzz = bzzz = p
    for _ in range(1000):
        U = 2 * by
        V = sqr(U)
        W = mul(U, V)
        S = mul(bx, V)
        xx = sqr(bx)
        M = 3 * xx
        need(4, 2 * S, "X3: 2S below 4p")
        X3 = sqr(M) + 4 * p
        if fp2:
            X3 = pr(X3)
        need(8, X3, "t: X3 below 8p")
        t = S + 8 * p
        Y3 = mulsub(Kms, M, t, W, by)
        ZZ3, ZZZ3 = mul(V, bzz), mul(W, bzzz)
        for v in (U, V, W, S, M, X3, t, Y3, ZZ3, ZZZ3):
            lim(v)
        nb = (max(bx, X3), max(by, Y3), max(bzz, ZZ3), max(bzzz, ZZZ3))
        if nb == (bx, by, bzz, bzzz):
            break
        bx, by, bzz, bzzz = nb
    else:
        raise AssertionError("bounds of the repeated doubling do not converge")
    out = {"X": log2(bx), "Y": log2(by), "ZZ": log2(bzz), "ZZZ": log2(bzzz), "limit": L * NL}
    if verbose:
        print(curve, "dbl G2" if fp2 else "dbl G1", {a: round(b, 2) for a, b in out.items()})
    return out


if __name__ == "__main__":
    for c in CURVES:
        for fp2 in (False, True):
            check_dbl(c, fp2, verbose=True)
    for c in CURVES:
        for fp2 in (False, True):
            check_mdbl(c, fp2, verbose=True)
    for c in CURVES:
        for fp2 in (False, True):
            check_add(c, fp2, verbose=True)
    for c in CURVES:
        for fp2 in (False, True):
            check(c, fp2, verbose=True)


# ---- plonk.hip.h::plonk_constraints29_kernel (round 5): the PLONK constraint expression on unreduced limbs, BN254's Fr -------------
BN254_R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
BLS12_381_R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
PLONK_SUB = dict(ord=8, zm1=2)   # f29_sub<8>(ll, rr), f29_sub<2>(z, 1)


def check_plonk_constraints(r=BN254_R, nb_bsb=16, L=29, NL=9):
    """Upper bound of every value of plonk_constraints29_kernel, statement by statement.  mulc(c, v) = f29_mul(32c mod r, v): the
    constant factor is canonical, v only has to be a normalized limb vector (< 2^261).  mulv(s, v) = f29_mul(32 s, v): s < 2^256.
    Either way the result is < factor * v / 2^261 + r.  Returns the bounds in units of r."""
    R = 1 << (L * NL)
    unit = 1 << (L * (NL - 1))

    def lim(v):
        assert v < R, ("value exceeds 2^261", v / r)
        return v

    def mulc(v):
        return (r * lim(v)) // R + r

    def mulv(s, v):
        assert s < (1 << 256), ("shifted factor reaches 2^256", s / r)
        return (32 * s * lim(v)) // R + r

    def sub(K, a, b, what):
        assert K * r - b > unit, (what, K, b / r)
        return lim(a + K * r)
    x = mulv(r, r)                     # lo * hi of the point tables
    xw = mulc(x)
    lro = r + r + mulc(x)              # L + bl0 + bl1 * x
    bz = lambda pt: r + mulv(pt, r + mulc(pt))
    z, zs = r + bz(x), r + bz(xw)
    gate = 3 * mulv(r, lro) + mulv(mulv(r, lro), lro) + r + nb_bsb * mulv(r, r)
    idv = mulc(x)
    a = r + lro + idv
    b = mulc(idv) + lro + r
    rr = mulv(z, mulv(b, mulv(a, b)))
    a2 = mulc(r) + lro + r
    ll = mulv(zs, mulv(a2, mulv(a2, a2)))
    ordv = sub(PLONK_SUB["ord"], ll, rr, "ord")
    lone = mulc(r)
    loc = mulv(lone, sub(PLONK_SUB["zm1"], z, r, "z - 1"))
    res = mulc(mulc(loc) + ordv) + gate
    lim(res)
    out = mulc(res)
    assert out < R                      # f29_reduce_3p takes any normalized value below 2^261
    return {k: v / r for k, v in dict(x=x, l=lro, z=z, gate=gate, a=a, rr=rr, ll=ll, ord=ordv, loc=loc, res=res, out=out, limit=R).items()}


if __name__ == "__main__" and False:
    print(check_plonk_constraints())
