"""Big-integer Python reference for the Groth16/PLONK prover hot path (TEST INFRASTRUCTURE ONLY).

This file is part of the *oracle*: it is imported only by tests/, by tools/ (constant and
fixture generators) and by __graft_entry__.smoke().  The product (gnark_amd/) never imports it.

It restates, with Python integers, the arithmetic that gnark delegates to gnark-crypto v0.21.0
(go.mod:10 of the reference; the module source is NOT in /root/reference, so the algorithms are
restated from their published definitions) and the pieces of the reference that sit on the path:

  * field / curve parameters ............ backend/groth16/bn254/solidity.go:64-65,
                                          std/math/emulated/emparams/emparams.go:142-157,225-241
  * G2 generators and [2^65]G2 literals .. std/algebra/emulated/sw_bn254/g2.go:75-97,
                                          std/algebra/emulated/sw_bls12381/g2.go:89-110
  * computeH ............................. backend/groth16/bn254/prove.go:346-389
  * Groth16 Setup / Prove ................ backend/groth16/bn254/setup.go:75-331, prove.go:52-315
  * point wire formats ................... backend/groth16/bn254/marshal.go:33-58 (+ [EXT] encoder)

Parity pinning: see oracle/README.md (fixtures the reference ships that this file is checked
against in tests/test_oracle_fixtures.py).
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field

# ----------------------------------------------------------------------------------------------
# curve parameter sets
# ----------------------------------------------------------------------------------------------


@dataclass(frozen=True)
class Curve:
    name: str
    cid: int            # id used by the C ABI (include/gnark_amd.h)
    p: int              # base field modulus
    r: int              # scalar field modulus
    fp_limbs: int       # 64-bit limbs of fp.Element
    fr_limbs: int       # 64-bit limbs of fr.Element (4 for both curves)
    b: int              # G1: y^2 = x^3 + b
    b2: tuple           # G2 twist: y^2 = x^3 + b2 (in Fp2 = Fp[u]/(u^2+1))
    g1: tuple           # G1 generator (affine)
    g2: tuple           # G2 generator ((x0,x1),(y0,y1))
    fr_gen: int         # fft.Domain.FrMultiplicativeGen (coset shift)
    fr_adicity: int     # 2-adicity of r-1

    @property
    def fp_bytes(self):
        return self.fp_limbs * 8

    def fr_root_of_unity(self, n: int) -> int:
        """generator of the order-n subgroup exactly as fft.NewDomain derives it [EXT]:
        w_max = fr_gen^((r-1)/2^adicity), w_n = w_max^(2^adicity / n)."""
        assert n & (n - 1) == 0 and n <= (1 << self.fr_adicity)
        wmax = pow(self.fr_gen, (self.r - 1) >> self.fr_adicity, self.r)
        return pow(wmax, (1 << self.fr_adicity) // n, self.r)


def _fp2_inv_raw(a, p):
    a0, a1 = a
    d = pow((a0 * a0 + a1 * a1) % p, -1, p)
    return (a0 * d % p, (-a1 * d) % p)


_BN_P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
_BN_R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
_t = _fp2_inv_raw((9, 1), _BN_P)
BN254 = Curve(
    name="bn254", cid=0, p=_BN_P, r=_BN_R, fp_limbs=4, fr_limbs=4, b=3,
    b2=(3 * _t[0] % _BN_P, 3 * _t[1] % _BN_P),
    g1=(1, 2),
    g2=((10857046999023057135944570762232829481370756359578518086990519993285655852781,
         11559732032986387107991004021392285783925812861821192530917403151452391805634),
        (8495653923123431417604973247489272438418190587263600148770280649306958101930,
         4082367875863433681332203403145435568316851327593401208105741076214120093531)),
    fr_gen=5, fr_adicity=28)

_BLS_P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
_BLS_R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
BLS12_381 = Curve(
    name="bls12-381", cid=1, p=_BLS_P, r=_BLS_R, fp_limbs=6, fr_limbs=4, b=4, b2=(4, 4),
    g1=(0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
        0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1),
    g2=((352701069587466618187139116011060144890029952792775240219908644239793785735715026873347600343865175952761926303160,
         3059144344244213709971259814753781636986470325476647558659373206291635324768958432433509563104347017837885763365758),
        (1985150602287291935568054521177171638300868978215655730859378665066344726373823718423869104263333984641494340347905,
         927553665492332455747201965776037880757740193453592970025027978793976877002675564980949289727957565575433344219582)),
    fr_gen=7, fr_adicity=32)

CURVES = {"bn254": BN254, "bls12-381": BLS12_381, 0: BN254, 1: BLS12_381}

# ----------------------------------------------------------------------------------------------
# generic short-Weierstrass (a = 0) arithmetic over Fp or Fp2, affine with None = infinity
# ----------------------------------------------------------------------------------------------


class FpOps:
    def __init__(self, p):
        self.p = p
        self.zero = 0
        self.one = 1

    def add(self, a, b): return (a + b) % self.p
    def sub(self, a, b): return (a - b) % self.p
    def mul(self, a, b): return a * b % self.p
    def neg(self, a): return (-a) % self.p
    def inv(self, a): return pow(a, -1, self.p)
    def muli(self, a, k): return a * k % self.p
    def is_zero(self, a): return a % self.p == 0


class Fp2Ops:
    """Fp2 = Fp[u]/(u^2+1) for both curves (E2{A0,A1} in gnark-crypto [EXT])."""

    def __init__(self, p):
        self.p = p
        self.zero = (0, 0)
        self.one = (1, 0)

    def add(self, a, b): return ((a[0] + b[0]) % self.p, (a[1] + b[1]) % self.p)
    def sub(self, a, b): return ((a[0] - b[0]) % self.p, (a[1] - b[1]) % self.p)

    def mul(self, a, b):
        p = self.p
        return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def neg(self, a): return ((-a[0]) % self.p, (-a[1]) % self.p)
    def inv(self, a): return _fp2_inv_raw(a, self.p)
    def muli(self, a, k): return (a[0] * k % self.p, a[1] * k % self.p)
    def is_zero(self, a): return a[0] % self.p == 0 and a[1] % self.p == 0


class Group:
    def __init__(self, F, b):
        self.F = F
        self.b = b

    def on_curve(self, P):
        if P is None:
            return True
        F = self.F
        x, y = P
        return F.sub(F.mul(y, y), F.add(F.mul(F.mul(x, x), x), self.b)) == F.zero

    def neg(self, P):
        return None if P is None else (P[0], self.F.neg(P[1]))

    def add(self, P, Q):
        F = self.F
        if P is None:
            return Q
        if Q is None:
            return P
        if P[0] == Q[0]:
            if P[1] == Q[1]:
                if F.is_zero(P[1]):
                    return None
                lam = F.mul(F.muli(F.mul(P[0], P[0]), 3), F.inv(F.muli(P[1], 2)))
            else:
                return None
        else:
            lam = F.mul(F.sub(Q[1], P[1]), F.inv(F.sub(Q[0], P[0])))
        x3 = F.sub(F.sub(F.mul(lam, lam), P[0]), Q[0])
        y3 = F.sub(F.mul(lam, F.sub(P[0], x3)), P[1])
        return (x3, y3)

    def mul(self, P, k):
        if k < 0:
            return self.mul(self.neg(P), -k)
        R = None
        while k:
            if k & 1:
                R = self.add(R, P)
            P = self.add(P, P)
            k >>= 1
        return R

    def msm(self, points, scalars):
        acc = None
        for P, s in zip(points, scalars):
            acc = self.add(acc, self.mul(P, s))
        return acc


def g1_group(c: Curve) -> Group:
    return Group(FpOps(c.p), c.b)


def g2_group(c: Curve) -> Group:
    return Group(Fp2Ops(c.p), c.b2)


# ----------------------------------------------------------------------------------------------
# memory images: Montgomery little-endian 64-bit limbs, as gnark-crypto lays fp/fr.Element out
# ----------------------------------------------------------------------------------------------


def to_mont_limbs(x: int, mod: int, nlimbs: int) -> list:
    v = (x << (64 * nlimbs)) % mod
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nlimbs)]


def from_mont_limbs(limbs, mod: int) -> int:
    n = len(limbs)
    v = sum(int(l) << (64 * i) for i, l in enumerate(limbs))
    return v * pow(1 << (64 * n), -1, mod) % mod


def to_limbs(x: int, nlimbs: int) -> list:
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nlimbs)]


def from_limbs(limbs) -> int:
    return sum(int(l) << (64 * i) for i, l in enumerate(limbs))


# ----------------------------------------------------------------------------------------------
# NTT with gnark-crypto's conventions [EXT fft.Domain.FFT/FFTInverse; call sites prove.go:362-386]
#   DIF: natural in  -> bit-reversed out.   DIT: bit-reversed in -> natural out.
#   FFTInverse multiplies by 1/n.  OnCoset: forward scales coefficient i by g^i before the
#   transform, inverse scales coefficient i by g^-i after it.
# ----------------------------------------------------------------------------------------------

DIF, DIT = 0, 1


def bitrev(i: int, logn: int) -> int:
    r = 0
    for _ in range(logn):
        r = (r << 1) | (i & 1)
        i >>= 1
    return r


def bitrev_permute(a):
    n = len(a)
    logn = n.bit_length() - 1
    return [a[bitrev(i, logn)] for i in range(n)]


def dft_naive(a, w, mod):
    n = len(a)
    return [sum(a[j] * pow(w, j * k, mod) for j in range(n)) % mod for k in range(n)]


def _ntt_natural(a, w, mod):
    """iterative radix-2, natural in -> natural out (textbook)."""
    n = len(a)
    logn = n.bit_length() - 1
    a = bitrev_permute(list(a))
    h = 1
    while h < n:
        wstep = pow(w, n // (2 * h), mod)
        for base in range(0, n, 2 * h):
            t = 1
            for j in range(h):
                x, y = a[base + j], a[base + j + h] * t % mod
                a[base + j] = (x + y) % mod
                a[base + j + h] = (x - y) % mod
                t = t * wstep % mod
        h *= 2
    return a


def fft(c: Curve, a, decimation, on_coset=False, inverse=False):
    """restates fft.Domain.FFT / FFTInverse on len(a) = domain cardinality."""
    mod = c.r
    n = len(a)
    w = c.fr_root_of_unity(n)
    g = c.fr_gen
    nat_in = list(a) if decimation == DIF else bitrev_permute(list(a))  # logical (natural) order
    if not inverse:
        if on_coset:
            nat_in = [x * pow(g, i, mod) % mod for i, x in enumerate(nat_in)]
        out = _ntt_natural(nat_in, w, mod)
    else:
        out = _ntt_natural(nat_in, pow(w, -1, mod), mod)
        ninv = pow(n, -1, mod)
        out = [x * ninv % mod for x in out]
        if on_coset:
            ginv = pow(g, -1, mod)
            out = [x * pow(ginv, i, mod) % mod for i, x in enumerate(out)]
    return bitrev_permute(out) if decimation == DIF else out


def compute_h(c: Curve, a, b, cc, n):
    """backend/groth16/bn254/prove.go:346-389; returns h in bit-reversed order (len n)."""
    mod = c.r
    pad = lambda v: list(v) + [0] * (n - len(v))
    a, b, cc = pad(a), pad(b), pad(cc)
    a = fft(c, a, DIF, inverse=True)
    b = fft(c, b, DIF, inverse=True)
    cc = fft(c, cc, DIF, inverse=True)
    a = fft(c, a, DIT, on_coset=True)
    b = fft(c, b, DIT, on_coset=True)
    cc = fft(c, cc, DIT, on_coset=True)
    den = pow((pow(c.fr_gen, n, mod) - 1) % mod, -1, mod)
    a = [((x * y - z) * den) % mod for x, y, z in zip(a, b, cc)]
    return fft(c, a, DIF, on_coset=True, inverse=True)


# ----------------------------------------------------------------------------------------------
# PLONK quotient: computeNumerator + divideByZH (backend/plonk/bn254/prove.go:841-1123,1287-1350)
# ----------------------------------------------------------------------------------------------
PLONK_IDS = ("L", "R", "O", "Z", "Ql", "Qr", "Qm", "Qo", "Qk", "S1", "S2", "S3")   # prove.go:44-59 without ZS (= Z shifted)


def _poly_eval(coeffs, x, mod):
    acc = 0
    for cf in reversed(coeffs):
        acc = (acc * x + cf) % mod
    return acc


def kzg_open(c: Curve, srs_g1, poly, z):
    """kzg.Open (gnark-crypto [EXT]; call sites backend/plonk/bn254/prove.go:681,788,827): claimed value p(z) and the commitment
    to the quotient (p(X) - p(z)) / (X - z) computed by the Horner recurrence q_{k-1} = p_k + z*q_k."""
    mod = c.r
    n = len(poly)
    q = [0] * max(n - 1, 0)
    acc = 0
    for k in range(n - 1, 0, -1):
        acc = (poly[k] + z * acc) % mod
        q[k - 1] = acc
    value = (poly[0] + z * acc) % mod if n else 0
    return value, g1_group(c).msm(srs_g1[: len(q)], q)


def plonk_rho(n: int) -> int:
    """domain1 = 8n below 6 constraints, else 4n (prove.go:247-251)"""
    return 8 if n < 6 else 4


def plonk_quotient(c: Curve, n: int, x: dict, qcp, pi2, bp: dict, alpha, beta, gamma):
    """x: canonical coefficients (n each) of PLONK_IDS; qcp/pi2: lists of canonical polynomials (Qcp_i, the committed
    polynomial of BSB22 gate i); bp: blinding polynomials {"Bl","Br","Bo" (2 coeffs), "Bz" (3 coeffs)}.  Returns the canonical
    coefficients (rho*n of them) of h, exactly what computeNumerator followed by divideByZH leaves in s.h.

    Follows prove.go:841-1123: for each of the rho cosets coset_i = g*w1^i of the small domain inside the big one, evaluate
    every polynomial on coset_i*H (the reference does it with ToCanonical / scale / ToLagrange, :1033-1058), apply
    allConstraints (:950-981) pointwise, store at the bit-reversed position (:1073); then divideByZH (:1287-1324)."""
    mod = c.r
    rho = plonk_rho(n)
    N = rho * n
    logN = N.bit_length() - 1
    w0 = c.fr_root_of_unity(n)
    w1 = c.fr_root_of_unity(N)
    g = c.fr_gen
    cs, css = g, g * g % mod                               # :891-893
    ninv = pow(n, -1, mod)
    tw = [pow(w0, j, mod) for j in range(n)]               # twiddles0, :845-858
    cres = [0] * N
    coset = 1
    nb_bsb = len(qcp)
    polys = dict(x)
    for i in range(nb_bsb):
        polys["Qc%d" % i], polys["Pi%d" % i] = qcp[i], pi2[i]
    for i in range(rho):
        coset = coset * (g if i == 0 else w1) % mod        # shifters, :936-941,998
        cexp = (pow(coset, n, mod) - 1) % mod              # cosetExponentiatedToNMinusOne, :999-1000
        den = [pow((coset * tw[j] - 1) % mod, -1, mod) for j in range(n)]   # precomputedDenominators, :1002-1007
        ev = {}
        for k, cf in polys.items():                        # batchApply, :1033-1058: evaluations on coset*H, natural order
            sc = [cf[j] * pow(coset, j, mod) % mod for j in range(n)]
            ev[k] = _ntt_natural(sc, w0, mod)
        # blinding polynomials scaled by (coset^n - 1) and by the coset powers (:1009-1017): Evaluate(w^j) = b(coset*w^j)*cexp
        bev = lambda name, j: _poly_eval(bp[name], coset * tw[j % n] % mod, mod) * cexp % mod
        for j in range(n):
            l = (ev["L"][j] + bev("Bl", j)) % mod          # :958-970
            r = (ev["R"][j] + bev("Br", j)) % mod
            o = (ev["O"][j] + bev("Bo", j)) % mod
            z = (ev["Z"][j] + bev("Bz", j)) % mod
            zs = (ev["Z"][(j + 1) % n] + bev("Bz", j + 1)) % mod   # ZS = Z shifted by one (:602,:972-973)
            s1, s2, s3 = ev["S1"][j] * beta % mod, ev["S2"][j] * beta % mod, ev["S3"][j] * beta % mod   # :953-955
            gate = (ev["Ql"][j] * l + ev["Qr"][j] * r + ev["Qm"][j] * l % mod * r + ev["Qo"][j] * o + ev["Qk"][j]) % mod   # :868-885
            for t in range(nb_bsb):
                gate = (gate + ev["Qc%d" % t][j] * ev["Pi%d" % t][j]) % mod
            idv = tw[j] * coset % mod * beta % mod          # :908
            rr = (gamma + l + idv) * ((idv * cs + r + gamma) % mod) % mod * ((idv * css + o + gamma) % mod) % mod * z % mod
            ll = (s1 + l + gamma) * ((s2 + r + gamma) % mod) % mod * ((s3 + o + gamma) % mod) % mod * zs % mod
            ordering = (ll - rr) % mod                      # :921
            lone = cexp * ninv % mod * den[j] % mod         # computeLagrangeOneOnCoset, :380-385
            local = (z - 1) * lone % mod                    # :926-934
            val = ((local * alpha + ordering) % mod * alpha + gate) % mod   # :978-980
            cres[bitrev(rho * j + i, logN)] = val           # :1073
    # divideByZH (:1287-1324): r[i] *= 1/((g*w1^k)^n - 1) with k = bitrev(i) % rho, then ToCanonical on the big coset
    xinv = [pow((pow(g * pow(w1, k, mod) % mod, n, mod) - 1) % mod, -1, mod) for k in range(rho)]   # :1327-1350
    nat = [0] * N
    for idx in range(N):
        m = bitrev(idx, logN)
        nat[m] = cres[idx] * xinv[m % rho] % mod
    co = _ntt_natural(nat, pow(w1, -1, mod), mod)
    Ninv, ginv = pow(N, -1, mod), pow(g, -1, mod)
    return [co[k] * Ninv % mod * pow(ginv, k, mod) % mod for k in range(N)]


def plonk_numerator_at(c: Curve, n: int, x: dict, qcp, pi2, bp: dict, alpha, beta, gamma, zeta):
    """The blinded constraint polynomial evaluated at one point from the canonical coefficients (independent of any FFT):
    gate + alpha*ordering + alpha^2*(Z-1)*L1.  For a satisfying instance it equals h(zeta)*(zeta^n - 1)."""
    mod = c.r
    w0 = c.fr_root_of_unity(n)
    zn1 = (pow(zeta, n, mod) - 1) % mod
    e = lambda k: _poly_eval(x[k], zeta, mod)
    b = lambda k, pt: _poly_eval(bp[k], pt, mod) * ((pow(pt, n, mod) - 1) % mod) % mod
    l, r, o = (e("L") + b("Bl", zeta)) % mod, (e("R") + b("Br", zeta)) % mod, (e("O") + b("Bo", zeta)) % mod
    z = (e("Z") + b("Bz", zeta)) % mod
    zw = zeta * w0 % mod
    zs = (_poly_eval(x["Z"], zw, mod) + b("Bz", zw)) % mod
    gate = (e("Ql") * l + e("Qr") * r + e("Qm") * l % mod * r + e("Qo") * o + e("Qk")) % mod
    for qc, pi in zip(qcp, pi2):
        gate = (gate + _poly_eval(qc, zeta, mod) * _poly_eval(pi, zeta, mod)) % mod
    g = c.fr_gen
    idv = zeta * beta % mod
    rr = (gamma + l + idv) * ((idv * g + r + gamma) % mod) % mod * ((idv * g * g + o + gamma) % mod) % mod * z % mod
    ll = (e("S1") * beta + l + gamma) * ((e("S2") * beta + r + gamma) % mod) % mod * ((e("S3") * beta + o + gamma) % mod) % mod * zs % mod
    lone = zn1 * pow(n, -1, mod) % mod * pow((zeta - 1) % mod, -1, mod) % mod
    return (((z - 1) * lone % mod * alpha + (ll - rr)) % mod * alpha + gate) % mod


def plonk_build_z(c: Curve, n: int, L, R, O, perm, beta, gamma):
    """iop.BuildRatioCopyConstraint (gnark-crypto [EXT], call site backend/plonk/bn254/prove.go:645-655): the grand-product
    polynomial in Lagrange form.  L,R,O: evaluations on the domain; perm: permutation of [0,3n) (s.trace.S).
    Z[0] = 1, Z[i+1] = Z[i] * prod_k (e_k[i] + beta*id_k(i) + gamma) / prod_k (e_k[i] + beta*id(perm[k*n+i]) + gamma)
    with id over the three cosets {1, g, g^2} * w^i."""
    mod = c.r
    w0, g = c.fr_root_of_unity(n), c.fr_gen
    ids = [pow(g, k, mod) * pow(w0, i, mod) % mod for k in range(3) for i in range(n)]
    ev = [L, R, O]
    Z = [1] * n
    for i in range(n - 1):
        num = den = 1
        for k in range(3):
            num = num * ((ev[k][i] + beta * ids[k * n + i] + gamma) % mod) % mod
            den = den * ((ev[k][i] + beta * ids[perm[k * n + i]] + gamma) % mod) % mod
        Z[i + 1] = Z[i] * num % mod * pow(den, -1, mod) % mod
    return Z


def plonk_synthetic_instance(c: Curve, n: int, seed: int, nb_bsb: int = 1):
    """A satisfying PLONK trace of size n built directly in Lagrange form (no circuit): random wire values with copy
    constraints (cycles over equal values), random selectors with Qk chosen so that every gate holds, S1..S3 from the
    permutation, Z from plonk_build_z.  Returns (lagrange dict, qcp, pi2, perm) -- all evaluation vectors."""
    mod = c.r
    rng = Xoshiro(seed)
    pool = [rng.field(mod) for _ in range(max(2, n // 2))]
    assign = [rng.next() % len(pool) for _ in range(3 * n)]
    wires = [pool[a] for a in assign]
    perm = list(range(3 * n))
    by_val = {}
    for pos, a in enumerate(assign):
        by_val.setdefault(a, []).append(pos)
    for cyc in by_val.values():
        for k, pos in enumerate(cyc):
            perm[pos] = cyc[(k + 1) % len(cyc)]
    L, R, O = wires[:n], wires[n:2 * n], wires[2 * n:]
    lag = dict(L=L, R=R, O=O)
    for k in ("Ql", "Qr", "Qm", "Qo"):
        lag[k] = [rng.field(mod) for _ in range(n)]
    qcp = [[rng.field(mod) if rng.next() % 4 == 0 else 0 for _ in range(n)] for _ in range(nb_bsb)]
    pi2 = [[rng.field(mod) for _ in range(n)] for _ in range(nb_bsb)]
    lag["Qk"] = [(-(lag["Ql"][i] * L[i] + lag["Qr"][i] * R[i] + lag["Qm"][i] * L[i] * R[i] + lag["Qo"][i] * O[i]
                    + sum(q[i] * p[i] for q, p in zip(qcp, pi2)))) % mod for i in range(n)]
    w0, g = c.fr_root_of_unity(n), c.fr_gen
    ids = [pow(g, k, mod) * pow(w0, i, mod) % mod for k in range(3) for i in range(n)]
    for k, name in enumerate(("S1", "S2", "S3")):
        lag[name] = [ids[perm[k * n + i]] for i in range(n)]
    return lag, qcp, pi2, perm, rng


# ----------------------------------------------------------------------------------------------
# point (de)compression [EXT gnark-crypto encoders; probed on the reference's fixtures, SURVEY 8c]
# ----------------------------------------------------------------------------------------------


def _sqrt_fp(a, p):
    # both base fields have p = 3 mod 4
    assert p % 4 == 3
    s = pow(a, (p + 1) // 4, p)
    return s if s * s % p == a % p else None


def _sqrt_fp2(a, p):
    """square root in Fp[u]/(u^2+1), p = 3 mod 4 (complex method)."""
    a0, a1 = a
    if a1 == 0:
        s = _sqrt_fp(a0, p)
        if s is not None:
            return (s, 0)
        s = _sqrt_fp((-a0) % p, p)
        return (0, s)
    norm = (a0 * a0 + a1 * a1) % p
    alpha = _sqrt_fp(norm, p)
    if alpha is None:
        return None
    inv2 = pow(2, -1, p)
    delta = (a0 + alpha) * inv2 % p
    x0 = _sqrt_fp(delta, p)
    if x0 is None:
        delta = (a0 - alpha) * inv2 % p
        x0 = _sqrt_fp(delta, p)
        if x0 is None:
            return None
    x1 = a1 * pow(2 * x0, -1, p) % p
    F = Fp2Ops(p)
    assert F.mul((x0, x1), (x0, x1)) == (a0 % p, a1 % p)
    return (x0, x1)


def _lex_larger_fp(y, p):
    return y > (p - 1) // 2


def _lex_larger_fp2(y, p):
    # gnark-crypto E2.LexicographicallyLargest: compare A1 first, A0 if A1 == 0 [EXT]
    if y[1] == 0:
        return _lex_larger_fp(y[0], p)
    return _lex_larger_fp(y[1], p)


def g1_decompress(c: Curve, data: bytes):
    """BN254: 32-byte BE x, top two bits 10 = smaller y, 11 = larger y, 01 = infinity.
    BLS12-381 (ZCash): 48-byte BE x, bit7 compressed, bit6 infinity, bit5 larger y."""
    n = c.fp_bytes
    assert len(data) == n
    if c.name == "bn254":
        flag = data[0] >> 6
        x = int.from_bytes(bytes([data[0] & 0x3F]) + data[1:], "big")
        if flag == 0b01:
            return None
        assert flag in (0b10, 0b11)
        large = flag == 0b11
    else:
        assert data[0] & 0x80, "not compressed"
        if data[0] & 0x40:
            return None
        large = bool(data[0] & 0x20)
        x = int.from_bytes(bytes([data[0] & 0x1F]) + data[1:], "big")
    y = _sqrt_fp((x * x * x + c.b) % c.p, c.p)
    assert y is not None, "x not on curve"
    if _lex_larger_fp(y, c.p) != large:
        y = c.p - y
    return (x, y)


def g1_compress(c: Curve, P) -> bytes:
    n = c.fp_bytes
    if c.name == "bn254":
        if P is None:
            return bytes([0x40]) + bytes(n - 1)
        x, y = P
        b = bytearray(x.to_bytes(n, "big"))
        b[0] |= 0xC0 if _lex_larger_fp(y, c.p) else 0x80
        return bytes(b)
    if P is None:
        return bytes([0xC0]) + bytes(n - 1)
    x, y = P
    b = bytearray(x.to_bytes(n, "big"))
    b[0] |= 0x80 | (0x20 if _lex_larger_fp(y, c.p) else 0)
    return bytes(b)


def g2_decompress(c: Curve, data: bytes):
    """A1 || A0, big-endian, flags in the first byte as for G1."""
    n = c.fp_bytes
    assert len(data) == 2 * n
    if c.name == "bn254":
        flag = data[0] >> 6
        if flag == 0b01:
            return None
        large = flag == 0b11
        x1 = int.from_bytes(bytes([data[0] & 0x3F]) + data[1:n], "big")
    else:
        assert data[0] & 0x80
        if data[0] & 0x40:
            return None
        large = bool(data[0] & 0x20)
        x1 = int.from_bytes(bytes([data[0] & 0x1F]) + data[1:n], "big")
    x0 = int.from_bytes(data[n:], "big")
    F = Fp2Ops(c.p)
    x = (x0, x1)
    y = _sqrt_fp2(F.add(F.mul(F.mul(x, x), x), c.b2), c.p)
    assert y is not None
    if _lex_larger_fp2(y, c.p) != large:
        y = F.neg(y)
    return (x, y)


def g2_compress(c: Curve, P) -> bytes:
    n = c.fp_bytes
    if P is None:
        first = 0x40 if c.name == "bn254" else 0xC0
        return bytes([first]) + bytes(2 * n - 1)
    (x0, x1), y = P
    b = bytearray(x1.to_bytes(n, "big") + x0.to_bytes(n, "big"))
    large = _lex_larger_fp2(y, c.p)
    if c.name == "bn254":
        b[0] |= 0xC0 if large else 0x80
    else:
        b[0] |= 0x80 | (0x20 if large else 0)
    return bytes(b)


# ----------------------------------------------------------------------------------------------
# deterministic test-vector PRNG shared with the C side (xoshiro256** seeded by splitmix64)
# ----------------------------------------------------------------------------------------------

_M64 = (1 << 64) - 1


class Xoshiro:
    def __init__(self, seed: int):
        s = seed & _M64
        st = []
        for _ in range(4):
            s = (s + 0x9E3779B97F4A7C15) & _M64
            z = s
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
            st.append(z ^ (z >> 31))
        self.s = st

    @staticmethod
    def _rotl(x, k):
        return ((x << k) | (x >> (64 - k))) & _M64

    def next(self) -> int:
        s = self.s
        result = (self._rotl((s[1] * 5) & _M64, 7) * 9) & _M64
        t = (s[1] << 17) & _M64
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = self._rotl(s[3], 45)
        return result

    def field(self, mod: int) -> int:
        """uniform in [0, mod) by rejection on the top bits (4 limbs, little-endian draw order)."""
        bits = mod.bit_length()
        nl = (bits + 63) // 64
        while True:
            v = 0
            for i in range(nl):
                v |= self.next() << (64 * i)
            v &= (1 << bits) - 1
            if v < mod:
                return v


# ----------------------------------------------------------------------------------------------
# Groth16 on a hand-written R1CS (tiny circuits) -- restates setup.go:75-331 and prove.go:52-315
# ----------------------------------------------------------------------------------------------


@dataclass
class R1CS:
    """L,R,O: list (per constraint) of dict wire->coeff.  Wire order: [1, public..., secret..., internal...]
    (constraint/bn254/solver.go:82-88)."""
    nb_public: int       # includes the constant-one wire (prove.go:235 / GetNbPublicVariables)
    nb_wires: int
    L: list
    R: list
    O: list
    commitments: list = field(default_factory=list)   # constraint.Groth16Commitments (constraint/commitment.go:9-14)


@dataclass
class Commitment:
    """constraint/commitment.go:9-14 Groth16Commitment"""
    public_and_commitment_committed: list   # sorted wire ids (public wires first, then earlier commitment wires)
    private_committed: list                 # sorted wire ids
    commitment_index: int                   # wire that receives the hash of the commitment
    nb_public_committed: int


def cubic_r1cs() -> R1CS:
    """examples/cubic/cubic.go:21-25 (x^3 + x + 5 == y) as gnark's r1cs builder emits it
    (SURVEY 8d config 1): wires {0:1, 1:y(public), 2:x(secret), 3:v1=x*x, 4:v2=v1*x}."""
    L = [{2: 1}, {3: 1}, {0: 1}]
    R = [{2: 1}, {2: 1}, {1: 1}]
    O = [{3: 1}, {4: 1}, {4: 1, 2: 1, 0: 5}]
    # third constraint: AssertIsEqual is encoded 1 * y == v2 + x + 5 (frontend/cs/r1cs/api_assertions.go:27-31);
    # no L/R swap applies (builder.go:190-195 only swaps when R has more terms than L).
    # => InfinityA = {y, v2}, InfinityB = {1, v1, v2}
    return R1CS(nb_public=2, nb_wires=5, L=L, R=R, O=O)


def cubic_witness(x=3):
    y = x ** 3 + x + 5
    return [1, y, x, x * x, x * x * x]


def commit_r1cs() -> R1CS:
    """A hand-written circuit with two api.Commit calls, shaped like the ones test/commitments_test.go builds: wires
    {0:1, 1:p (public), 2:x, 3:y (secret), 4:v=x*y, 5:cm0, 6:t=cm0*x, 7:cm1, 8:u=cm1*v}; commitment 0 commits to the private
    wires {x, y} and the public wire p, commitment 1 to the private wire v and to commitment 0's wire.  Commitment wires are
    hint outputs, i.e. internal wires for the solver and public inputs for the Groth16 setup (setup.go:94-98)."""
    L = [{2: 1}, {4: 1}, {5: 1}, {7: 1}]
    R = [{3: 1}, {0: 1}, {2: 1}, {4: 1}]
    O = [{4: 1}, {1: 1}, {6: 1}, {8: 1}]
    cms = [Commitment([1], [2, 3], 5, 1), Commitment([5], [4], 7, 0)]
    return R1CS(nb_public=2, nb_wires=9, L=L, R=R, O=O, commitments=cms)


def commit_solve(c: Curve, cs: R1CS, x: int, y: int, hint) -> list:
    """Solve commit_r1cs in wire order; hint(i, w) returns the value of commitment i's wire (the bsb22 hint, prove.go:72-100)."""
    w = [0] * cs.nb_wires
    w[0], w[2], w[3] = 1, x % c.r, y % c.r
    w[4] = w[2] * w[3] % c.r
    w[1] = w[4]
    w[5] = hint(0, w)
    w[6] = w[5] * w[2] % c.r
    w[7] = hint(1, w)
    w[8] = w[7] * w[4] % c.r
    return w


def r1cs_solve(c: Curve, cs: R1CS, w):
    """A,B,C = <L,w>, <R,w>, <O,w> per constraint (constraint/bn254/solver.go:1085-1171)."""
    ev = lambda row: sum(k * w[i] for i, k in row.items()) % c.r
    A = [ev(r) for r in cs.L]
    B = [ev(r) for r in cs.R]
    C = [ev(r) for r in cs.O]
    for a, b, cc in zip(A, B, C):
        assert a * b % c.r == cc, "witness does not satisfy the R1CS"
    return A, B, C


@dataclass
class ProvingKey:
    """setup.go:25-48 (values are affine python points; None = infinity)."""
    curve: Curve
    n: int
    alpha1: tuple
    beta1: tuple
    delta1: tuple
    A: list
    B: list
    Z: list
    K: list
    beta2: tuple
    delta2: tuple
    B2: list
    infinityA: list
    infinityB: list
    commitment_keys: list = field(default_factory=list)   # [(Basis, BasisExpSigma)] pedersen.ProvingKey (setup.go:276-287)


@dataclass
class VerifyingKey:
    alpha1: tuple
    beta2: tuple
    gamma2: tuple
    delta2: tuple
    K: list
    commitment_g2: tuple = None                 # pedersen.VerifyingKey.G (shared, setup.go:272-281)
    commitment_g2_sigma_neg: list = field(default_factory=list)   # [-sigma_i]G per commitment key


def groth16_setup(c: Curve, cs: R1CS, toxic):
    """setup.go:75-331 with injected toxic waste (alpha,beta,gamma,delta,tau)."""
    alpha, beta, gamma, delta, tau = toxic[:5]   # then one sigma per commitment, then the dlog of the pedersen G2 point
    mod = c.r
    m = len(cs.L)
    n = 1
    while n < m:
        n *= 2
    w = c.fr_root_of_unity(n)
    # Lagrange basis at tau over the domain (setupABC, setup.go:346-428)
    tn1 = (pow(tau, n, mod) - 1) % mod
    ninv = pow(n, -1, mod)
    lag = [tn1 * pow(w, i, mod) % mod * ninv % mod * pow((tau - pow(w, i, mod)) % mod, -1, mod) % mod
           for i in range(n)]
    nw = cs.nb_wires
    Av, Bv, Cv = [0] * nw, [0] * nw, [0] * nw
    for i in range(m):
        for wi, k in cs.L[i].items():
            Av[wi] = (Av[wi] + k * lag[i]) % mod
        for wi, k in cs.R[i].items():
            Bv[wi] = (Bv[wi] + k * lag[i]) % mod
        for wi, k in cs.O[i].items():
            Cv[wi] = (Cv[wi] + k * lag[i]) % mod
    G1, G2 = g1_group(c), g2_group(c)
    g1, g2 = c.g1, c.g2
    dinv, ginv = pow(delta, -1, mod), pow(gamma, -1, mod)
    # K scalars (setup.go:133-178): public wires and commitment wires / gamma -> vk.K; private committed wires / gamma ->
    # the commitment bases ckK[i]; every other private wire / delta -> pk.K
    kk = [(beta * Av[i] + alpha * Bv[i] + Cv[i]) % mod for i in range(nw)]
    com_wires = {cm.commitment_index for cm in cs.commitments}
    owner = {wi: ci for ci, cm in enumerate(cs.commitments) for wi in cm.private_committed}
    vk_s, pk_s, ck_s = [], [], [[] for _ in cs.commitments]
    for i in range(nw):
        if i < cs.nb_public or i in com_wires:
            vk_s.append(kk[i] * ginv % mod)
        elif i in owner:
            ck_s[owner[i]].append(kk[i] * ginv % mod)
        else:
            pk_s.append(kk[i] * dinv % mod)
    vkK = [G1.mul(g1, k) for k in vk_s]
    pkK = [G1.mul(g1, k) for k in pk_s]
    # pedersen.Setup per commitment (setup.go:280-287; gnark-crypto [EXT]): BasisExpSigma = [sigma]Basis, vk.GSigmaNeg = [-sigma]G
    sigmas = list(toxic[5:5 + len(cs.commitments)]) if len(toxic) > 5 else []
    assert len(sigmas) == len(cs.commitments), "one sigma per commitment must be injected after (alpha,beta,gamma,delta,tau)"
    cg2_dlog = toxic[5 + len(cs.commitments)] if len(toxic) > 5 + len(cs.commitments) else 1
    cks = []
    for ci in range(len(cs.commitments)):
        basis = [G1.mul(g1, k) for k in ck_s[ci]]
        cks.append((basis, [G1.mul(P, sigmas[ci]) for P in basis]))
    # Z scalars tau^i (tau^n - 1)/delta (setup.go:181-192), stored bit-reversed (:247), n-1 kept (:248-249)
    zs = [pow(tau, i, mod) * tn1 % mod * dinv % mod for i in range(n)]
    Zp = bitrev_permute([G1.mul(g1, z) for z in zs])[: n - 1]
    infA = [a == 0 for a in Av]
    infB = [b == 0 for b in Bv]
    pk = ProvingKey(
        curve=c, n=n,
        alpha1=G1.mul(g1, alpha), beta1=G1.mul(g1, beta), delta1=G1.mul(g1, delta),
        A=[G1.mul(g1, a) for a in Av if a != 0],
        B=[G1.mul(g1, b) for b in Bv if b != 0],
        Z=Zp, K=pkK,
        beta2=G2.mul(g2, beta), delta2=G2.mul(g2, delta),
        B2=[G2.mul(g2, b) for b in Bv if b != 0],
        infinityA=infA, infinityB=infB, commitment_keys=cks)
    cg2 = G2.mul(g2, cg2_dlog)
    vk = VerifyingKey(alpha1=pk.alpha1, beta2=pk.beta2, gamma2=G2.mul(g2, gamma), delta2=pk.delta2, K=vkK,
                      commitment_g2=cg2, commitment_g2_sigma_neg=[G2.mul(cg2, (-sg) % mod) for sg in sigmas])
    dlog = dict(A=[a for a in Av if a], B=[b for b in Bv if b], Z=bitrev_permute(zs)[: n - 1],
                K=pk_s, CK=ck_s, kk=kk, alpha=alpha, beta=beta, delta=delta, sigmas=sigmas)
    return pk, vk, dlog


# ---- BSB22 commitments: hash-to-field and the solver-side hint (prove.go:60-127) ------------------------------------------
COMMITMENT_DST = b"bsb22-commitment"   # constraint/commitment.go:7
FOLD_DST = b"G16-BSB22"                # prove.go:123


def expand_message_xmd(msg: bytes, dst: bytes, n: int) -> bytes:
    """RFC 9380 5.3.1 with SHA-256 (gnark-crypto field/hash.ExpandMsgXmd [EXT]); pinned by the vectors of
    std/hash/expand/expand_test.go:52-140."""
    ell = (n + 31) // 32
    assert ell <= 255 and len(dst) <= 255
    dst_prime = dst + bytes([len(dst)])
    b0 = hashlib.sha256(bytes(64) + msg + n.to_bytes(2, "big") + b"\x00" + dst_prime).digest()
    out, bi = b"", b""
    for i in range(1, ell + 1):
        x = b0 if i == 1 else bytes(a ^ b for a, b in zip(b0, bi))
        bi = hashlib.sha256(x + bytes([i]) + dst_prime).digest()
        out += bi
    return out[:n]


def fr_hash(c: Curve, msg: bytes, dst: bytes, count: int = 1) -> list:
    """fr.Hash (same template as internal/smallfields/tinyfield/element.go:456-481): L = 16 + fr.Bytes bytes per element,
    big-endian, reduced mod r."""
    L = 16 + (c.r.bit_length() + 7) // 8
    b = expand_message_xmd(msg, dst, count * L)
    return [int.from_bytes(b[i * L:(i + 1) * L], "big") % c.r for i in range(count)]


def g1_marshal_uncompressed(c: Curve, P) -> bytes:
    """G1Affine.Marshal() [EXT]: big-endian x | y; infinity = flag byte 0x40 then zeros."""
    nb = c.fp_bytes
    if P is None:
        return bytes([0x40]) + bytes(2 * nb - 1)
    return P[0].to_bytes(nb, "big") + P[1].to_bytes(nb, "big")


def commitment_hint(pk: ProvingKey, cs: R1CS, i: int, w) -> tuple:
    """The bsb22 hint override of prove.go:72-100 for commitment i on the partially solved wire vector w: returns
    (commitment point, value of the commitment wire)."""
    c = pk.curve
    cm = cs.commitments[i]
    G1 = g1_group(c)
    com = G1.msm(pk.commitment_keys[i][0], [w[j] for j in cm.private_committed])
    fbytes = (c.r.bit_length() - 1) // 8 + 1
    msg = g1_marshal_uncompressed(c, com) + b"".join(int(w[j]).to_bytes(fbytes, "big") for j in cm.public_and_commitment_committed)
    return com, fr_hash(c, msg, COMMITMENT_DST, 1)[0]


def groth16_prove(pk: ProvingKey, cs: R1CS, w, r, s):
    """prove.go:52-315 with injected (r, s).  Returns affine (Ar, Bs, Krs); with commitments use groth16_prove_bsb22."""
    c = pk.curve
    mod = c.r
    G1, G2 = g1_group(c), g2_group(c)
    A, B, C = r1cs_solve(c, cs, w)
    h = compute_h(c, A, B, C, pk.n)
    wA = [w[i] for i in range(len(w)) if not pk.infinityA[i]]
    wB = [w[i] for i in range(len(w)) if not pk.infinityB[i]]
    kr = (-(r * s)) % mod
    deltas = [G1.mul(pk.delta1, k) for k in (r, s, kr)]
    bs1 = G1.add(G1.add(G1.msm(pk.B, wB), pk.beta1), deltas[1])
    ar = G1.add(G1.add(G1.msm(pk.A, wA), pk.alpha1), deltas[0])
    krs2 = G1.msm(pk.Z, h[: pk.n - 1])
    removed = {j for cm in cs.commitments for j in cm.private_committed} | {cm.commitment_index for cm in cs.commitments}
    krs = G1.msm(pk.K, [w[i] for i in range(cs.nb_public, len(w)) if i not in removed])   # filterHeap, prove.go:231-235
    krs = G1.add(krs, deltas[2])
    krs = G1.add(krs, krs2)
    krs = G1.add(krs, G1.mul(ar, s))
    krs = G1.add(krs, G1.mul(bs1, r))
    bs = G2.add(G2.add(G2.msm(pk.B2, wB), G2.mul(pk.delta2, s)), pk.beta2)
    return ar, bs, krs


def groth16_prove_bsb22(pk: ProvingKey, cs: R1CS, w, r, s):
    """prove.go:52-315 for a circuit with commitments; w is the full solution (commitment wires already set by
    commitment_hint).  Returns (Ar, Bs, Krs, commitments, folded PoK)."""
    c = pk.curve
    G1 = g1_group(c)
    coms, poks = [], []
    for i, cm in enumerate(cs.commitments):
        vals = [w[j] for j in cm.private_committed]
        com, hv = commitment_hint(pk, cs, i, w)
        assert hv == w[cm.commitment_index], "commitment wire does not hold the hash of the commitment"
        coms.append(com)
        poks.append(G1.msm(pk.commitment_keys[i][1], vals))                     # ProveKnowledge, prove.go:112-117
    ser = b"".join(int(w[cm.commitment_index]).to_bytes(32, "big") for cm in cs.commitments)   # prove.go:119-122
    pok = None
    if cs.commitments:
        ch = fr_hash(c, ser, FOLD_DST, 1)[0]
        pok = G1.msm(poks, [pow(ch, i, c.r) for i in range(len(poks))])      # Fold, prove.go:127
    ar, bs, krs = groth16_prove(pk, cs, w, r, s)
    return ar, bs, krs, coms, pok


def proof_bytes(c: Curve, ar, bs, krs, commitments=(), pok=None) -> bytes:
    """Proof.WriteTo, marshal.go:33-58: Ar | Bs | Krs | u32 len(commitments) | commitments | CommitmentPok."""
    return (g1_compress(c, ar) + g2_compress(c, bs) + g1_compress(c, krs) + len(commitments).to_bytes(4, "big")
            + b"".join(g1_compress(c, P) for P in commitments) + g1_compress(c, pok))


def g2_marshal_uncompressed(c: Curve, P) -> bytes:
    """G2Affine.RawBytes [EXT]: x.A1 | x.A0 | y.A1 | y.A0 big-endian; infinity = 0x40 then zeros."""
    nb = c.fp_bytes
    if P is None:
        return bytes([0x40]) + bytes(4 * nb - 1)
    (x0, x1), (y0, y1) = P
    return b"".join(v.to_bytes(nb, "big") for v in (x1, x0, y1, y0))


def proof_bytes_raw(c: Curve, ar, bs, krs, commitments=(), pok=None) -> bytes:
    """Proof.WriteRawTo, marshal.go:25-30: the WriteTo layout with uncompressed points."""
    return (g1_marshal_uncompressed(c, ar) + g2_marshal_uncompressed(c, bs) + g1_marshal_uncompressed(c, krs)
            + len(commitments).to_bytes(4, "big") + b"".join(g1_marshal_uncompressed(c, P) for P in commitments)
            + g1_marshal_uncompressed(c, pok))


def sha_tag(*parts) -> str:
    h = hashlib.sha256()
    for p in parts:
        h.update(p if isinstance(p, bytes) else repr(p).encode())
    return h.hexdigest()


# ----------------------------------------------------------------------------------------------
# pairing and the Groth16 verifier (backend/groth16/bn254/verify.go:38-145), so that proof BYTES can be checked the way the
# reference checks them -- by Verify -- instead of through known toxic waste.  The pairing itself lives in gnark-crypto [EXT]:
# restated here as the ate pairing a_T(Q, P) = f_{T,Q}(P)^((p^12-1)/r) with T = t - 1 (= 6x^2 for BN254, = x for BLS12-381;
# Hess-Smart-Vercauteren).  It is not gnark-crypto's optimal-ate VALUE (BN254's differs by a fixed power), but any
# non-degenerate bilinear pairing decides the verifier's product-of-pairings equation identically, and that predicate is what
# the 12 tuples of backend/groth16/bellman_test.go:26-84 pin (tests/test_oracle_fixtures.py).
# Fp12 = Fp[w]/(w^12 - 2*a*w^6 + (a^2+1)) with xi = a + u = w^6 (a = 9 BN254, 1 BLS12-381), coefficient lists of length 12.
# ----------------------------------------------------------------------------------------------
_PAIRING = {
    "bn254": dict(a=9, T=6 * 4965661367192848881 ** 2, mtwist=False),   # E'/Fp2: y^2 = x^3 + 3/xi   (D-type): psi(x,y) = (x w^2, y w^3)
    "bls12-381": dict(a=1, T=0xD201000000010000, mtwist=True),          # E'/Fp2: y^2 = x^3 + 4 xi   (M-type): psi(x,y) = (x/w^2, y/w^3); T = |x|
}


class Fp12Ops:
    def __init__(self, c: Curve):
        self.p = c.p
        self.a = _PAIRING[c.name]["a"]
        self.c6 = 2 * self.a                    # w^12 = c6*w^6 - c0
        self.c0 = self.a * self.a + 1
        self.one = [1] + [0] * 11

    def mul(self, x, y):
        p = self.p
        t = [0] * 23
        for i, xi in enumerate(x):
            if xi:
                for j, yj in enumerate(y):
                    if yj:
                        t[i + j] += xi * yj
        for k in range(22, 11, -1):             # w^k = c6*w^(k-6) - c0*w^(k-12)
            v = t[k] % p
            if v:
                t[k - 6] += self.c6 * v
                t[k - 12] -= self.c0 * v
        return [v % p for v in t[:12]]

    def conj(self, x):                          # x^(p^6): w -> -w
        return [(-v) % self.p if i & 1 else v for i, v in enumerate(x)]

    def pow(self, x, e):
        r = self.one
        while e:
            if e & 1:
                r = self.mul(r, x)
            x = self.mul(x, x)
            e >>= 1
        return r

    def inv(self, x):
        """x * conj(x) lies in Fp6 = Fp[w^2]; invert there by solving a 6x6 linear system, then x^-1 = conj(x) * (x conj x)^-1"""
        p = self.p
        n = self.mul(x, self.conj(x))           # even powers only
        assert all(n[i] == 0 for i in range(1, 12, 2))
        # multiplication-by-n matrix on the basis w^0, w^2, ..., w^10
        cols = []
        for k in range(6):
            e = [0] * 12
            e[2 * k] = 1
            cols.append(self.mul(n, e)[0::2])
        M = [[cols[k][r] for k in range(6)] + [1 if r == 0 else 0] for r in range(6)]
        for i in range(6):                      # Gauss-Jordan mod p
            piv = next(r for r in range(i, 6) if M[r][i] % p)
            M[i], M[piv] = M[piv], M[i]
            iv = pow(M[i][i], -1, p)
            M[i] = [v * iv % p for v in M[i]]
            for r in range(6):
                if r != i and M[r][i]:
                    f = M[r][i]
                    M[r] = [(vr - f * vi) % p for vr, vi in zip(M[r], M[i])]
        ninv = [0] * 12
        for k in range(6):
            ninv[2 * k] = M[k][6]
        return self.mul(self.conj(x), ninv)

    def embed_fp2(self, v, k):
        """v = v0 + v1*u (u = w^6 - a) times w^k as a sparse coefficient list"""
        out = [0] * 12
        out[k] = (v[0] - self.a * v[1]) % self.p
        out[k + 6] = v[1] % self.p
        return out


def miller_loop(c: Curve, P, Q):
    """f_{T,Q}(P) for P in G1 (affine Fp), Q in G2 (affine over Fp2, on the twist); infinity on either side gives 1.
    The running point stays on the twist; a line through twist points with slope m untwists to (scaled by an element of a proper
    subfield, which the final exponentiation kills):  D-type: -yP + (m xP) w + (yR - m xR) w^3;  M-type: (yR - m xR) + (m xP) w^2 - yP w^3."""
    F12, F2 = Fp12Ops(c), Fp2Ops(c.p)
    cfg = _PAIRING[c.name]
    f = F12.one
    if P is None or Q is None:
        return f
    xP, yP = P
    G2 = g2_group(c)

    def line(R, S):
        if R[0] == S[0] and R[1] == S[1]:
            m = F2.mul(F2.muli(F2.mul(R[0], R[0]), 3), F2.inv(F2.muli(R[1], 2)))
        else:
            m = F2.mul(F2.sub(S[1], R[1]), F2.inv(F2.sub(S[0], R[0])))
        a1 = F2.muli(m, xP)
        a0 = F2.sub(R[1], F2.mul(m, R[0]))
        if cfg["mtwist"]:
            l = F12.embed_fp2(a0, 0)
            for i, v in enumerate(F12.embed_fp2(a1, 2)):
                l[i] = (l[i] + v) % c.p
            l[3] = (l[3] - yP) % c.p
        else:
            l = F12.embed_fp2(a1, 1)
            for i, v in enumerate(F12.embed_fp2(a0, 3)):
                l[i] = (l[i] + v) % c.p
            l[0] = (l[0] - yP) % c.p
        return l

    R = Q
    T = cfg["T"]
    for i in range(T.bit_length() - 2, -1, -1):
        f = F12.mul(F12.mul(f, f), line(R, R))
        R = G2.add(R, R)
        if (T >> i) & 1:
            if R[0] == Q[0] and R[1] != Q[1]:   # R = -Q: vertical line (an element of a proper subfield), R becomes infinity -- last step only
                R = None
                break
            f = F12.mul(f, line(R, Q))
            R = G2.add(R, Q)
    return f


def final_exponentiation(c: Curve, f):
    F12 = Fp12Ops(c)
    g = F12.mul(F12.conj(f), F12.inv(f))                      # ^(p^6 - 1)
    e = (c.p ** 2 + 1) * ((c.p ** 4 - c.p ** 2 + 1) // c.r)
    assert (c.p ** 4 - c.p ** 2 + 1) % c.r == 0
    return F12.pow(g, e)


def pairing(c: Curve, P, Q):
    return final_exponentiation(c, miller_loop(c, P, Q))


def pairing_check(c: Curve, pairs) -> bool:
    """prod e(P_i, Q_i) == 1 with one final exponentiation (curve.PairingCheck [EXT])"""
    F12 = Fp12Ops(c)
    f = F12.one
    for P, Q in pairs:
        f = F12.mul(f, miller_loop(c, P, Q))
    return final_exponentiation(c, f) == F12.one


def groth16_verify(c: Curve, vk: "VerifyingKey", proof, public_witness, public_and_commitment_committed=()) -> bool:
    """verify.go:38-145.  proof = (Ar, Bs, Krs, commitments, pok) affine points (commitments / pok may be () / None);
    public_witness excludes the constant-one wire; public_and_commitment_committed = vk.PublicAndCommitmentCommitted.
    Returns False where the reference returns an error."""
    ar, bs, krs = proof[0], proof[1], proof[2]
    coms = list(proof[3]) if len(proof) > 3 and proof[3] else []
    pok = proof[4] if len(proof) > 4 else None
    G1, G2 = g1_group(c), g2_group(c)
    pacc = [list(x) for x in public_and_commitment_committed]
    nb_public = len(vk.K) - len(pacc)
    if len(vk.commitment_g2_sigma_neg) != len(pacc) or len(coms) != len(pacc) or len(public_witness) != nb_public - 1:   # :46-58
        return False
    # proof.isValid (:63-65): on the curve and in the prime-order subgroups
    for P in [ar, krs] + coms + ([pok] if coms else []):
        if not G1.on_curve(P) or G1.mul(P, c.r) is not None:
            return False
    if not G2.on_curve(bs) or G2.mul(bs, c.r) is not None:
        return False
    pw = [int(x) % c.r for x in public_witness]
    fbytes = (c.r.bit_length() - 1) // 8 + 1
    ser = b""
    for i, idx in enumerate(pacc):             # solveCommitmentWire (:86-103)
        msg = g1_marshal_uncompressed(c, coms[i]) + b"".join(pw[j - 1].to_bytes(fbytes, "big") for j in idx)
        h = fr_hash(c, msg, COMMITMENT_DST, 1)[0]
        pw.append(h)
        ser += h.to_bytes(fbytes, "big")
    if pacc:                                   # pedersen.BatchVerifyMultiVk (:104-112) [EXT]: prod e(ch^i C_i, [-sigma_i]G) e(pok, G) == 1
        ch = fr_hash(c, ser, FOLD_DST, 1)[0]
        pairs = [(G1.mul(C, pow(ch, i, c.r)), vk.commitment_g2_sigma_neg[i]) for i, C in enumerate(coms)] + [(pok, vk.commitment_g2)]
        if not pairing_check(c, pairs):
            return False
    ksum = G1.add(G1.msm(vk.K[1:], pw), vk.K[0])            # :115-119
    for C in coms:
        ksum = G1.add(ksum, C)                               # :121-123
    # e(Krs, -delta) e(Ar, Bs) e(kSum, -gamma) == e(alpha, beta)   (:70-73,128-139)
    return pairing_check(c, [(krs, G2.neg(vk.delta2)), (ar, bs), (ksum, G2.neg(vk.gamma2)), (G1.neg(vk.alpha1), vk.beta2)])


def vk_public_and_commitment_committed(cs: "R1CS") -> list:
    """vk.PublicAndCommitmentCommitted as Setup fills it (setup.go:289, constraint/commitment.go:52-75): public wire ids stay,
    the wire id of an earlier commitment becomes nbPublic + its position among the commitment wires -- i.e. its index in the
    verifier's public-witness vector (public inputs, then the commitment hashes in order; verify.go:86-103 reads [idx - 1])."""
    com_wires = [cm.commitment_index for cm in cs.commitments]
    out = []
    for cm in cs.commitments:
        row = list(cm.public_and_commitment_committed[: cm.nb_public_committed])
        row += [cs.nb_public + com_wires.index(j) for j in cm.public_and_commitment_committed[cm.nb_public_committed:]]
        out.append(row)
    return out


def vk_read(c: Curve, data: bytes):
    """VerifyingKey.ReadFrom, marshal.go:151-230 (compressed points): [alpha]1 [beta]1 [beta]2 [gamma]2 [delta]1 [delta]2,
    u32 len + [K]1, then -- when present -- PublicAndCommitmentCommitted ([][]uint64: u32 count, per row u32 len + u64s),
    u32 nbCommitments and per commitment a pedersen.VerifyingKey (G, GSigmaNeg in G2) [EXT].
    Returns (VerifyingKey, public_and_commitment_committed, beta1, delta1, bytes consumed)."""
    nb = c.fp_bytes
    off = 0

    def g1():
        nonlocal off
        P = g1_decompress(c, data[off:off + nb])
        off += nb
        return P

    def g2():
        nonlocal off
        Q = g2_decompress(c, data[off:off + 2 * nb])
        off += 2 * nb
        return Q

    def u32():
        nonlocal off
        v = int.from_bytes(data[off:off + 4], "big")
        off += 4
        return v
    alpha1, beta1, beta2, gamma2, delta1, delta2 = g1(), g1(), g2(), g2(), g1(), g2()
    K = [g1() for _ in range(u32())]
    pacc, cg2, sneg = [], None, []
    if off < len(data):
        for _ in range(u32()):
            row = []
            for _ in range(u32()):
                row.append(int.from_bytes(data[off:off + 8], "big"))
                off += 8
            pacc.append(row)
        for _ in range(u32()):
            cg2 = g2()
            sneg.append(g2())
    # (like the Go decoder, reading stops here: the bellman_test.go keys carry zero padding behind the last field)
    return VerifyingKey(alpha1=alpha1, beta2=beta2, gamma2=gamma2, delta2=delta2, K=K, commitment_g2=cg2, commitment_g2_sigma_neg=sneg), pacc, beta1, delta1, off


# ----------------------------------------------------------------------------------------------
# proving-key wire formats (backend/groth16/bn254/marshal.go:231-539) and Proof.ReadFrom (:62-86)
# Framing owned by gnark-crypto v0.21.0 [EXT, restated from its published code, see gnark_amd/csrc/keyio.hip.h]:
# Encoder integers big-endian, []G1Affine = u32 BE length + points, []bool one byte per entry without a
# length, fft.Domain.WriteTo = cardinality + 5 fr elements (+ withPrecompute byte), unsafe.WriteSlice = u64 LE
# length + memory image, unsafe.WriteMarker = uint64(0xdeadbeef) in native byte order.
# ----------------------------------------------------------------------------------------------


def domain_bytes(c: Curve, n: int, with_precompute_byte: bool = True) -> bytes:
    """fft.Domain.WriteTo: Cardinality, CardinalityInv, Generator, GeneratorInv, FrMultiplicativeGen, FrMultiplicativeGenInv"""
    w = c.fr_root_of_unity(n)
    elems = [pow(n, -1, c.r), w, pow(w, -1, c.r), c.fr_gen, pow(c.fr_gen, -1, c.r)]
    out = n.to_bytes(8, "big") + b"".join(e.to_bytes(32, "big") for e in elems)
    return out + (b"\x01" if with_precompute_byte else b"")


def _enc_g1(c, P, raw):
    return g1_marshal_uncompressed(c, P) if raw else g1_compress(c, P)


def _enc_g2(c, P, raw):
    return g2_marshal_uncompressed(c, P) if raw else g2_compress(c, P)


def _enc_vec(c, pts, raw, g2=False):
    f = _enc_g2 if g2 else _enc_g1
    return len(pts).to_bytes(4, "big") + b"".join(f(c, P, raw) for P in pts)


def _key_tail(pk: "ProvingKey") -> bytes:
    nw = len(pk.infinityA)
    return (nw.to_bytes(8, "big") + sum(map(bool, pk.infinityA)).to_bytes(8, "big") + sum(map(bool, pk.infinityB)).to_bytes(8, "big")
            + bytes(1 if x else 0 for x in pk.infinityA) + bytes(1 if x else 0 for x in pk.infinityB)
            + len(pk.commitment_keys).to_bytes(4, "big"))


def pk_write(pk: "ProvingKey", raw: bool = False, with_precompute_byte: bool = True) -> bytes:
    """ProvingKey.WriteTo / WriteRawTo, marshal.go:243-299"""
    c = pk.curve
    out = domain_bytes(c, pk.n, with_precompute_byte)
    out += _enc_g1(c, pk.alpha1, raw) + _enc_g1(c, pk.beta1, raw) + _enc_g1(c, pk.delta1, raw)
    out += _enc_vec(c, pk.A, raw) + _enc_vec(c, pk.B, raw) + _enc_vec(c, pk.Z, raw) + _enc_vec(c, pk.K, raw)
    out += _enc_g2(c, pk.beta2, raw) + _enc_g2(c, pk.delta2, raw) + _enc_vec(c, pk.B2, raw, g2=True)
    out += _key_tail(pk)
    for basis, sigma in pk.commitment_keys:      # pedersen.ProvingKey.WriteTo / WriteRawTo: Basis, BasisExpSigma
        out += _enc_vec(c, basis, raw) + _enc_vec(c, sigma, raw)
    return out


def _mem_g1(c, P):
    if P is None:
        return bytes(2 * c.fp_bytes)
    return b"".join(int(l).to_bytes(8, "little") for v in P for l in to_mont_limbs(v, c.p, c.fp_limbs))


def _mem_g2(c, P):
    if P is None:
        return bytes(4 * c.fp_bytes)
    (x0, x1), (y0, y1) = P
    return b"".join(int(l).to_bytes(8, "little") for v in (x0, x1, y0, y1) for l in to_mont_limbs(v, c.p, c.fp_limbs))


def _dump_slice(c, pts, g2=False):
    f = _mem_g2 if g2 else _mem_g1
    return len(pts).to_bytes(8, "little") + b"".join(f(c, P) for P in pts)


def pk_write_dump(pk: "ProvingKey", with_precompute_byte: bool = True) -> bytes:
    """ProvingKey.WriteDump, marshal.go:378-445: marker, domain, header points (raw encoding), nbWires / infinity masks,
    then the slices as memory images (Montgomery limbs, little-endian: gnark-crypto's fp.Element layout on amd64)"""
    c = pk.curve
    out = (0xdeadbeef).to_bytes(8, "little") + domain_bytes(c, pk.n, with_precompute_byte)
    out += _enc_g1(c, pk.alpha1, True) + _enc_g1(c, pk.beta1, True) + _enc_g1(c, pk.delta1, True)
    out += _enc_g2(c, pk.beta2, True) + _enc_g2(c, pk.delta2, True)
    out += _key_tail(pk)
    out += _dump_slice(c, pk.A) + _dump_slice(c, pk.B) + _dump_slice(c, pk.Z) + _dump_slice(c, pk.K) + _dump_slice(c, pk.B2, g2=True)
    for basis, sigma in pk.commitment_keys:
        out += _dump_slice(c, basis) + _dump_slice(c, sigma)
    return out


def g1_unmarshal(c: Curve, data: bytes):
    """uncompressed G1 point: x | y big-endian, 0x40 = infinity"""
    n = c.fp_bytes
    if data[0] & 0x40 and not (data[0] & 0x80):
        return None
    return (int.from_bytes(data[:n], "big"), int.from_bytes(data[n:2 * n], "big"))


def g2_unmarshal(c: Curve, data: bytes):
    n = c.fp_bytes
    if data[0] & 0x40 and not (data[0] & 0x80):
        return None
    x1, x0, y1, y0 = (int.from_bytes(data[i * n:(i + 1) * n], "big") for i in range(4))
    return ((x0, x1), (y0, y1))


def proof_read(c: Curve, data: bytes):
    """Proof.ReadFrom, marshal.go:62-86: (ar, bs, krs, commitments, pok); points compressed or uncompressed by their flag bits"""
    n = c.fp_bytes
    raw = (data[0] >> 6) == 0 if c.name == "bn254" else not (data[0] & 0x80)
    s1, s2 = (2 * n, 4 * n) if raw else (n, 2 * n)
    g1 = (lambda b: g1_unmarshal(c, b)) if raw else (lambda b: g1_decompress(c, b))
    g2 = (lambda b: g2_unmarshal(c, b)) if raw else (lambda b: g2_decompress(c, b))
    off = 0
    ar = g1(data[off:off + s1]); off += s1
    bs = g2(data[off:off + s2]); off += s2
    krs = g1(data[off:off + s1]); off += s1
    k = int.from_bytes(data[off:off + 4], "big"); off += 4
    coms = []
    for _ in range(k):
        coms.append(g1(data[off:off + s1])); off += s1
    pok = g1(data[off:off + s1]); off += s1
    return ar, bs, krs, coms, pok, off
