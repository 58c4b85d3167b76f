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
from dataclasses import dataclass

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


@dataclass
class VerifyingKey:
    alpha1: tuple
    beta2: tuple
    gamma2: tuple
    delta2: tuple
    K: list


def groth16_setup(c: Curve, cs: R1CS, toxic):
    """setup.go:75-331 with injected toxic waste (alpha,beta,gamma,delta,tau)."""
    alpha, beta, gamma, delta, tau = toxic
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
    # K scalars (setup.go:142-178): public part / gamma (vk), private part / delta (pk)
    kk = [(beta * Av[i] + alpha * Bv[i] + Cv[i]) % mod for i in range(nw)]
    vkK = [G1.mul(g1, kk[i] * ginv % mod) for i in range(cs.nb_public)]
    pkK = [G1.mul(g1, kk[i] * dinv % mod) for i in range(cs.nb_public, nw)]
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
        infinityA=infA, infinityB=infB)
    vk = VerifyingKey(alpha1=pk.alpha1, beta2=pk.beta2, gamma2=G2.mul(g2, gamma), delta2=pk.delta2, K=vkK)
    dlog = dict(A=[a for a in Av if a], B=[b for b in Bv if b], Z=bitrev_permute(zs)[: n - 1],
                K=[kk[i] * dinv % mod for i in range(cs.nb_public, nw)],
                alpha=alpha, beta=beta, delta=delta)
    return pk, vk, dlog


def groth16_prove(pk: ProvingKey, cs: R1CS, w, r, s):
    """prove.go:52-315 with injected (r, s), no commitments.  Returns affine (Ar, Bs, Krs)."""
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
    krs = G1.msm(pk.K, w[cs.nb_public:])
    krs = G1.add(krs, deltas[2])
    krs = G1.add(krs, krs2)
    krs = G1.add(krs, G1.mul(ar, s))
    krs = G1.add(krs, G1.mul(bs1, r))
    bs = G2.add(G2.add(G2.msm(pk.B2, wB), G2.mul(pk.delta2, s)), pk.beta2)
    return ar, bs, krs


def proof_bytes(c: Curve, ar, bs, krs) -> bytes:
    """Proof.WriteTo, marshal.go:33-58: Ar | Bs | Krs | u32 len(commitments)=0 | CommitmentPok (infinity)."""
    return (g1_compress(c, ar) + g2_compress(c, bs) + g1_compress(c, krs) + (0).to_bytes(4, "big")
            + g1_compress(c, None))


def sha_tag(*parts) -> str:
    h = hashlib.sha256()
    for p in parts:
        h.update(p if isinstance(p, bytes) else repr(p).encode())
    return h.hexdigest()
