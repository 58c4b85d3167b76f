"""The C oracle (oracle/oracle.c) against the independent big-integer restatement (oracle/pyref.py)."""
import numpy as np
import pytest

import oracle
import pyref
from helpers import BLS12_381, BN254, arr_to_fr, arr_to_g1_affine, arr_to_g2_affine, fr_to_arr, gen_of, group_of, jac_to_affine_py, pts_to_arr

CURVES = [BN254, BLS12_381]


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_field_and_generator(c):
    rng = pyref.Xoshiro(1)
    a = [rng.field(c.r) for _ in range(32)]
    b = [rng.field(c.r) for _ in range(32)]
    got = arr_to_fr(c, oracle.fr_mul(c.cid, fr_to_arr(c, a), fr_to_arr(c, b)))
    assert got == [x * y % c.r for x, y in zip(a, b)]
    assert oracle.fr_dot(c.cid, fr_to_arr(c, a), fr_to_arr(c, b, mont=False)) == sum(x * y for x, y in zip(a, b)) % c.r
    for group in (0, 1):
        k = rng.field(c.r)
        got = jac_to_affine_py(c, group, oracle.generator_mul(c.cid, group, k))
        assert got == group_of(c, group).mul(gen_of(c, group), k)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("group", [0, 1], ids=["G1", "G2"])
def test_msm_pippenger_vs_naive_vs_python(c, group):
    rng = pyref.Xoshiro(77 + group)
    n = 24
    G, g = group_of(c, group), gen_of(c, group)
    pts = [G.mul(g, rng.next() & 0xFFFFF) for _ in range(n)]
    sc = [rng.field(c.r) for _ in range(n)]
    sc[0], sc[1], sc[2] = 0, 1, c.r - 1
    pts[3] = None
    pts[5] = pts[4]
    P, S = pts_to_arr(c, group, pts), fr_to_arr(c, sc)
    want = G.msm(pts, sc)
    assert jac_to_affine_py(c, group, oracle.msm(c.cid, group, P, S)) == want
    assert jac_to_affine_py(c, group, oracle.msm(c.cid, group, P, S, nthreads=4)) == want
    assert jac_to_affine_py(c, group, oracle.msm(c.cid, group, P, S, naive=True)) == want


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_msm_fast_batch_affine_vs_python_and_simple(c):
    """msm_fast.c (bench.py's CPU baseline: signed digits, batch-affine buckets, window x range tasks) against the big-integer
    MSM, the simple Pippenger and the closed form [sum s_i k_i]G -- with every exceptional case of the AFFINE addition inside one
    bucket list: equal points (tangent), P and -P (nothing left), infinity among the bases, zero and maximal scalars, and lists
    that become empty in the middle of a chunk's tree."""
    import numpy as np
    rng = pyref.Xoshiro(91)
    G, g = group_of(c, 0), gen_of(c, 0)
    n = 40
    pts = [G.mul(g, rng.next() & 0xFFFFF) for _ in range(n)]
    sc = [rng.field(c.r) for _ in range(n)]
    sc[0], sc[1], sc[2] = 0, 1, c.r - 1
    pts[3] = None
    pts[5] = pts[4]
    sc[5] = sc[4]                       # the same point twice in the same bucket of every window: tangent
    pts[7] = G.neg(pts[6])
    sc[7] = sc[6]                       # P and -P in the same bucket of every window: they cancel
    pts[9] = pts[8]
    sc[9] = c.r - sc[8]                 # [s]P + [r - s]P = infinity through opposite digits
    for k in range(10, 18):
        pts[k], sc[k] = pts[10], 5      # eight copies in one bucket: three tangent rounds in a row
    P, S = pts_to_arr(c, 0, pts), fr_to_arr(c, sc)
    want = G.msm(pts, sc)
    assert jac_to_affine_py(c, 0, oracle.msm(c.cid, 0, P, S)) == want
    for force_c, force_splits, threads in ((0, 0, 1), (0, 0, 8), (3, 4, 8), (7, 1, 2), (16, 3, 3), (5, 40, 4)):
        got = oracle.msm_fast(c.cid, P, S, nthreads=threads, force_c=force_c, force_splits=force_splits)
        assert jac_to_affine_py(c, 0, got) == want, (force_c, force_splits, threads)
    # everything cancels / nothing to add
    assert jac_to_affine_py(c, 0, oracle.msm_fast(c.cid, P[6:8], S[6:8])) is None
    assert jac_to_affine_py(c, 0, oracle.msm_fast(c.cid, P[:0], S[:0])) is None
    # a few thousand points with known discrete logs: the closed form, canonical (non-Montgomery) scalars, ragged splits
    m = 3001
    ks = np.array([rng.next() >> 1 for _ in range(m)], dtype=np.uint64)
    Pb = oracle.gen_bases(c.cid, 0, ks)
    sc2 = [rng.field(c.r) for _ in range(m)]
    e = sum(int(k) * s for k, s in zip(ks, sc2)) % c.r
    want2 = G.mul(g, e)
    S2 = fr_to_arr(c, sc2)
    for threads, fs in ((1, 0), (5, 0), (8, 7)):
        assert jac_to_affine_py(c, 0, oracle.msm_fast(c.cid, Pb, S2, nthreads=threads, force_splits=fs)) == want2
    Sc = np.array([[(s >> (64 * i)) & (2**64 - 1) for i in range(4)] for s in sc2], dtype=np.uint64)
    assert jac_to_affine_py(c, 0, oracle.msm_fast(c.cid, Pb, Sc, mont=False, nthreads=4)) == want2
    # a boolean-like witness: 6000 operations on two buckets -- the conflict queue (4096) overflows into the Jacobian side sums
    m3 = 6000
    ks3 = np.array([rng.next() >> 1 for _ in range(m3)], dtype=np.uint64)
    Pb3 = oracle.gen_bases(c.cid, 0, ks3)
    sc3 = [1 + (i % 2) for i in range(m3)]
    want3 = G.mul(g, sum(int(k) * s for k, s in zip(ks3, sc3)) % c.r)
    for threads, fc in ((1, 0), (4, 16), (3, 5)):
        assert jac_to_affine_py(c, 0, oracle.msm_fast(c.cid, Pb3, fr_to_arr(c, sc3), nthreads=threads, force_c=fc)) == want3
    cbits, nwin, splits = oracle.msm_fast_plan(c.cid, 1 << 24, 64)
    assert 12 <= cbits <= 16 and nwin * splits >= 64      # more cores than windows: the point range is split so that all work


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_fft_conventions(c):
    rng = pyref.Xoshiro(3)
    for logn in (0, 1, 4, 7):
        n = 1 << logn
        a = [rng.field(c.r) for _ in range(n)]
        for dec in (pyref.DIF, pyref.DIT):
            for coset in (False, True):
                for inv in (False, True):
                    got = arr_to_fr(c, oracle.fft(c.cid, fr_to_arr(c, a), int(inv), dec, coset))
                    assert got == pyref.fft(c, a, dec, on_coset=coset, inverse=inv), (logn, dec, coset, inv)
    # O(n^2) DFT definition
    a = [rng.field(c.r) for _ in range(64)]
    w = c.fr_root_of_unity(64)
    nat = pyref.dft_naive(a, w, c.r)
    assert arr_to_fr(c, oracle.fft(c.cid, fr_to_arr(c, a), 0, pyref.DIF, False)) == pyref.bitrev_permute(nat)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_compute_h_identity(c):
    """A(x)B(x) - C(x) = H(x)(x^n - 1) at a random point, H from the C oracle (bit-reversed coefficients)."""
    rng = pyref.Xoshiro(9)
    m, n = 50, 64
    A = [rng.field(c.r) for _ in range(m)]
    B = [rng.field(c.r) for _ in range(m)]
    Cc = [x * y % c.r for x, y in zip(A, B)]
    h = arr_to_fr(c, oracle.compute_h(c.cid, fr_to_arr(c, A), fr_to_arr(c, B), fr_to_arr(c, Cc), n))
    assert h == pyref.compute_h(c, A, B, Cc, n)
    hc = pyref.bitrev_permute(h)   # natural coefficient order
    assert hc[n - 1] == 0          # deg H <= n-2 (setup.go:247-249)
    pad = lambda v: v + [0] * (n - m)
    coef = lambda ev: pyref.fft(c, pyref.fft(c, pad(ev), pyref.DIF, inverse=True), pyref.DIT, inverse=False) and \
        pyref.bitrev_permute(pyref.fft(c, pad(ev), pyref.DIF, inverse=True))
    x = rng.field(c.r)
    ev = lambda co: sum(v * pow(x, i, c.r) for i, v in enumerate(co)) % c.r
    lhs = (ev(coef(A)) * ev(coef(B)) - ev(coef(Cc))) % c.r
    assert lhs == ev(hc) * (pow(x, n, c.r) - 1) % c.r


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_groth16_cubic_matches_python_and_dlog(c):
    rng = pyref.Xoshiro(2024)
    cs, w = pyref.cubic_r1cs(), pyref.cubic_witness(3)
    toxic = [rng.field(c.r) for _ in range(5)]
    pk, vk, dl = pyref.groth16_setup(c, cs, toxic)
    r, s = rng.field(c.r), rng.field(c.r)
    ar, bs, krs = pyref.groth16_prove(pk, cs, w, r, s)
    A, B, Cc = pyref.r1cs_solve(c, cs, w)
    key = dict(n=pk.n, alpha1=pts_to_arr(c, 0, [pk.alpha1]), beta1=pts_to_arr(c, 0, [pk.beta1]), delta1=pts_to_arr(c, 0, [pk.delta1]),
               A=pts_to_arr(c, 0, pk.A), B=pts_to_arr(c, 0, pk.B), Z=pts_to_arr(c, 0, pk.Z), K=pts_to_arr(c, 0, pk.K),
               beta2=pts_to_arr(c, 1, [pk.beta2]), delta2=pts_to_arr(c, 1, [pk.delta2]), B2=pts_to_arr(c, 1, pk.B2),
               infinityA=pk.infinityA, infinityB=pk.infinityB)
    gar, gbs, gkrs = oracle.groth16_prove(c.cid, key, fr_to_arr(c, w), fr_to_arr(c, A), fr_to_arr(c, B), fr_to_arr(c, Cc),
                                          cs.nb_public, fr_to_arr(c, [r]), fr_to_arr(c, [s]))
    assert arr_to_g1_affine(c, gar) == ar
    assert arr_to_g2_affine(c, gbs) == bs
    assert arr_to_g1_affine(c, gkrs) == krs
    # closed form in the exponent (Groth16 equation): with a = alpha + A(tau) + r*delta etc. the proof verifies iff
    #   a*b = alpha*beta + (sum_pub w_i K_i) + krs*delta     (all as discrete logs)
    alpha, beta, gamma, delta, tau = toxic
    mod = c.r
    G1, G2 = group_of(c, 0), group_of(c, 1)
    a_dl = (alpha + sum(wv * k for wv, k in zip([w[i] for i in range(len(w)) if not pk.infinityA[i]], dl["A"])) + r * delta) % mod
    b_dl = (beta + sum(wv * k for wv, k in zip([w[i] for i in range(len(w)) if not pk.infinityB[i]], dl["B"])) + s * delta) % mod
    assert G1.mul(c.g1, a_dl) == ar and G2.mul(c.g2, b_dl) == bs
    h = pyref.compute_h(c, A, B, Cc, pk.n)
    krs_dl = (sum(wv * k for wv, k in zip(w[cs.nb_public:], dl["K"])) + sum(hv * z for hv, z in zip(h, dl["Z"]))
              + s * a_dl + r * b_dl - r * s * delta) % mod
    assert G1.mul(c.g1, krs_dl) == krs
    # pairing equation in the exponent: e(A,B) = e(alpha,beta) * e(sum_pub w_i vkK_i, gamma) * e(Krs, delta)
    n = pk.n
    wroot = c.fr_root_of_unity(n)
    tn1 = (pow(tau, n, mod) - 1) % mod
    lag = [tn1 * pow(wroot, i, mod) * pow(n, -1, mod) * pow((tau - pow(wroot, i, mod)) % mod, -1, mod) % mod for i in range(n)]
    Av, Bv, Cv = [0] * cs.nb_wires, [0] * cs.nb_wires, [0] * cs.nb_wires
    for i in range(len(cs.L)):
        for wi, k in cs.L[i].items(): Av[wi] = (Av[wi] + k * lag[i]) % mod
        for wi, k in cs.R[i].items(): Bv[wi] = (Bv[wi] + k * lag[i]) % mod
        for wi, k in cs.O[i].items(): Cv[wi] = (Cv[wi] + k * lag[i]) % mod
    pub = sum(w[i] * (beta * Av[i] + alpha * Bv[i] + Cv[i]) for i in range(cs.nb_public)) % mod
    assert a_dl * b_dl % mod == (alpha * beta + pub + krs_dl * delta) % mod


@pytest.mark.parametrize("c", [BN254, BLS12_381], ids=lambda c: c.name)
def test_c_oracle_plonk_quotient_matches_python(c):
    """oracle.c's computeNumerator + divideByZH (the checker used at 2^14 on the GPU) == pyref's restatement, coefficient for
    coefficient, on a satisfying trace and on random polynomials"""
    mod = c.r
    for n, nb, satisfying in ((8, 1, True), (64, 2, True), (32, 0, False)):
        lag, qcp, pi2, perm, rng = pyref.plonk_synthetic_instance(c, n, 17 + n, nb)
        beta, gamma, alpha = rng.field(mod), rng.field(mod), rng.field(mod)
        lag["Z"] = pyref.plonk_build_z(c, n, lag["L"], lag["R"], lag["O"], perm, beta, gamma)
        if not satisfying:
            lag = {k: [rng.field(mod) for _ in range(n)] for k in lag}
        w0, ninv = c.fr_root_of_unity(n), pow(n, -1, mod)
        can = lambda v: [x * ninv % mod for x in pyref._ntt_natural(v, pow(w0, -1, mod), mod)]
        x = {k: can(v) for k, v in lag.items()}
        qc, pi = [can(v) for v in qcp], [can(v) for v in pi2]
        bp = {k: [rng.field(mod) for _ in range(3 if k == "Bz" else 2)] for k in ("Bl", "Br", "Bo", "Bz")}
        want = pyref.plonk_quotient(c, n, x, qc, pi, bp, alpha, beta, gamma)
        polys = [fr_to_arr(c, x[k]) for k in pyref.PLONK_IDS]
        for a, b in zip(qc, pi):
            polys += [fr_to_arr(c, a), fr_to_arr(c, b)]
        got = oracle.plonk_quotient(c.cid, n, polys, fr_to_arr(c, bp["Bl"]), fr_to_arr(c, bp["Br"]), fr_to_arr(c, bp["Bo"]),
                                    fr_to_arr(c, bp["Bz"]), fr_to_arr(c, [alpha]), fr_to_arr(c, [beta]), fr_to_arr(c, [gamma]), nb)
        assert arr_to_fr(c, got) == want
