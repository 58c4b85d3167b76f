"""BASELINE-size checkers (SURVEY 8c "oracle self-checks", 8d configs 3/4/5).  TEST INFRASTRUCTURE: imported by tests/ and by
bench.py's checker leg (outside every timed region) -- never by gnark_amd/.

The reference asserts prover correctness only through Verify and gnark cannot run here, so at sizes where the
oracle's own prover would take minutes the product is checked through closed forms the CPU evaluates in O(n):
  * Groth16: key of bases [k_i]G with known exponents (gnark_amd/synth.py) => the proof points are [e]G with e from field dot
    products (oracle.fr_dot); the formula itself is cross-checked against the oracle's prover at small sizes
    (tests/test_emu_kernels.py::test_emu_groth16_known_dlogs_checker).
  * computeH: A(x)B(x) - C(x) = H(x)(x^n - 1) at a point outside the domain, A, B, C evaluated from their values by the
    barycentric formula (oracle.fr_eval_lagrange) -- no transform on the checking side.
  * PLONK quotient: h(zeta)(zeta^n - 1) = gate + alpha*ordering + alpha^2 (Z-1) L1 on a satisfying synthetic trace.
"""
import numpy as np

import oracle
import pyref
from gnark_amd import fft, groth16, plonk


def fr_to_arr(c, vals):
    return np.array([pyref.to_mont_limbs(v, c.r, 4) for v in vals], dtype=np.uint64).reshape(-1, 4)


def arr_to_fr(c, arr):
    return [pyref.from_mont_limbs(row, c.r) for row in np.asarray(arr, dtype=np.uint64).reshape(-1, 4)]


def check_compute_h_identity(c, A, B, Cc, h_bitrev, n, nthreads=1, xv=0x1234567890ABCDEF1234567890ABCDEF0F1E2D3C4B5A6978):
    """size-independent property of computeH (prove.go:346-389): A(x)B(x) - C(x) == H(x)(x^n - 1) at a point outside the domain,
    with A, B, C evaluated from their VALUES by the barycentric formula (no transform involved) and H from its bit-reversed
    coefficients; deg H <= n - 2 (setup.go:247-249)."""
    xv %= c.r
    x = fr_to_arr(c, [xv])[0]
    pad = lambda v: v if v.shape[0] == n else np.concatenate([v, np.zeros((n - v.shape[0], 4), np.uint64)])
    ea, eb, ec = arr_to_fr(c, oracle.fr_eval_lagrange(c.cid, [pad(A), pad(B), pad(Cc)], x, nthreads))
    hx = pyref.from_mont_limbs(oracle.fr_eval_bitrev(c.cid, h_bitrev, x), c.r)
    assert not h_bitrev[n - 1].any()          # bitrev(n-1) = n-1: the coefficient of x^(n-1) is zero
    assert (ea * eb - ec) % c.r == hx * (pow(xv, n, c.r) - 1) % c.r


def check_groth16_known_dlogs(ctx, c, logn, nthreads=1, seed=0x5EED0005, also_oracle_prover=False, proofs=1, inst_kw=None, **pk_kw):
    """Groth16 Prove on the synthetic known-dlog instance: the three proof points must equal [ar]G1, [bs]G2, [krs]G1 with the
    exponents computed by the CPU oracle's dot products (gnark_amd/synth.py), and h must satisfy the polynomial identity."""
    from gnark_amd import synth
    inst = synth.make_instance(ctx, c.name, logn, seed, **(inst_kw or {}))   # inst_kw: e.g. inf_b = the wires that do not appear in B
    n = inst.n
    sol = inst.solution
    assert np.array_equal(sol.C[:64], oracle.fr_mul(c.cid, sol.A[:64], sol.B[:64]))   # ga_fr_vec_mul really multiplied
    d = fft.Domain(ctx, c.name, n)
    try:
        h = d.compute_h(sol.A, sol.B, sol.C)
    finally:
        d.close()
    check_compute_h_identity(c, sol.A, sol.B, sol.C, h, n, nthreads)
    exp = synth.expected_exponents(inst, h, lambda a, b: oracle.fr_dot(c.cid, a, b))
    pk = inst.proving_key(ctx, **pk_kw)
    try:
        part = groth16.ProvePartial(pk, sol, inst.nb_public)
        for _ in range(proofs):   # repeated proofs reuse the context's scratch: the last one is the one compared
            proof = groth16.Prove(pk, sol, inst.nb_public, inst.r, inst.s)
    finally:
        pk.FreeGPUResources()
    pt = lambda group, k: oracle.jac_to_affine(c.cid, group, oracle.generator_mul(c.cid, group, k))
    assert np.array_equal(proof.Ar, pt(0, exp["Ar"])), "Ar"
    assert np.array_equal(proof.Bs, pt(1, exp["Bs"])), "Bs"
    assert np.array_equal(proof.Krs, pt(0, exp["Krs"])), "Krs"
    # the four pre-randomisation sums (what a multi-GPU run all-gathers): A | B1 | K+Z | B2
    fp = c.fp_limbs
    cuts = [(0, 3 * fp, 0, "partial_A"), (3 * fp, 6 * fp, 0, "partial_B1"), (6 * fp, 9 * fp, 0, "partial_KZ"), (9 * fp, 15 * fp, 1, "partial_B2")]
    for lo, hi, grp, name in cuts:
        assert np.array_equal(oracle.jac_to_affine(c.cid, grp, part[lo:hi]), pt(grp, exp[name])), name
    if also_oracle_prover:   # small sizes: the C oracle's prover on the same key agrees with the dlog closed form
        key = dict(inst.key, n=n)
        want = oracle.groth16_prove(c.cid, key, sol.W, sol.A, sol.B, sol.C, inst.nb_public, inst.r, inst.s, nthreads=max(1, nthreads))
        assert np.array_equal(proof.Ar, want[0]) and np.array_equal(proof.Bs, want[1]) and np.array_equal(proof.Krs, want[2])
    return proof


# ---- PLONK quotient at BASELINE size (config 5, n = 2^22): vectorised satisfying trace + identity at a random point --------------
def plonk_synthetic_instance_np(ctx, c, n, seed, threads=1):
    """the numpy/C-oracle counterpart of pyref.plonk_synthetic_instance for sizes where Python integers are too slow: random
    wire values drawn from a pool (equal values form the copy-constraint cycles), random selectors, Qk chosen so that every gate
    holds, one BSB22 pair, S1..S3 = id(perm).  Everything is an (n, 4) fr image (Montgomery), Lagrange form, regular order."""
    from gnark_amd import synth
    cid = c.cid
    pool = synth._gen_scalars(ctx, cid, max(2, n // 2), seed)
    rng = np.random.default_rng(seed)
    assign = rng.integers(0, pool.shape[0], size=3 * n)
    wires = pool[assign]
    order = np.argsort(assign, kind="stable")                 # positions grouped by value
    sa = assign[order]
    first = np.r_[True, sa[1:] != sa[:-1]]                    # group starts
    start_idx = np.maximum.accumulate(np.where(first, np.arange(3 * n), 0))
    last = np.r_[first[1:], True]
    nxt = np.where(last, order[start_idx], np.r_[order[1:], order[:1]])   # successor inside the group, the last wraps to the first
    perm = np.empty(3 * n, dtype=np.int64)
    perm[order] = nxt
    lag = dict(L=wires[:n].copy(), R=wires[n:2 * n].copy(), O=wires[2 * n:].copy())
    for k, sd in (("Ql", 1), ("Qr", 2), ("Qm", 3), ("Qo", 4)):
        lag[k] = synth._gen_scalars(ctx, cid, n, seed + sd)
    qcp = synth._gen_scalars(ctx, cid, n, seed + 5)
    qcp[rng.random(n) < 0.75] = 0                              # Qcp is sparse in real circuits
    pi2 = synth._gen_scalars(ctx, cid, n, seed + 6)
    mul, add = (lambda a, b: oracle.fr_mul(cid, a, b)), (lambda a, b: oracle.fr_add(cid, a, b))
    tot = add(add(mul(lag["Ql"], lag["L"]), mul(lag["Qr"], lag["R"])), add(mul(mul(lag["Qm"], lag["L"]), lag["R"]), mul(lag["Qo"], lag["O"])))
    tot = add(tot, mul(qcp, pi2))
    lag["Qk"] = oracle.fr_sub(cid, np.zeros_like(tot), tot)
    w0, g = c.fr_root_of_unity(n), c.fr_gen
    ids = np.concatenate([oracle.fr_powers(cid, fr_to_arr(c, [w0])[0], fr_to_arr(c, [pow(g, k, c.r)])[0], n) for k in range(3)])
    for k, name in enumerate(("S1", "S2", "S3")):
        lag[name] = ids[perm[k * n:(k + 1) * n]]
    return lag, qcp, pi2, perm


def check_plonk_quotient_identity(ctx, c, logn, nthreads=1, seed=2024, pinned=False):
    """h(zeta) (zeta^n - 1) == gate + alpha*ordering + alpha^2 (Z-1) L1 at a random zeta (prove.go:950-981,1287-1350): Z from the
    device grand product (spot-checked against the recurrence and for closure), h from ga_plonk_quotient(_pinned), every
    polynomial evaluated on the CPU from its VALUES by the barycentric formula -- independent of the device transforms."""
    n = 1 << logn
    mod = c.r
    lag, qcp, pi2, perm = plonk_synthetic_instance_np(ctx, c, n, seed, nthreads)
    rng = pyref.Xoshiro(seed)
    beta, gamma, alpha, zeta = (rng.field(mod) for _ in range(4))
    d0, d1 = fft.Domain(ctx, c.name, n), fft.Domain(ctx, c.name, plonk.Rho(n) * n)
    try:
        Z = plonk.BuildRatioCopyConstraint(d0, lag["L"], lag["R"], lag["O"], perm, fr_to_arr(c, [beta]), fr_to_arr(c, [gamma]))
        lag["Z"] = Z
        w0, g = c.fr_root_of_unity(n), c.fr_gen
        ids = lambda pos: pow(g, int(pos) // n, mod) * pow(w0, int(pos) % n, mod) % mod
        val = lambda name, i: pyref.from_mont_limbs(lag[name][i], mod)

        def ratio(i):
            num = den = 1
            for k, name in enumerate(("L", "R", "O")):
                num = num * ((val(name, i) + beta * ids(k * n + i) + gamma) % mod) % mod
                den = den * ((val(name, i) + beta * ids(perm[k * n + i]) + gamma) % mod) % mod
            return num * pow(den, -1, mod) % mod
        assert val("Z", 0) == 1 and val("Z", n - 1) * ratio(n - 1) % mod == 1
        for i in sorted({0, 1, 255 % (n - 1), 256 % (n - 1), 4097 % (n - 1), n - 2}):
            assert val("Z", i + 1) == val("Z", i) * ratio(i) % mod
        bp = {"Bl": [rng.field(mod) for _ in range(2)], "Br": [rng.field(mod) for _ in range(2)],
              "Bo": [rng.field(mod) for _ in range(2)], "Bz": [rng.field(mod) for _ in range(3)]}
        kw = dict(bp={k: fr_to_arr(c, v) for k, v in bp.items()}, alpha=fr_to_arr(c, [alpha]), beta=fr_to_arr(c, [beta]), gamma=fr_to_arr(c, [gamma]))
        if pinned:
            pk = plonk.ProvingKey(d0, d1, {k: lag[k] for k in plonk.FIXED_IDS}, [qcp], lagrange=plonk.FIXED_IDS + ("Qcp0",))
            try:
                h = pk.ComputeQuotient({k: lag[k] for k in plonk.PROOF_IDS}, [pi2], lagrange=plonk.PROOF_IDS + ("Pi20",), **kw)
            finally:
                pk.close()
        else:
            h = plonk.ComputeQuotient(d0, d1, {k: lag[k] for k in plonk.IDS}, [qcp], [pi2], lagrange=tuple(plonk.IDS) + ("Qcp0", "Pi20"), **kw)
    finally:
        d0.close()
        d1.close()
    names = list(plonk.IDS)
    zm = fr_to_arr(c, [zeta])[0]
    evs = arr_to_fr(c, oracle.fr_eval_lagrange(c.cid, [lag[k] for k in names] + [qcp, pi2], zm, nthreads))
    e = dict(zip(names + ["Qcp", "Pi2"], evs))
    zw = zeta * w0 % mod
    z_w = arr_to_fr(c, oracle.fr_eval_lagrange(c.cid, [lag["Z"]], fr_to_arr(c, [zw])[0], nthreads))[0]
    zn1 = (pow(zeta, n, mod) - 1) % mod
    b = lambda k, pt: pyref._poly_eval(bp[k], pt, mod) * ((pow(pt, n, mod) - 1) % mod) % mod
    l, r, o = (e["L"] + b("Bl", zeta)) % mod, (e["R"] + b("Br", zeta)) % mod, (e["O"] + b("Bo", zeta)) % mod
    z, zs = (e["Z"] + b("Bz", zeta)) % mod, (z_w + b("Bz", zw)) % mod
    gate = (e["Ql"] * l + e["Qr"] * r + e["Qm"] * l % mod * r + e["Qo"] * o + e["Qk"] + e["Qcp"] * e["Pi2"]) % mod
    idv = zeta * beta % mod
    rr = (gamma + l + idv) * ((idv * g + r + gamma) % mod) % mod * ((idv * g * g + o + gamma) % mod) % mod * z % mod
    ll = (e["S1"] * beta + l + gamma) * ((e["S2"] * beta + r + gamma) % mod) % mod * ((e["S3"] * beta + o + gamma) % mod) % mod * zs % mod
    lone = zn1 * pow(n, -1, mod) % mod * pow((zeta - 1) % mod, -1, mod) % mod
    want = (((z - 1) * lone % mod * alpha + (ll - rr)) % mod * alpha + gate) % mod
    hz = pyref.from_mont_limbs(oracle.fr_horner(c.cid, h[: 3 * n + 6], zm), mod)
    assert hz * zn1 % mod == want
    assert not h[3 * n + 6:].any()          # deg h = 3n + 5
    return True
