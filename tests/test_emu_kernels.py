"""Kernel-logic checks of the product sources under the functional HIP emulation (tests/emu), against the
big-integer oracle.  These run on CPU (`-m "not gpu"`); the same cases run on the real GPU in test_gpu_*.py."""
import os

import numpy as np
import pytest

import oracle
import pyref
from gnark_amd import ecc, fft, groth16, plonk
from gnark_amd.device import affine_words
from helpers import BLS12_381, BN254, arr_to_fr, arr_to_g1_affine, arr_to_g2_affine, fr_to_arr, gen_of, group_of, jac_to_affine_py, pts_to_arr

CURVES = [BN254, BLS12_381]


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("logn", [0, 1, 3, 5])
def test_emu_fft_all_modes(emu_ctx, c, logn):
    n = 1 << logn
    rng = pyref.Xoshiro(100 + logn)
    a = [rng.field(c.r) for _ in range(n)]
    d = fft.Domain(emu_ctx, c.name, n)
    try:
        for dec in (pyref.DIF, pyref.DIT):
            for coset in (False, True):
                for inv in (False, True):
                    want = pyref.fft(c, a, dec, on_coset=coset, inverse=inv)
                    run = d.FFTInverse if inv else d.FFT
                    got = arr_to_fr(c, run(fr_to_arr(c, a), dec, coset))
                    assert got == want, (dec, coset, inv)
    finally:
        d.close()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_emu_fft_multi_pass(emu_ctx, c):
    # 2^12 > one LDS tile (2^10): exercises the strided upper pass and the fused scaling of both passes
    n = 1 << 12
    rng = pyref.Xoshiro(7)
    a = [rng.field(c.r) for _ in range(n)]
    d = fft.Domain(emu_ctx, c.name, n)
    try:
        got = arr_to_fr(c, d.FFT(fr_to_arr(c, a), pyref.DIF, True))
        assert got == pyref.fft(c, a, pyref.DIF, on_coset=True)
        got = arr_to_fr(c, d.FFTInverse(fr_to_arr(c, a), pyref.DIF, True))
        assert got == pyref.fft(c, a, pyref.DIF, on_coset=True, inverse=True)
        got = arr_to_fr(c, d.FFT(fr_to_arr(c, a), pyref.DIT, True))
        assert got == pyref.fft(c, a, pyref.DIT, on_coset=True)
    finally:
        d.close()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("logn", [2, 4, 6, 7, 9, 11, 13, 14])
def test_emu_fft_pass_shapes_vs_c_oracle(emu_ctx, c, logn):
    """every pass shape of the radix-4 kernel -- even / odd stage counts (the radix-2 tail), one, two and three passes, the
    natural -> bit-reversed transform on Cooley-Tukey butterflies with block-indexed twiddles (ntt.hip.h) -- element for element
    against the C oracle's radix-2 transform, all eight mode combinations"""
    n = 1 << logn
    rng = np.random.default_rng(900 + logn)
    a = fr_to_arr(c, [int.from_bytes(rng.bytes(40), "little") % c.r for _ in range(n)])
    d = fft.Domain(emu_ctx, c.name, n)
    try:
        for dec in (pyref.DIF, pyref.DIT):
            for coset in (False, True):
                for inv in (False, True):
                    want = oracle.fft(c.cid, a, 1 if inv else 0, dec, coset, nthreads=2)
                    got = (d.FFTInverse if inv else d.FFT)(a, dec, coset)
                    assert np.array_equal(got, want), (logn, dec, coset, inv)
    finally:
        d.close()


@pytest.mark.parametrize("knobs", [{}, {"GA_NTT_DIRECT": "0"}, {"GA_NTT_WAVE_LOCAL": "0"}, {"GA_NTT_WAVE_LOCAL": "0", "GA_NTT_DIRECT": "0"}],
                         ids=["default", "no-direct", "no-wave-local", "round3"])
@pytest.mark.parametrize("logn", [10, 12, 17, 18])
def test_emu_fft_wave_local_rounds(emu_ctx, monkeypatch, logn, knobs):
    """round 4's pass structure on full 1024-slot tiles: rounds below slot bit 8 exchange inside a wave (no workgroup barrier),
    bits 8/9 form one radix-4 round behind the only __syncthreads, first / last rounds move quads straight between registers and
    HBM -- every plan shape the 2^24 transforms use (10 stages alone, 10+2, 10+7, 10+8 / 9+9 with a single-stage round at slot
    bit 7 or 8), each of the two A/B knobs off as well, against the C oracle's radix-2 transform in all eight modes."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    c = BN254
    n = 1 << logn
    rng = np.random.default_rng(4100 + logn)
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= (1 << 60) - 1   # below r as 256-bit integers (Montgomery images: any residue is a valid input)
    d = fft.Domain(emu_ctx, c.name, n)
    try:
        for dec in (pyref.DIF, pyref.DIT):
            for coset in (False, True):
                for inv in (False, True):
                    want = oracle.fft(c.cid, a, 1 if inv else 0, dec, coset, nthreads=4)
                    got = (d.FFTInverse if inv else d.FFT)(a, dec, coset)
                    assert np.array_equal(got, want), (logn, dec, coset, inv)
    finally:
        d.close()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_emu_compute_h(emu_ctx, c):
    rng = pyref.Xoshiro(11)
    m, n = 13, 16
    A = [rng.field(c.r) for _ in range(m)]
    B = [rng.field(c.r) for _ in range(m)]
    Cc = [x * y % c.r for x, y in zip(A, B)]
    d = fft.Domain(emu_ctx, c.name, m)
    try:
        assert d.Cardinality == n
        got = arr_to_fr(c, d.compute_h(fr_to_arr(c, A), fr_to_arr(c, B), fr_to_arr(c, Cc)))
        assert got == pyref.compute_h(c, A, B, Cc, n)
    finally:
        d.close()


def _random_points(c, group, n, rng):
    G, g = group_of(c, group), gen_of(c, group)
    ks = [rng.next() & 0xFFFFFF for _ in range(n)]
    return [G.mul(g, k) for k in ks], ks


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("group", [0, 1], ids=["G1", "G2"])
def test_emu_msm_matches_naive(emu_ctx, c, group):
    rng = pyref.Xoshiro(1000 + group)
    n = 40
    pts, ks = _random_points(c, group, n, rng)
    scalars = [rng.field(c.r) for _ in range(n)]
    # edge cases the reference handles: zero / one / r-1 scalars, infinity bases, repeated bases (DummySetup-like)
    scalars[0], scalars[1], scalars[2] = 0, 1, c.r - 1
    pts[3] = None
    pts[5] = pts[4]
    pts[7] = group_of(c, group).neg(pts[6])
    scalars[7] = scalars[6]
    got = ecc.MultiExp(emu_ctx, c.name, group, pts_to_arr(c, group, pts), fr_to_arr(c, scalars))
    want = group_of(c, group).msm(pts, scalars)
    assert jac_to_affine_py(c, group, got) == want
    # canonical (non-Montgomery) scalars give the same point
    got2 = ecc.MultiExp(emu_ctx, c.name, group, pts_to_arr(c, group, pts), fr_to_arr(c, scalars, mont=False), montgomery=False)
    assert jac_to_affine_py(c, group, got2) == want


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("group", [0, 1], ids=["G1", "G2"])
def test_emu_msm_lazy_window_reduction(emu_ctx, c, group, monkeypatch):
    """the lazy window reduction (msm_reduce_groups29_kernel + per-bit sums + exact redo of the groups with an exceptional
    addition) forced on inputs where most buckets are empty, so both the lazy path and the redo path run"""
    monkeypatch.setenv("GA_REDUCE_LAZY_MIN", "0")
    test_emu_msm_matches_naive(emu_ctx, c, group)
    test_emu_msm_precomputed_table(emu_ctx, c, group)


def test_emu_msm_hot_bucket_and_windows(emu_ctx):
    # witness-like scalars: many equal small values -> one bucket holds most points (task splitting + hot merge)
    c, group = BN254, 0
    rng = pyref.Xoshiro(5)
    n = 3000
    G = group_of(c, group)
    base_pts, _ = _random_points(c, group, 8, rng)
    pts = [base_pts[i % 8] for i in range(n)]
    scalars = [1 if i % 10 else (rng.next() & 0xFFFF) for i in range(n)]
    P, S = pts_to_arr(c, group, pts), fr_to_arr(c, scalars)
    got = ecc.MultiExp(emu_ctx, c.name, group, P, S)
    # expected via grouping by base
    want = None
    for j in range(8):
        k = sum(s for i, s in enumerate(scalars) if i % 8 == j) % c.r
        want = G.add(want, G.mul(base_pts[j], k))
    assert jac_to_affine_py(c, group, got) == want
    # window-sharded evaluation (multi-GPU partitioning A) recombines to the same point
    cbits, nwin = ecc.plan(c.name, group, n, lib=emu_ctx.lib)
    halves = []
    mid = nwin // 2
    for lo, hi in ((0, mid), (mid, nwin)):
        w, cb, nw = ecc.MultiExpWindows(emu_ctx, c.name, group, P, S, n, lo, hi)
        assert (cb, nw) == (cbits, nwin)
        halves.append(w)
    comb = ecc.combine_windows(c.name, group, np.concatenate(halves), cbits, lib=emu_ctx.lib)
    assert jac_to_affine_py(c, group, comb) == want


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("group", [0, 1], ids=["G1", "G2"])
def test_emu_msm_precomputed_table(emu_ctx, c, group):
    """ga_msm_table_*: windows share one bucket set over [2^(c*w)]P tables; same point as the plain MSM."""
    rng = pyref.Xoshiro(31 + group)
    n = 33
    pts, _ = _random_points(c, group, n, rng)
    scalars = [rng.field(c.r) for _ in range(n)]
    scalars[0], scalars[1], scalars[2] = 0, 1, c.r - 1
    pts[3] = None
    pts[5] = pts[4]
    t = ecc.PrecomputedBases(emu_ctx, c.name, group, pts_to_arr(c, group, pts))
    try:
        info = t.info()
        assert info["windows"] == c.r.bit_length() // info["window_bits"] + 1
        got = t.MultiExp(fr_to_arr(c, scalars))
        assert jac_to_affine_py(c, group, got) == group_of(c, group).msm(pts, scalars)
        got = t.MultiExp(fr_to_arr(c, [0] * n))
        assert jac_to_affine_py(c, group, got) is None
    finally:
        t.free()


def test_emu_msm_chunked_and_ragged_sizes(emu_ctx, monkeypatch, sizes=(2, 7, 65)):
    """point-axis chunking (the msmChunkedG1 analogue) and sizes that are not multiples of anything"""
    c, group = BN254, 0
    rng = pyref.Xoshiro(99)
    G = group_of(c, group)
    pts, _ = _random_points(c, group, 9, rng)
    for n in sizes:
        P = [pts[i % 9] for i in range(n)]
        S = [rng.field(c.r) for _ in range(n)]
        want = None
        for j in range(9):
            want = G.add(want, G.mul(pts[j], sum(s for i, s in enumerate(S) if i % 9 == j) % c.r))
        for cap in (None, "5"):
            if cap:
                monkeypatch.setenv("GA_MSM_MAX_CHUNK", cap)
            else:
                monkeypatch.delenv("GA_MSM_MAX_CHUNK", raising=False)
            got = ecc.MultiExp(emu_ctx, c.name, group, pts_to_arr(c, group, P), fr_to_arr(c, S))
            assert jac_to_affine_py(c, group, got) == want, (n, cap)
    monkeypatch.delenv("GA_MSM_MAX_CHUNK", raising=False)
    # all bases at infinity, all scalars zero
    z = ecc.MultiExp(emu_ctx, c.name, group, pts_to_arr(c, group, [None] * 5), fr_to_arr(c, [3] * 5))
    assert jac_to_affine_py(c, group, z) is None


@pytest.mark.parametrize("world", [1, 3])
def test_emu_groth16_sharded_key_single_process(emu_ctx, world):
    """key sharded by base-point range (multi-GPU partition B), all shards driven from one process: partials add up to the
    same proof bytes as the unsharded prover / the oracle"""
    c = BN254
    rng = pyref.Xoshiro(777)
    cs, w = pyref.cubic_r1cs(), pyref.cubic_witness(3)
    pk, _, _ = pyref.groth16_setup(c, cs, [rng.field(c.r) for _ in range(5)])
    r, s = rng.field(c.r), rng.field(c.r)
    A, B, Cc = pyref.r1cs_solve(c, cs, w)
    sol = groth16.Solution(W=fr_to_arr(c, w), A=fr_to_arr(c, A), B=fr_to_arr(c, B), C=fr_to_arr(c, Cc))
    kw = dict(domain_cardinality=pk.n, alpha1=pts_to_arr(c, 0, [pk.alpha1]), beta1=pts_to_arr(c, 0, [pk.beta1]),
              delta1=pts_to_arr(c, 0, [pk.delta1]), A=pts_to_arr(c, 0, pk.A), B=pts_to_arr(c, 0, pk.B), Z=pts_to_arr(c, 0, pk.Z),
              K=pts_to_arr(c, 0, pk.K), beta2=pts_to_arr(c, 1, [pk.beta2]), delta2=pts_to_arr(c, 1, [pk.delta2]),
              B2=pts_to_arr(c, 1, pk.B2), infinityA=pk.infinityA, infinityB=pk.infinityB)
    parts, keys = [], []
    for k in range(world):
        dpk = groth16.ProvingKey(emu_ctx, c.name, shard=(k, world), **kw)
        keys.append(dpk)
        parts.append(groth16.ProvePartial(dpk, sol, cs.nb_public))
    total = groth16.SumPartials(c.name, parts, lib=emu_ctx.lib)
    proof = groth16.Finish(keys[0], total, fr_to_arr(c, [r]), fr_to_arr(c, [s]))
    if world > 1:
        from gnark_amd import GnarkAmdError
        with pytest.raises(GnarkAmdError, match="shard"):
            groth16.Prove(keys[0], sol, cs.nb_public, fr_to_arr(c, [r]), fr_to_arr(c, [s]))
    for dpk in keys:
        dpk.FreeGPUResources()
    assert proof.WriteTo() == pyref.proof_bytes(c, *pyref.groth16_prove(pk, cs, w, r, s))


def test_emu_error_behaviour(emu_ctx):
    """errors mirror the reference's: bad sizes are rejected with a message, nothing is computed"""
    from gnark_amd import GnarkAmdError
    c = BN254
    with pytest.raises(ValueError):
        fft.Domain(emu_ctx, c.name, 8).FFT(fr_to_arr(c, [1, 2, 3]), fft.DIF)          # len != cardinality
    with pytest.raises(ValueError):
        ecc.MultiExp(emu_ctx, "bw6-761", 0, np.zeros((1, 8), np.uint64), np.zeros((1, 4), np.uint64))   # curve not built
    with pytest.raises(GnarkAmdError, match="power of two|cardinality"):
        import ctypes as C
        h = C.c_void_p()
        emu_ctx.lib.check(emu_ctx.lib.ga_domain_create(emu_ctx.handle, 0, 12, C.byref(h)))
    with pytest.raises(GnarkAmdError, match="len\\(G1.Z\\)|n-1"):
        groth16.ProvingKey(emu_ctx, c.name, domain_cardinality=8, alpha1=np.zeros((1, 8)), beta1=np.zeros((1, 8)), delta1=np.zeros((1, 8)),
                           A=np.zeros((2, 8)), B=np.zeros((2, 8)), Z=np.zeros((3, 8)), K=np.zeros((1, 8)), beta2=np.zeros((1, 16)),
                           delta2=np.zeros((1, 16)), B2=np.zeros((2, 16)), infinityA=[0, 0], infinityB=[0, 0])


def test_emu_msm_empty_and_single(emu_ctx):
    c = BN254
    z = ecc.MultiExp(emu_ctx, c.name, 0, np.zeros((0, 8), np.uint64), np.zeros((0, 4), np.uint64))
    assert jac_to_affine_py(c, 0, z) is None
    got = ecc.MultiExp(emu_ctx, c.name, 0, pts_to_arr(c, 0, [c.g1]), fr_to_arr(c, [12345]))
    assert jac_to_affine_py(c, 0, got) == group_of(c, 0).mul(c.g1, 12345)
    with pytest.raises(ValueError):
        ecc.MultiExp(emu_ctx, c.name, 0, pts_to_arr(c, 0, [c.g1]), fr_to_arr(c, [1, 2]))


class _share_pct:
    """GA_G16_SHARE_MIN_PCT for the duration of a key pin: 0 forces the wire-indexed tables + single witness sort for every
    base vector (even sparse ones), 101 forbids it"""
    def __init__(self, pct):
        self.pct = pct

    def __enter__(self):
        import os
        self.old = os.environ.get("GA_G16_SHARE_MIN_PCT")
        if self.pct is not None:
            os.environ["GA_G16_SHARE_MIN_PCT"] = str(self.pct)

    def __exit__(self, *a):
        import os
        if self.old is None:
            os.environ.pop("GA_G16_SHARE_MIN_PCT", None)
        else:
            os.environ["GA_G16_SHARE_MIN_PCT"] = self.old


@pytest.mark.parametrize("precompute", [1, -1, "shared-sort"], ids=["tables", "no-tables", "tables-shared-sort"])
@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_emu_groth16_cubic(emu_ctx, c, precompute):
    """config 1: examples/cubic through the whole prover core, proof bytes identical to the oracle's."""
    share = 0 if precompute == "shared-sort" else None
    precompute = 1 if precompute == "shared-sort" else precompute
    rng = pyref.Xoshiro(2024)
    cs, w = pyref.cubic_r1cs(), pyref.cubic_witness(3)
    toxic = [rng.field(c.r) for _ in range(5)]
    pk, vk, _ = pyref.groth16_setup(c, cs, toxic)
    r, s = rng.field(c.r), rng.field(c.r)
    ar, bs, krs = pyref.groth16_prove(pk, cs, w, r, s)
    A, B, Cc = pyref.r1cs_solve(c, cs, w)
    with _share_pct(share):
        dpk = groth16.ProvingKey(
            emu_ctx, c.name, domain_cardinality=pk.n,
            alpha1=pts_to_arr(c, 0, [pk.alpha1]), beta1=pts_to_arr(c, 0, [pk.beta1]), delta1=pts_to_arr(c, 0, [pk.delta1]),
            A=pts_to_arr(c, 0, pk.A), B=pts_to_arr(c, 0, pk.B), Z=pts_to_arr(c, 0, pk.Z), K=pts_to_arr(c, 0, pk.K),
            beta2=pts_to_arr(c, 1, [pk.beta2]), delta2=pts_to_arr(c, 1, [pk.delta2]), B2=pts_to_arr(c, 1, pk.B2),
            infinityA=pk.infinityA, infinityB=pk.infinityB, precompute=precompute)
    try:
        sol = groth16.Solution(W=fr_to_arr(c, w), A=fr_to_arr(c, A), B=fr_to_arr(c, B), C=fr_to_arr(c, Cc))
        proof = groth16.Prove(dpk, sol, cs.nb_public, fr_to_arr(c, [r]), fr_to_arr(c, [s]))
    finally:
        dpk.FreeGPUResources()
    assert arr_to_g1_affine(c, proof.Ar) == ar
    assert arr_to_g2_affine(c, proof.Bs) == bs
    assert arr_to_g1_affine(c, proof.Krs) == krs
    assert proof.WriteTo() == pyref.proof_bytes(c, ar, bs, krs)
    assert proof.WriteRawTo() == pyref.proof_bytes_raw(c, ar, bs, krs)
    # and the way the reference itself asserts a prover: Verify accepts the proof BYTES -- the oracle's restatement of
    # verify.go:38-145, pinned by the twelve bellman tuples (tests/test_oracle_fixtures.py) -- and rejects a wrong public input
    got = pyref.proof_read(c, proof.WriteTo())[:5]
    assert pyref.groth16_verify(c, vk, got, w[1:cs.nb_public])
    assert not pyref.groth16_verify(c, vk, got, [(w[1] ^ 1) % c.r])


def _xmd_vectors():
    import json, os
    with open(os.path.join(os.path.dirname(__file__), "golden", "expand_msg_xmd.json")) as f:
        return json.load(f)


def test_emu_hash_to_field(emu_ctx):
    """the library's host hash-to-field: expand_message_xmd against the reference's vectors
    (std/hash/expand/expand_test.go:52-140), fr.Hash against the oracle for both scalar fields"""
    import ctypes as C
    lib = emu_ctx.lib
    g = _xmd_vectors()
    dst = np.frombuffer(g["dst"].encode(), dtype=np.uint8)
    for v in g["vectors"]:
        msg = np.frombuffer(v["msg"].encode(), dtype=np.uint8) if v["msg"] else np.zeros(1, dtype=np.uint8)
        out = np.zeros(v["len_in_bytes"], dtype=np.uint8)
        lib.check(lib.ga_expand_message_xmd(msg.ctypes.data_as(C.c_void_p), len(v["msg"]), dst.ctypes.data_as(C.c_void_p), dst.size,
                                            out.size, out.ctypes.data_as(C.c_void_p)))
        assert out.tobytes().hex() == v["uniform_bytes_hex"]
    rng = pyref.Xoshiro(31)
    for c in CURVES:
        for ln in (0, 1, 55, 56, 64, 200):
            msg = bytes(rng.next() & 0xFF for _ in range(ln))
            got = groth16.HashToField(c.name, msg, groth16.FOLD_DST, 3, lib=lib)
            assert arr_to_fr(c, got) == pyref.fr_hash(c, msg, pyref.FOLD_DST, 3)


@pytest.mark.parametrize("precompute", [1, -1, "shared-sort"], ids=["tables", "no-tables", "tables-shared-sort"])
@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_emu_groth16_bsb22_commitments(emu_ctx, c, precompute):
    """a circuit with two api.Commit calls.  The solver-side hint calls ProvingKey.Commit (device MSMs over
    the pinned pedersen bases) and hashes the commitment; Prove leaves the committed wires out of the K MSM; the folded proof
    of knowledge and the proof bytes equal the oracle's (prove.go:60-127,231-235, marshal.go:33-58)."""
    lib = emu_ctx.lib
    share = 0 if precompute == "shared-sort" else None
    precompute = 1 if precompute == "shared-sort" else precompute
    rng = pyref.Xoshiro(4242)
    cs = pyref.commit_r1cs()
    toxic = [rng.field(c.r) for _ in range(5 + len(cs.commitments) + 1)]
    pk, vk, _ = pyref.groth16_setup(c, cs, toxic)
    removed = sorted({j for cm in cs.commitments for j in cm.private_committed} | {cm.commitment_index for cm in cs.commitments})
    with _share_pct(share):
        dpk = groth16.ProvingKey(
            emu_ctx, c.name, domain_cardinality=pk.n,
            alpha1=pts_to_arr(c, 0, [pk.alpha1]), beta1=pts_to_arr(c, 0, [pk.beta1]), delta1=pts_to_arr(c, 0, [pk.delta1]),
            A=pts_to_arr(c, 0, pk.A), B=pts_to_arr(c, 0, pk.B), Z=pts_to_arr(c, 0, pk.Z), K=pts_to_arr(c, 0, pk.K),
            beta2=pts_to_arr(c, 1, [pk.beta2]), delta2=pts_to_arr(c, 1, [pk.delta2]), B2=pts_to_arr(c, 1, pk.B2),
            infinityA=pk.infinityA, infinityB=pk.infinityB, precompute=precompute,
            commitment_keys=[(pts_to_arr(c, 0, b), pts_to_arr(c, 0, e)) for b, e in pk.commitment_keys], k_remove=removed)
    coms, poks = {}, {}
    fbytes = (c.r.bit_length() - 1) // 8 + 1

    def hint(i, w):   # the bsb22 hint override (prove.go:72-100) with the device doing the MSMs
        cm = cs.commitments[i]
        coms[i], poks[i] = dpk.Commit(i, fr_to_arr(c, [w[j] for j in cm.private_committed]))
        msg = groth16.MarshalG1(c.name, coms[i], lib=lib) + b"".join(int(w[j]).to_bytes(fbytes, "big") for j in cm.public_and_commitment_committed)
        return arr_to_fr(c, groth16.HashToField(c.name, msg, groth16.COMMITMENT_DST, 1, lib=lib))[0]

    try:
        w = pyref.commit_solve(c, cs, 3, 11, hint)
        assert w == pyref.commit_solve(c, cs, 3, 11, lambda i, ww: pyref.commitment_hint(pk, cs, i, ww)[1])
        r, s = rng.field(c.r), rng.field(c.r)
        A, B, Cc = pyref.r1cs_solve(c, cs, w)
        sol = groth16.Solution(W=fr_to_arr(c, w), A=fr_to_arr(c, A), B=fr_to_arr(c, B), C=fr_to_arr(c, Cc))
        proof = groth16.Prove(dpk, sol, cs.nb_public, fr_to_arr(c, [r]), fr_to_arr(c, [s]))
        with pytest.raises(Exception, match="values for a basis"):
            dpk.Commit(0, fr_to_arr(c, [1, 2, 3]))
        with pytest.raises(Exception, match="commitment 2 of 2"):
            dpk.Commit(2, fr_to_arr(c, [1]))
    finally:
        dpk.FreeGPUResources()
    ser = b"".join(int(w[cm.commitment_index]).to_bytes(32, "big") for cm in cs.commitments)
    challenge = groth16.HashToField(c.name, ser, groth16.FOLD_DST, 1, lib=lib)
    proof.Commitments = np.stack([coms[i] for i in range(len(cs.commitments))])
    proof.CommitmentPok = groth16.FoldPok(c.name, np.stack([poks[i] for i in range(len(cs.commitments))]), challenge, lib=lib)
    ar, bs, krs, ocoms, opok = pyref.groth16_prove_bsb22(pk, cs, w, r, s)
    assert [arr_to_g1_affine(c, x) for x in proof.Commitments] == ocoms
    assert arr_to_g1_affine(c, proof.CommitmentPok) == opok
    assert (arr_to_g1_affine(c, proof.Ar), arr_to_g2_affine(c, proof.Bs), arr_to_g1_affine(c, proof.Krs)) == (ar, bs, krs)
    assert proof.WriteTo() == pyref.proof_bytes(c, ar, bs, krs, ocoms, opok)
    assert proof.WriteRawTo() == pyref.proof_bytes_raw(c, ar, bs, krs, ocoms, opok)
    # Verify (verify.go:38-145 restated, pinned by the bellman tuples) accepts the proof bytes: commitment hashes recomputed from the
    # commitments in the proof, the folded pedersen proof of knowledge checked by pairings, then the Groth16 equation
    got = pyref.proof_read(c, proof.WriteTo())[:5]
    pacc = pyref.vk_public_and_commitment_committed(cs)
    assert pyref.groth16_verify(c, vk, got, w[1:cs.nb_public], pacc)
    assert not pyref.groth16_verify(c, vk, got, [(w[1] ^ 1) % c.r], pacc)
    # pedersen verification in the exponent: pok_i = [sigma_i] commitment_i, folded with the challenge powers
    G1 = group_of(c, 0)
    sig = toxic[5:7]
    ch = pyref.fr_hash(c, ser, pyref.FOLD_DST, 1)[0]
    assert opok == G1.msm(ocoms, [sg * pow(ch, i, c.r) % c.r for i, sg in enumerate(sig)])


def _plonk_case(c, n, seed, nb_bsb):
    """a satisfying synthetic PLONK trace (oracle) with its challenges and blinding polynomials, canonical + Lagrange forms"""
    mod = c.r
    lag, qcp, pi2, perm, rng = pyref.plonk_synthetic_instance(c, n, seed, nb_bsb)
    beta, gamma, alpha = rng.field(mod), rng.field(mod), rng.field(mod)
    lag["Z"] = pyref.plonk_build_z(c, n, lag["L"], lag["R"], lag["O"], perm, beta, gamma)
    w0 = c.fr_root_of_unity(n)
    ninv = pow(n, -1, mod)
    tocan = lambda v: [x * ninv % mod for x in pyref._ntt_natural(v, pow(w0, -1, mod), mod)]
    can = {k: tocan(v) for k, v in lag.items()}
    qc_can, pi_can = [tocan(v) for v in qcp], [tocan(v) for v in pi2]
    bp = {"Bl": [rng.field(mod) for _ in range(2)], "Br": [rng.field(mod) for _ in range(2)],
          "Bo": [rng.field(mod) for _ in range(2)], "Bz": [rng.field(mod) for _ in range(3)]}
    return dict(lag=lag, can=can, qcp=qcp, pi2=pi2, qc_can=qc_can, pi_can=pi_can, perm=perm, bp=bp, alpha=alpha, beta=beta,
                gamma=gamma, rng=rng)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n,nb_bsb", [(4, 0), (8, 1), (64, 2), (512, 1)])
def test_emu_plonk_quotient(emu_ctx, c, n, nb_bsb, seed=11):
    """computeNumerator + divideByZH on the device == the oracle's restatement of prove.go:841-1123,1287-1350,
    coefficient for coefficient, from canonical inputs, from Lagrange inputs and from a mix; and h(x)(x^n-1) equals the blinded
    constraint polynomial at a random point (the instance satisfies its gates and copy constraints)."""
    mod = c.r
    T = _plonk_case(c, n, seed, nb_bsb)
    want = pyref.plonk_quotient(c, n, T["can"], T["qc_can"], T["pi_can"], T["bp"], T["alpha"], T["beta"], T["gamma"])
    zeta = T["rng"].field(mod)
    assert pyref._poly_eval(want, zeta, mod) * (pow(zeta, n, mod) - 1) % mod == pyref.plonk_numerator_at(
        c, n, T["can"], T["qc_can"], T["pi_can"], T["bp"], T["alpha"], T["beta"], T["gamma"], zeta)
    d0 = fft.Domain(emu_ctx, c.name, n)
    d1 = fft.Domain(emu_ctx, c.name, plonk.Rho(n) * n)
    try:
        kw = dict(bp={k: fr_to_arr(c, v) for k, v in T["bp"].items()}, alpha=fr_to_arr(c, [T["alpha"]]), beta=fr_to_arr(c, [T["beta"]]),
                  gamma=fr_to_arr(c, [T["gamma"]]))
        names = list(plonk.IDS) + [x for i in range(nb_bsb) for x in (f"Qcp{i}", f"Pi2{i}")]
        for lagr in ((), tuple(names), tuple(names[::2])):
            pick = lambda nm, canv, lagv: fr_to_arr(c, lagv if nm in lagr else canv)
            polys = {k: pick(k, T["can"][k], T["lag"][k]) for k in plonk.IDS}
            qc = [pick(f"Qcp{i}", T["qc_can"][i], T["qcp"][i]) for i in range(nb_bsb)]
            pi = [pick(f"Pi2{i}", T["pi_can"][i], T["pi2"][i]) for i in range(nb_bsb)]
            got = plonk.ComputeQuotient(d0, d1, polys, qc, pi, lagrange=lagr, **kw)
            assert arr_to_fr(c, got) == want, lagr
            # the same proof with the circuit constants pinned (ga_plonk_pk_create / ga_plonk_quotient_pinned), two proofs per key
            ppk = plonk.ProvingKey(d0, d1, {k: polys[k] for k in plonk.FIXED_IDS}, qc, lagrange=[x for x in lagr if x in plonk.FIXED_IDS or x.startswith("Qcp")])
            try:
                for _ in range(2):
                    got = ppk.ComputeQuotient({k: polys[k] for k in plonk.PROOF_IDS}, pi, lagrange=[x for x in lagr if x in plonk.PROOF_IDS or x.startswith("Pi2")], **kw)
                    assert arr_to_fr(c, got) == want, ("pinned", lagr)
            finally:
                ppk.close()
    finally:
        d0.close()
        d1.close()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n", [2, 8, 512, 1000])
def test_emu_plonk_build_z_and_batch_invert(emu_ctx, c, n):
    """iop.BuildRatioCopyConstraint (prove.go:645-655) and batchInvert (prove.go:1134-1147) against the oracle; the grand
    product closes (Z[n-1] * ratio[n-1] == 1) because the permutation only links equal values"""
    mod = c.r
    vals = [0, 1, mod - 1] + [pyref.Xoshiro(n).field(mod) for _ in range(max(0, n - 3))]
    got = plonk.BatchInvert(emu_ctx, c.name, fr_to_arr(c, vals[:n]))
    assert arr_to_fr(c, got) == [pow(v, -1, mod) if v else 0 for v in vals[:n]]
    if n & (n - 1):
        return
    lag, _, _, perm, rng = pyref.plonk_synthetic_instance(c, n, 3 + n, 0)
    beta, gamma = rng.field(mod), rng.field(mod)
    want = pyref.plonk_build_z(c, n, lag["L"], lag["R"], lag["O"], perm, beta, gamma)
    d0 = fft.Domain(emu_ctx, c.name, n)
    try:
        got = plonk.BuildRatioCopyConstraint(d0, fr_to_arr(c, lag["L"]), fr_to_arr(c, lag["R"]), fr_to_arr(c, lag["O"]), perm,
                                             fr_to_arr(c, [beta]), fr_to_arr(c, [gamma]))
        with pytest.raises(Exception, match="outside"):
            plonk.BuildRatioCopyConstraint(d0, fr_to_arr(c, lag["L"]), fr_to_arr(c, lag["R"]), fr_to_arr(c, lag["O"]),
                                           [3 * n] + list(perm[1:]), fr_to_arr(c, [beta]), fr_to_arr(c, [gamma]))
    finally:
        d0.close()
    assert arr_to_fr(c, got) == want


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_emu_groth16_synthetic_vs_c_oracle(emu_ctx, c, monkeypatch, logn=7):
    """Synthetic instance (SURVEY 8d config 3 shape, scaled down; 2^7 under the emulation, 2^10 on the GPU): bases [k_i]G
    generated on the device, proof points identical to the C oracle's prover, with and without window tables."""
    ctx = emu_ctx
    lib = ctx.lib
    n = 1 << logn
    nw = n
    nb_public = 3

    def gen(group, count, seed):
        buf = ctx.malloc(count * affine_words(c.cid, group) * 8)
        lib.check(lib.ga_gen_bases(ctx.handle, c.cid, group, seed, count, buf.ptr, None))
        h = buf.to_host((count, affine_words(c.cid, group)))
        buf.free()
        return h

    def scal(count, seed):
        buf = ctx.malloc(count * 32)
        lib.check(lib.ga_gen_scalars(ctx.handle, c.cid, seed, count, buf.ptr))
        h = buf.to_host((count, 4))
        buf.free()
        return h
    infA = np.zeros(nw, np.uint8)
    infB = np.zeros(nw, np.uint8)
    infA[[1, 5, nw - 1]] = 1
    infB[[0, 7]] = 1
    m1, m2 = gen(0, 3, 1), gen(1, 2, 2)
    key = dict(n=n, alpha1=m1[0:1], beta1=m1[1:2], delta1=m1[2:3], A=gen(0, nw - 3, 3), B=gen(0, nw - 2, 4), Z=gen(0, n - 1, 5),
               K=gen(0, nw - nb_public, 6), beta2=m2[0:1], delta2=m2[1:2], B2=gen(1, nw - 2, 7), infinityA=infA, infinityB=infB)
    m = n - 9
    W, A, B = scal(nw, 10), scal(m, 11), scal(m, 12)
    Cc = oracle.fr_mul(c.cid, A, B)
    rs = scal(2, 13)
    want = oracle.groth16_prove(c.cid, key, W, A, B, Cc, nb_public, rs[0], rs[1], nthreads=8)
    for precompute in (1, -1):
        pk = groth16.ProvingKey(ctx, c.name, domain_cardinality=n, precompute=precompute, **{k: v for k, v in key.items() if k != "n"})
        try:
            proof = groth16.Prove(pk, groth16.Solution(W, A, B, Cc), nb_public, rs[0], rs[1])
        finally:
            pk.FreeGPUResources()
        assert np.array_equal(proof.Ar, want[0]) and np.array_equal(proof.Bs, want[1]) and np.array_equal(proof.Krs, want[2]), precompute
    # precompute = 0 with a budget that does not hold all five tables (a 2^26-constraint key on one GPU; GA_G16_TABLE_BUDGET_PCT scales the
    # situation down): the library builds tables for a PREFIX of A, B, K, Z, G2.B and proves with the rest as plain vectors -- wire-indexed
    # tables over the shared sort, compact tables and un-pinned MSMs side by side in one proof, same proof points
    order = ("A", "B", "K", "Z", "B2")
    sizes = []
    for pct in (0, 5, 30, 55, 80):   # (0 = no limit; a G1 table is ~1/6 of the five tables' bytes, G2.B 2/6)
        if pct:
            monkeypatch.setenv("GA_G16_TABLE_BUDGET_PCT", str(pct))
        pk = groth16.ProvingKey(ctx, c.name, domain_cardinality=n, precompute=0, **{k: v for k, v in key.items() if k != "n"})
        try:
            lay = groth16.ShardLayout(pk)
            proof = groth16.Prove(pk, groth16.Solution(W, A, B, Cc), nb_public, rs[0], rs[1])
        finally:
            pk.FreeGPUResources()
        have = [k for k in order if lay["tables"][k]]
        assert have == list(order[:len(have)]), lay["tables"]                      # a prefix of the preference order
        assert all(lay["tables"][k] for k, v in lay["wire_indexed"].items() if v)  # wire-indexed implies a table
        sizes.append(len(have))
        assert np.array_equal(proof.Ar, want[0]) and np.array_equal(proof.Bs, want[1]) and np.array_equal(proof.Krs, want[2]), (pct, lay["tables"])
    monkeypatch.delenv("GA_G16_TABLE_BUDGET_PCT", raising=False)
    assert sizes == [5, 0, 1, 3, 4], sizes   # none / A / A, B, K / A, B, K, Z: partial table sets, growing with the budget
    assert len(proof.WriteTo()) == (164 if c.cid == 0 else 244)


def _device_inputs(ctx, c, group, n, seed):
    lib = ctx.lib
    wa = affine_words(c.cid, group)
    bases, dlogs, scal = ctx.malloc(n * wa * 8), ctx.malloc(n * 32), ctx.malloc(n * 32)
    lib.check(lib.ga_gen_bases(ctx.handle, c.cid, group, seed, n, bases.ptr, dlogs.ptr))
    lib.check(lib.ga_gen_scalars(ctx.handle, c.cid, seed + 1, n, scal.ptr))
    return bases, dlogs, scal


def _expect_from_dlogs(c, group, S_host, K_host):
    k = oracle.fr_dot(c.cid, S_host, K_host)
    return oracle.jac_to_affine(c.cid, group, oracle.generator_mul(c.cid, group, k))


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("group", [0, 1], ids=["G1", "G2"])
def test_emu_msm_table_batch(emu_ctx, c, group, n=300, k=3, batched=False):
    """ga_msm_table_run_batch: k scalar vectors over one table in one pass == k separate runs == [sum s_i k_i]G; host and device
    scalars, edge vectors (all zero, all ones), argument validation; batched = the GA_TABLE_BATCHED planning hint"""
    ctx = emu_ctx
    bases, dlogs, scal = _device_inputs(ctx, c, group, n, 0xBA7C + group)
    K = dlogs.to_host((n, 4))
    one = np.array(pyref.to_mont_limbs(1, c.r, 4), dtype=np.uint64)
    vecs = [scal.to_host((n, 4))]
    for j in range(1, k):
        b = ctx.malloc(n * 32)
        ctx.lib.check(ctx.lib.ga_gen_scalars(ctx.handle, c.cid, 0x51 + j, n, b.ptr))
        vecs.append(b.to_host((n, 4)))
        b.free()
    vecs.append(np.zeros((n, 4), dtype=np.uint64))
    vecs.append(np.tile(one, (n, 1)))
    t = ecc.PrecomputedBases(ctx, c.name, group, bases, n=n, batched=batched)
    devs = [ctx.to_device(v) for v in vecs]
    try:
        single = [t.MultiExp(v) for v in vecs]
        got_h = t.MultiExpBatch(vecs)
        got_d = t.MultiExpBatch(devs)
        for j, v in enumerate(vecs):
            want = _expect_from_dlogs(c, group, v, K)
            assert np.array_equal(oracle.jac_to_affine(c.cid, group, got_h[j]), want), j
            assert np.array_equal(oracle.jac_to_affine(c.cid, group, got_d[j]), want), j
            assert np.array_equal(oracle.jac_to_affine(c.cid, group, single[j]), want), j
        assert np.array_equal(oracle.jac_to_affine(c.cid, group, t.MultiExpBatch(vecs[:1])[0]), _expect_from_dlogs(c, group, vecs[0], K))
        with pytest.raises(Exception, match="batch size"):
            t.MultiExpBatch([vecs[0]] * 17)
        with pytest.raises(ValueError):
            t.MultiExpBatch([vecs[0], devs[1]])
    finally:
        t.free()
        for b in [bases, dlogs, scal] + devs:
            b.free()


def test_emu_msm_degenerate_bases_exact_kernel(emu_ctx, monkeypatch, c=BN254, group=0, n=600):
    """the last line of defence, msm_accumulate29_redo_kernel (complete formulas in exact arithmetic), is what GA_MSM_EXACT_REDO=1
    sends every flagged task to: same degenerate inputs, same results"""
    monkeypatch.setenv("GA_MSM_EXACT_REDO", "1")
    test_emu_msm_degenerate_bases(emu_ctx, c, group, n=n)
    test_emu_msm_degenerate_bases(emu_ctx, BLS12_381, 1, n=200)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("group", [0, 1], ids=["G1", "G2"])
def test_emu_msm_degenerate_bases(emu_ctx, c, group, n=1500):
    """DummySetup-like key (backend/groth16/bn254/setup.go:517-543): every base the same point, so every bucket meets P + P and P - P.
    The fast loop flags (nearly) every task, the complete lazy loop (madd29_complete / mdbl29) re-runs them; the table is then
    remembered as degenerate and the SECOND run goes through the complete loop directly.  Also a half-degenerate vector."""
    ctx = emu_ctx
    bases, dlogs, scal = _device_inputs(ctx, c, group, n, 0xD0 + group)
    wa = affine_words(c.cid, group)
    P, K, S = bases.to_host((n, wa)), dlogs.to_host((n, 4)), scal.to_host((n, 4))
    for variant in ("all-equal", "half-equal"):
        Pv, Kv = P.copy(), K.copy()
        if variant == "all-equal":
            Pv[:], Kv[:] = P[0], K[0]
        else:
            Pv[::2], Kv[::2] = P[0], K[0]
        want = _expect_from_dlogs(c, group, S, Kv)
        dv = ctx.to_device(Pv)
        t = ecc.PrecomputedBases(ctx, c.name, group, dv, n=n)
        try:
            for _ in range(2):   # second run: degenerate table -> complete loop from the start
                assert np.array_equal(oracle.jac_to_affine(c.cid, group, t.MultiExp(scal)), want), variant
            assert np.array_equal(oracle.jac_to_affine(c.cid, group, ecc.MultiExp(ctx, c.name, group, dv, scal, n=n)), want), variant
        finally:
            t.free()
            dv.free()
    for b in (bases, dlogs, scal):
        b.free()


def test_emu_msm_table_two_callers(emu_ctx, c=BN254, group=0, n=2000, rounds=3):
    """three host threads commit DIFFERENT scalar vectors over one pinned table at the same time (the PLONK prover's goroutines): the
    second caller runs on lane 1 of the context beside the first; every result equals the one computed alone"""
    import threading
    ctx = emu_ctx
    bases, dlogs, scal = _device_inputs(ctx, c, group, n, 0x7C0 + group)
    vecs = [scal.to_host((n, 4))]
    for j in range(1, 3):
        b = ctx.malloc(n * 32)
        ctx.lib.check(ctx.lib.ga_gen_scalars(ctx.handle, c.cid, 0x91 + j, n, b.ptr))
        vecs.append(b.to_host((n, 4)))
        b.free()
    t = ecc.PrecomputedBases(ctx, c.name, group, bases, n=n)
    try:
        # (compared as affine points: from 2^25 pairs the Jacobian representative of a result is not fixed, gnark_amd.h ga_msm)
        aff = lambda p: oracle.jac_to_affine(c.cid, group, p)
        want = [t.MultiExp(v) for v in vecs]
        want_aff = [aff(p) for p in want]
        bad = []

        def worker(tid):
            for k in range(rounds):
                j = (tid + k) % 3
                if not np.array_equal(aff(t.MultiExp(vecs[j])), want_aff[j]):
                    bad.append((tid, k, j))

        th = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not bad, bad
        K = dlogs.to_host((n, 4))
        assert np.array_equal(oracle.jac_to_affine(c.cid, group, want[1]), _expect_from_dlogs(c, group, vecs[1], K))
    finally:
        t.free()
        for b in (bases, dlogs, scal):
            b.free()


@pytest.mark.parametrize("c,group,table", [(BN254, 0, True), (BN254, 0, False), (BLS12_381, 1, False)], ids=["bn254-G1-table", "bn254-G1-raw", "bls-G2-raw"])
def test_emu_msm_very_hot_bucket(emu_ctx, c, group, table, n=36000):
    """boolean-heavy witness: 60 % of the scalars equal to one, 10 % zero -> the digit-1 bucket of window 0 holds 0.6 n points, i.e.
    hundreds of tasks: the cooperative task-list writer (msm_task_list_long_kernel) and the two-stage merge of very hot buckets
    (msm_vhot_stage1/2_kernel, > 512 partial sums) against [sum s_i k_i]G"""
    ctx = emu_ctx
    bases, dlogs, scal = _device_inputs(ctx, c, group, n, 0x1207 + group)
    S = scal.to_host((n, 4))
    K = dlogs.to_host((n, 4))
    one = np.array(pyref.to_mont_limbs(1, c.r, 4), dtype=np.uint64)
    u = np.random.default_rng(7).random(n)
    S[u < 0.6] = one
    S[(u >= 0.6) & (u < 0.7)] = 0
    sdev = ctx.to_device(S)
    try:
        if table:
            t = ecc.PrecomputedBases(ctx, c.name, group, bases, n=n)
            try:
                got = t.MultiExp(sdev)
            finally:
                t.free()
        else:
            got = ecc.MultiExp(ctx, c.name, group, bases, sdev, n=n)
        assert np.array_equal(oracle.jac_to_affine(c.cid, group, got), _expect_from_dlogs(c, group, S, K))
    finally:
        for b in (bases, dlogs, scal, sdev):
            b.free()


@pytest.mark.parametrize("xcd", [0, 7], ids=["no-xcd-placement", "xcd-slices-and-swizzle"])
@pytest.mark.parametrize("c,group", [(BN254, 0), (BLS12_381, 1)], ids=["bn254-G1", "bls12-381-G2"])
def test_emu_msm_fused_first_sort_pass(emu_ctx, c, group, xcd, monkeypatch, n=1300, table_c=16):
    """msm.hip.h 1b -- the digit extraction fused with the first radix-sort pass (histogram of the low key bits from the scalars,
    LDS-partitioned tiles, one library pass for the high bits) -- forced on a small table MSM (GA_MSM_FUSE_MIN=0; GA_TABLE_C=16:
    16 windows, 2^15 buckets, partial tiles of 832 scalars) and compared with the same MSM through the plain digits + sort
    sequence and with [sum s_i k_i]G.  Scalars: uniform, 0, 1, r-1, a hot value, canonical and Montgomery inputs, window ranges."""
    ctx = emu_ctx
    monkeypatch.setenv("GA_TABLE_C", str(table_c))
    monkeypatch.setenv("GA_MSM_XCD", str(xcd))   # per-XCD slices of the first level (bit 2: also below 2^24 pairs), XCD swizzle of the second: placement only
    monkeypatch.setenv("GA_MSM_P1_GRID", "1" if xcd == 0 else "9")   # one block walks every tile / nine (-> 16) blocks with XCD classes: the tile loop
    if xcd == 0:
        monkeypatch.setenv("GA_MSM_TASK_EXACT_MIN", "0")   # the task list ordered by exact length (what MSMs of 2^25 pairs and more do); the other case: quantised
    bases, dlogs, scal = _device_inputs(ctx, c, group, n, 0xF05E + group)
    S = scal.to_host((n, 4))
    K = dlogs.to_host((n, 4))
    mont = lambda v: np.array(pyref.to_mont_limbs(v, c.r, 4), dtype=np.uint64)
    S[0], S[1], S[2] = mont(0), mont(1), mont(c.r - 1)
    S[100:700] = mont(1)          # a hot bucket (window 0, digit 1) and many zero digits above it
    S[700:900] = mont(0)
    sdev = ctx.to_device(S)
    t = ecc.PrecomputedBases(ctx, c.name, group, bases, n=n)
    try:
        info = t.info()
        assert info["window_bits"] == table_c and info["windows"] <= 16
        want = _expect_from_dlogs(c, group, S, K)
        monkeypatch.setenv("GA_MSM_FUSE_MIN", str(1 << 40))
        plain = t.MultiExp(sdev)
        monkeypatch.setenv("GA_MSM_FUSE_MIN", "0")
        ctx.profile(True)
        ctx.profile_reset()
        fused = t.MultiExp(sdev)
        ctx.sync()
        stages = [name for name, _ in ctx.profile_read()]
        ctx.profile(False)
        assert "msm_digits_pass1" in stages and "msm_digits" not in stages   # the fused path is what ran
        assert np.array_equal(oracle.jac_to_affine(c.cid, group, plain), want)
        assert np.array_equal(oracle.jac_to_affine(c.cid, group, fused), want)
        # canonical scalars, and the window ranges of a window-sharded key (carries below the range, partial results add up)
        rinv, m64 = pow(1 << 256, -1, c.r), (1 << 64) - 1
        ints = [(int(a) | int(b) << 64 | int(d) << 128 | int(e) << 192) * rinv % c.r for a, b, d, e in S.tolist()]
        canon = np.array([[v & m64, v >> 64 & m64, v >> 128 & m64, v >> 192] for v in ints], dtype=np.uint64)
        got = t.MultiExp(ctx.to_device(canon), montgomery=False)
        assert np.array_equal(oracle.jac_to_affine(c.cid, group, got), want)
        nw = info["windows"]
        parts = [t.MultiExpWindows(sdev, lo, hi) for lo, hi in ((0, 5), (5, nw - 1), (nw - 1, nw))]
        acc = parts[0]
        for q in parts[1:]:
            acc = ecc.jac_add(c.name, group, acc, q, lib=ctx.lib)
        assert np.array_equal(oracle.jac_to_affine(c.cid, group, acc), want)
        # raw (un-pinned) bases: one bucket set per window, keys = window * 2^(c-1) + digit - 1.  The fused pass applies to window
        # ranges of at most 16 windows whose bucket sets fit 22 key bits: the two halves of a window-sharded MSM recombine to the point
        monkeypatch.delenv("GA_TABLE_C")
        cbits, nwin = ecc.plan(c.name, group, n, lib=ctx.lib)
        cuts = list(range(0, nwin, 16)) + [nwin]
        ctx.profile(True)
        ctx.profile_reset()
        wins = [ecc.MultiExpWindows(ctx, c.name, group, bases, sdev, n, lo, hi)[0] for lo, hi in zip(cuts[:-1], cuts[1:])]
        ctx.sync()
        stages = [name for name, _ in ctx.profile_read()]
        ctx.profile(False)
        if 8 <= cbits <= 18:   # 16 windows x 2^(c-1) buckets: 12..22 key bits
            assert "msm_digits_pass1" in stages
        comb = ecc.combine_windows(c.name, group, np.concatenate(wins), cbits, lib=ctx.lib)
        assert np.array_equal(oracle.jac_to_affine(c.cid, group, comb), want)
    finally:
        t.free()
        for b in (bases, dlogs, scal, sdev):
            b.free()


@pytest.mark.parametrize("xcd", [0, 7], ids=["no-xcd-placement", "xcd-slices-and-swizzle"])
@pytest.mark.parametrize("table_c,batch", [(22, 3), (23, 3)], ids=["23-bit-keys-11-bit-level", "24-bit-keys-12-bit-level"])
def test_emu_msm_fused_sort_wide_keys(emu_ctx, monkeypatch, table_c, batch, xcd, c=BN254, group=0, n=257):
    """msm.hip.h 1b / 1c beyond 22 key bits and beyond one scalar vector (round 4): a batch of `batch` scalar vectors over one table
    stacks its bucket sets in ONE key space -- 3 x 2^21 keys (PLONK's batched commitments: 23 key bits, 11-bit first level with 3073
    high parts in the second) and 3 x 2^22 keys (24 key bits: the 12-bit first level).  Vectors: uniform; every scalar the same (all
    pairs of a window share ONE key: one bin, one run, segments of a single key); zeros and ones (skip bucket + a hot bucket); r - 1.
    The top window of a 254-bit scalar has 12 live bits at c = 22 (2^24 of 12 x 2^24 pairs in buckets 0..4095 at full size) -- inherent
    in every vector here.  Fused == the library sort == [sum s_i k_i]G."""
    ctx = emu_ctx
    monkeypatch.setenv("GA_TABLE_C", str(table_c))
    monkeypatch.setenv("GA_MSM_XCD", str(xcd))
    bases, dlogs, scal = _device_inputs(ctx, c, group, n, 0x57DE + table_c)
    K = dlogs.to_host((n, 4))
    mont = lambda v: np.array(pyref.to_mont_limbs(v, c.r, 4), dtype=np.uint64)
    U = scal.to_host((n, 4))
    same = np.tile(mont(0x1234567890ABCDEF1234567890ABCDEF % c.r), (n, 1))
    zo = np.tile(mont(1), (n, 1))
    zo[::3] = mont(0)
    zo[5] = mont(c.r - 1)
    vecs = [U, same, zo][:batch]
    devs = [ctx.to_device(v) for v in vecs]
    t = ecc.PrecomputedBases(ctx, c.name, group, bases, n=n)
    try:
        assert t.info()["window_bits"] == table_c
        want = [_expect_from_dlogs(c, group, v, K) for v in vecs]
        monkeypatch.setenv("GA_MSM_FUSE_MIN", str(1 << 40))
        plain = t.MultiExpBatch(devs)
        monkeypatch.setenv("GA_MSM_FUSE_MIN", "0")
        ctx.profile(True)
        ctx.profile_reset()
        fused = t.MultiExpBatch(devs)
        ctx.sync()
        stages = [name for name, _ in ctx.profile_read()]
        ctx.profile(False)
        assert "msm_digits_pass1" in stages and "msm_digits" not in stages   # the fused path took the batch
        for j in range(batch):
            assert np.array_equal(oracle.jac_to_affine(c.cid, group, plain[j]), want[j]), j
            assert np.array_equal(oracle.jac_to_affine(c.cid, group, fused[j]), want[j]), j
    finally:
        t.free()
        for b in [bases, dlogs, scal] + devs:
            b.free()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("group", [0, 1], ids=["G1", "G2"])
def test_emu_msm_vs_c_oracle_and_dlogs(emu_ctx, c, group, logn=11):
    """device-generated bases [k_i]G with known discrete logs: MSM == C oracle's Pippenger == [sum s_i k_i]G (2^11 under the
    emulation, 2^14 on the GPU), device-resident and host-pointer inputs"""
    gpu_ctx = emu_ctx
    n = 1 << logn
    bases, dlogs, scal = _device_inputs(gpu_ctx, c, group, n, 0xABC0 + group)
    P = bases.to_host((n, affine_words(c.cid, group)))
    S = scal.to_host((n, 4))
    K = dlogs.to_host((n, 4))
    # the generated bases really are [k_i]G (spot-check against the oracle's generator multiplication)
    for i in (0, 1, n - 1):
        want = oracle.jac_to_affine(c.cid, group, oracle.generator_mul(c.cid, group, int(K[i, 0])))
        assert np.array_equal(P[i], want)
    got = oracle.jac_to_affine(c.cid, group, ecc.MultiExp(gpu_ctx, c.name, group, bases, scal, n=n))
    want = oracle.jac_to_affine(c.cid, group, oracle.msm(c.cid, group, P, S, nthreads=8))
    assert np.array_equal(got, want)
    assert np.array_equal(got, _expect_from_dlogs(c, group, S, K))
    # host-pointer path (what a cgo caller passes) gives the same point
    got2 = oracle.jac_to_affine(c.cid, group, ecc.MultiExp(gpu_ctx, c.name, group, P, S))
    assert np.array_equal(got2, want)
    for b in (bases, dlogs, scal):
        b.free()


def test_emu_groth16_sharded_key_with_commitments(emu_ctx):
    """base-range sharded key (multi-GPU partition B) on the circuit with BSB22 commitments: the K filter is sliced per shard;
    summed partials + finish give the oracle's proof"""
    c = BN254
    rng = pyref.Xoshiro(99)
    cs = pyref.commit_r1cs()
    pk, _, _ = pyref.groth16_setup(c, cs, [rng.field(c.r) for _ in range(8)])
    w = pyref.commit_solve(c, cs, 4, 9, lambda i, ww: pyref.commitment_hint(pk, cs, i, ww)[1])
    r, s_ = rng.field(c.r), rng.field(c.r)
    A, B, Cc = pyref.r1cs_solve(c, cs, w)
    sol = groth16.Solution(W=fr_to_arr(c, w), A=fr_to_arr(c, A), B=fr_to_arr(c, B), C=fr_to_arr(c, Cc))
    removed = sorted({j for cm in cs.commitments for j in cm.private_committed} | {cm.commitment_index for cm in cs.commitments})
    kw = dict(domain_cardinality=pk.n, alpha1=pts_to_arr(c, 0, [pk.alpha1]), beta1=pts_to_arr(c, 0, [pk.beta1]),
              delta1=pts_to_arr(c, 0, [pk.delta1]), A=pts_to_arr(c, 0, pk.A), B=pts_to_arr(c, 0, pk.B), Z=pts_to_arr(c, 0, pk.Z),
              K=pts_to_arr(c, 0, pk.K), beta2=pts_to_arr(c, 1, [pk.beta2]), delta2=pts_to_arr(c, 1, [pk.delta2]),
              B2=pts_to_arr(c, 1, pk.B2), infinityA=pk.infinityA, infinityB=pk.infinityB, k_remove=removed,
              commitment_keys=[(pts_to_arr(c, 0, b), pts_to_arr(c, 0, e)) for b, e in pk.commitment_keys])
    world, parts, keys = 2, [], []
    for k in range(world):
        dpk = groth16.ProvingKey(emu_ctx, c.name, shard=(k, world), **kw)
        keys.append(dpk)
        parts.append(groth16.ProvePartial(dpk, sol, cs.nb_public))
    proof = groth16.Finish(keys[0], groth16.SumPartials(c.name, parts, lib=emu_ctx.lib), fr_to_arr(c, [r]), fr_to_arr(c, [s_]))
    for dpk in keys:
        dpk.FreeGPUResources()
    ar, bs, krs, _, _ = pyref.groth16_prove_bsb22(pk, cs, w, r, s_)
    assert (arr_to_g1_affine(c, proof.Ar), arr_to_g2_affine(c, proof.Bs), arr_to_g1_affine(c, proof.Krs)) == (ar, bs, krs)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n", [1, 2, 7, 150])
def test_emu_kzg_open(emu_ctx, c, n, srs_len=None):
    """kzg.Open over a pinned monomial SRS with known tau (test/unsafekzg-style, kzgsrs.go:186-200): the device division by
    (X - z) + table MSM against the oracle's Horner recurrence, for a random point, z = 0 and a polynomial shorter than the SRS;
    and H = [(p(tau) - p(z)) / (tau - z)]G in the exponent"""
    mod = c.r
    rng = pyref.Xoshiro(1000 + n)
    tau = rng.field(mod)
    srs_len = srs_len or max(n, 4)
    G1 = group_of(c, 0)
    taus = [pow(tau, i, mod) for i in range(srs_len)]
    if srs_len <= 512:
        srs_pts = [G1.mul(c.g1, t) for t in taus]
        srs_arr = pts_to_arr(c, 0, srs_pts)
    else:   # large SRS: the C oracle multiplies the generator
        srs_arr = np.stack([oracle.jac_to_affine(c.cid, 0, oracle.generator_mul(c.cid, 0, t)) for t in taus])
        srs_pts = None
    srs = ecc.PrecomputedBases(emu_ctx, c.name, 0, srs_arr)
    try:
        for z in (rng.field(mod), 0, 1):
            poly = [rng.field(mod) for _ in range(n)]
            val, H = srs.KzgOpen(fr_to_arr(c, poly), fr_to_arr(c, [z]))
            pz = pyref._poly_eval(poly, z, mod)
            assert arr_to_fr(c, val.reshape(1, 4))[0] == pz
            got = jac_to_affine_py(c, 0, H)
            if n > 1:
                qt = (pyref._poly_eval(poly, tau, mod) - pz) * pow((tau - z) % mod, -1, mod) % mod
                assert got == G1.mul(c.g1, qt)
            if srs_pts is not None:
                assert got == pyref.kzg_open(c, srs_pts, poly, z)[1]
        with pytest.raises(Exception, match="at least"):
            srs.KzgOpen(fr_to_arr(c, [1] * (srs_len + 2)), fr_to_arr(c, [5]))
    finally:
        srs.free()


def test_emu_abi_exception_barrier(emu_ctx, monkeypatch, c=BN254, n=300):
    """No C++ exception crosses the C ABI (the reference turns device errors into Go errors, icicle.go:122-208; an exception unwinding
    into cgo would abort the process).  GA_FAULT_THROW=<entry point> makes the next device-scratch request under that entry point
    throw std::bad_alloc deep inside the call; every entry point is a function-try-block, so the call returns GA_ERR_NOMEM (-3), the
    locks / lanes / staged buffers are released by the unwinding, and the SAME call on the same context succeeds right after with
    the result it had before."""
    ctx = emu_ctx
    rng = pyref.Xoshiro(77)
    bases, dlogs, scal = _device_inputs(ctx, c, 0, n, 0xE1C)
    table = ecc.PrecomputedBases(ctx, c.name, 0, bases, n=n)
    dom = fft.Domain(ctx, c.name, 64)
    T = _plonk_case(c, 16, 5, 1)
    d0, d1 = fft.Domain(ctx, c.name, 16), fft.Domain(ctx, c.name, 4 * 16)
    polys = {k: fr_to_arr(c, T["can"][k]) for k in plonk.IDS}
    kw = dict(bp={k: fr_to_arr(c, v) for k, v in T["bp"].items()}, alpha=fr_to_arr(c, [T["alpha"]]), beta=fr_to_arr(c, [T["beta"]]),
              gamma=fr_to_arr(c, [T["gamma"]]))
    ppk = plonk.ProvingKey(d0, d1, {k: polys[k] for k in plonk.FIXED_IDS}, [fr_to_arr(c, T["qc_can"][0])])
    v = fr_to_arr(c, [rng.field(c.r) for _ in range(64)])
    poly = fr_to_arr(c, [rng.field(c.r) for _ in range(n)])
    z = fr_to_arr(c, [rng.field(c.r)])
    A = fr_to_arr(c, [rng.field(c.r) for _ in range(60)])

    def fft_host():
        w = v.copy()
        dom.FFT(w, fft.DIF)
        return w
    calls = {
        "ga_msm": lambda: ecc.MultiExp(ctx, c.name, 0, bases, scal, n=n),
        "ga_msm_table_run": lambda: table.MultiExp(scal),
        "ga_msm_table_run_batch": lambda: table.MultiExpBatch([scal, scal]),
        "ga_fft": fft_host,
        "ga_compute_h": lambda: dom.compute_h(A, A, oracle.fr_mul(c.cid, A, A)),
        "ga_plonk_quotient_pinned": lambda: ppk.ComputeQuotient({k: polys[k] for k in plonk.PROOF_IDS}, [fr_to_arr(c, T["pi_can"][0])], **kw),
        "ga_plonk_quotient": lambda: plonk.ComputeQuotient(d0, d1, polys, [fr_to_arr(c, T["qc_can"][0])], [fr_to_arr(c, T["pi_can"][0])], **kw),
        "ga_kzg_open": lambda: np.concatenate([np.asarray(x).ravel() for x in table.KzgOpen(poly, z)]),
        "ga_fr_batch_invert": lambda: plonk.BatchInvert(ctx, c.name, v),
    }
    norm = lambda name, r: ecc.jac_to_affine(c.cid, 0, r, lib=ctx.lib) if name in ("ga_msm", "ga_msm_table_run") else (
        np.stack([ecc.jac_to_affine(c.cid, 0, x, lib=ctx.lib) for x in r]) if name == "ga_msm_table_run_batch" else np.asarray(r))
    try:
        for name, call in calls.items():
            want = norm(name, call())
            monkeypatch.setenv("GA_FAULT_THROW", name)
            with pytest.raises(Exception, match=r"error -3: out of host memory \(std::bad_alloc\) under " + name):
                call()
            monkeypatch.setenv("GA_FAULT_THROW", "ga_some_other_entry_point")   # a fault armed for another entry point does not fire here
            assert np.array_equal(norm(name, call()), want), name
            monkeypatch.delenv("GA_FAULT_THROW")
            assert np.array_equal(norm(name, call()), want), name
    finally:
        monkeypatch.delenv("GA_FAULT_THROW", raising=False)
        ppk.close()
        for d in (dom, d0, d1):
            d.close()
        table.free()
        for b in (bases, dlogs, scal):
            b.free()


def test_emu_abi_exception_barrier_groth16(emu_ctx, monkeypatch, c=BN254):
    """the same for a proof: ga_g16_prove with a throwing scratch request returns GA_ERR_NOMEM, releases its slot, lanes and helper
    thread, and the next proof on the same key is byte-identical to the one before the fault"""
    from gnark_amd import synth
    ctx = emu_ctx
    inst = synth.make_instance(ctx, c.cid, 7, 0xFA17, want_dlogs=False)
    pk = inst.proving_key(ctx, precompute=1)
    try:
        args = (pk, inst.solution, inst.nb_public, inst.r, inst.s)
        before = groth16.Prove(*args).raw()
        monkeypatch.setenv("GA_FAULT_THROW", "ga_g16_prove")
        with pytest.raises(Exception, match=r"error -3"):
            groth16.Prove(*args)
        monkeypatch.delenv("GA_FAULT_THROW")
        assert np.array_equal(groth16.Prove(*args).raw(), before)
    finally:
        monkeypatch.delenv("GA_FAULT_THROW", raising=False)
        pk.FreeGPUResources()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_emu_fr_linear_combination(emu_ctx, c, n=300):
    mod = c.r
    rng = pyref.Xoshiro(5)
    for k in (1, 3, 16):
        vs = [[rng.field(mod) for _ in range(n)] for _ in range(k)]
        sc = [rng.field(mod) for _ in range(k)]
        got = plonk.LinearCombination(emu_ctx, c.name, fr_to_arr(c, sc), [fr_to_arr(c, v) for v in vs])
        assert arr_to_fr(c, got) == [sum(s * v[i] for s, v in zip(sc, vs)) % mod for i in range(n)]
    for m in (1, 2, 257, 1000):   # polynomial evaluation (iop.Polynomial.Evaluate)
        poly, z = [rng.field(mod) for _ in range(m)], rng.field(mod)
        got = plonk.Evaluate(emu_ctx, c.name, fr_to_arr(c, poly), fr_to_arr(c, [z]))
        assert arr_to_fr(c, got.reshape(1, 4))[0] == pyref._poly_eval(poly, z, mod)
    with pytest.raises(Exception, match="1..16"):
        plonk.LinearCombination(emu_ctx, c.name, fr_to_arr(c, [1] * 17), [fr_to_arr(c, [1, 2])] * 17)


# ---- the BASELINE-size checkers (oracle/checkers.py) at emulation sizes ------------------------------------------------------
import checkers
from checkers import check_compute_h_identity, check_groth16_known_dlogs, check_plonk_quotient_identity  # noqa: F401,E402 (reused by test_gpu_parity)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_emu_groth16_known_dlogs_checker(emu_ctx, c):
    """the full-size checker itself, at 2^7 under the emulation, cross-checked against the C oracle's prover"""
    check_groth16_known_dlogs(emu_ctx, c, 7, nthreads=4, also_oracle_prover=True)


def test_emu_fr_vec_mul(emu_ctx):
    c = BN254
    rng = pyref.Xoshiro(3)
    a = fr_to_arr(c, [rng.field(c.r) for _ in range(100)])
    b = fr_to_arr(c, [rng.field(c.r) for _ in range(100)])
    out = np.zeros_like(a)
    emu_ctx.lib.check(emu_ctx.lib.ga_fr_vec_mul(emu_ctx.handle, c.cid, a.ctypes.data, b.ctypes.data, 100, out.ctypes.data, 0))
    assert np.array_equal(out, oracle.fr_mul(c.cid, a, b))


@pytest.mark.parametrize("pinned", [False, True], ids=["plain", "pinned"])
def test_emu_plonk_quotient_identity_checker(emu_ctx, pinned):
    """the full-size PLONK checker at n = 2^6 under the emulation"""
    check_plonk_quotient_identity(emu_ctx, BN254, 6, nthreads=2, pinned=pinned)


# ---- staged key construction (ga_g16_builder_*): the cgo-safe call pattern ----------------------------------------------------
def _commit_key_kwargs(c, pk, cs):
    removed = sorted({j for cm in cs.commitments for j in cm.private_committed} | {cm.commitment_index for cm in cs.commitments})
    return dict(domain_cardinality=pk.n, alpha1=pts_to_arr(c, 0, [pk.alpha1]), beta1=pts_to_arr(c, 0, [pk.beta1]),
                delta1=pts_to_arr(c, 0, [pk.delta1]), A=pts_to_arr(c, 0, pk.A), B=pts_to_arr(c, 0, pk.B), Z=pts_to_arr(c, 0, pk.Z),
                K=pts_to_arr(c, 0, pk.K), beta2=pts_to_arr(c, 1, [pk.beta2]), delta2=pts_to_arr(c, 1, [pk.delta2]),
                B2=pts_to_arr(c, 1, pk.B2), infinityA=pk.infinityA, infinityB=pk.infinityB,
                commitment_keys=[(pts_to_arr(c, 0, b), pts_to_arr(c, 0, e)) for b, e in pk.commitment_keys], k_remove=removed)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("precompute", [1, -1], ids=["tables", "no-tables"])
def test_emu_groth16_staged_builder(emu_ctx, c, precompute):
    """the key streamed through ga_g16_builder_* in chunks of 3 points (one flat pointer per call, the source buffer wiped after
    every call) gives the same proofs and commitments as the struct-of-pointers ga_g16_pk_create, unsharded and as shard 1 of 3"""
    rng = pyref.Xoshiro(77)
    cs = pyref.commit_r1cs()
    pk, _, _ = pyref.groth16_setup(c, cs, [rng.field(c.r) for _ in range(8)])
    w = pyref.commit_solve(c, cs, 5, 6, lambda i, ww: pyref.commitment_hint(pk, cs, i, ww)[1])
    A, B, Cc = pyref.r1cs_solve(c, cs, w)
    sol = groth16.Solution(W=fr_to_arr(c, w), A=fr_to_arr(c, A), B=fr_to_arr(c, B), C=fr_to_arr(c, Cc))
    r, s = fr_to_arr(c, [rng.field(c.r)]), fr_to_arr(c, [rng.field(c.r)])
    kw = _commit_key_kwargs(c, pk, cs)
    vals = fr_to_arr(c, [w[j] for j in cs.commitments[0].private_committed])
    out = {}
    for name, extra in (("struct", {}), ("staged", {"staged_chunk": 3})):
        dpk = groth16.ProvingKey(emu_ctx, c.name, precompute=precompute, **kw, **extra)
        try:
            out[name] = (groth16.Prove(dpk, sol, cs.nb_public, r, s).raw(), dpk.Commit(0, vals))
        finally:
            dpk.FreeGPUResources()
        dpk = groth16.ProvingKey(emu_ctx, c.name, precompute=precompute, shard=(1, 3), **kw, **extra)
        try:
            out[name + "-shard"] = groth16.ProvePartial(dpk, sol, cs.nb_public)
        finally:
            dpk.FreeGPUResources()
    assert np.array_equal(out["struct"][0], out["staged"][0])
    assert np.array_equal(out["struct"][1][0], out["staged"][1][0]) and np.array_equal(out["struct"][1][1], out["staged"][1][1])
    assert np.array_equal(out["struct-shard"], out["staged-shard"])
    ar, bs, krs, _, _ = pyref.groth16_prove_bsb22(pk, cs, w, arr_to_fr(c, r)[0], arr_to_fr(c, s)[0])
    fp = c.fp_limbs
    assert (arr_to_g1_affine(c, out["staged"][0][:2 * fp]), arr_to_g2_affine(c, out["staged"][0][2 * fp:6 * fp]),
            arr_to_g1_affine(c, out["staged"][0][6 * fp:])) == (ar, bs, krs)


def test_emu_groth16_builder_errors(emu_ctx):
    """state machine of the builder: append before reserve, overflow, finish on an incomplete key, null arguments"""
    import ctypes as C
    lib, c = emu_ctx.lib, BN254
    b = C.c_void_p()
    assert lib.ga_g16_builder_create(emu_ctx.handle, 7, 4, 5, 0, 1, C.byref(b)) != 0            # unknown curve
    assert lib.ga_g16_builder_create(emu_ctx.handle, c.cid, 4, 5, 3, 3, C.byref(b)) != 0        # shard_index >= shard_count
    lib.check(lib.ga_g16_builder_create(emu_ctx.handle, c.cid, 4, 5, 0, 1, C.byref(b)))
    pts = pts_to_arr(c, 0, [c.g1] * 4)
    assert lib.ga_g16_builder_append(b, 0, pts.ctypes.data, 1) == -4                             # GA_ERR_STATE: not reserved
    lib.check(lib.ga_g16_builder_reserve(b, 0, 3))
    assert lib.ga_g16_builder_reserve(b, 0, 3) == -4                                             # reserved twice
    assert lib.ga_g16_builder_reserve(b, 9, 3) == -1
    lib.check(lib.ga_g16_builder_append(b, 0, pts.ctypes.data, 2))
    assert lib.ga_g16_builder_append(b, 0, pts.ctypes.data, 2) == -1                             # overflows the reserved length
    assert lib.ga_g16_builder_append(b, 0, None, 1) == -1
    assert lib.ga_g16_builder_set_point(b, 7, pts.ctypes.data) == -1
    mask = np.zeros(4, dtype=np.uint8)
    assert lib.ga_g16_builder_set_infinity(b, 0, mask.ctypes.data, 4) == -1                      # nb_wires is 5
    h = C.c_void_p()
    assert lib.ga_g16_builder_finish(b, 0, C.byref(h)) == -4 and not h.value                     # incomplete; the builder is consumed
    assert b"incomplete" in lib.ga_last_error()
    b2 = C.c_void_p()
    lib.check(lib.ga_g16_builder_create(emu_ctx.handle, c.cid, 4, 5, 0, 1, C.byref(b2)))
    lib.ga_g16_builder_destroy(b2)                                                               # abandon: frees the staged buffers


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_emu_groth16_tiny_domains_vs_oracle(emu_ctx, c):
    """the smallest circuits (domain 2, 4, 8; one public wire; duplicate infinity positions): window plans with a single point,
    transforms of one and two stages, empty K slices -- with and without tables, against the C oracle's prover"""
    from gnark_amd import synth
    for logn in (1, 2, 3):
        inst = synth.make_instance(emu_ctx, c.name, logn, 0x77, want_dlogs=False, nb_public=1)
        key = dict(inst.key, n=inst.n)
        sol = inst.solution
        want = oracle.groth16_prove(c.cid, key, sol.W, sol.A, sol.B, sol.C, inst.nb_public, inst.r, inst.s, nthreads=1)
        want = np.concatenate([np.asarray(w).reshape(-1) for w in want])
        for pre in (1, -1):
            pk = inst.proving_key(emu_ctx, precompute=pre)
            try:
                assert np.array_equal(groth16.Prove(pk, sol, inst.nb_public, inst.r, inst.s).raw(), want), (logn, pre)
            finally:
                pk.FreeGPUResources()


def test_emu_groth16_two_keys_two_curves_interleaved(emu_ctx, logn_a=6, logn_b=8, rounds=3):
    """Two pinned keys of different sizes AND different curves on one context, proved alternately from two host threads (each
    thread switches key every proof): the context's scratch is shared by name across keys and grows / is reused across element
    sizes, the lanes are taken by whichever caller comes first -- every proof must equal the proof of its key computed alone."""
    import threading
    from gnark_amd import synth
    ia = synth.make_instance(emu_ctx, BN254.name, logn_a, 0x2B01, want_dlogs=False)
    ib = synth.make_instance(emu_ctx, BLS12_381.name, logn_b, 0x2B02, want_dlogs=False)
    pka, pkb = ia.proving_key(emu_ctx, precompute=1), ib.proving_key(emu_ctx, precompute=-1)
    try:
        jobs = [(pka, ia), (pkb, ib)]
        want = [groth16.Prove(pk, i.solution, i.nb_public, i.r, i.s).raw() for pk, i in jobs]
        bad = []

        def prover(tid):
            for k in range(rounds):
                j = (tid + k) % 2
                pk, i = jobs[j]
                if not np.array_equal(groth16.Prove(pk, i.solution, i.nb_public, i.r, i.s).raw(), want[j]):
                    bad.append((tid, k, j))

        th = [threading.Thread(target=prover, args=(t,)) for t in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not bad, bad
    finally:
        pka.FreeGPUResources()
        pkb.FreeGPUResources()


# ---- two proofs in flight on one context (lanes) with DIFFERENT solutions ---------------------------------------------------------
@pytest.mark.parametrize("c,precompute", [(BN254, 1), (BLS12_381, -1)], ids=["bn254-tables", "bls12-381-no-tables"])
def test_emu_groth16_two_callers_distinct_solutions(emu_ctx, c, precompute, logn=7, rounds=2):
    """Two host threads prove DIFFERENT solutions on one key at the same time (the second caller runs on lane 1 of the context,
    groth16.hip ga_g16_prove): every proof must equal the proof of ITS solution computed alone -- identical inputs on both
    threads would hide a buffer shared between the lanes.  A third thread keeps the context busy with transforms."""
    import threading
    from gnark_amd import synth
    inst = synth.make_instance(emu_ctx, c.name, logn, 0x4C41, nb_constraints=(1 << logn) - 3)
    others = [synth.make_instance(emu_ctx, c.name, logn, 0x4C42 + k, nb_constraints=(1 << logn) - 3, want_dlogs=False) for k in range(2)]
    sols = [inst.solution] + [o.solution for o in others]          # same key (inst's), three unrelated solutions
    rs = [(inst.r, inst.s)] + [(o.r, o.s) for o in others]
    pk = inst.proving_key(emu_ctx, precompute=precompute)
    dom = fft.Domain(emu_ctx, c.name, inst.n)
    try:
        want = [groth16.Prove(pk, sol, inst.nb_public, r, s_).raw() for sol, (r, s_) in zip(sols, rs)]
        assert not np.array_equal(want[0], want[1]) and not np.array_equal(want[1], want[2])
        fft_in = np.zeros((inst.n, 4), dtype=np.uint64)
        fft_in[: sols[0].A.shape[0]] = sols[0].A
        fft_want = dom.FFT(fft_in, fft.DIF)
        bad = []

        def prover(tid):
            for k in range(rounds):
                j = (tid + k) % len(sols)
                got = groth16.Prove(pk, sols[j], inst.nb_public, rs[j][0], rs[j][1]).raw()
                if not np.array_equal(got, want[j]):
                    bad.append((tid, k, j))

        def transformer():
            for _ in range(rounds):
                if not np.array_equal(dom.FFT(fft_in, fft.DIF), fft_want):
                    bad.append(("fft",))

        before = emu_ctx.lane_stats()
        th = [threading.Thread(target=prover, args=(t,)) for t in range(3)] + [threading.Thread(target=transformer)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not bad, bad
        after = emu_ctx.lane_stats()
        ran = sum(after[k] - before[k] for k in ("lanes01_proofs", "lanes23_proofs", "queued_proofs"))
        assert ran == 3 * rounds, (before, after)                        # every call is counted exactly once
        assert after["lanes23_scratch_bytes"] >= 0 and after["lanes01_scratch_bytes"] > 0
    finally:
        dom.close()
        pk.FreeGPUResources()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_emu_groth16_batched_witness_tables(emu_ctx, c, monkeypatch, logn=6):
    """The wire-indexed G1 tables A, B1, K go through ONE pass of the bucket kernel / merge / window reduction over the shared
    witness sort (msm_table_device_reuse_multi; prove.go:194,207,237 are three MultiExp over the same wireValues): proof points equal
    to the C oracle's prover and to the one-pass-per-table schedule (GA_G16_BATCH_TABLES=0) -- on a key with REPEATED and OPPOSITE
    bases inside A, B and K, so that each table flags its own tasks in the pass (per-table redo lists) --, with a 0/1-heavy witness
    (one very hot bucket: the merge's hot lists are shared by the tables), and on a DummySetup-like key (every base of a table the
    same point): the first proof marks the tables degenerate inside the batched pass, the second one takes them out of it."""
    ctx, lib = emu_ctx, emu_ctx.lib
    n = 1 << logn
    nw, nb_public = n, 3

    def gen(group, count, seed):
        buf = ctx.malloc(count * affine_words(c.cid, group) * 8)
        lib.check(lib.ga_gen_bases(ctx.handle, c.cid, group, seed, count, buf.ptr, None))
        h = buf.to_host((count, affine_words(c.cid, group)))
        buf.free()
        return h

    def scal(count, seed):
        buf = ctx.malloc(count * 32)
        lib.check(lib.ga_gen_scalars(ctx.handle, c.cid, seed, count, buf.ptr))
        h = buf.to_host((count, 4))
        buf.free()
        return h

    def neg_y(pt):   # -P of a G1 affine image (Montgomery limbs): y -> p - y
        fp = c.fp_limbs
        y = sum(int(v) << (64 * i) for i, v in enumerate(pt[fp:]))
        q = pt.copy()
        q[fp:] = [((c.p - y) >> (64 * i)) & (2**64 - 1) for i in range(fp)]
        return q
    infA, infB = np.zeros(nw, np.uint8), np.zeros(nw, np.uint8)
    infA[[2, nw - 1]] = 1
    infB[[4]] = 1
    m1, m2 = gen(0, 3, 21), gen(1, 2, 22)
    A, B, K = gen(0, nw - 2, 23), gen(0, nw - 1, 24), gen(0, nw - nb_public, 25)
    for V in (A, B, K):   # runs of equal points and a P, -P pair: wires with equal scalars below make them meet in one bucket
        V[10:20] = V[10]
        V[31] = neg_y(V[30])
    key = dict(n=n, alpha1=m1[0:1], beta1=m1[1:2], delta1=m1[2:3], A=A, B=B, Z=gen(0, n - 1, 26), K=K, beta2=m2[0:1], delta2=m2[1:2],
               B2=gen(1, nw - 1, 27), infinityA=infA, infinityB=infB)
    m = n - 5
    W, Av, Bv = scal(nw, 30), scal(m, 31), scal(m, 32)
    W[8:40] = W[8]                       # equal scalars over the equal / opposite bases
    Cc = oracle.fr_mul(c.cid, Av, Bv)
    rs = scal(2, 33)
    one = fr_to_arr(c, [1])[0]
    W01 = W.copy()
    W01[nw // 4:] = one                  # a boolean-heavy witness: the digit-1 bucket of window 0 holds most of the wires
    W01[nw // 2:] = 0
    dummy = dict(key, A=np.repeat(A[:1], nw - 2, axis=0), B=np.repeat(B[:1], nw - 1, axis=0), K=np.repeat(K[:1], nw - nb_public, axis=0))
    for name, kd, wit, rounds in (("exceptional", key, W, 1), ("boolean", key, W01, 1), ("dummy", dummy, W, 2)):
        want = oracle.groth16_prove(c.cid, kd, wit, Av, Bv, Cc, nb_public, rs[0], rs[1], nthreads=8)
        pk = groth16.ProvingKey(ctx, c.name, domain_cardinality=n, precompute=1, **{k: v for k, v in kd.items() if k != "n"})
        try:
            lay = groth16.ShardLayout(pk)
            assert lay["wire_indexed"]["A"] and lay["wire_indexed"]["B"] and lay["wire_indexed"]["K"], lay
            for batched in ["1"] * rounds + ["0"]:
                monkeypatch.setenv("GA_G16_BATCH_TABLES", batched)
                proof = groth16.Prove(pk, groth16.Solution(wit, Av, Bv, Cc), nb_public, rs[0], rs[1])
                assert np.array_equal(proof.Ar, want[0]) and np.array_equal(proof.Bs, want[1]) and np.array_equal(proof.Krs, want[2]), (name, batched)
        finally:
            monkeypatch.delenv("GA_G16_BATCH_TABLES", raising=False)
            pk.FreeGPUResources()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_emu_groth16_prove_oneshot(emu_ctx, c, logn=6):
    """ga_g16_prove_oneshot -- the key uploaded by a helper thread WHILE the proof runs, every MSM waiting for its own vector, the
    key dropped afterwards (the reference's default PinToGPU = false, icicle.go:797-805): proof bytes identical to pinning the
    plain vectors, proving and freeing, and to the pinned-with-tables proof; two callers at once on one context; bad keys are
    errors that leave the context usable; the examples/cubic bytes of the oracle."""
    import threading
    from gnark_amd import synth
    inst = synth.make_instance(emu_ctx, c.name, logn, 0x4F53, nb_constraints=(1 << logn) - 3)
    other = synth.make_instance(emu_ctx, c.name, logn, 0x4F54, nb_constraints=(1 << logn) - 3, want_dlogs=False)
    pk = inst.proving_key(emu_ctx, precompute=-1)
    pkt = inst.proving_key(emu_ctx, precompute=1)
    try:
        want = groth16.Prove(pk, inst.solution, inst.nb_public, inst.r, inst.s).raw()
        assert np.array_equal(groth16.Prove(pkt, inst.solution, inst.nb_public, inst.r, inst.s).raw(), want)
        want_other = groth16.Prove(pk, other.solution, inst.nb_public, other.r, other.s).raw()
    finally:
        pk.FreeGPUResources()
        pkt.FreeGPUResources()
    for _ in range(2):   # (the second call gets the first one's NTT domain back from the context)
        assert np.array_equal(inst.prove_oneshot(emu_ctx).raw(), want)
    assert np.array_equal(inst.prove_oneshot(emu_ctx, other.solution, other.r, other.s).raw(), want_other)
    bad = []

    def caller(k):
        for j in range(2):
            sol, r, s_, w = ((inst.solution, inst.r, inst.s, want), (other.solution, other.r, other.s, want_other))[(k + j) % 2]
            if not np.array_equal(inst.prove_oneshot(emu_ctx, sol, r, s_).raw(), w):
                bad.append((k, j))
    th = [threading.Thread(target=caller, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad, bad
    # a key whose masks disagree with its vectors: an error (after the buffers were reserved), nothing left behind, context usable
    broken = dict(inst.key, infinityA=np.zeros_like(inst.key["infinityA"]))
    with pytest.raises(Exception, match="Infinity|len"):
        groth16.ProveOneShot(emu_ctx, c.name, inst.solution, inst.nb_public, inst.r, inst.s, domain_cardinality=inst.n, **broken)
    with pytest.raises(Exception, match="nbWires|len\\(W\\)"):
        groth16.ProveOneShot(emu_ctx, c.name, groth16.Solution(inst.solution.W[:-1], inst.solution.A, inst.solution.B, inst.solution.C),
                             inst.nb_public, inst.r, inst.s, domain_cardinality=inst.n, **inst.key)
    assert np.array_equal(inst.prove_oneshot(emu_ctx).raw(), want)
    # examples/cubic (BASELINE config 1): the bytes of the oracle's prover
    rng = pyref.Xoshiro(5)
    cs, w = pyref.cubic_r1cs(), pyref.cubic_witness(3)
    opk, _, _ = pyref.groth16_setup(c, cs, [rng.field(c.r) for _ in range(5)])
    r, s_ = rng.field(c.r), rng.field(c.r)
    a_, b_, c_ = pyref.r1cs_solve(c, cs, w)
    proof = groth16.ProveOneShot(
        emu_ctx, c.name, groth16.Solution(fr_to_arr(c, w), fr_to_arr(c, a_), fr_to_arr(c, b_), fr_to_arr(c, c_)), cs.nb_public,
        fr_to_arr(c, [r]), fr_to_arr(c, [s_]), domain_cardinality=opk.n, alpha1=pts_to_arr(c, 0, [opk.alpha1]),
        beta1=pts_to_arr(c, 0, [opk.beta1]), delta1=pts_to_arr(c, 0, [opk.delta1]), A=pts_to_arr(c, 0, opk.A), B=pts_to_arr(c, 0, opk.B),
        Z=pts_to_arr(c, 0, opk.Z), K=pts_to_arr(c, 0, opk.K), beta2=pts_to_arr(c, 1, [opk.beta2]), delta2=pts_to_arr(c, 1, [opk.delta2]),
        B2=pts_to_arr(c, 1, opk.B2), infinityA=opk.infinityA, infinityB=opk.infinityB)
    assert proof.WriteTo() == pyref.proof_bytes(c, *pyref.groth16_prove(opk, cs, w, r, s_))


def test_emu_groth16_second_caller_without_memory_queues_instead_of_failing(emu_ctx, monkeypatch, logn=6, rounds=2):
    """precompute = 0 fills the HBM with tables beside ONE caller's scratch; a second concurrent ga_g16_prove caller is sent
    to lanes 2/3, whose scratch may then not fit.  GA_FAULT_LANE2_NOMEM makes every scratch request of lanes 2/3 fail as if HBM were
    exhausted: such a caller must give back what lanes 2/3 hold and queue for the device -- its proof is the right one, never an
    error -- and be counted as queued, not as a lane-2 proof."""
    import threading
    from gnark_amd import synth
    c = BN254
    inst = synth.make_instance(emu_ctx, c.name, logn, 0x4E4F, nb_constraints=(1 << logn) - 3)
    other = synth.make_instance(emu_ctx, c.name, logn, 0x4E50, nb_constraints=(1 << logn) - 3, want_dlogs=False)
    sols, rs = [inst.solution, other.solution], [(inst.r, inst.s), (other.r, other.s)]
    pk = inst.proving_key(emu_ctx, precompute=1)
    try:
        want = [groth16.Prove(pk, sol, inst.nb_public, r, s_).raw() for sol, (r, s_) in zip(sols, rs)]
        monkeypatch.setenv("GA_FAULT_LANE2_NOMEM", "1")
        groth16.Prove(pk, sols[0], inst.nb_public, *rs[0])             # (a lane-0 call reads the knobs)
        bad, errs = [], []

        def prover(tid):
            try:
                for k in range(rounds):
                    j = (tid + k) % 2
                    if not np.array_equal(groth16.Prove(pk, sols[j], inst.nb_public, rs[j][0], rs[j][1]).raw(), want[j]):
                        bad.append((tid, k, j))
            except Exception as e:   # noqa: BLE001 -- the point of the test: no caller may see GA_ERR_NOMEM
                errs.append(repr(e))

        before = emu_ctx.lane_stats()
        th = [threading.Thread(target=prover, args=(t,)) for t in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        after = emu_ctx.lane_stats()
        assert not errs, errs
        assert not bad, bad
        assert after["lanes23_proofs"] == before["lanes23_proofs"]          # nobody proved on lanes 2/3 ...
        assert after["lanes23_scratch_bytes"] == 0                           # ... and what they held was given back
        ran = sum(after[k] - before[k] for k in ("lanes01_proofs", "lanes23_proofs", "queued_proofs"))
        assert ran == 3 * rounds, (before, after)                            # every call counted exactly once
        monkeypatch.delenv("GA_FAULT_LANE2_NOMEM")
        groth16.Prove(pk, sols[0], inst.nb_public, *rs[0])
    finally:
        monkeypatch.delenv("GA_FAULT_LANE2_NOMEM", raising=False)
        pk.FreeGPUResources()


@pytest.mark.parametrize("c,precompute", [(BN254, 1), (BLS12_381, -1)], ids=["bn254-tables", "bls12-381-no-tables"])
def test_emu_groth16_split_schedule_and_late_free(emu_ctx, c, precompute, monkeypatch, logn=7):
    """A proof's H side (computeH, Z MSM, possibly K) runs on the partner lane from a helper thread (prove_partial): the proof must
    equal the single-lane one (GA_G16_SPLIT=0), and ga_g16_lane_stats must say the split happened.
    Then FreeGPUResources from one thread while another is still proving on the key: the destroy waits (use-after-free)."""
    import threading
    from gnark_amd import synth
    inst = synth.make_instance(emu_ctx, c.name, logn, 0x5350, nb_constraints=(1 << logn) - 1)
    pk = inst.proving_key(emu_ctx, precompute=precompute)
    s0 = emu_ctx.lane_stats()
    split = groth16.Prove(pk, inst.solution, inst.nb_public, inst.r, inst.s).raw()
    s1 = emu_ctx.lane_stats()
    assert s1["split_proofs"] == s0["split_proofs"] + 1 and s1["lanes01_proofs"] == s0["lanes01_proofs"] + 1
    monkeypatch.setenv("GA_G16_SPLIT", "0")
    single = groth16.Prove(pk, inst.solution, inst.nb_public, inst.r, inst.s).raw()
    assert emu_ctx.lane_stats()["split_proofs"] == s1["split_proofs"]
    monkeypatch.delenv("GA_G16_SPLIT")
    assert np.array_equal(split, single)   # (both also equal the known-dlog closed form: test_emu_groth16_known_dlogs_checker)
    # free while proving: the prover thread is inside ga_g16_prove when the main thread calls ga_g16_pk_destroy
    out, started = [], threading.Event()

    refused = []

    def prover():
        started.set()
        try:
            out.append(groth16.Prove(pk, inst.solution, inst.nb_public, inst.r, inst.s).raw())
        except Exception as e:   # the key died first: an error code, not a crash
            refused.append(str(e))

    t = threading.Thread(target=prover)
    t.start()
    started.wait()
    pk.FreeGPUResources()
    t.join()
    # either the proof was already in flight (the destroy waited and the proof is right) or it started after the key died (refused)
    assert (out and np.array_equal(out[0], split)) or (refused and ("destroyed" in refused[0] or "freed" in refused[0])), (len(out), refused)


# ---- one proof over several devices from one process (ga_g16_prove_multi) ------------------------------------------------------
# (under the emulation a subset of the curve x shard-count x key-mode grid, sized for the CPU suite; the GPU suite runs both curves
# with 2 and 3 shards and both key modes at 2^12 constraints)
@pytest.mark.parametrize("c,nshards,precompute", [(BN254, 2, 1), (BN254, 3, -1), (BLS12_381, 3, 1), (BLS12_381, 5, -1)],
                         ids=["bn254-2-tables", "bn254-3-no-tables", "bls12-381-3-tables", "bls12-381-5-no-tables"])
def test_emu_groth16_prove_multi(emu_ctx, c, nshards, precompute, logn=7):
    """shard i of N in its own context (one context per device; here all on device 0): the native multi-device prover -- one
    host thread per shard, computeH's chains on the first three, peer copies of b, c and of the h slices -- returns the proof of
    the unsharded key, which equals the known-dlog closed form; also the pieces API (witness / chain / combine / z) by hand"""
    from gnark_amd import synth
    from gnark_amd.device import Context
    if logn == 7 and c is BLS12_381 and precompute == 1:
        logn = 6   # (the 381-bit table build dominates this case under the emulation)
    inst = synth.make_instance(emu_ctx, c.name, logn, 0x77 + nshards, nb_constraints=(1 << logn) - 5)
    sol = inst.solution
    pk1 = inst.proving_key(emu_ctx, precompute=precompute)
    try:
        want = groth16.Prove(pk1, sol, inst.nb_public, inst.r, inst.s)
    finally:
        pk1.FreeGPUResources()
    ctxs = [Context(0, lib=emu_ctx.lib) for _ in range(nshards)]
    pks = []
    try:
        for i, cx in enumerate(ctxs):
            pks.append(inst.proving_key(cx, precompute=precompute, shard=(i, nshards), staged_chunk=50))
        got = groth16.ProveMulti(pks, sol, inst.nb_public, inst.r, inst.s)
        assert np.array_equal(got.raw(), want.raw())
        # wire ranges: together they cover what the bases need, each about 1/N of W
        lays = [groth16.ShardLayout(p) for p in pks]
        assert lays[0]["off_z"] == 0 and lays[-1]["off_z"] + lays[-1]["len_z"] == inst.n - 1
        assert all(l["w_hi"] - l["w_lo"] <= inst.nb_wires // nshards + 4 for l in lays)
        # the same proof assembled by hand from the pieces, h on shard 0 only
        n = inst.n
        bufs = [ctxs[0].malloc(n * 32) for _ in range(3)]
        for v, b in zip((sol.A, sol.B, sol.C), bufs):
            groth16.HChain(pks[0], v, b.ptr)
        groth16.HCombine(pks[0], bufs[0].ptr, bufs[1].ptr, bufs[2].ptr)
        h = bufs[0].to_host((n, 4))
        parts = []
        for p, cx, lay in zip(pks, ctxs, lays):
            hs = cx.to_device(h[lay["off_z"]: lay["off_z"] + lay["len_z"]])
            wpart = groth16.WitnessPartial(p, sol.W, inst.nb_public)
            z = groth16.ZPartial(p, hs.ptr)
            fp = c.fp_limbs
            wpart[6 * fp: 9 * fp] = ecc.jac_add(c.name, 0, np.ascontiguousarray(wpart[6 * fp: 9 * fp]), z, lib=emu_ctx.lib)
            parts.append(wpart)
            hs.free()
        byhand = groth16.Finish(pks[0], groth16.SumPartials(c.name, parts, lib=emu_ctx.lib), inst.r, inst.s)
        assert np.array_equal(byhand.raw(), want.raw())
        for b in bufs:
            b.free()
        # argument validation: shards out of order, shared context
        with pytest.raises(Exception, match="must be shard"):
            groth16.ProveMulti(pks[::-1], sol, inst.nb_public, inst.r, inst.s)
        with pytest.raises(Exception, match="one share of a sharded key"):
            groth16.Prove(pks[0], sol, inst.nb_public, inst.r, inst.s)
        # partition A (BASELINE config 4 "window-sharded"): whole key on every device, share i of the windows of every MSM
        for p in pks:
            p.FreeGPUResources()
        pks = [inst.proving_key(cx, precompute=precompute, window_shard=(i, nshards), staged_chunk=(64 if i % 2 else 0)) for i, cx in enumerate(ctxs)]
        lays = [groth16.ShardLayout(p) for p in pks]
        assert all(l["len_z"] == inst.n - 1 and l["off_z"] == 0 and l["win_count"] == nshards for l in lays)
        gotw = groth16.ProveMulti(pks, sol, inst.nb_public, inst.r, inst.s)
        assert np.array_equal(gotw.raw(), want.raw())
        with pytest.raises(Exception, match="cannot be combined"):
            inst.proving_key(ctxs[0], shard=(0, 2), window_shard=(0, 2))
    finally:
        for p in pks:
            p.FreeGPUResources()
        for cx in ctxs:
            cx.close()
    # the closed form from the key's discrete logs (oracle dot products)
    d = fft.Domain(emu_ctx, c.name, inst.n)
    try:
        hh = d.compute_h(sol.A, sol.B, sol.C)
    finally:
        d.close()
    exp = synth.expected_exponents(inst, hh, lambda a, b: oracle.fr_dot(c.cid, a, b))
    pt = lambda group, k: oracle.jac_to_affine(c.cid, group, oracle.generator_mul(c.cid, group, k))
    assert np.array_equal(got.Ar, pt(0, exp["Ar"])) and np.array_equal(got.Bs, pt(1, exp["Bs"])) and np.array_equal(got.Krs, pt(0, exp["Krs"]))


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("group", [0, 1], ids=["G1", "G2"])
def test_emu_msm_table_window_ranges(emu_ctx, c, group):
    """window ranges on the pinned-table path (multi-GPU partition A on the fast path): the partial results of disjoint ranges
    ADD UP to the full MSM -- no Horner step, because the 2^(c*w) factors are part of the table; an empty range is infinity"""
    n = 300
    bases, dlogs, scal = _device_inputs(emu_ctx, c, group, n, 0x71AB + group)
    t = ecc.PrecomputedBases(emu_ctx, c.name, group, bases, n=n)
    try:
        nwin = t.info()["windows"]
        full = oracle.jac_to_affine(c.cid, group, t.MultiExp(scal))
        assert np.array_equal(full, _expect_from_dlogs(c, group, scal.to_host((n, 4)), dlogs.to_host((n, 4))))
        for cuts in ((0, nwin // 3, nwin), (0, 1, 2, nwin - 1, nwin), (0, 0, nwin)):
            acc = None
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                part = t.MultiExpWindows(scal, lo, hi)
                acc = part if acc is None else ecc.jac_add(c.name, group, acc, part, lib=emu_ctx.lib)
            assert np.array_equal(oracle.jac_to_affine(c.cid, group, acc), full), cuts
        assert not t.MultiExpWindows(scal, 2, 2)[-c.fp_limbs:].any()          # empty range: Z = 0
        with pytest.raises(Exception, match="window range"):
            t.MultiExpWindows(scal, 0, nwin + 1)
    finally:
        t.free()
        for b in (bases, dlogs, scal):
            b.free()


# ---- proving-key files and proof bytes (marshal.go:62-86,231-539) --------------------------------------------------
def _py_key_fields(c, pk):
    return dict(domain_cardinality=pk.n, alpha1=pts_to_arr(c, 0, [pk.alpha1]), beta1=pts_to_arr(c, 0, [pk.beta1]),
                delta1=pts_to_arr(c, 0, [pk.delta1]), A=pts_to_arr(c, 0, pk.A), B=pts_to_arr(c, 0, pk.B), Z=pts_to_arr(c, 0, pk.Z),
                K=pts_to_arr(c, 0, pk.K), beta2=pts_to_arr(c, 1, [pk.beta2]), delta2=pts_to_arr(c, 1, [pk.delta2]),
                B2=pts_to_arr(c, 1, pk.B2), infinityA=pk.infinityA, infinityB=pk.infinityB,
                commitment_keys=[(pts_to_arr(c, 0, b), pts_to_arr(c, 0, e)) for b, e in pk.commitment_keys])


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("circuit", ["cubic", "commit"])
def test_emu_proving_key_files(emu_ctx, c, circuit, tmp_path):
    """the three key layouts of marshal.go (WriteTo compressed, WriteRawTo, WriteDump), both ways against the oracle's
    restatement: the library's writer produces the oracle's bytes; a key READ from the oracle's bytes -- from memory and from a
    file descriptor, whole and as shard 1 of 2, with and without the withPrecompute byte of the domain block -- proves exactly like
    the key built from the host arrays (proof bytes == oracle)."""
    rng = pyref.Xoshiro(31337)
    cs = pyref.cubic_r1cs() if circuit == "cubic" else pyref.commit_r1cs()
    pk, _, _ = pyref.groth16_setup(c, cs, [rng.field(c.r) for _ in range(5 + len(cs.commitments) + 1)])
    if circuit == "cubic":
        w = pyref.cubic_witness(3)
        removed = []
    else:
        w = pyref.commit_solve(c, cs, 4, 9, lambda i, ww: pyref.commitment_hint(pk, cs, i, ww)[1])
        removed = sorted({j for cm in cs.commitments for j in cm.private_committed} | {cm.commitment_index for cm in cs.commitments})
    A, B, Cc = pyref.r1cs_solve(c, cs, w)
    sol = groth16.Solution(W=fr_to_arr(c, w), A=fr_to_arr(c, A), B=fr_to_arr(c, B), C=fr_to_arr(c, Cc))
    r, s = fr_to_arr(c, [rng.field(c.r)]), fr_to_arr(c, [rng.field(c.r)])
    fields = _py_key_fields(c, pk)
    ref = groth16.ProvingKey(emu_ctx, c.name, k_remove=removed, **fields)
    try:
        want = groth16.Prove(ref, sol, cs.nb_public, r, s).raw()
        want_part = None
    finally:
        ref.FreeGPUResources()
    images = {groth16.KEY_FORMAT_COMPRESSED: pyref.pk_write(pk, raw=False), groth16.KEY_FORMAT_RAW: pyref.pk_write(pk, raw=True),
              groth16.KEY_FORMAT_DUMP: pyref.pk_write_dump(pk)}
    for fmt, img in images.items():
        path = tmp_path / f"key{fmt}.bin"
        with open(path, "wb") as f:                         # writer: byte parity with the oracle
            n = groth16.WriteKey(emu_ctx, c.name, f, fmt, **fields)
        assert n == len(img) and open(path, "rb").read() == img, fmt
        for source in ("mem", "fd"):
            if source == "mem":
                dpk = groth16.ProvingKey.ReadFrom(emu_ctx, c.name, img, k_remove=removed)
            else:
                with open(path, "rb") as f:
                    dpk = groth16.ProvingKey.ReadFrom(emu_ctx, c.name, f, k_remove=removed)
            try:
                assert dpk.bytes_read == len(img) and dpk.nb_wires == len(pk.infinityA)
                assert np.array_equal(groth16.Prove(dpk, sol, cs.nb_public, r, s).raw(), want), (fmt, source)
            finally:
                dpk.FreeGPUResources()
    # older gnark-crypto domain block (no withPrecompute byte), and a sharded read
    old = pyref.pk_write(pk, raw=False, with_precompute_byte=False)
    dpk = groth16.ProvingKey.ReadFrom(emu_ctx, c.name, old, k_remove=removed)
    try:
        assert dpk.bytes_read == len(old)
        assert np.array_equal(groth16.Prove(dpk, sol, cs.nb_public, r, s).raw(), want)
    finally:
        dpk.FreeGPUResources()
    parts = []
    for k in range(2):
        spk = groth16.ProvingKey.ReadFrom(emu_ctx, c.name, images[groth16.KEY_FORMAT_DUMP if k else groth16.KEY_FORMAT_COMPRESSED],
                                          shard=(k, 2), k_remove=removed)
        try:
            parts.append(groth16.ProvePartial(spk, sol, cs.nb_public))
            fin = groth16.Finish(spk, groth16.SumPartials(c.name, parts, lib=emu_ctx.lib), r, s) if k else None
        finally:
            spk.FreeGPUResources()
    assert np.array_equal(fin.raw(), want)
    # malformed input fails cleanly
    bad = bytearray(images[groth16.KEY_FORMAT_COMPRESSED])
    bad[8 + 160 + 1 + 3 * c.fp_bytes + 4 + 5] ^= 0x55        # a byte inside the first point of G1.A: (almost surely) not on the curve
    with pytest.raises(Exception, match="do not decode|does not decode"):
        groth16.ProvingKey.ReadFrom(emu_ctx, c.name, bytes(bad), k_remove=removed)
    with pytest.raises(Exception, match="end of input"):
        groth16.ProvingKey.ReadFrom(emu_ctx, c.name, images[groth16.KEY_FORMAT_RAW][:-7], k_remove=removed)
    # a vector whose FIRST point is infinity (an unused first private wire): on BN254 the infinity flag is the same byte in the
    # compressed and the raw encoding, so the stream's mode must come from [alpha]1, not from the vector's first byte
    if circuit == "cubic":
        import copy
        pk0 = copy.copy(pk)
        pk0.K = [None] + list(pk.K[1:])
        f0 = _py_key_fields(c, pk0)
        ref = groth16.ProvingKey(emu_ctx, c.name, **f0)
        try:
            want0 = groth16.Prove(ref, sol, cs.nb_public, r, s).raw()
        finally:
            ref.FreeGPUResources()
        assert not np.array_equal(want0, want)
        for raw in (False, True):
            dpk = groth16.ProvingKey.ReadFrom(emu_ctx, c.name, pyref.pk_write(pk0, raw=raw))
            try:
                assert np.array_equal(groth16.Prove(dpk, sol, cs.nb_public, r, s).raw(), want0), raw
            finally:
                dpk.FreeGPUResources()
        # an all-zero uncompressed point is accepted as infinity too (lenient reading of RawBytes)
        img = bytearray(pyref.pk_write(pk0, raw=True))
        nbp = c.fp_bytes
        k_off = 8 + 160 + 1 + 3 * 2 * nbp + sum(4 + 2 * nbp * len(v) for v in (pk0.A, pk0.B, pk0.Z)) + 4
        assert img[k_off] == 0x40 and not any(img[k_off + 1:k_off + 2 * nbp])
        img[k_off] = 0
        dpk = groth16.ProvingKey.ReadFrom(emu_ctx, c.name, bytes(img))
        try:
            assert np.array_equal(groth16.Prove(dpk, sol, cs.nb_public, r, s).raw(), want0)
        finally:
            dpk.FreeGPUResources()
        # a key whose vectors cannot belong to its wire count is refused before anything is laid out by wire id
        bad_fields = dict(fields)
        bad_fields["K"] = np.concatenate([fields["K"]] * 4)
        with pytest.raises(Exception, match="do not fit|inconsistent|len"):
            groth16.ProvingKey(emu_ctx, c.name, **bad_fields)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_emu_proving_key_file_fuzz(emu_ctx, c):
    """The key reader on damaged input (marshal.go:305-373,449-539 read untrusted files): every truncation point class and 24
    random mutations (byte flips, overwritten length words, spliced garbage) of each of the three layouts must come back as an
    error -- or as a key, when the damage happens to decode -- without crashing, hanging or allocating by an untrusted length;
    a key that does load must survive a proof call (error or proof)."""
    rng = pyref.Xoshiro(0xF022)
    cs = pyref.commit_r1cs()
    pk, _, _ = pyref.groth16_setup(c, cs, [rng.field(c.r) for _ in range(5 + len(cs.commitments) + 1)])
    w = pyref.commit_solve(c, cs, 4, 9, lambda i, ww: pyref.commitment_hint(pk, cs, i, ww)[1])
    removed = sorted({j for cm in cs.commitments for j in cm.private_committed} | {cm.commitment_index for cm in cs.commitments})
    A, B, Cc = pyref.r1cs_solve(c, cs, w)
    sol = groth16.Solution(W=fr_to_arr(c, w), A=fr_to_arr(c, A), B=fr_to_arr(c, B), C=fr_to_arr(c, Cc))
    r, s = fr_to_arr(c, [rng.field(c.r)]), fr_to_arr(c, [rng.field(c.r)])
    images = [pyref.pk_write(pk, raw=False), pyref.pk_write(pk, raw=True), pyref.pk_write_dump(pk)]
    prng = np.random.default_rng(20260924)
    loaded = errors = 0

    def attempt(data):
        nonlocal loaded, errors
        try:
            dpk = groth16.ProvingKey.ReadFrom(emu_ctx, c.name, bytes(data), k_remove=removed)
        except Exception:
            errors += 1
            return
        loaded += 1
        try:
            groth16.Prove(dpk, sol, cs.nb_public, r, s)
        except Exception:
            pass
        finally:
            dpk.FreeGPUResources()

    for img in images:
        n = len(img)
        for cut in sorted({0, 1, 7, 8, 9, 100, 167, 168, 169, 170, n // 3, n // 2, n - 33, n - 4, n - 1} & set(range(n))):
            attempt(img[:cut])
        for _ in range(24):
            d = bytearray(img)
            kind = int(prng.integers(0, 4))
            pos = int(prng.integers(0, n))
            if kind == 0:                                   # a flipped byte anywhere
                d[pos] ^= int(prng.integers(1, 256))
            elif kind == 1:                                 # a huge big-endian length / count word somewhere
                d[pos:pos + 8] = (int(prng.integers(1 << 20, 1 << 62))).to_bytes(8, "big")[: max(0, min(8, n - pos))]
            elif kind == 2:                                 # random garbage spliced in
                d[pos:pos] = bytes(prng.integers(0, 256, size=int(prng.integers(1, 64)), dtype=np.uint8))
            else:                                           # a run of 0xFF
                d[pos:pos + 16] = b"\xff" * min(16, n - pos)
            attempt(d)
    assert errors > 40          # most damage is detected (points off the curve, lengths that do not fit, short input)
    whole = groth16.ProvingKey.ReadFrom(emu_ctx, c.name, images[0], k_remove=removed)   # and the reader is still usable afterwards
    try:
        assert whole.bytes_read == len(images[0])
    finally:
        whole.FreeGPUResources()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_emu_proof_unmarshal(emu_ctx, c):
    """Proof.ReadFrom (marshal.go:62-86) on the oracle's WriteTo and WriteRawTo bytes, with and without commitments; the single
    point decoder on the reference's serialized verifying keys (backend/solidity/testdata/*.vk)"""
    import os
    lib = emu_ctx.lib
    rng = pyref.Xoshiro(99)
    G1, G2 = group_of(c, 0), group_of(c, 1)
    pt1 = lambda: G1.mul(c.g1, rng.field(c.r))
    pt2 = lambda: G2.mul(c.g2, rng.field(c.r))
    for ncom in (0, 3):
        ar, bs, krs = pt1(), pt2(), pt1()
        coms = [pt1() for _ in range(ncom)]
        pok = pt1() if ncom else None
        for data in (pyref.proof_bytes(c, ar, bs, krs, coms, pok), pyref.proof_bytes_raw(c, ar, bs, krs, coms, pok)):
            p = groth16.ParseProof(c.name, data + b"trailing", lib=lib)
            assert p.bytes_read == len(data)
            assert (arr_to_g1_affine(c, p.Ar), arr_to_g2_affine(c, p.Bs), arr_to_g1_affine(c, p.Krs)) == (ar, bs, krs)
            assert [arr_to_g1_affine(c, x) for x in p.Commitments] == coms and arr_to_g1_affine(c, p.CommitmentPok) == pok
            assert pyref.proof_read(c, data)[:5] == (ar, bs, krs, coms, pok)
            assert p.WriteTo() == pyref.proof_bytes(c, ar, bs, krs, coms, pok)      # and back out through the marshaller
    with pytest.raises(Exception, match="does not decode|end of input|malformed"):
        groth16.ParseProof(c.name, b"\x00" * 20, lib=lib)
    # the reference's own serialized keys: alpha1 beta1 beta2 gamma2 delta1 delta2, u32 4, 4 x K  (marshal.go:99-125)
    name = "bn254" if c.cid == 0 else "bls12381"
    raw = open(os.path.join(os.path.dirname(__file__), "golden", f"vk_blank_groth16_{name}_nocommit.bin"), "rb").read()
    b, off = c.fp_bytes, 0
    import ctypes as C
    for group in (0, 0, 1, 1, 0, 1, None, 0, 0, 0, 0):
        if group is None:
            off += 4
            continue
        ln = b if group == 0 else 2 * b
        out = np.zeros(2 * c.fp_limbs * (1 if group == 0 else 2), dtype=np.uint64)
        used = C.c_size_t()
        buf = np.frombuffer(raw[off:], dtype=np.uint8)
        lib.check(lib.ga_point_unmarshal(c.cid, group, buf.ctypes.data, buf.shape[0], out.ctypes.data, C.byref(used)))
        assert used.value == ln
        want = pyref.g1_decompress(c, raw[off:off + ln]) if group == 0 else pyref.g2_decompress(c, raw[off:off + ln])
        assert (arr_to_g1_affine(c, out) if group == 0 else arr_to_g2_affine(c, out)) == want
        off += ln


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_emu_proof_unmarshal_fuzz(emu_ctx, c):
    """Proof.ReadFrom on damaged bytes (proofs arrive over the network): truncations at every length and random mutations of a
    compressed and a raw proof with commitments -- an error or a proof, never a crash; an absurd commitment count is refused
    before anything is allocated for it."""
    lib = emu_ctx.lib
    rng = pyref.Xoshiro(1234)
    G1, G2 = group_of(c, 0), group_of(c, 1)
    pt1 = lambda: G1.mul(c.g1, rng.field(c.r))
    ar, bs, krs, coms, pok = pt1(), G2.mul(c.g2, rng.field(c.r)), pt1(), [pt1(), pt1()], pt1()
    prng = np.random.default_rng(7)
    ok = bad = 0
    for data in (pyref.proof_bytes(c, ar, bs, krs, coms, pok), pyref.proof_bytes_raw(c, ar, bs, krs, coms, pok)):
        cases_ = [data[:k] for k in range(len(data))]
        for _ in range(150):
            d = bytearray(data)
            for _ in range(int(prng.integers(1, 4))):
                d[int(prng.integers(0, len(d)))] = int(prng.integers(0, 256))
            cases_.append(bytes(d))
        for blob in cases_:
            try:
                p = groth16.ParseProof(c.name, blob, lib=lib)
                assert p.bytes_read <= len(blob)
                ok += 1
            except Exception:
                bad += 1
    # the commitment count sits right after Ar | Bs | Krs: 0xFFFFFFFF commitments must be an error, not an allocation
    comp = pyref.proof_bytes(c, ar, bs, krs, [], None)
    head = 4 * c.fp_bytes
    assert comp[head:head + 4] == b"\x00\x00\x00\x00"
    with pytest.raises(Exception):
        groth16.ParseProof(c.name, comp[:head] + b"\xff\xff\xff\xff" + comp[head + 4:], lib=lib)
    assert bad > 200 and ok >= 0


def test_emu_msm_unreduced_canonical_scalars(emu_ctx):
    """canonical (non-Montgomery) scalars need not be below r -- r itself, 2r+5 and 2^256-1 give the same point as
    their residues"""
    c, group, n = BN254, 0, 64
    rng = pyref.Xoshiro(8)
    ks = np.array([rng.next() for _ in range(n)], dtype=np.uint64)
    P = oracle.gen_bases(c.cid, group, ks)
    vals = [rng.field(c.r) for _ in range(n)]
    vals[0], vals[1], vals[2], vals[3] = c.r, 2 * c.r + 5, (1 << 256) - 1, c.r - 1
    S = np.array([pyref.to_limbs(v, 4) for v in vals], dtype=np.uint64)
    got = jac_to_affine_py(c, group, ecc.MultiExp(emu_ctx, c.name, group, P, S, montgomery=False))
    red = fr_to_arr(c, [v % c.r for v in vals])
    want = jac_to_affine_py(c, group, oracle.msm(c.cid, group, P, red))
    assert got == want


def test_emu_plonk_build_z_rejects_bad_device_permutation(emu_ctx):
    c, n = BN254, 8
    d0 = fft.Domain(emu_ctx, c.name, n)
    try:
        v = emu_ctx.to_device(fr_to_arr(c, list(range(1, n + 1))))
        perm = np.arange(3 * n, dtype=np.int64)
        perm[5] = 3 * n            # out of range
        dp = emu_ctx.to_device(perm)
        one = fr_to_arr(c, [1])
        out = emu_ctx.malloc(n * 32)
        rc = emu_ctx.lib.ga_plonk_build_z(d0.handle, v.ptr, v.ptr, v.ptr, dp.ptr, one.ctypes.data, one.ctypes.data, 1, out.ptr)
        assert rc == -1 and b"outside [0, 3n)" in emu_ctx.lib.ga_last_error()
        for b in (v, dp, out):
            b.free()
    finally:
        d0.close()


def test_emu_chunked_sharded_key_generation(emu_ctx):
    """synth.pin_key_chunked (what every rank of `bench.py --gpus N` uses): the key of make_instance generated on the device chunk by
    chunk, only the slice a shard keeps (ga_gen_bases_at + ga_g16_builder_append(NULL) for the rest), proves like the key uploaded from
    host arrays -- whole, and as three base-range shards whose partial sums are added; a null-pointer append INSIDE the shard's slice is
    refused"""
    import ctypes as C
    from gnark_amd import synth
    c = BN254
    inst = synth.make_instance(emu_ctx, c.name, 7, 0x77, want_dlogs=False)
    pk = inst.proving_key(emu_ctx, precompute=1)
    want = groth16.Prove(pk, inst.solution, inst.nb_public, inst.r, inst.s).raw()
    pk.FreeGPUResources()
    lean = synth.make_instance(emu_ctx, c.name, 7, 0x77, want_dlogs=False, with_key=False)
    assert "A" not in lean.key and np.array_equal(lean.solution.W, inst.solution.W)
    pk2 = synth.pin_key_chunked(emu_ctx, lean, precompute=1, chunk=50)
    assert np.array_equal(groth16.Prove(pk2, lean.solution, lean.nb_public, lean.r, lean.s).raw(), want)
    pk2.FreeGPUResources()
    parts = []
    for k in range(3):
        spk = synth.pin_key_chunked(emu_ctx, lean, shard=(k, 3), precompute=-1, chunk=20)
        parts.append(groth16.ProvePartial(spk, lean.solution, lean.nb_public))
        if k == 2:
            fin = groth16.Finish(spk, groth16.SumPartials(c.name, parts, lib=emu_ctx.lib), lean.r, lean.s)
        spk.FreeGPUResources()
    assert np.array_equal(fin.raw(), want)
    lib = emu_ctx.lib
    b = C.c_void_p()
    lib.check(lib.ga_g16_builder_create(emu_ctx.handle, c.cid, 128, 128, 1, 2, C.byref(b)))
    lib.check(lib.ga_g16_builder_reserve(b, 0, 100))
    lib.check(lib.ga_g16_builder_append(b, 0, None, 50))            # points [0, 50): shard 1 of 2 keeps [50, 100)
    assert lib.ga_g16_builder_append(b, 0, None, 10) != 0 and b"null pointer" in lib.ga_last_error()
    lib.ga_g16_builder_destroy(b)


def test_emu_gnark_fixture_directories(emu_ctx, tmp_path):
    """tests/gnark_fixture.py: case directories in the layout gen_fixtures_test.go writes from a gnark checkout (pk.WriteRawTo bytes,
    R1CSSolution.WriteTo bytes, r, s, gnark's proof bytes) proved through ga_g16_pk_read_mem -> ga_g16_prove (+ ga_g16_commit /
    ga_g16_fold_pok) and compared byte for byte.  No Go toolchain exists here, so the directories of this test come from the oracle in
    the same layout (examples/cubic and the two-commitment circuit, both curves); directories under tests/golden/gnark/ -- the real
    thing, once a box with Go has produced them -- are consumed the same way."""
    import gnark_fixture
    ctx = emu_ctx
    cases_ = gnark_fixture.oracle_cases(str(tmp_path)) + gnark_fixture.case_dirs(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gnark"))
    assert len(cases_) >= 4
    for d in cases_:
        for precompute in (-1, 1):
            got, got_raw, want, want_raw = gnark_fixture.run_case(ctx, d, precompute=precompute)
            assert got == want, d
            assert want_raw is None or got_raw == want_raw, d
    # the parsers refuse damaged inputs instead of proving something else
    c = pyref.BN254
    good = open(os.path.join(cases_[0], "solution.bin"), "rb").read()
    for bad in (good[:-1], good + b"\0", good[:4] + b"\xff" * 32 + good[36:]):
        with pytest.raises(ValueError):
            gnark_fixture.parse_solution(c, bad)
