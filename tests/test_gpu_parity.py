"""Parity tests proper: the hipcc-built libgnark_amd.so on a real MI355X, through the C ABI, against the oracle.
Small cases compare with the big-integer / C oracle point-for-point; BASELINE-size cases use known discrete logs
(MSM(s, [k_i]G) == [sum s_i k_i]G, an O(n) field dot product on the CPU oracle) and transform identities."""
import numpy as np
import pytest

import oracle
import pyref
import test_emu_kernels as cases
from gnark_amd import _lib, ecc, fft, groth16
from gnark_amd.device import affine_words
from helpers import BLS12_381, BN254, arr_to_fr, fr_to_arr, pts_to_arr

pytestmark = pytest.mark.gpu
CURVES = [BN254, BLS12_381]


# ---- the emulation cases, now on the device ---------------------------------------------------------------
@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("logn", [0, 1, 3, 5])
def test_fft_all_modes(gpu_ctx, c, logn):
    cases.test_emu_fft_all_modes(gpu_ctx, c, logn)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_fft_multi_pass(gpu_ctx, c):
    cases.test_emu_fft_multi_pass(gpu_ctx, c)


@pytest.mark.parametrize("knobs", [{}, {"GA_NTT_DIRECT": "0"}, {"GA_NTT_WAVE_LOCAL": "0", "GA_NTT_DIRECT": "0"}], ids=["default", "no-direct", "round3"])
@pytest.mark.parametrize("logn", [10, 12, 17, 18])
def test_fft_wave_local_rounds(gpu_ctx, monkeypatch, logn, knobs):
    cases.test_emu_fft_wave_local_rounds(gpu_ctx, monkeypatch, logn, knobs)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_compute_h_small(gpu_ctx, c):
    cases.test_emu_compute_h(gpu_ctx, c)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("group", [0, 1], ids=["G1", "G2"])
def test_msm_matches_naive(gpu_ctx, c, group):
    cases.test_emu_msm_matches_naive(gpu_ctx, c, group)


def test_msm_hot_bucket_and_windows(gpu_ctx):
    cases.test_emu_msm_hot_bucket_and_windows(gpu_ctx)


def test_msm_chunked_and_ragged_sizes(gpu_ctx, monkeypatch):
    cases.test_emu_msm_chunked_and_ragged_sizes(gpu_ctx, monkeypatch, sizes=(2, 3, 7, 65, 257, 1000))


def test_error_behaviour(gpu_ctx):
    cases.test_emu_error_behaviour(gpu_ctx)


def test_msm_empty_and_single(gpu_ctx):
    cases.test_emu_msm_empty_and_single(gpu_ctx)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_groth16_batched_witness_tables_2_12(gpu_ctx, c, monkeypatch):
    """round 6: A, B1, K in ONE pass of the bucket kernel / merge / reduction (msm_table_device_reuse_multi) vs the C oracle's prover
    and vs one pass per table -- exceptional additions inside every table, a boolean-heavy witness, a DummySetup-like key"""
    cases.test_emu_groth16_batched_witness_tables(gpu_ctx, c, monkeypatch, logn=12)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_groth16_prove_oneshot_2_14(gpu_ctx, c):
    """round 6: ga_g16_prove_oneshot (the key uploaded while the proof runs, then dropped -- the reference's default, PinToGPU false):
    same bytes as pin + prove + free and as the pinned-with-tables proof, two callers at once, bad keys as errors, examples/cubic"""
    cases.test_emu_groth16_prove_oneshot(gpu_ctx, c, logn=14)


def test_groth16_second_caller_without_memory_queues(gpu_ctx, monkeypatch):
    """a second concurrent ga_g16_prove caller whose lanes 2/3 cannot allocate (GA_FAULT_LANE2_NOMEM) gives their scratch back and
    queues for the device: right proofs, no error"""
    cases.test_emu_groth16_second_caller_without_memory_queues_instead_of_failing(gpu_ctx, monkeypatch, logn=14, rounds=4)


@pytest.mark.parametrize("precompute", [1, -1, "shared-sort"], ids=["tables", "no-tables", "tables-shared-sort"])
@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_groth16_cubic_bytes(gpu_ctx, c, precompute):
    cases.test_emu_groth16_cubic(gpu_ctx, c, precompute)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("group", [0, 1], ids=["G1", "G2"])
def test_msm_precomputed_table_small(gpu_ctx, c, group):
    cases.test_emu_msm_precomputed_table(gpu_ctx, c, group)


@pytest.mark.parametrize("c,group,logn", [(BN254, 0, 20), (BN254, 1, 16), (BLS12_381, 0, 16), (BLS12_381, 1, 14)],
                         ids=["bn254-G1-2^20", "bn254-G2-2^16", "bls-G1-2^16", "bls-G2-2^14"])
def test_msm_precomputed_table_vs_plain_and_dlog(gpu_ctx, c, group, logn):
    n = 1 << logn
    bases, dlogs, scal = _device_inputs(gpu_ctx, c, group, n, 0x7AB1E + group)
    t = ecc.PrecomputedBases(gpu_ctx, c.name, group, bases, n=n)
    try:
        got = oracle.jac_to_affine(c.cid, group, t.MultiExp(scal))
        plain = oracle.jac_to_affine(c.cid, group, ecc.MultiExp(gpu_ctx, c.name, group, bases, scal, n=n))
        assert np.array_equal(got, plain)
        assert np.array_equal(got, _expect_from_dlogs(c, group, scal.to_host((n, 4)), dlogs.to_host((n, 4))))
    finally:
        t.free()
        for b in (bases, dlogs, scal):
            b.free()


# ---- larger sizes -------------------------------------------------------------------------------------------
_device_inputs, _expect_from_dlogs = cases._device_inputs, cases._expect_from_dlogs


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("group", [0, 1], ids=["G1", "G2"])
def test_msm_2_14_vs_c_oracle(gpu_ctx, c, group):
    cases.test_emu_msm_vs_c_oracle_and_dlogs(gpu_ctx, c, group, logn=14)


@pytest.mark.parametrize("dist", ["uniform", "zero", "one", "rminus1", "witness", "same_base", "some_inf"])
def test_msm_2_20_bn254_g1_distributions(gpu_ctx, dist):
    """BASELINE config 2 (2^20, BN254 G1) under the scalar/base distributions of SURVEY 8d."""
    c, group, n = BN254, 0, 1 << 20
    ctx = gpu_ctx
    bases, dlogs, scal = _device_inputs(ctx, c, group, n, 0x5EED0002)
    wa = affine_words(c.cid, group)
    P = bases.to_host((n, wa))
    K = dlogs.to_host((n, 4))
    S = scal.to_host((n, 4))
    rng = np.random.default_rng(1)
    one = np.array(pyref.to_mont_limbs(1, c.r, 4), dtype=np.uint64)
    if dist == "zero":
        S[:] = 0
    elif dist == "one":
        S[:] = one
    elif dist == "rminus1":
        S[:] = np.array(pyref.to_mont_limbs(c.r - 1, c.r, 4), dtype=np.uint64)
    elif dist == "witness":   # 30 % in {0,1}, 20 % < 2^32, rest uniform
        u = rng.random(n)
        S[u < 0.15] = 0
        S[(u >= 0.15) & (u < 0.30)] = one
        small = np.where((u >= 0.30) & (u < 0.50))[0]
        S[small] = fr_to_arr(c, [int(v) for v in rng.integers(0, 1 << 32, size=small.shape[0])]) if small.shape[0] < 4096 else \
            np.array([pyref.to_mont_limbs(int(v), c.r, 4) for v in rng.integers(0, 1 << 32, size=small.shape[0])], dtype=np.uint64)
    elif dist == "same_base":   # DummySetup-like: every base identical
        P[:] = P[0]
        K[:] = K[0]
    elif dist == "some_inf":
        idx = rng.integers(0, n, size=1000)
        P[idx] = 0
        K[idx] = 0
    got = oracle.jac_to_affine(c.cid, group, ecc.MultiExp(ctx, c.name, group, P, S))
    assert np.array_equal(got, _expect_from_dlogs(c, group, S, K))
    for b in (bases, dlogs, scal):
        b.free()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("group", [0, 1], ids=["G1", "G2"])
def test_msm_degenerate_bases_2_16(gpu_ctx, c, group):
    """all bases equal / every second base equal at 2^16 points: complete lazy loop (retry, then direct on the remembered table)"""
    cases.test_emu_msm_degenerate_bases(gpu_ctx, c, group, n=1 << 16)


@pytest.mark.parametrize("c,group", [(BN254, 0), (BLS12_381, 1)], ids=["bn254-G1", "bls12-381-G2"])
def test_msm_table_two_callers_2_18(gpu_ctx, c, group):
    """three threads, three different scalar vectors, one pinned table of 2^18 points: concurrent results == sequential results"""
    cases.test_emu_msm_table_two_callers(gpu_ctx, c, group, n=1 << 18, rounds=6)


def test_msm_degenerate_bases_exact_kernel(gpu_ctx, monkeypatch):
    """GA_MSM_EXACT_REDO=1: the exact-arithmetic re-run kernel on all-equal bases (BN254 G1 2^14, BLS12-381 G2 2^12)"""
    cases.test_emu_msm_degenerate_bases_exact_kernel(gpu_ctx, monkeypatch, n=1 << 14)


@pytest.mark.parametrize("xcd", [0, 7], ids=["no-xcd-placement", "xcd-slices-and-swizzle"])
@pytest.mark.parametrize("c,group,n,table_c", [(BN254, 0, 1 << 20, 22), (BLS12_381, 1, 1 << 18, 20), (BN254, 1, (1 << 16) + 77, 16)],
                         ids=["bn254-G1-2^20-c22", "bls12-381-G2-2^18-c20", "bn254-G2-ragged-c16"])
def test_msm_fused_first_sort_pass(gpu_ctx, monkeypatch, c, group, n, table_c, xcd):
    """the digit extraction fused with the first level of the sort (msm.hip.h 1b; the default from 2^21 pairs, forced here on smaller
    inputs; with and without the XCD placement of round 5) == the plain digits + two-pass sort sequence == [sum s_i k_i]G: production shape (c = 22, 12 windows), 13 windows,
    16 windows with partial tiles and a ragged length; hot / zero / canonical scalars, window ranges; the same inputs as a raw-bases
    MSM in ranges of 16 windows (one bucket set per window)"""
    cases.test_emu_msm_fused_first_sort_pass(gpu_ctx, c, group, xcd, monkeypatch, n=n, table_c=table_c)


@pytest.mark.parametrize("xcd", [0, 7], ids=["no-xcd-placement", "xcd-slices-and-swizzle"])
@pytest.mark.parametrize("table_c,batch,n", [(22, 3, 1 << 20), (23, 3, (1 << 18) + 5)], ids=["23-bit-keys-2^20", "24-bit-keys-ragged"])
def test_msm_fused_sort_wide_keys(gpu_ctx, monkeypatch, table_c, batch, n, xcd):
    """the fused sort on 23- and 24-bit key spaces and batches of scalar vectors (PLONK's grouped commitments; the 12-bit first level):
    uniform / all-equal / zero-one vectors, fused == library sort == known discrete logs"""
    cases.test_emu_msm_fused_sort_wide_keys(gpu_ctx, monkeypatch, table_c, batch, xcd, n=n)


def test_abi_exception_barrier(gpu_ctx, monkeypatch):
    """GA_FAULT_THROW on the hipcc-built library: std::bad_alloc thrown inside ga_msm, ga_msm_table_run(_batch), ga_fft, ga_compute_h,
    ga_plonk_quotient(_pinned), ga_kzg_open, ga_fr_batch_invert comes back as GA_ERR_NOMEM and the context keeps working"""
    cases.test_emu_abi_exception_barrier(gpu_ctx, monkeypatch, n=5000)


def test_abi_exception_barrier_groth16(gpu_ctx, monkeypatch):
    cases.test_emu_abi_exception_barrier_groth16(gpu_ctx, monkeypatch)


def test_raw_msm_2_24_takes_the_fused_sort(gpu_ctx):
    """a raw-bases (no table) BN254 G1 MSM of 2^24 points -- 13 windows x 2^19 buckets, 23 key bits -- runs on the fused sort since
    round 4 (no library sort on any BASELINE path) and equals [sum s_i k_i]G"""
    c, n = BN254, 1 << 24
    bases, dlogs, scal = cases._device_inputs(gpu_ctx, c, 0, n, 0x24F5)
    try:
        gpu_ctx.profile(True)
        gpu_ctx.profile_reset()
        got = ecc.MultiExp(gpu_ctx, c.name, 0, bases, scal, n=n)
        gpu_ctx.sync()
        stages = [name for name, _ in gpu_ctx.profile_read()]
        gpu_ctx.profile(False)
        assert "msm_digits_pass1" in stages and "msm_digits" not in stages, stages
        want = cases._expect_from_dlogs(c, 0, scal.to_host((n, 4)), dlogs.to_host((n, 4)))
        assert np.array_equal(oracle.jac_to_affine(c.cid, 0, got), want)
    finally:
        for b in (bases, dlogs, scal):
            b.free()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("group", [0, 1], ids=["G1", "G2"])
def test_msm_table_batch_2_18(gpu_ctx, c, group):
    """three commitments' worth of scalars + an all-zero and an all-one vector over one pinned SRS in one pass (2^18 points)"""
    cases.test_emu_msm_table_batch(gpu_ctx, c, group, n=1 << 18, k=3)


def test_msm_table_batched_hint_2_22(gpu_ctx):
    """GA_TABLE_BATCHED on a 2^22-point SRS (BASELINE config 5's size): a narrower window is planned (c = 20 instead of 22) and the
    batched and single runs over it still equal [sum s_i k_i]G"""
    c, n = BN254, 1 << 22
    bases, dlogs, scal = _device_inputs(gpu_ctx, c, 0, n, 0xBA7D)
    plain = ecc.PrecomputedBases(gpu_ctx, c.name, 0, bases, n=n)
    hinted = ecc.PrecomputedBases(gpu_ctx, c.name, 0, bases, n=n, batched=True)
    try:
        assert plain.info()["window_bits"] == 22 and hinted.info()["window_bits"] == 20, (plain.info(), hinted.info())
    finally:
        plain.free()
        hinted.free()
        for b in (bases, dlogs, scal):
            b.free()
    cases.test_emu_msm_table_batch(gpu_ctx, c, 0, n=n, k=3, batched=True)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("group", [0, 1], ids=["G1", "G2"])
@pytest.mark.parametrize("table", [True, False], ids=["table", "raw"])
def test_msm_2_20_boolean_heavy_witness(gpu_ctx, c, group, table):
    """60 % ones + 10 % zeros at 2^20 points: one bucket of 0.6 n points (thousands of tasks, two-stage merge) on every MSM shape"""
    cases.test_emu_msm_very_hot_bucket(gpu_ctx, c, group, table, n=1 << 20)


def test_msm_2_24_bn254_g1_dlog(gpu_ctx):
    """BASELINE headline size: 2^24 points, result == [sum s_i k_i]G."""
    c, group, n = BN254, 0, 1 << 24
    bases, dlogs, scal = _device_inputs(gpu_ctx, c, group, n, 0x5EED0005)
    got = oracle.jac_to_affine(c.cid, group, ecc.MultiExp(gpu_ctx, c.name, group, bases, scal, n=n))
    S, K = scal.to_host((n, 4)), dlogs.to_host((n, 4))
    assert np.array_equal(got, _expect_from_dlogs(c, group, S, K))
    # window-sharded evaluation recombines to the same point (multi-GPU partitioning A on one device)
    cbits, nwin = ecc.plan(c.name, group, n)
    parts = [ecc.MultiExpWindows(gpu_ctx, c.name, group, bases, scal, n, lo, hi)[0] for lo, hi in ((0, nwin // 2), (nwin // 2, nwin))]
    comb = oracle.jac_to_affine(c.cid, group, ecc.combine_windows(c.name, group, np.concatenate(parts), cbits))
    assert np.array_equal(comb, got)
    for b in (bases, dlogs, scal):
        b.free()


def test_msm_2_22_raw_bases_take_the_fused_sort_pass_by_default(gpu_ctx):
    """2^22 un-pinned bases (c = 17: 15 bucket sets, 21 key bits, 63 M pairs): the library fuses the digit extraction with the
    first sort pass on its own (no knob) and the result equals [sum s_i k_i]G"""
    c, group, n = BN254, 0, 1 << 22
    bases, dlogs, scal = _device_inputs(gpu_ctx, c, group, n, 0x5EED0022)
    gpu_ctx.profile(True)
    gpu_ctx.profile_reset()
    got = oracle.jac_to_affine(c.cid, group, ecc.MultiExp(gpu_ctx, c.name, group, bases, scal, n=n))
    gpu_ctx.sync()
    stages = [name for name, _ in gpu_ctx.profile_read()]
    gpu_ctx.profile(False)
    assert "msm_digits_pass1" in stages and "msm_digits" not in stages
    assert np.array_equal(got, _expect_from_dlogs(c, group, scal.to_host((n, 4)), dlogs.to_host((n, 4))))
    for b in (bases, dlogs, scal):
        b.free()


def test_msm_2_26_bn254_g1_table_and_raw_dlog(gpu_ctx):
    """four times the headline size (2^26 points: 4 GiB of bases, a 48 GiB window table, 805 M (bucket, point) pairs): the
    pinned-table MSM and the raw-bases MSM both equal [sum s_i k_i]G"""
    c, group, n = BN254, 0, 1 << 26
    bases, dlogs, scal = _device_inputs(gpu_ctx, c, group, n, 0x5EED0026)
    S, K = scal.to_host((n, 4)), dlogs.to_host((n, 4))
    want = _expect_from_dlogs(c, group, S, K)
    raw = oracle.jac_to_affine(c.cid, group, ecc.MultiExp(gpu_ctx, c.name, group, bases, scal, n=n))
    assert np.array_equal(raw, want)
    table = ecc.PrecomputedBases(gpu_ctx, c.name, group, bases, n=n)
    try:
        got = oracle.jac_to_affine(c.cid, group, table.MultiExp(scal))
    finally:
        table.free()
    assert np.array_equal(got, want)
    for b in (bases, dlogs, scal):
        b.free()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_fft_2_16_vs_c_oracle(gpu_ctx, c):
    n = 1 << 16
    rng = np.random.default_rng(5)
    scal = gpu_ctx.malloc(n * 32)
    gpu_ctx.lib.check(gpu_ctx.lib.ga_gen_scalars(gpu_ctx.handle, c.cid, 99, n, scal.ptr))
    a = scal.to_host((n, 4))
    scal.free()
    d = fft.Domain(gpu_ctx, c.name, n)
    try:
        for dec in (0, 1):
            for coset in (False, True):
                for inv in (0, 1):
                    got = (d.FFTInverse if inv else d.FFT)(a, dec, coset)
                    assert np.array_equal(got, oracle.fft(c.cid, a, inv, dec, coset)), (dec, coset, inv)
        m = n - 3
        A, B = a[:m], np.roll(a, 7, axis=0)[:m]
        Cc = oracle.fr_mul(c.cid, A, B)
        assert np.array_equal(d.compute_h(A, B, Cc), oracle.compute_h(c.cid, A, B, Cc, n))
    finally:
        d.close()


def test_fft_2_22_roundtrip_and_oracle_2_20(gpu_ctx):
    c = BN254
    n = 1 << 20
    scal = gpu_ctx.malloc(4 * n * 32)
    gpu_ctx.lib.check(gpu_ctx.lib.ga_gen_scalars(gpu_ctx.handle, c.cid, 7, 4 * n, scal.ptr))
    a4 = scal.to_host((4 * n, 4))
    scal.free()
    d = fft.Domain(gpu_ctx, c.name, n)
    a = a4[:n]
    assert np.array_equal(d.FFT(a, 0, True), oracle.fft(c.cid, a, 0, 0, True))
    d.close()
    d = fft.Domain(gpu_ctx, c.name, 4 * n)
    y = d.FFT(a4, 0, True)                     # DIF on coset: natural -> bit-reversed
    back = d.FFTInverse(y, 1, True)            # DIT inverse on coset: bit-reversed -> natural
    assert np.array_equal(back, a4)
    d.close()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_groth16_synthetic_2_10_vs_oracle(gpu_ctx, c, monkeypatch):
    """2^10 synthetic instance vs the C oracle's prover: all tables, no tables, and (round 5) the partial table sets of precompute = 0
    under a budget -- wire-indexed tables, compact tables and plain vectors in one proof"""
    cases.test_emu_groth16_synthetic_vs_c_oracle(gpu_ctx, c, monkeypatch, logn=10)


@pytest.mark.parametrize("precompute", [1, -1, "shared-sort"], ids=["tables", "no-tables", "tables-shared-sort"])
@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_groth16_bsb22_commitments_bytes(gpu_ctx, c, precompute):
    cases.test_emu_groth16_bsb22_commitments(gpu_ctx, c, precompute)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("group", [0, 1], ids=["G1", "G2"])
def test_msm_lazy_window_reduction_forced(gpu_ctx, c, group, monkeypatch):
    cases.test_emu_msm_lazy_window_reduction(gpu_ctx, c, group, monkeypatch)


def test_hash_to_field_host(gpu_ctx):
    cases.test_emu_hash_to_field(gpu_ctx)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_groth16_bsb22_synthetic_2_14_vs_oracle(gpu_ctx, c):
    """Commitments at a size where the MSM pipeline (not the tiny-input path) runs: 2 commitment keys of 2^12 and 37 points,
    a K MSM with the committed wires filtered out on device; everything against the C oracle (commitment / PoK = plain MSMs,
    Krs via a K vector with infinity at the removed wires)."""
    ctx, lib = gpu_ctx, gpu_ctx.lib
    n = 1 << 14
    nw, nb_public = n, 3
    wa = affine_words(c.cid, 0)

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
    rng = np.random.default_rng(8)
    committed = [np.sort(rng.choice(np.arange(nb_public, nw - 2), size=sz, replace=False)) for sz in (1 << 12, 37)]
    committed[1] = np.setdiff1d(committed[1], committed[0])
    com_wires = np.array([nw - 2, nw - 1])
    removed = np.sort(np.concatenate(committed + [com_wires]))
    keep = np.setdiff1d(np.arange(nb_public, nw), removed)
    infA = np.zeros(nw, np.uint8)
    infB = np.zeros(nw, np.uint8)
    infA[[1, 5]] = 1
    infB[[0, 7, 9]] = 1
    m1, m2 = gen(0, 3, 1), gen(1, 2, 2)
    Kc = gen(0, keep.size, 6)
    Kfull = np.zeros((nw - nb_public, wa), np.uint64)   # oracle view: infinity where the wire is committed
    Kfull[keep - nb_public] = Kc
    key = dict(n=n, alpha1=m1[0:1], beta1=m1[1:2], delta1=m1[2:3], A=gen(0, nw - 2, 3), B=gen(0, nw - 3, 4), Z=gen(0, n - 1, 5),
               K=Kfull, beta2=m2[0:1], delta2=m2[1:2], B2=gen(1, nw - 3, 7), infinityA=infA, infinityB=infB)
    cks = [(gen(0, len(cw), 20 + i), gen(0, len(cw), 30 + i)) for i, cw in enumerate(committed)]
    m = n - 5
    W, A, B = scal(nw, 10), scal(m, 11), scal(m, 12)
    Cc = oracle.fr_mul(c.cid, A, B)
    rs = scal(2, 13)
    want = oracle.groth16_prove(c.cid, key, W, A, B, Cc, nb_public, rs[0], rs[1], nthreads=8)
    pk = groth16.ProvingKey(ctx, c.name, domain_cardinality=n, precompute=1, commitment_keys=cks, k_remove=removed,
                            **{k: (Kc if k == "K" else v) for k, v in key.items() if k != "n"})
    try:
        for i, cw in enumerate(committed):
            com, pok = pk.Commit(i, W[cw])
            assert np.array_equal(com, oracle.jac_to_affine(c.cid, 0, oracle.msm(c.cid, 0, cks[i][0], W[cw], nthreads=8)))
            assert np.array_equal(pok, oracle.jac_to_affine(c.cid, 0, oracle.msm(c.cid, 0, cks[i][1], W[cw], nthreads=8)))
        proof = groth16.Prove(pk, groth16.Solution(W, A, B, Cc), nb_public, rs[0], rs[1])
    finally:
        pk.FreeGPUResources()
    assert np.array_equal(proof.Ar, want[0]) and np.array_equal(proof.Bs, want[1]) and np.array_equal(proof.Krs, want[2])


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n,nb_bsb", [(4, 0), (8, 1), (64, 2), (1024, 1)])
def test_plonk_quotient_vs_oracle(gpu_ctx, c, n, nb_bsb):
    cases.test_emu_plonk_quotient(gpu_ctx, c, n, nb_bsb, seed=77 + n)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n", [2, 8, 512, 1000, 4096])
def test_plonk_build_z_and_batch_invert(gpu_ctx, c, n):
    cases.test_emu_plonk_build_z_and_batch_invert(gpu_ctx, c, n)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_plonk_quotient_2_14_vs_c_oracle(gpu_ctx, c):
    """n = 2^14 (big domain 2^16: multi-pass transforms), random polynomials (the quotient map is defined for any input), one
    BSB22 gate: every one of the 65536 coefficients of h equals the C oracle's, for the plain call and for the pinned key"""
    from gnark_amd import plonk
    n, lib = 1 << 14, gpu_ctx.lib

    def scal(count, seed):
        buf = gpu_ctx.malloc(count * 32)
        lib.check(lib.ga_gen_scalars(gpu_ctx.handle, c.cid, seed, count, buf.ptr))
        h = buf.to_host((count, 4))
        buf.free()
        return h
    names = list(plonk.IDS) + ["Qcp0", "Pi20"]
    P = {k: scal(n, 700 + i) for i, k in enumerate(names)}
    small = scal(12, 999)
    bp = {"Bl": small[0:2], "Br": small[2:4], "Bo": small[4:6], "Bz": small[6:9]}
    alpha, beta, gamma = small[9:10], small[10:11], small[11:12]
    want = oracle.plonk_quotient(c.cid, n, [P[k] for k in names], bp["Bl"], bp["Br"], bp["Bo"], bp["Bz"], alpha, beta, gamma, 1)
    d0, d1 = fft.Domain(gpu_ctx, c.name, n), fft.Domain(gpu_ctx, c.name, 4 * n)
    try:
        kw = dict(bp=bp, alpha=alpha, beta=beta, gamma=gamma)
        got = plonk.ComputeQuotient(d0, d1, {k: P[k] for k in plonk.IDS}, [P["Qcp0"]], [P["Pi20"]], **kw)
        assert np.array_equal(got, want)
        ppk = plonk.ProvingKey(d0, d1, {k: P[k] for k in plonk.FIXED_IDS}, [P["Qcp0"]])
        try:
            got = ppk.ComputeQuotient({k: P[k] for k in plonk.PROOF_IDS}, [P["Pi20"]], **kw)
        finally:
            ppk.close()
        assert np.array_equal(got, want)
    finally:
        d0.close()
        d1.close()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n,srs_len", [(1, 4), (2, 4), (7, 7), (300, 300), (4096, 4100)])
def test_kzg_open_vs_oracle(gpu_ctx, c, n, srs_len):
    cases.test_emu_kzg_open(gpu_ctx, c, n, srs_len)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_fr_linear_combination(gpu_ctx, c):
    cases.test_emu_fr_linear_combination(gpu_ctx, c, n=5000)


def test_kzg_open_2_20_known_dlogs(gpu_ctx):
    """kzg.Open at 2^20 coefficients over 2^20 pinned bases [k_i]G with known k_i: claimed value == Horner (C oracle) and
    H == [sum q_i k_i]G with the quotient coefficients from the sequential recurrence on the host"""
    c, n = BN254, 1 << 20
    mod = c.r
    bases, dlogs, scal = _device_inputs(gpu_ctx, c, 0, n, 0x4B5A)
    poly_arr, K = scal.to_host((n, 4)), dlogs.to_host((n, 4))
    z = pyref.Xoshiro(77).field(mod)
    table = ecc.PrecomputedBases(gpu_ctx, c.name, 0, bases, n=n)
    try:
        val, H = table.KzgOpen(poly_arr, fr_to_arr(c, [z]))
    finally:
        table.free()
        for b in (bases, dlogs, scal):
            b.free()
    assert np.array_equal(val, oracle.fr_horner(c.cid, poly_arr, fr_to_arr(c, [z])[0]))
    poly = arr_to_fr(c, poly_arr)
    q, acc = [0] * n, 0
    for k in range(n - 1, 0, -1):
        acc = (poly[k] + z * acc) % mod
        q[k - 1] = acc
    want = oracle.jac_to_affine(c.cid, 0, oracle.generator_mul(c.cid, 0, oracle.fr_dot(c.cid, fr_to_arr(c, q), K)))
    assert np.array_equal(oracle.jac_to_affine(c.cid, 0, H), want)


def test_plonk_quotient_2_16_identity(gpu_ctx):
    """n = 2^16 (4n = 2^18: multi-pass transforms): a satisfying synthetic trace built with the C oracle's FFT, Z from the device
    grand product (checked against a Python prefix product at sampled positions and to close), then
    h(zeta) * (zeta^n - 1) == gate + alpha*ordering + alpha^2*(Z-1)*L1 at a random zeta, every polynomial evaluated by the C
    oracle's Horner -- independent of the device transforms."""
    from gnark_amd import plonk
    c, n = BN254, 1 << 16
    mod = c.r
    lag, qcp, pi2, perm, rng = pyref.plonk_synthetic_instance(c, n, 2024, 1)
    beta, gamma, alpha, zeta = (rng.field(mod) for _ in range(4))
    d0, d1 = fft.Domain(gpu_ctx, c.name, n), fft.Domain(gpu_ctx, c.name, 4 * n)
    try:
        arr = {k: fr_to_arr(c, v) for k, v in lag.items()}
        Z = plonk.BuildRatioCopyConstraint(d0, arr["L"], arr["R"], arr["O"], perm, fr_to_arr(c, [beta]), fr_to_arr(c, [gamma]))
        arr["Z"] = Z
        zi = arr_to_fr(c, Z)
        w0, g = c.fr_root_of_unity(n), c.fr_gen
        ids = lambda pos: pow(g, pos // n, mod) * pow(w0, pos % n, mod) % mod
        ev = [lag["L"], lag["R"], lag["O"]]

        def ratio(i):
            num = den = 1
            for k in range(3):
                num = num * ((ev[k][i] + beta * ids(k * n + i) + gamma) % mod) % mod
                den = den * ((ev[k][i] + beta * ids(perm[k * n + i]) + gamma) % mod) % mod
            return num * pow(den, -1, mod) % mod
        assert zi[0] == 1 and zi[n - 1] * ratio(n - 1) % mod == 1
        for i in (0, 1, 255, 256, 4097, n - 2):
            assert zi[i + 1] == zi[i] * ratio(i) % mod
        bp = {"Bl": [rng.field(mod) for _ in range(2)], "Br": [rng.field(mod) for _ in range(2)],
              "Bo": [rng.field(mod) for _ in range(2)], "Bz": [rng.field(mod) for _ in range(3)]}
        names = list(plonk.IDS) + ["Qcp0", "Pi20"]
        h = plonk.ComputeQuotient(d0, d1, {k: arr[k] for k in plonk.IDS}, [fr_to_arr(c, qcp[0])], [fr_to_arr(c, pi2[0])],
                                  bp={k: fr_to_arr(c, v) for k, v in bp.items()}, alpha=fr_to_arr(c, [alpha]), beta=fr_to_arr(c, [beta]),
                                  gamma=fr_to_arr(c, [gamma]), lagrange=tuple(names))
    finally:
        d0.close()
        d1.close()
    horner = lambda coeffs, x: arr_to_fr(c, oracle.fr_horner(c.cid, coeffs, fr_to_arr(c, [x])[0]).reshape(1, 4))[0]

    def lag_eval(evals_arr, x):
        """evaluate the interpolant of evals on the domain at x: C oracle inverse FFT (DIF: bit-reversed output), then Horner"""
        nat = oracle.fft(c.cid, evals_arr, 1, 0, False)
        idx = np.array([pyref.bitrev(i, 16) for i in range(n)])
        return horner(nat[idx], x)
    e = {k: lag_eval(arr[k], zeta) for k in plonk.IDS}
    zw = zeta * w0 % mod
    z_w = lag_eval(arr["Z"], zw)
    qc_z, pi_z = lag_eval(fr_to_arr(c, qcp[0]), zeta), lag_eval(fr_to_arr(c, pi2[0]), zeta)
    zn1 = (pow(zeta, n, mod) - 1) % mod
    b = lambda k, pt: pyref._poly_eval(bp[k], pt, mod) * ((pow(pt, n, mod) - 1) % mod) % mod
    l, r, o = (e["L"] + b("Bl", zeta)) % mod, (e["R"] + b("Br", zeta)) % mod, (e["O"] + b("Bo", zeta)) % mod
    z, zs = (e["Z"] + b("Bz", zeta)) % mod, (z_w + b("Bz", zw)) % mod
    gate = (e["Ql"] * l + e["Qr"] * r + e["Qm"] * l % mod * r + e["Qo"] * o + e["Qk"] + qc_z * pi_z) % mod
    idv = zeta * beta % mod
    rr = (gamma + l + idv) * ((idv * g + r + gamma) % mod) % mod * ((idv * g * g + o + gamma) % mod) % mod * z % mod
    ll = (e["S1"] * beta + l + gamma) * ((e["S2"] * beta + r + gamma) % mod) % mod * ((e["S3"] * beta + o + gamma) % mod) % mod * zs % mod
    lone = zn1 * pow(n, -1, mod) % mod * pow((zeta - 1) % mod, -1, mod) % mod
    want = (((z - 1) * lone % mod * alpha + (ll - rr)) % mod * alpha + gate) % mod
    assert horner(h, zeta) * zn1 % mod == want
    assert not h[3 * n + 6:].any()          # deg h = 3n + 5


def test_loaded_library_is_the_hip_build(gpu_ctx):
    assert gpu_ctx.lib.path.endswith("gnark_amd/libgnark_amd.so")
    info = gpu_ctx.info()
    assert "gfx950" in info["name"], info
    mb = gpu_ctx.microbench()
    assert mb.get("v_mad_u64_u32_Gops", 0) > 0


def test_kzg_ceremony_relations_on_device(gpu_ctx):
    """The reference's EIP-4844 fixture (tests/golden/kzg4096_bls12381.npz, from std/evmprecompiles/kzg_trusted_setup.json)
    through the HIP path: sum L_i = G, sum w^i L_i = [tau]G, and MSM(p, monomial) == MSM(NTT(p), lagrange)."""
    import os
    c = BLS12_381
    kzg = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kzg4096_bls12381.npz"))
    n = 4096
    L, M = kzg["g1_lagrange"], kzg["g1_monomial"]
    aff = lambda jac: oracle.jac_to_affine(c.cid, 0, jac)
    one = fr_to_arr(c, [1] * n)
    assert np.array_equal(aff(ecc.MultiExp(gpu_ctx, c.name, 0, L, one)), M[0])
    w = c.fr_root_of_unity(n)
    pw = fr_to_arr(c, [pow(w, i, c.r) for i in range(n)])
    assert np.array_equal(aff(ecc.MultiExp(gpu_ctx, c.name, 0, L, pw)), M[1])
    rng = pyref.Xoshiro(4844)
    p = fr_to_arr(c, [rng.field(c.r) for _ in range(n)])
    d = fft.Domain(gpu_ctx, c.name, n)
    ev_bitrev = d.FFT(p, fft.DIF)
    d.close()
    idx = np.array([pyref.bitrev(i, 12) for i in range(n)])
    lhs = aff(ecc.MultiExp(gpu_ctx, c.name, 0, M, p))
    rhs = aff(ecc.MultiExp(gpu_ctx, c.name, 0, L, ev_bitrev[idx]))
    assert np.array_equal(lhs, rhs) and lhs.any()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_kzg_commit_shape_plonk(gpu_ctx, c):
    """PLONK's kzg.Commit (backend/plonk/bn254/prove.go:438-476,532,1267-1277) is MultiExp(pk.G1[:len(p)], p) over the
    monomial SRS [tau^i]G (test/unsafekzg/kzgsrs.go:186-200): with a known tau the commitment must equal [p(tau)]G.
    Also exercises ToLagrange/ToCanonical round trips through the pinned-SRS table path."""
    n = 1 << 12
    rng = pyref.Xoshiro(0x5125)
    tau = rng.field(c.r)
    # SRS on the host via the oracle (known discrete logs tau^i); committed polynomial p
    pw = [pow(tau, i, c.r) for i in range(n)]
    srs = np.stack([oracle.jac_to_affine(c.cid, 0, oracle.generator_mul(c.cid, 0, k)) for k in pw])
    p = [rng.field(c.r) for _ in range(n)]
    P = fr_to_arr(c, p)
    want = oracle.jac_to_affine(c.cid, 0, oracle.generator_mul(c.cid, 0, sum(a * b for a, b in zip(p, pw)) % c.r))
    table = ecc.PrecomputedBases(gpu_ctx, c.name, 0, srs)
    try:
        assert np.array_equal(oracle.jac_to_affine(c.cid, 0, table.MultiExp(P)), want)
        assert np.array_equal(oracle.jac_to_affine(c.cid, 0, ecc.MultiExp(gpu_ctx, c.name, 0, srs, P)), want)
        # Lagrange-basis commitment: Commit_lagrange(NTT(p)) == Commit(p) with the Lagrange SRS [L_i(tau)]G
        w = c.fr_root_of_unity(n)
        ninv = pow(n, -1, c.r)
        tn1 = (pow(tau, n, c.r) - 1) % c.r
        lag = [tn1 * pow(w, i, c.r) % c.r * ninv % c.r * pow((tau - pow(w, i, c.r)) % c.r, -1, c.r) % c.r for i in range(n)]
        srs_l = np.stack([oracle.jac_to_affine(c.cid, 0, oracle.generator_mul(c.cid, 0, k)) for k in lag])
        d = fft.Domain(gpu_ctx, c.name, n)
        ev = d.FFT(P, fft.DIF)                                            # bit-reversed evaluations
        idx = np.array([pyref.bitrev(i, 12) for i in range(n)])
        got = oracle.jac_to_affine(c.cid, 0, ecc.MultiExp(gpu_ctx, c.name, 0, srs_l, ev[idx]))
        assert np.array_equal(got, want)
        back = d.FFTInverse(ev, fft.DIT)                                  # ToCanonical
        assert np.array_equal(back, P)
        d.close()
    finally:
        table.free()


def test_plain_c_client_of_the_abi(tmp_path):
    """include/gnark_amd.h is a genuine C ABI: a C99 program (what cgo compiles) links libgnark_amd.so and runs an MSM
    (raw bases and precomputed table) + NTT round trip without Python, torch or C++ in the client."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "abi_client")
    lib = os.path.join(root, "gnark_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c_abi", "abi_client.c"),
                           "-L", lib, "-lgnark_amd", "-Wl,-rpath," + lib, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ABI_CLIENT_OK" in r.stdout, r.stdout + r.stderr


def test_cgo_call_pattern_and_concurrency(tmp_path):
    """tests/c_abi/cgo_pattern.c on the device at 2^14 constraints: staged key (transient source buffers), struct key, poisoned
    solutions, 4 host threads over 2 contexts on one GPU, proofs interleaved with ga_fft -- all proofs identical"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "cgo_pattern")
    lib = os.path.join(root, "gnark_amd")
    subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-pthread", "-DCGO_LOGN=14", "-DCGO_THREAD_ROUNDS=8", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c_abi", "cgo_pattern.c"), "-L", lib, "-lgnark_amd", "-Wl,-rpath," + lib, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CGO_PATTERN_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("curve", ["GA_BN254", "GA_BLS12_381"])
def test_plonk_call_pattern(tmp_path, curve):
    """tests/c_abi/plonk_pattern.c on the device at n = 2^14: the patched PLONK prover's call order (prove.patch) with transient,
    poisoned buffers, the struct arguments in C heap, batched == single commitments, pinned == un-pinned quotient, two replays equal"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "plonk_pattern")
    lib = os.path.join(root, "gnark_amd")
    subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-DPLONK_LOGN=14", "-DPLONK_CURVE=" + curve, "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c_abi", "plonk_pattern.c"), "-L", lib, "-lgnark_amd", "-Wl,-rpath," + lib, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PLONK_PATTERN_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("precompute", [1, -1], ids=["tables", "no-tables"])
def test_groth16_staged_builder(gpu_ctx, c, precompute):
    cases.test_emu_groth16_staged_builder(gpu_ctx, c, precompute)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("precompute", [1, -1], ids=["tables", "no-tables"])
def test_groth16_two_callers_distinct_solutions(gpu_ctx, c, precompute):
    """three host threads proving three different solutions on one key + a fourth running transforms on the same context, at
    2^16 constraints: every proof equals the proof of its own solution computed alone (lanes do not share buffers)"""
    cases.test_emu_groth16_two_callers_distinct_solutions(gpu_ctx, c, precompute, logn=16, rounds=6)


def test_groth16_two_keys_two_curves_interleaved(gpu_ctx):
    """a 2^16 BN254 key (tables) and a 2^18 BLS12-381 key (plain vectors) pinned on one context, proved alternately from two host
    threads: shared scratch names across keys, element sizes and lanes"""
    cases.test_emu_groth16_two_keys_two_curves_interleaved(gpu_ctx, logn_a=16, logn_b=18, rounds=6)


def test_groth16_soak_no_device_memory_growth(gpu_ctx):
    """600 small proofs from three host threads on one key (every lane pair, slot hand-offs, helper threads, per-proof events):
    all byte-equal to the first proof, and the device's free memory after the run is what it was after the warm-up -- nothing a
    proof allocates (scratch is reused, events / streams are released) accumulates."""
    import threading
    from gnark_amd import synth
    inst = synth.make_instance(gpu_ctx, "bn254", 12, 0x50AC, want_dlogs=False)
    pk = inst.proving_key(gpu_ctx, precompute=1)
    sol, nbp, r, s = inst.solution, inst.nb_public, inst.r, inst.s
    try:
        want = groth16.Prove(pk, sol, nbp, r, s).raw()
        bad = []

        def prover(count):
            for _ in range(count):
                if not np.array_equal(groth16.Prove(pk, sol, nbp, r, s).raw(), want):
                    bad.append(1)

        def run(count):
            th = [threading.Thread(target=prover, args=(count,)) for _ in range(3)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            gpu_ctx.sync()

        run(5)                                   # warm-up: every lane's scratch exists now
        free0 = gpu_ctx.info()["free_bytes"]
        run(200)
        free1 = gpu_ctx.info()["free_bytes"]
        assert not bad
        assert free1 >= free0 - (8 << 20), (free0, free1)
    finally:
        pk.FreeGPUResources()


def test_out_of_device_memory_is_an_error_not_a_crash(gpu_ctx):
    """HBM exhausted (down to the reserve the library keeps for the runtime): pinning a key (tables asked for explicitly) and
    proving on a key pinned earlier must come back as library errors -- no crash, no leaked lock, no poisoned context, no stale
    HIP error picked up by the next launch -- and once the memory is back the same calls succeed with the same proof bytes."""
    from gnark_amd import synth
    from gnark_amd._lib import GnarkAmdError
    from gnark_amd.device import Context
    inst = synth.make_instance(gpu_ctx, "bn254", 16, 0x00D1, want_dlogs=False)
    sol, nbp, r, s = inst.solution, inst.nb_public, inst.r, inst.s
    shared_ctx, gpu_ctx = gpu_ctx, Context(0)           # a context of its own: no scratch left over from earlier tests
    pk = inst.proving_key(gpu_ctx, precompute=-1)      # plain vectors; no proof yet, so its scratch does not exist
    ballast = []
    try:
        # fill the device down to the library's reserve (GA_HBM_RESERVE_MB, 1 GiB: what the ROCm runtime needs for the kernels'
        # private segments -- it aborts the process when it cannot get it, tools/exp/oom_repro.py): ever smaller pieces until a
        # 16 MiB one is refused (smaller ones may use half of the reserve), so that less than a 2^16 proof's scratch and far less
        # than a key with tables is left
        step = 64 << 30
        while step >= (16 << 20):
            try:
                ballast.append(gpu_ctx.malloc(step))
            except GnarkAmdError:
                step //= 2
        assert gpu_ctx.info()["free_bytes"] >= (1 << 30) - (64 << 20)      # the reserve itself stays free
        with pytest.raises(GnarkAmdError):
            inst.proving_key(gpu_ctx, precompute=1)
        with pytest.raises(GnarkAmdError):
            groth16.Prove(pk, sol, nbp, r, s)
        with pytest.raises(GnarkAmdError):               # a second failure: the first one left no lock or lane behind
            groth16.Prove(pk, sol, nbp, r, s)
    finally:
        for b in ballast:
            b.free()
    try:
        got = groth16.Prove(pk, sol, nbp, r, s).raw()
        pk2 = inst.proving_key(gpu_ctx, precompute=1)
        try:
            assert np.array_equal(groth16.Prove(pk2, sol, nbp, r, s).raw(), got)
        finally:
            pk2.FreeGPUResources()
    finally:
        pk.FreeGPUResources()
        gpu_ctx.close()


def test_groth16_builder_errors(gpu_ctx):
    cases.test_emu_groth16_builder_errors(gpu_ctx)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("nshards,precompute", [(2, 1), (3, 1), (3, -1)], ids=["2-tables", "3-tables", "3-no-tables"])
def test_groth16_prove_multi_contexts_on_one_device(gpu_ctx, c, nshards, precompute):
    """ga_g16_prove_multi at 2^12 constraints with one context per shard (all on this box's single GPU: the peer copies become
    device-to-device copies): base-range shards, window shards and the pieces API give the unsharded proof"""
    cases.test_emu_groth16_prove_multi(gpu_ctx, c, nshards, precompute, logn=12)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("group", [0, 1], ids=["G1", "G2"])
def test_msm_table_window_ranges(gpu_ctx, c, group):
    cases.test_emu_msm_table_window_ranges(gpu_ctx, c, group)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("circuit", ["cubic", "commit"])
def test_proving_key_files(gpu_ctx, c, circuit, tmp_path):
    cases.test_emu_proving_key_files(gpu_ctx, c, circuit, tmp_path)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_proof_unmarshal(gpu_ctx, c):
    cases.test_emu_proof_unmarshal(gpu_ctx, c)


@pytest.mark.parametrize("fmt", [0, 1, 2], ids=["WriteTo", "WriteRawTo", "WriteDump"])
def test_proving_key_file_2_20_roundtrip(gpu_ctx, fmt, tmp_path):
    """a 2^20-constraint BN254 key (5.2 M points) written in each of the three layouts by the library's writer, read back from
    the file descriptor straight into HBM (compressed points decoded by the device kernel: one square root each), and used:
    the proof equals the proof of the key pinned from the host arrays"""
    import time
    from gnark_amd import synth
    c = BN254
    inst = synth.make_instance(gpu_ctx, c.name, 20, 0xF11E, want_dlogs=False)
    ref = inst.proving_key(gpu_ctx)
    try:
        want = groth16.Prove(ref, inst.solution, inst.nb_public, inst.r, inst.s).raw()
    finally:
        ref.FreeGPUResources()
    path = tmp_path / "key.bin"
    with open(path, "wb") as f:
        size = groth16.WriteKey(gpu_ctx, c.name, f, fmt, domain_cardinality=inst.n, **inst.key)
    t0 = time.perf_counter()
    with open(path, "rb") as f:
        pk = groth16.ProvingKey.ReadFrom(gpu_ctx, c.name, f)
    load_s = time.perf_counter() - t0
    try:
        assert pk.bytes_read == size
        got = groth16.Prove(pk, inst.solution, inst.nb_public, inst.r, inst.s).raw()
    finally:
        pk.FreeGPUResources()
    assert np.array_equal(got, want)
    print("key file format %d: %.0f MiB loaded and pinned in %.2f s" % (fmt, size / 2**20, load_s))


def test_compute_h_2_20_polynomial_identity(gpu_ctx):
    """size-independent property of computeH at 2^20: A(x)B(x) - C(x) == H(x)(x^n - 1) at a random point, with A, B, C
    interpolated by the CPU oracle and H (bit-reversed coefficients, deg <= n-2) from the device."""
    c = BN254
    n = 1 << 20
    m = n - 12345
    buf = gpu_ctx.malloc(3 * n * 32)
    gpu_ctx.lib.check(gpu_ctx.lib.ga_gen_scalars(gpu_ctx.handle, c.cid, 4242, 3 * n, buf.ptr))
    v = buf.to_host((3 * n, 4))
    buf.free()
    A, B = v[:m], v[n:n + m]
    Cc = oracle.fr_mul(c.cid, A, B)
    d = fft.Domain(gpu_ctx, c.name, m)
    assert d.Cardinality == n
    h_bitrev = d.compute_h(A, B, Cc)
    d.close()
    idx = np.arange(n, dtype=np.uint64)
    rev = np.zeros(n, dtype=np.int64)
    for b in range(20):
        rev |= (((idx >> np.uint64(b)) & np.uint64(1)).astype(np.int64)) << (19 - b)
    h = h_bitrev[rev]                                   # natural coefficient order
    assert not h[n - 1].any()                           # deg H <= n-2 (setup.go:247-249)
    pad = lambda x: np.concatenate([x, np.zeros((n - x.shape[0], 4), np.uint64)])
    coef = lambda ev: oracle.fft(c.cid, pad(ev), 1, 0, False)[rev]     # iFFT (DIF -> bit-reversed) -> natural order
    xv = 0x1234567890ABCDEF1234567890ABCDEF % c.r
    x = fr_to_arr(c, [xv])[0]
    ev = lambda co: pyref.from_mont_limbs(oracle.fr_horner(c.cid, co, x), c.r)
    lhs = (ev(coef(A)) * ev(coef(B)) - ev(coef(Cc))) % c.r
    assert lhs == ev(h) * (pow(xv, n, c.r) - 1) % c.r


def test_groth16_sharded_key_on_device(gpu_ctx):
    cases.test_emu_groth16_sharded_key_single_process(gpu_ctx, 3)


# ---- BASELINE configs 3, 4, 5 at their STATED sizes ("no parity check at their stated size") ------------------------
import os as _os

_NT = max(1, min(64, _os.cpu_count() or 1))


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_groth16_2_24_known_dlogs(gpu_ctx, c):
    """configs 3 (BN254) and 4 (BLS12-381, single-GPU leg): Groth16 Prove at 2^24 constraints on the known-dlog synthetic key with
    C = A o B: Ar, Bs, Krs and the four pre-randomisation sums equal the points whose exponents the CPU oracle
    computes with O(n) dot products; h (2^24 coefficients) satisfies A(x)B(x) - C(x) = H(x)(x^n - 1) with A, B, C evaluated by
    the oracle's barycentric formula.  Exercises the c = 22 shared-bucket plans, the Fe2 table kernel, the shared witness sort
    and the 3-pass NTT at the judged size."""
    cases.check_groth16_known_dlogs(gpu_ctx, c, 24, nthreads=_NT, proofs=2)


def test_groth16_2_26_known_dlogs_beyond_the_table_budget(gpu_ctx):
    """four times the headline size: 2^26 constraints, BN254.  The five window tables (288 GiB) no longer fit HBM together;
    precompute = 0 builds the ones that fit beside the proof's scratch -- on an empty device A, B and K, which share one witness sort;
    in the middle of this suite, with the scratch of the earlier tests resident, fewer -- and the rest (Z, G2.B, ...) run as un-pinned
    MSMs (per-window bucket sets over 2^26 points); 2^26-point transforms (4-pass plans).  The proof must equal the closed form from
    the key's discrete logs whichever mix of paths it took; h satisfies the identity.  (`tools/size_sweep.py` ran the same check at
    2^27 -- 230 GiB of HBM -- profiles/README.md, round 3 batch M.)"""
    cases.check_groth16_known_dlogs(gpu_ctx, BN254, 26, nthreads=_NT, proofs=1, precompute=0)


@pytest.mark.parametrize("c,group", [(BN254, 1), (BLS12_381, 0), (BLS12_381, 1)], ids=["bn254-G2", "bls12-381-G1", "bls12-381-G2"])
def test_msm_2_24_other_shapes_dlog(gpu_ctx, c, group):
    """the MSM shapes of configs 3/4 besides BN254 G1, 2^24 points each, raw bases (ga_msm) and pinned table (ga_msm_table_run)"""
    n = 1 << 24
    bases, dlogs, scal = _device_inputs(gpu_ctx, c, group, n, 0x5EED0100 + 16 * c.cid + group)
    S, K = scal.to_host((n, 4)), dlogs.to_host((n, 4))
    dlogs.free()
    want = _expect_from_dlogs(c, group, S, K)
    del S, K
    raw = oracle.jac_to_affine(c.cid, group, ecc.MultiExp(gpu_ctx, c.name, group, bases, scal, n=n))
    assert np.array_equal(raw, want)
    table = ecc.PrecomputedBases(gpu_ctx, c.name, group, bases, n=n)
    try:
        got = oracle.jac_to_affine(c.cid, group, table.MultiExp(scal))
    finally:
        table.free()
    assert np.array_equal(got, want)
    for b in (bases, scal):
        b.free()


@pytest.mark.parametrize("c,modes", [(BN254, ((0, 0, 1), (1, 1, 1), (1, 0, 0))), (BLS12_381, ((0, 1, 1),))], ids=["bn254", "bls12-381"])
def test_fft_2_24_vs_c_oracle(gpu_ctx, c, modes):
    """2^24-point transforms (3 HBM passes) against the C oracle's radix-2 transform, element for element; modes are
    (inverse, decimation, on_coset) -- the ones computeH uses plus a plain inverse"""
    n = 1 << 24
    scal = gpu_ctx.malloc(n * 32)
    gpu_ctx.lib.check(gpu_ctx.lib.ga_gen_scalars(gpu_ctx.handle, c.cid, 2424, n, scal.ptr))
    a = scal.to_host((n, 4))
    scal.free()
    d = fft.Domain(gpu_ctx, c.name, n)
    try:
        for inv, dec, coset in modes:
            got = (d.FFTInverse if inv else d.FFT)(a, dec, bool(coset))
            want = oracle.fft(c.cid, a, inv, dec, bool(coset), nthreads=_NT)
            assert np.array_equal(got, want), (inv, dec, coset)
    finally:
        d.close()


@pytest.mark.parametrize("pinned", [False, True], ids=["plain", "pinned"])
def test_plonk_quotient_2_22_identity(gpu_ctx, pinned):
    """config 5 at its stated size (n = 2^22 gates, 4n = 2^24): grand product + quotient on the device, identity at a random
    point with every polynomial evaluated by the CPU oracle from its values"""
    cases.check_plonk_quotient_identity(gpu_ctx, BN254, 22, nthreads=_NT, pinned=pinned)


def test_rccl_collectives_and_sharded_proof_on_one_gpu():
    """The multi-GPU prover's RCCL path on a 1-GPU box: a ONE-rank process group over the nccl (= RCCL) backend, every collective
    gnark_amd/multigpu.py uses on DEVICE tensors, one sharded MSM exchange, and a 2^16 sharded proof walked through the full
    collective schedule (sliced uploads gathered on the chain owner, scatter of h, all_gather of the partial sums) whose bytes
    must equal the plain single-GPU proof.  Runs in a child process (bench.py --nccl-selftest-worker: the same code the N = 1 bench
    line reports as "nccl_selftest")."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GA_SELFTEST_BACKEND="nccl", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--nccl-selftest-worker"], capture_output=True, text=True, env=env, cwd=root, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("NCCL_SELFTEST ")]
    assert lines, (r.returncode, r.stdout[-1000:], r.stderr[-3000:])
    d = json.loads(lines[-1][len("NCCL_SELFTEST "):])
    assert d.get("ok") is True, d
    assert d["collectives"]["backend"] == "nccl" and d["collectives"]["device_tensors"] is True
    assert all(d["collectives"][k] is True for k in ("all_gather", "gather", "scatter", "broadcast", "all_reduce"))
    assert d["sharded_proof_same_bytes"] is True and d["msm_all_gather"] is True and d["constraints"] == 1 << 16


def test_gnark_fixture_directories(gpu_ctx, tmp_path):
    """golden directories in gnark's own formats (tests/gnark_fixture.py; go/.../internal/fixtures/gen_fixtures_test.go writes them from a
    gnark checkout with the groth16_rs.patch hook): key bytes -> HBM, solution -> proof, proof BYTES == the directory's.  Consumes
    tests/golden/gnark/* when present (none yet: no Go toolchain) and the same layout written by the oracle."""
    cases.test_emu_gnark_fixture_directories(gpu_ctx, tmp_path)
