#!/usr/bin/env python3
"""BASELINE config 5: the kernel work of one PLONK BN254 proof at 2^22 gates:
10 G1 MSMs over an SRS of 2^22 points + 108 NTTs of size 2^22 (4 cosets x 12 polynomials x {iFFT, coset FFT} + 12) +
one iNTT of size 2^24, everything resident in HBM.  The PLONK round logic itself stays in Go."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnark_amd  # noqa: E402
from gnark_amd import _lib, ecc, fft  # noqa: E402


def run(ctx, logn=22, reps=3, reference_count=True):
    """returns [fused-pipeline result, (optionally) the reference's transform count issued one by one]"""
    results = []
    n = 1 << logn
    lib = ctx.lib
    srs = ctx.malloc(n * 64)
    lib.check(lib.ga_gen_bases(ctx.handle, 0, 0, 0x5EED0007, n, srs.ptr, None))
    srs_table = ecc.PrecomputedBases(ctx, "bn254", ecc.G1, srs, n=n, batched=True)   # the KZG SRS is a pinned key
    polys = [ctx.malloc(n * 32) for _ in range(12)]
    for i, p in enumerate(polys):
        lib.check(lib.ga_gen_scalars(ctx.handle, 0, 100 + i, n, p.ptr))
    big = ctx.malloc(4 * n * 32)
    lib.check(lib.ga_gen_scalars(ctx.handle, 0, 999, 4 * n, big.ptr))
    d = fft.Domain(ctx, "bn254", n)
    d4 = fft.Domain(ctx, "bn254", 4 * n)

    def proof_kernels():
        for k in range(10):                       # commitToLRO x3, Z, quotient x3, opening x2, linearised
            srs_table.MultiExp(polys[k % 12])
        for _ in range(4):                        # computeNumerator: per coset, every polynomial iFFT -> coset FFT
            for p in polys:
                d.FFTInverse(p, fft.DIF)
                d.FFT(p, fft.DIT, on_coset=True)
        for p in polys:                           # restore canonical form (prove.go:1101-1107, 1420)
            d.FFTInverse(p, fft.DIF)
        d4.FFTInverse(big, fft.DIF, on_coset=True)   # divideByZH: one size-4n inverse transform

    # ---- the same proof with the quotient computed by the fused device pipeline (ga_plonk_quotient):
    # 12 inverse transforms once + 12 coset transforms per coset + one size-4n inverse, instead of 108 + 1
    import ctypes as C
    qin = _lib.PlonkQuotientIn()
    for k, pbuf in zip(("l", "r", "o", "z", "ql", "qr", "qm", "qo", "qk", "s1", "s2", "s3"), polys):
        setattr(qin, k, pbuf.ptr)
    small = ctx.malloc(16 * 32)
    lib.check(lib.ga_gen_scalars(ctx.handle, 0, 4242, 16, small.ptr))
    host_small = small.to_host((16, 4))
    qin.bl, qin.br, qin.bo, qin.bz = (host_small[i:].ctypes.data for i in (0, 2, 4, 6))
    qin.alpha, qin.beta, qin.gamma = (host_small[i:].ctypes.data for i in (9, 10, 11))
    qin.lagrange_mask = (1 << 12) - 1          # s.x[...] are in Lagrange form when computeNumerator starts
    qin.flags = 1                              # GA_PLONK_ON_DEVICE
    perm = ctx.malloc(3 * n * 8)
    import numpy as np
    host_perm = np.random.default_rng(1).permutation(3 * n).astype(np.int64)
    ctx.lib.check(ctx.lib.ga_copy_to_device(ctx.handle, perm.ptr, host_perm.ctypes.data, 3 * n * 8))
    zbuf = ctx.malloc(n * 32)

    # circuit constants (Ql, Qr, Qm, Qo, S1, S2, S3) evaluated on the four cosets once and pinned: ga_plonk_pk_create
    ppk = C.c_void_p()
    t_pin = time.perf_counter()
    lib.check(lib.ga_plonk_pk_create(d.handle, d4.handle, C.byref(qin), C.byref(ppk)))
    pin_s = time.perf_counter() - t_pin

    def proof_fused():
        if pinned:   # the prover's own grouping (prove.go:404-489,558-633): [L],[R],[O] together, [Z], [H0],[H1],[H2] together, 3 openings
            srs_table.MultiExpBatch(polys[0:3])
            srs_table.MultiExp(polys[3])
            srs_table.MultiExpBatch(polys[4:7])
            for k in range(7, 10):
                srs_table.MultiExp(polys[k])
        else:
            for k in range(10):
                srs_table.MultiExp(polys[k % 12])
        lib.check(lib.ga_plonk_build_z(d.handle, polys[0].ptr, polys[1].ptr, polys[2].ptr, perm.ptr, host_small[10:].ctypes.data,
                                       host_small[11:].ctypes.data, 1, zbuf.ptr))
        if pinned:
            lib.check(lib.ga_plonk_quotient_pinned(ppk, C.byref(qin), big.ptr))
        else:
            lib.check(lib.ga_plonk_quotient(d.handle, d4.handle, C.byref(qin), big.ptr))

    for pinned in (True, False):
        proof_fused()
        ctx.profile(True)
        ctx.profile_reset()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            proof_fused()
        ctx.sync()
        el = (time.perf_counter() - t0) / reps
        st = {}
        for name, ms in ctx.profile_read():
            st[name] = st.get(name, 0.0) + ms / reps
        results.append(({"workload": "PLONK BN254 2^%d, %s: 10 G1 MSM (%s) + BuildRatioCopyConstraint + computeNumerator/divideByZH on device" % (
                              logn, "circuit constants pinned on all cosets (ga_plonk_pk_create, %.2f s once)" % pin_s if pinned else "fused quotient, nothing pinned",
                              "[L,R,O] and [H0,H1,H2] as two batches of three over the pinned SRS, ga_msm_table_run_batch" if pinned else "one by one"),
                          "ms_per_proof_kernels": round(el * 1e3, 2),
                          "msm_ms": round(sum(v for k, v in st.items() if k.startswith("msm_")), 2),
                          "ntt_ms": round(sum(v for k, v in st.items() if k.startswith("ntt_")), 2),
                          "plonk_pointwise_ms": round(sum(v for k, v in st.items() if k.startswith("plonk_")), 2),
                          "stages_ms": {k: round(v, 3) for k, v in st.items()}}))
        ctx.profile(False)
        if not reference_count:
            break   # bench.py: only the pinned variant

    def cleanup():
        lib.ga_plonk_pk_destroy(ppk)
        for b in polys + [srs, big, small, perm, zbuf]:
            b.free()
        srs_table.free()
        d.close()
        d4.close()

    if not reference_count:
        cleanup()
        return results
    proof_kernels()
    ctx.profile(True)
    ctx.profile_reset()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        proof_kernels()
    ctx.sync()
    el = (time.perf_counter() - t0) / reps
    st = {}
    for name, ms in ctx.profile_read():
        st[name] = st.get(name, 0.0) + ms / reps
    alg = 10 * 96 * n + 108 * 64 * n + 64 * 4 * n
    msm_ms = sum(v for k, v in st.items() if k.startswith("msm_"))
    ntt_ms = sum(v for k, v in st.items() if k.startswith("ntt_"))
    results.append(({"workload": "PLONK BN254 2^%d kernels: 10 G1 MSM + 108 NTT(2^%d) + 1 iNTT(2^%d)" % (logn, logn, logn + 2),
                      "ms_per_proof_kernels": round(el * 1e3, 2), "msm_ms": round(msm_ms, 2), "ntt_ms": round(ntt_ms, 2),
                      "algorithmic_bytes": alg, "hbm_frac": round(alg / el / 8e12, 5),
                      "ntt_hbm_frac": round((108 * 64 * n + 64 * 4 * n) / (ntt_ms * 1e-3) / 8e12, 5),
                      "stages_ms": {k: round(v, 3) for k, v in st.items()}}))
    ctx.profile(False)
    cleanup()
    return results


if __name__ == "__main__":
    with gnark_amd.Context(0) as _ctx:
        for _r in run(_ctx, int(os.environ.get("GA_PLONK_LOGN", "22"))):
            print(json.dumps(_r))
