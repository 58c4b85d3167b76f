#!/usr/bin/env python3
"""A/B timing of the hot kernels for one build of the library (GA_LIB_PATH selects a tools/build_variant.sh build) and one set
of run-time knobs (GA_NTT_PLAN, GA_G16_SPLIT, ...), one JSON line per run -- the harness behind the variant tables in
profiles/README.md.  Results carry a hash of the outputs, so that variants can be compared bit for bit with the shipped build.

  python tools/ab_kernels.py --parts ntt,msm --tag base
  GA_NTT_PLAN=8,8,8 python tools/ab_kernels.py --parts ntt --tag plan888
  GA_LIB_PATH=$PWD/gnark_amd/variants/libgnark_amd_twu.so python tools/ab_kernels.py --parts ntt --tag twu
  python tools/ab_kernels.py --parts g16 --tag split         # GA_G16_SPLIT=0 for the single-lane schedule
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def stage_avg(recs):
    agg = {}
    for name, ms in recs:
        a = agg.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += ms
    return {k: {"n": v[0], "avg_ms": round(v[1] / v[0], 4), "total_ms": round(v[1], 3)} for k, v in agg.items()}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", default="ntt,msm")
    ap.add_argument("--log-n", type=int, default=24)
    ap.add_argument("--curve", default="bn254")
    ap.add_argument("--tag", default="")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--proofs", type=int, default=5)
    ap.add_argument("--knob", default="", help="g16ab: NAME=v1,v2[,v3]: a run-time knob of the library toggled IN this process, the settings interleaved round by round")
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    import gnark_amd
    from gnark_amd import _lib, ecc, fft
    from gnark_amd.device import curve_id
    cid = curve_id(args.curve)
    ctx = gnark_amd.Context(0)
    lib = ctx.lib
    n = 1 << args.log_n
    out = {"tag": args.tag, "lib": os.path.basename(lib.path), "curve": args.curve, "log_n": args.log_n,
           "env": {k: v for k, v in os.environ.items() if k.startswith("GA_") and k != "GA_LIB_PATH"}}
    parts = args.parts.split(",")

    if "ntt" in parts:
        d = fft.Domain(ctx, cid, n)
        v = ctx.malloc(n * 32)
        lib.check(lib.ga_gen_scalars(ctx.handle, cid, 0xABCD, n, v.ptr))
        res = {}
        # the three transform shapes computeH uses: iFFT DIF, coset FFT DIT, coset iFFT DIF -- each on the previous one's output
        shapes = [("ifft_dif", fft.Domain.FFTInverse, fft.DIF, False), ("fft_dit_coset", fft.Domain.FFT, fft.DIT, True),
                  ("ifft_dif_coset", fft.Domain.FFTInverse, fft.DIF, True), ("fft_dif", fft.Domain.FFT, fft.DIF, False)]
        for name, fn, dec, coset in shapes:
            fn(d, v, dec, coset)   # warm
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                fn(d, v, dec, coset)
            ctx.sync()
            res[name + "_ms"] = round((time.perf_counter() - t0) * 1e3 / args.reps, 4)
        # deterministic content check: fresh input through the chain computeH applies
        lib.check(lib.ga_gen_scalars(ctx.handle, cid, 0xABCD, n, v.ptr))
        d.FFTInverse(v, fft.DIF, False)
        res["sha_ifft_dif"] = sha(v.to_host((n, 4)))
        d.FFT(v, fft.DIT, True)
        res["sha_fft_dit_coset"] = sha(v.to_host((n, 4)))
        d.FFTInverse(v, fft.DIF, True)
        res["sha_ifft_dif_coset"] = sha(v.to_host((n, 4)))
        ctx.profile(True)
        ctx.profile_reset()
        d.FFTInverse(v, fft.DIF, False)
        d.FFT(v, fft.DIT, True)
        ctx.sync()
        res["passes"] = [(k, round(ms, 4)) for k, ms in ctx.profile_read()]
        ctx.profile(False)
        # computeH end to end on device buffers
        a, b, c = (ctx.malloc(n * 32) for _ in range(3))
        for k, buf in enumerate((a, b, c)):
            lib.check(lib.ga_gen_scalars(ctx.handle, cid, 0x1000 + k, n, buf.ptr))
        lib.check(lib.ga_compute_h(d.handle, a.ptr, b.ptr, c.ptr, n, a.ptr, 1))
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            lib.check(lib.ga_compute_h(d.handle, a.ptr, b.ptr, c.ptr, n, a.ptr, 1))
        ctx.sync()
        res["compute_h_ms"] = round((time.perf_counter() - t0) * 1e3 / args.reps, 4)
        for buf in (a, b, c, v):
            buf.free()
        d.close()
        out["ntt"] = res

    if "msm" in parts:
        res = {}
        for group, gname in ((_lib.G1, "g1"), (_lib.G2, "g2")):
            words = gnark_amd.device.affine_words(cid, group)
            bases = ctx.malloc(n * words * 8)
            scal = ctx.malloc(n * 32)
            lib.check(lib.ga_gen_bases(ctx.handle, cid, group, 0x5EED0002, n, bases.ptr, None))
            lib.check(lib.ga_gen_scalars(ctx.handle, cid, 0x5EED0001, n, scal.ptr))
            table = ecc.PrecomputedBases(ctx, cid, group, bases, n=n)
            bases.free()
            r0 = table.MultiExp(scal)
            ctx.profile(True)
            ctx.profile_reset()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                r = table.MultiExp(scal)
            ctx.sync()
            el = (time.perf_counter() - t0) * 1e3 / args.reps
            st = stage_avg(ctx.profile_read())
            ctx.profile(False)
            res[gname] = {"msm_ms": round(el, 3), "accumulate_ms": st.get("msm_accumulate", {}).get("avg_ms"),
                          "reduce_ms": st.get("msm_reduce", {}).get("avg_ms"), "sort_ms": st.get("msm_sort", {}).get("avg_ms"),
                          "sha": sha(ecc.jac_to_affine(cid, group, r)),
                          # (as affine points: the Jacobian representative is not fixed from call to call, include/gnark_amd.h ga_msm)
                          "stable": bool(np.array_equal(ecc.jac_to_affine(cid, group, r), ecc.jac_to_affine(cid, group, r0)))}
            table.free()
            scal.free()
        out["msm"] = res

    if "g16" in parts:
        from gnark_amd import groth16, synth
        inst = synth.make_instance(ctx, cid, args.log_n, 0x5EED0005, want_dlogs=False)
        pk = inst.proving_key(ctx, precompute=1)
        sol, nbp, r, s = inst.solution, inst.nb_public, inst.r, inst.s
        res = {}
        for mode, envv in (("split", "1"), ("single_lane", "0")):
            os.environ["GA_G16_SPLIT"] = envv
            for _ in range(2):
                proof = groth16.Prove(pk, sol, nbp, r, s)
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(args.proofs):
                proof = groth16.Prove(pk, sol, nbp, r, s)
            ctx.sync()
            res[mode + "_ms"] = round((time.perf_counter() - t0) * 1e3 / args.proofs, 2)
            res[mode + "_sha"] = sha(proof.raw())

            def run_pair(count):
                def prover():
                    for _ in range(count):
                        groth16.Prove(pk, sol, nbp, r, s)
                th = [threading.Thread(target=prover) for _ in range(2)]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                ctx.sync()
            run_pair(2)
            l0 = ctx.lane_stats()
            t0 = time.perf_counter()
            run_pair(args.proofs)
            res[mode + "_two_callers_ms"] = round((time.perf_counter() - t0) * 1e3 / (2 * args.proofs), 2)
            l1 = ctx.lane_stats()
            res[mode + "_two_callers_lanes"] = {k: l1[k] - l0[k] for k in ("lanes01_proofs", "lanes23_proofs", "queued_proofs", "split_proofs")}
        os.environ.pop("GA_G16_SPLIT", None)
        pk.FreeGPUResources()
        out["g16"] = res
    if "g16ab" in parts:
        # one pinned key, one process, one box: the settings of a run-time knob interleaved (A B A B ...), so that drift and box spread
        # cancel; per setting the one-caller and two-caller proof times of every round, the proof hash, and the stage table of a
        # profiled (single-lane) proof
        from gnark_amd import groth16, synth
        name, vals = args.knob.split("=")
        vals = vals.split(",")
        inst = synth.make_instance(ctx, cid, args.log_n, 0x5EED0005, want_dlogs=False)
        pk = inst.proving_key(ctx, precompute=1)
        sol, nbp, r, s = inst.solution, inst.nb_public, inst.r, inst.s
        res = {v: {"one_caller_ms": [], "two_callers_ms": []} for v in vals}

        def run_pair(count):
            def prover():
                for _ in range(count):
                    groth16.Prove(pk, sol, nbp, r, s)
            th = [threading.Thread(target=prover) for _ in range(2)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            ctx.sync()
        for v in vals:   # warm both lane pairs under every setting (scratch sizes differ)
            os.environ[name] = v
            for _ in range(2):
                groth16.Prove(pk, sol, nbp, r, s)
            run_pair(2)
        for rnd in range(args.rounds):
            for v in vals:
                os.environ[name] = v
                groth16.Prove(pk, sol, nbp, r, s)   # (a lane-0 call reads the knobs)
                ctx.sync()
                t0 = time.perf_counter()
                for _ in range(args.proofs):
                    proof = groth16.Prove(pk, sol, nbp, r, s)
                ctx.sync()
                res[v]["one_caller_ms"].append(round((time.perf_counter() - t0) * 1e3 / args.proofs, 2))
                res[v]["sha"] = sha(proof.raw())
                t0 = time.perf_counter()
                run_pair(args.proofs)
                res[v]["two_callers_ms"].append(round((time.perf_counter() - t0) * 1e3 / (2 * args.proofs), 2))
        for v in vals:
            os.environ[name] = v
            groth16.Prove(pk, sol, nbp, r, s)
            ctx.profile(True)
            ctx.profile_reset()
            for _ in range(2):
                groth16.Prove(pk, sol, nbp, r, s)
            ctx.sync()
            res[v]["stages_ms_per_proof"] = {k: round(x["total_ms"] / 2, 3) for k, x in stage_avg(ctx.profile_read()).items()}
            ctx.profile(False)
            res[v]["one_caller_best_ms"] = min(res[v]["one_caller_ms"])
            res[v]["two_callers_best_ms"] = min(res[v]["two_callers_ms"])
        os.environ.pop(name, None)
        pk.FreeGPUResources()
        out["g16ab"] = {"knob": name, "settings": res}
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
