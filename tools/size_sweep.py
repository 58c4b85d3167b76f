#!/usr/bin/env python3
"""Groth16 proof time against the number of constraints (one GPU, synthetic known-dlog key, W/A/B/C on the host, proof on the host;
precompute = 0: the library decides whether the window tables fit HBM).  One JSON object per size on stdout.

  python tools/size_sweep.py --curve bn254 --logs 16,18,20,22,24,25,26 [--check-max 26]

--check-max L: sizes up to 2^L are also CHECKED against the closed form from the key's discrete logs (oracle/checkers.py; CPU
time grows with n) -- the checker is test infrastructure, outside every timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", default="bn254")
    ap.add_argument("--logs", default="16,18,20,22,24")
    ap.add_argument("--proofs", type=int, default=3)
    ap.add_argument("--precompute", type=int, default=0)
    ap.add_argument("--check-max", type=int, default=0)
    ap.add_argument("--one-shot", action="store_true", help="also time pin (plain vectors) + prove + free as ONE operation")
    ap.add_argument("--b-density", type=float, default=1.0, help="fraction of the wires that appear in B (pk.InfinityB elsewhere): real circuits "
                    "are sparse on the B side; the default is the dense worst case of BASELINE config 3")
    args = ap.parse_args()
    from gnark_amd import groth16, synth
    from gnark_amd.device import Context
    for logn in [int(x) for x in args.logs.split(",")]:
        ctx = Context(0)   # a context per size: the scratch a smaller size left behind must not eat into a larger key's table budget
        info0 = ctx.info()
        t0 = time.perf_counter()
        checked = None
        kw = {}
        if args.b_density < 1.0:
            import numpy as np
            rng = np.random.default_rng(1234)
            kw["inf_b"] = np.flatnonzero(rng.random(1 << logn) >= args.b_density)
        if logn <= args.check_max:   # (in a context of its own: the check's proof must not leave scratch behind that the timed key's table budget sees)
            import checkers
            import pyref
            t0 = time.perf_counter()
            cctx = Context(0)
            checkers.check_groth16_known_dlogs(cctx, pyref.CURVES[args.curve], logn, nthreads=min(64, os.cpu_count() or 1), proofs=1, precompute=args.precompute, inst_kw=kw)
            cctx.close()
            checked = round(time.perf_counter() - t0, 1)
            info0 = ctx.info()
        inst = synth.make_instance(ctx, args.curve, logn, 0x5EED0005, want_dlogs=False, **kw)
        t1 = time.perf_counter()
        pk = inst.proving_key(ctx, precompute=args.precompute)
        ctx.sync()
        t2 = time.perf_counter()
        sol, nbp, r, s = inst.solution, inst.nb_public, inst.r, inst.s
        for _ in range(2):
            groth16.Prove(pk, sol, nbp, r, s)
        ctx.sync()
        t3 = time.perf_counter()
        for _ in range(args.proofs):
            groth16.Prove(pk, sol, nbp, r, s)
        ctx.sync()
        ms = (time.perf_counter() - t3) * 1e3 / args.proofs
        info1 = ctx.info()
        lanes = ctx.lane_stats()
        lay = groth16.ShardLayout(pk)
        pk.FreeGPUResources()
        one_shot = None
        if args.one_shot:   # the Go package's default (PinToGPU = false): upload the plain key, prove, free -- every proof
            for k in range(3):
                t4 = time.perf_counter()
                pk1 = inst.proving_key(ctx, precompute=-1)
                groth16.Prove(pk1, sol, nbp, r, s)
                pk1.FreeGPUResources()
                ctx.sync()
                one_shot = round((time.perf_counter() - t4) * 1e3, 2)   # the last of three
        print(json.dumps({"curve": args.curve, "log_n": logn, "ms_per_proof": round(ms, 3), "proofs_per_s": round(1e3 / ms, 3),
                          "constraints_per_s": round((1 << logn) / ms * 1e3), "key_pin_s": round(t2 - t1, 2),
                          "hbm_used_gib": round((info0["free_bytes"] - info1["free_bytes"]) / 2**30, 2),
                          "scratch_gib": round((lanes["lanes01_scratch_bytes"] + lanes["lanes23_scratch_bytes"]) / 2**30, 2),
                          "tables": "".join(k + ("*" if lay["wire_indexed"].get(k) else "") + " " for k, v in lay["tables"].items() if v).strip() or "none",
                          "checked_known_dlogs_s": checked, "b_density": args.b_density,
                          "one_shot_pin_prove_free_ms": one_shot}), flush=True)
        del inst, sol
        ctx.close()


if __name__ == "__main__":
    main()
