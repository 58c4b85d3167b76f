#!/usr/bin/env python3
"""Where a SMALL Groth16 proof spends its time: wall time per proof (split and single-lane schedule) against the sum of the device
stages (stage profiler, single-lane) -- the difference is launch latency, host round trips and thread hand-offs.

  python tools/small_profile.py --logs 14,16,18,20 [--curve bn254]
"""
import argparse
import collections
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", default="bn254")
    ap.add_argument("--logs", default="14,16,18,20")
    ap.add_argument("--proofs", type=int, default=20)
    args = ap.parse_args()
    from gnark_amd import groth16, synth
    from gnark_amd.device import Context
    ctx = Context(0)
    for logn in [int(x) for x in args.logs.split(",")]:
        inst = synth.make_instance(ctx, args.curve, logn, 0x5EED0005, want_dlogs=False)
        pk = inst.proving_key(ctx, precompute=1)
        sol, nbp, r, s = inst.solution, inst.nb_public, inst.r, inst.s
        out = {"curve": args.curve, "log_n": logn}
        for mode, envv in (("split", "1"), ("single_lane", "0")):
            os.environ["GA_G16_SPLIT"] = envv
            for _ in range(3):
                groth16.Prove(pk, sol, nbp, r, s)
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(args.proofs):
                groth16.Prove(pk, sol, nbp, r, s)
            ctx.sync()
            out[mode + "_ms"] = round((time.perf_counter() - t0) * 1e3 / args.proofs, 3)
        os.environ.pop("GA_G16_SPLIT", None)
        ctx.profile(True)
        ctx.profile_reset()
        t0 = time.perf_counter()
        for _ in range(4):
            groth16.Prove(pk, sol, nbp, r, s)
        ctx.sync()
        out["profiled_ms"] = round((time.perf_counter() - t0) * 1e3 / 4, 3)
        st = collections.OrderedDict()
        for k, v in ctx.profile_read():
            e = st.setdefault(k, [0, 0.0])
            e[0] += 1
            e[1] += v
        ctx.profile(False)
        out["stages_ms_per_proof"] = {k: [e[0] // 4, round(e[1] / 4, 4)] for k, e in st.items()}
        out["stage_sum_ms"] = round(sum(e[1] for e in st.values()) / 4, 3)
        out["launches_per_proof"] = sum(e[0] for e in st.values()) // 4
        pk.FreeGPUResources()
        print(json.dumps(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
