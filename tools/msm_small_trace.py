#!/usr/bin/env python3
"""Raw (un-pinned, `ga_msm`) and table MSMs at a given size, R repetitions each -- run under `rocprofv3 --kernel-trace --stats`
to see which kernels a SMALL MSM spends its time in; prints wall ms per MSM and the stage profile.

  python tools/msm_small_trace.py --log-n 20 --reps 20 [--mode raw|table|both]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", default="bn254")
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--mode", default="both")
    ap.add_argument("--group", type=int, default=0)
    args = ap.parse_args()
    import gnark_amd.device
    from gnark_amd import ecc
    from gnark_amd.device import Context, curve_id
    ctx = Context(0)
    lib = ctx.lib
    cid = curve_id(args.curve)
    n = 1 << args.log_n
    words = gnark_amd.device.affine_words(cid, args.group)
    bases = ctx.malloc(n * words * 8)
    scal = ctx.malloc(n * 32)
    lib.check(lib.ga_gen_bases(ctx.handle, cid, args.group, 0x5EED0002, n, bases.ptr, None))
    lib.check(lib.ga_gen_scalars(ctx.handle, cid, 0x5EED0001, n, scal.ptr))
    out = {"log_n": args.log_n, "curve": args.curve, "group": args.group}

    def timed(fn, tag):
        for _ in range(3):
            fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            fn()
        ctx.sync()
        out[tag + "_ms"] = round((time.perf_counter() - t0) * 1e3 / args.reps, 4)
        ctx.profile(True)
        ctx.profile_reset()
        fn()
        ctx.sync()
        out[tag + "_stages"] = [(k, round(ms, 4)) for k, ms in ctx.profile_read()]
        ctx.profile(False)

    if args.mode in ("raw", "both"):
        timed(lambda: ecc.MultiExp(ctx, cid, args.group, bases, scal, n=n), "raw")
    if args.mode in ("table", "both"):
        table = ecc.PrecomputedBases(ctx, cid, args.group, bases, n=n)
        timed(lambda: table.MultiExp(scal), "table")
        table.free()
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
