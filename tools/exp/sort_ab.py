#!/usr/bin/env python3
"""Same-box A/B of the MSM front end (digits + sort + task lists) over the run-time knobs, on the three key shapes that matter:
  raw    ga_msm on 2^log_n un-pinned bases      (13 windows x 2^19 buckets at 2^24: 6.8 M keys, the Go shim's default path)
  table  ga_msm_table_run over a pinned table   (one shared set of 2^21 buckets at 2^24: the headline MSM, every MSM of a proof)
  batch  ga_msm_table_run_batch, 3 vectors of 2^(log_n - 2) over a GA_TABLE_BATCHED table  (PLONK's grouped commitments)
One JSON line per (shape, setting): wall ms per MSM and the library's stage profile; results are compared as affine points.

  python tools/exp/sort_ab.py [--log-n 24] [--modes 0,3] [--shapes raw,table,batch]        (GA_LIB_PATH selects another build)
(profiles/r05_a_sort_ab_2p24.txt was made with the one-batch run-time modes of `GA_MSM_SORT_MODE`, which no longer exist.)
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=24)
    ap.add_argument("--modes", default="3", help="values of GA_MSM_XCD to time (bit 0 per-XCD slices in the first level, bit 1 XCD swizzle in the second)")
    ap.add_argument("--shapes", default="raw,table,batch")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--library", type=int, default=1, help="also time the library sort (GA_MSM_FUSE_MIN above every size)")
    ap.add_argument("--extra-env", default="", help="K=V,K=V applied to every run")
    args = ap.parse_args()
    import gnark_amd
    from gnark_amd import _lib, ecc
    from gnark_amd.device import affine_words
    for kv in filter(None, args.extra_env.split(",")):
        k, v = kv.split("=")
        os.environ[k] = v
    ctx = gnark_amd.Context(0)
    lib = ctx.lib
    cid, group = 0, _lib.G1
    n = 1 << args.log_n
    words = affine_words(cid, group)
    bases = ctx.malloc(n * words * 8)
    lib.check(lib.ga_gen_bases(ctx.handle, cid, group, 0x5EED0002, n, bases.ptr, None))
    scal = [ctx.malloc(n * 32) for _ in range(3)]
    for k, s in enumerate(scal):
        lib.check(lib.ga_gen_scalars(ctx.handle, cid, 0x5EED0001 + k, n, s.ptr))
    aff = lambda r: ecc.jac_to_affine(cid, group, r, lib=lib)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]

    def timed(fn):
        fn()
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            r = fn()
        ctx.sync()
        wall = (time.perf_counter() - t0) * 1e3 / args.reps
        ctx.profile(True)
        ctx.profile_reset()
        for _ in range(2):
            fn()
        ctx.sync()
        agg = {}
        for name, ms in ctx.profile_read():
            agg[name] = agg.get(name, 0.0) + ms / 2
        ctx.profile(False)
        return wall, {k: round(v, 3) for k, v in agg.items()}, r

    settings = [("xcd%d" % int(m), {"GA_MSM_XCD": m, "GA_MSM_FUSE_MIN": "0"}) for m in args.modes.split(",")]
    if args.library:
        settings.append(("library", {"GA_MSM_FUSE_MIN": str(1 << 40)}))
    shapes = args.shapes.split(",")
    tables = {}
    if "table" in shapes:
        tables["table"] = ecc.PrecomputedBases(ctx, cid, group, bases, n=n)
    if "batch" in shapes:
        nq = n >> 2
        tables["batch"] = ecc.PrecomputedBases(ctx, cid, group, bases, n=nq, batched=True)
    for shape in shapes:
        ref = None
        for tag, env in settings:
            for k, v in env.items():
                os.environ[k] = v
            if shape == "raw":
                fn = lambda: ecc.MultiExp(ctx, cid, group, bases, scal[0], n=n)
            elif shape == "table":
                fn = lambda: tables["table"].MultiExp(scal[0])
            else:
                fn = lambda: tables["batch"].MultiExpBatch(scal)
            wall, st, r = timed(fn)
            pts = [aff(x) for x in (r if shape == "batch" else [r])]
            h = sha(np.concatenate(pts))
            ref = ref or h
            front = sum(v for k, v in st.items() if k in ("msm_digits", "msm_digits_pass1", "msm_sort", "msm_tasks"))
            print(json.dumps({"lib": os.path.basename(lib.path), "shape": shape, "log_n": args.log_n, "setting": tag, "ms_per_call": round(wall, 3), "front_end_ms": round(front, 3),
                              "stages_ms": st, "same_points": h == ref}), flush=True)
            for k in env:
                os.environ.pop(k, None)
    ctx.close()


if __name__ == "__main__":
    main()
