#!/usr/bin/env python3
"""What pinning a proving key costs and what it buys: window-table build times per base vector at 2^logn and
the Groth16 proof time with (precompute = 1) and without (precompute = -1) the tables.  One JSON line per curve.
   python tools/pin_cost.py [logn]        -> profiles/r02_e_pin_cost.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from gnark_amd import _lib, ecc, groth16, synth  # noqa: E402
from gnark_amd.device import Context, affine_words  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << logn
ctx = Context(0)
lib = ctx.lib
for cname, cid in (("bn254", 0), ("bls12-381", 1)):
    res = {"curve": cname, "log_n": logn}
    for gname, group in (("G1", 0), ("G2", 1)):
        words = affine_words(cid, group)
        b = ctx.malloc(n * words * 8)
        lib.check(lib.ga_gen_bases(ctx.handle, cid, group, 77, n, b.ptr, None))
        ctx.sync()
        t0 = time.perf_counter()
        t = ecc.PrecomputedBases(ctx, cid, group, b, n=n)
        ctx.sync()
        res["table_build_s_" + gname] = round(time.perf_counter() - t0, 3)
        res["table_GiB_" + gname] = round(t.info()["table_bytes"] / 2**30, 1)
        host = b.to_host((n, words))
        t0 = time.perf_counter()
        d = ctx.to_device(host)
        ctx.sync()
        res["upload_only_s_" + gname] = round(time.perf_counter() - t0, 3)
        d.free()
        t.free()
        b.free()
        del host
    inst = synth.make_instance(ctx, cid, logn, 0x5EED0005, want_dlogs=False)
    for pre in (1, -1):
        t0 = time.perf_counter()
        pk = inst.proving_key(ctx, precompute=pre)
        ctx.sync()
        pin = time.perf_counter() - t0
        groth16.Prove(pk, inst.solution, inst.nb_public, inst.r, inst.s)
        t0 = time.perf_counter()
        for _ in range(3):
            p = groth16.Prove(pk, inst.solution, inst.nb_public, inst.r, inst.s)
        ms = (time.perf_counter() - t0) / 3 * 1e3
        key = "tables" if pre == 1 else "plain_bases"
        res[key] = {"pin_s": round(pin, 2), "ms_per_proof": round(ms, 1), "sha": __import__("hashlib").sha256(p.WriteTo()).hexdigest()[:12]}
        pk.FreeGPUResources()
    d_pin = res["tables"]["pin_s"] - res["plain_bases"]["pin_s"]
    d_ms = res["plain_bases"]["ms_per_proof"] - res["tables"]["ms_per_proof"]
    res["break_even_proofs"] = int(d_pin * 1e3 / d_ms) + 1 if d_ms > 0 else None
    print(json.dumps(res), flush=True)
    del inst
