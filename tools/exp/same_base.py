#!/usr/bin/env python3
"""MSM over a DummySetup-like key: every base identical (backend/groth16/bn254/setup.go:517-543 fills pk with one point)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gnark_amd import _lib, ecc
from gnark_amd.device import Context, affine_words
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << logn
ctx = Context(0); lib = ctx.lib
for group in (0, 1):
    wa = affine_words(0, group)
    b = ctx.malloc(n * wa * 8)
    lib.check(lib.ga_gen_bases(ctx.handle, 0, group, 5, n, b.ptr, None))
    P = b.to_host((n, wa))
    P[:] = P[0]
    same = ctx.to_device(P)
    s = ctx.malloc(n * 32)
    lib.check(lib.ga_gen_scalars(ctx.handle, 0, 9, n, s.ptr))
    for name, bases in (("distinct", b), ("all-equal", same)):
        t = ecc.PrecomputedBases(ctx, 0, group, bases, n=n)
        t.MultiExp(s)
        ctx.profile(True); ctx.profile_reset()
        t0 = time.perf_counter()
        t.MultiExp(s)
        el = (time.perf_counter() - t0) * 1e3
        st = {}
        for k, v in ctx.profile_read():
            st[k] = st.get(k, 0) + v
        ctx.profile(False)
        print("2^%d G%d %-10s %.2f ms  %s" % (logn, group + 1, name, el, {k: round(v, 2) for k, v in st.items()}), flush=True)
        t.free()
    b.free(); same.free(); s.free()
