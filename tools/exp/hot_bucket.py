#!/usr/bin/env python3
"""How a boolean-heavy witness (half the scalars equal to 1, a quarter 0) changes the MSM time: the digit-1 bucket of window 0
becomes one bucket of n/2 points (split into tasks, then merged)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gnark_amd import _lib, ecc
from gnark_amd.device import Context, affine_words
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << logn
ctx = Context(0); lib = ctx.lib
for group in (0, 1):
    b = ctx.malloc(n * affine_words(0, group) * 8)
    lib.check(lib.ga_gen_bases(ctx.handle, 0, group, 5, n, b.ptr, None))
    t = ecc.PrecomputedBases(ctx, 0, group, b, n=n)
    s = ctx.malloc(n * 32)
    lib.check(lib.ga_gen_scalars(ctx.handle, 0, 9, n, s.ptr))
    S = s.to_host((n, 4))
    one_mont = np.array([(((1 << 256) % 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    rng = np.random.default_rng(1)
    sel = rng.integers(0, 4, n)
    W = S.copy(); W[sel < 2] = one_mont; W[sel == 2] = 0
    sw = ctx.to_device(W)
    for name, sc in (("uniform", s), ("boolean-heavy", sw)):
        t.MultiExp(sc)
        ctx.profile(True); ctx.profile_reset()
        t0 = time.perf_counter()
        for _ in range(3):
            t.MultiExp(sc)
        el = (time.perf_counter() - t0) / 3 * 1e3
        st = {}
        for k, v in ctx.profile_read():
            st[k] = st.get(k, 0) + v / 3
        ctx.profile(False)
        print("G%d %-14s %.2f ms  %s" % (group + 1, name, el, {k: round(v, 2) for k, v in st.items()}))
    t.free(); b.free(); s.free(); sw.free()
