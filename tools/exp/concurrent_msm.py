#!/usr/bin/env python3
"""Experiment: do two MSMs issued from two contexts (two streams) on one GPU overlap usefully?  Total time of 2 x K MSMs run
back to back on one context vs K MSMs on each of two contexts from two host threads."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gnark_amd import _lib, ecc
from gnark_amd.device import Context

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
K = 5
n = 1 << logn
lib = _lib.load()
ctxs = [Context(0, lib), Context(0, lib)]
tabs, scal = [], []
for i, ctx in enumerate(ctxs):
    b = ctx.malloc(n * 64)
    lib.check(lib.ga_gen_bases(ctx.handle, 0, _lib.G1, 0x1234 + i, n, b.ptr, None))
    s = ctx.malloc(n * 32)
    lib.check(lib.ga_gen_scalars(ctx.handle, 0, 0x99 + i, n, s.ptr))
    tabs.append(ecc.PrecomputedBases(ctx, 0, _lib.G1, b, n=n))
    scal.append(s)
    b.free()

def run(i, k):
    for _ in range(k):
        tabs[i].MultiExp(scal[i])

run(0, 1); run(1, 1)
t0 = time.time(); run(0, K); run(1, K); seq = time.time() - t0
t0 = time.time()
th = [threading.Thread(target=run, args=(i, K)) for i in range(2)]
[t.start() for t in th]; [t.join() for t in th]
par = time.time() - t0
print("sequential %.2f ms per MSM, two streams %.2f ms per MSM (%.1f %%)" % (seq * 1e3 / (2 * K), par * 1e3 / (2 * K), 100 * (seq - par) / seq))
