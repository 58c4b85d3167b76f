#!/usr/bin/env python3
"""Experiment: the kernels of one Groth16 2^24 proof split over two contexts (two streams) of one GPU,
lane A = A, B1 (G1) + B2 (G2) MSMs, lane B = 7 NTTs + Z, K (G1) MSMs, against the same work back to back."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gnark_amd import _lib, ecc, fft
from gnark_amd.device import Context

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << logn
lib = _lib.load()
cA, cB = Context(0, lib), Context(0, lib)

def table(ctx, group, seed):
    words = 8 if group == 0 else 16
    b = ctx.malloc(n * words * 8)
    lib.check(lib.ga_gen_bases(ctx.handle, 0, group, seed, n, b.ptr, None))
    t = ecc.PrecomputedBases(ctx, 0, group, b, n=n)
    b.free()
    return t

def scalars(ctx, seed):
    s = ctx.malloc(n * 32)
    lib.check(lib.ga_gen_scalars(ctx.handle, 0, seed, n, s.ptr))
    return s

tA1, tA2, tG2 = table(cA, 0, 1), table(cA, 0, 2), table(cA, 1, 3)
tB1, tB2 = table(cB, 0, 4), table(cB, 0, 5)
sA, sB = scalars(cA, 7), scalars(cB, 8)
dom = fft.NewDomain(cB, 0, n)
vB = scalars(cB, 9)

def laneA():
    tA1.MultiExp(sA); tA2.MultiExp(sA); tG2.MultiExp(sA)

def laneB():
    for k in range(7):
        dom.FFT(vB, fft.DIF)
    tB1.MultiExp(sB); tB2.MultiExp(sB)

laneA(); laneB()
K = 4
t0 = time.time()
for _ in range(K):
    laneA(); laneB()
seq = (time.time() - t0) / K
t0 = time.time()
for _ in range(K):
    th = [threading.Thread(target=f) for f in (laneA, laneB)]
    [t.start() for t in th]; [t.join() for t in th]
par = (time.time() - t0) / K
t0 = time.time(); [laneA() for _ in range(K)]; a = (time.time() - t0) / K
t0 = time.time(); [laneB() for _ in range(K)]; b = (time.time() - t0) / K
print("lane A alone %.1f ms, lane B alone %.1f ms; back to back %.1f ms, two streams %.1f ms (%.1f %%)" % (a * 1e3, b * 1e3, seq * 1e3, par * 1e3, 100 * (seq - par) / seq))
