#!/usr/bin/env python3
"""pin (plain vectors) / first prove / second prove / free, timed separately, three rounds (the Go package's default path)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnark_amd import groth16, synth
from gnark_amd.device import Context
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ctx = Context(0)
inst = synth.make_instance(ctx, "bn254", logn, 0x5EED0005, want_dlogs=False)
sol, nbp, r, s = inst.solution, inst.nb_public, inst.r, inst.s
for k in range(3):
    t0 = time.perf_counter()
    pk = inst.proving_key(ctx, precompute=-1)
    ctx.sync()
    t1 = time.perf_counter()
    groth16.Prove(pk, sol, nbp, r, s)
    t2 = time.perf_counter()
    groth16.Prove(pk, sol, nbp, r, s)
    t3 = time.perf_counter()
    pk.FreeGPUResources()
    ctx.sync()
    t4 = time.perf_counter()
    print(json.dumps({"round": k, "pin_ms": round((t1 - t0) * 1e3, 1), "first_prove_ms": round((t2 - t1) * 1e3, 1), "second_prove_ms": round((t3 - t2) * 1e3, 1),
                      "free_ms": round((t4 - t3) * 1e3, 1)}), flush=True)
