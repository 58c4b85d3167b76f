#!/usr/bin/env python3
"""stage table of a proof on a key pinned as PLAIN vectors (precompute = -1): where the default path's 165 ms go"""
import json, sys, time
sys.path.insert(0, ".")
from gnark_amd import groth16, synth
from gnark_amd.device import Context
ctx = Context(0)
inst = synth.make_instance(ctx, "bn254", 24, 0x5EED0005, want_dlogs=False)
pk = inst.proving_key(ctx, precompute=-1)
for _ in range(2):
    groth16.Prove(pk, inst.solution, inst.nb_public, inst.r, inst.s)
t0 = time.perf_counter()
for _ in range(3):
    groth16.Prove(pk, inst.solution, inst.nb_public, inst.r, inst.s)
print("ms per proof", round((time.perf_counter() - t0) / 3 * 1e3, 2))
ctx.profile(True); ctx.profile_reset()
for _ in range(2):
    groth16.Prove(pk, inst.solution, inst.nb_public, inst.r, inst.s)
ctx.sync()
agg = {}
for name, ms in ctx.profile_read():
    a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += ms
print(json.dumps({k: {"n_per_proof": v[0] / 2, "ms_per_proof": round(v[1] / 2, 3)} for k, v in agg.items()}, indent=0))
