#!/usr/bin/env python3
"""What would accumulating a witness MSM in two halves cost?  (round 4, VERDICT item 7: hide the W upload of a single caller by
starting on the first half of W while the second half crosses PCIe.)  One 2^24 table MSM against two 2^23 table MSMs over the two
halves of the same bases -- each half a complete MSM with its own sort, task lists, bucket pass, merge and window reduction, i.e.
an UPPER bound of the cost of a chained two-pass accumulation (which would share the reduction but pay one merge per bucket).
Prints one JSON line per group: ms for the whole MSM, for the two halves, and the stage split."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import gnark_amd
    import gnark_amd.device
    from gnark_amd import _lib, ecc
    ctx = gnark_amd.Context(0)
    lib = ctx.lib
    cid, log_n = 0, 24
    n = 1 << log_n
    for group, gname in ((_lib.G1, "g1"), (_lib.G2, "g2")):
        words = gnark_amd.device.affine_words(cid, group)
        bases = ctx.malloc(n * words * 8)
        scal = ctx.malloc(n * 32)
        lib.check(lib.ga_gen_bases(ctx.handle, cid, group, 0x5EED0002, n, bases.ptr, None))
        lib.check(lib.ga_gen_scalars(ctx.handle, cid, 0x5EED0001, n, scal.ptr))
        whole = ecc.PrecomputedBases(ctx, cid, group, bases, n=n)
        half = [ecc.PrecomputedBases(ctx, cid, group, bases.offset(k * (n // 2) * words * 8), n=n // 2) for k in range(2)]
        sc_half = [scal.offset(k * (n // 2) * 32) for k in range(2)]

        def timed(fn, reps=5):
            fn()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            ctx.sync()
            ms = (time.perf_counter() - t0) * 1e3 / reps
            ctx.profile(True)
            ctx.profile_reset()
            fn()
            ctx.sync()
            st = {}
            for k, v in ctx.profile_read():
                st[k] = round(st.get(k, 0.0) + v, 3)
            ctx.profile(False)
            return round(ms, 3), st

        one, st_one = timed(lambda: whole.MultiExp(scal))
        two, st_two = timed(lambda: [half[k].MultiExp(sc_half[k]) for k in range(2)])
        print(json.dumps({"group": gname, "log_n": log_n, "one_msm_ms": one, "two_half_msms_ms": two, "split_cost_ms": round(two - one, 3),
                          "stages_one": st_one, "stages_two_halves": st_two}))
        sys.stdout.flush()
        whole.free()
        for h in half:
            h.free()
        bases.free()
        scal.free()


if __name__ == "__main__":
    main()
