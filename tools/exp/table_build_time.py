import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gnark_amd import _lib, ecc
from gnark_amd.device import Context, affine_words
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << logn
ctx = Context(0); lib = ctx.lib
out = []
for cid in (0, 1):
    for group in (0, 1):
        b = ctx.malloc(n * affine_words(cid, group) * 8)
        lib.check(lib.ga_gen_bases(ctx.handle, cid, group, 77, n, b.ptr, None)); ctx.sync()
        ctx.profile(True); ctx.profile_reset()
        t0 = time.perf_counter(); t = ecc.PrecomputedBases(ctx, cid, group, b, n=n); ctx.sync()
        el = time.perf_counter() - t0
        k = sum(r[1] for r in ctx.profile_read() if r[0] == 'msm_table_build'); ctx.profile(False)
        out.append("%s G%d call %.3f s kernel %.3f s" % (("bn254", "bls")[cid], group + 1, el, k / 1e3))
        t.free(); b.free()
print(os.environ.get("GA_LIB_PATH", "default")[-24:], " | ".join(out))
