"""Synthetic Groth16 instances with KNOWN DISCRETE LOGS (SURVEY 8d config 3/4, 8c "known discrete log"): test and bench
support, not part of the prover.

No Go toolchain exists here, so there is no compiled circuit at 2^24 constraints; the *solution* is synthesised directly
(A, B uniform, C = A o B, W uniform with W[0] = 1) and the proving key is made of bases [k_i]G generated on the device
(`ga_gen_bases`) whose exponents k_i are kept.  With the exponents known, the three proof points have closed forms
(prove.go:185-292 in the exponent):

    ar  = alpha + <W|_A, a>  + r*delta
    bs  = beta  + <W|_B, b>  + s*delta          (once in G1 with G1.B's exponents, once in G2 with G2.B's)
    krs = <W[nbPublic:], k> + <h[:n-1], z> + s*ar + r*bs1 - r*s*delta

so a 2^24-constraint proof is checked with five O(n) field dot products plus one polynomial identity for h -- exactly
what SURVEY 8d prescribes ("Expected outputs via dlog dot products + oracle NTT").  The dot products and the generator
multiplication are passed in by the caller (the CPU oracle in tests and in bench.py's checker leg); this module never
imports the oracle.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import _lib, groth16
from .device import FP_LIMBS, Context, affine_words, curve_id

FR_MODULUS = {
    _lib.BN254: 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001,
    _lib.BLS12_381: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
}


def limbs_to_int(a) -> int:
    return sum(int(v) << (64 * i) for i, v in enumerate(np.asarray(a, dtype=np.uint64).reshape(-1)))


def from_mont(curve: int, a) -> int:
    """fr.Element image (Montgomery, R = 2^256) -> canonical integer"""
    r = FR_MODULUS[curve]
    return limbs_to_int(a) * pow(1 << 256, -1, r) % r


@dataclass
class Instance:
    curve: int
    n: int
    nb_wires: int
    nb_public: int
    key: dict                      # host arrays: alpha1 beta1 delta1 A B Z K beta2 delta2 B2 infinityA infinityB
    dlogs: dict                    # canonical exponents (len, 4) uint64 for A B Z K B2, python ints for alpha1 beta1 delta1 beta2 delta2
    solution: groth16.Solution
    r: np.ndarray
    s: np.ndarray
    extra: dict = field(default_factory=dict)

    def proving_key(self, ctx: Context, **kw) -> groth16.ProvingKey:
        return groth16.ProvingKey(ctx, self.curve, domain_cardinality=self.n, **self.key, **kw)

    def prove_oneshot(self, ctx: Context, solution=None, r=None, s=None) -> groth16.Proof:
        """the key uploaded while the proof runs, then dropped (ga_g16_prove_oneshot)"""
        return groth16.ProveOneShot(ctx, self.curve, solution or self.solution, self.nb_public, self.r if r is None else r,
                                    self.s if s is None else s, domain_cardinality=self.n, **self.key)


def _gen_bases(ctx: Context, cid: int, group: int, count: int, seed: int, want_dlogs: bool):
    words = affine_words(cid, group)
    buf = ctx.malloc(max(count, 1) * words * 8)
    dl = ctx.malloc(max(count, 1) * 32) if want_dlogs else None
    ctx.lib.check(ctx.lib.ga_gen_bases(ctx.handle, cid, group, seed, count, buf.ptr, dl.ptr if dl else None))
    host = buf.to_host((count, words))
    buf.free()
    k = None
    if dl:
        k = dl.to_host((count, 4))
        dl.free()
    return host, k


def _gen_scalars(ctx: Context, cid: int, count: int, seed: int) -> np.ndarray:
    buf = ctx.malloc(max(count, 1) * 32)
    ctx.lib.check(ctx.lib.ga_gen_scalars(ctx.handle, cid, seed, count, buf.ptr))
    host = buf.to_host((count, 4))
    buf.free()
    return host


def _vector_plan(nw: int, n: int, nb_public: int, infA, infB):
    """(name, group, length, seed offset) of the five base vectors of the synthetic key"""
    return (("A", 0, nw - int(infA.sum()), 1), ("B", 0, nw - int(infB.sum()), 2), ("Z", 0, n - 1, 3), ("K", 0, nw - nb_public, 4),
            ("B2", 1, nw - int(infB.sum()), 5))


def make_instance(ctx: Context, curve, logn: int, seed: int = 0x5EED0005, *, nb_constraints: int | None = None, nb_public: int = 2,
                  inf_a=None, inf_b=None, want_dlogs: bool = True, product_c: bool = True, with_key: bool = True) -> Instance:
    """2^logn-constraint instance in the shape of SURVEY 8d config 3: nbWires = n, two infinity entries in A and in B like the
    squaring-chain circuit of backend/groth16/groth16_test.go:120-132, len(K) = nbWires - nbPublic, len(Z) = n - 1.
    with_key=False leaves the five base vectors out (12 GiB of host memory at 2^24): pin_key_chunked generates and pins them
    chunk by chunk, and only the slice a shard keeps."""
    cid = curve_id(curve)
    n = 1 << logn
    nw = n
    m = n if nb_constraints is None else int(nb_constraints)
    inf_a = np.asarray([1, nw - 1] if inf_a is None else inf_a, dtype=np.int64)
    inf_b = np.asarray([0, nw - 2] if inf_b is None else inf_b, dtype=np.int64)
    infA = np.zeros(nw, dtype=np.uint8)
    infB = np.zeros(nw, dtype=np.uint8)
    infA[inf_a] = 1
    infB[inf_b] = 1
    key, dl = {}, {}
    for name, group, count, sd in _vector_plan(nw, n, nb_public, infA, infB):
        if with_key:
            key[name], dl[name] = _gen_bases(ctx, cid, group, count, seed + sd, want_dlogs)
    m1, k1 = _gen_bases(ctx, cid, 0, 3, seed + 6, want_dlogs)
    m2, k2 = _gen_bases(ctx, cid, 1, 2, seed + 7, want_dlogs)
    key.update(alpha1=m1[0:1], beta1=m1[1:2], delta1=m1[2:3], beta2=m2[0:1], delta2=m2[1:2], infinityA=infA, infinityB=infB)
    if want_dlogs:
        dl.update(alpha1=limbs_to_int(k1[0]), beta1=limbs_to_int(k1[1]), delta1=limbs_to_int(k1[2]),
                  beta2=limbs_to_int(k2[0]), delta2=limbs_to_int(k2[1]))
    W = _gen_scalars(ctx, cid, nw, seed + 10)
    one = np.array([(((1 << 256) % FR_MODULUS[cid]) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    W[0] = one                                       # the constant-one wire (solver.go:82-88)
    a = _gen_scalars(ctx, cid, m, seed + 11)
    b = _gen_scalars(ctx, cid, m, seed + 12)
    if product_c:                                    # C = A o B: the instance is satisfiable, h is a polynomial of degree <= n-2
        c = np.zeros_like(a)
        ctx.lib.check(ctx.lib.ga_fr_vec_mul(ctx.handle, cid, a.ctypes.data, b.ctypes.data, m, c.ctypes.data, 0))
    else:
        c = _gen_scalars(ctx, cid, m, seed + 13)
    rs = _gen_scalars(ctx, cid, 2, seed + 14)
    return Instance(cid, n, nw, nb_public, key, dl, groth16.Solution(W, a, b, c), rs[0].copy(), rs[1].copy(), extra={"seed": seed})


def pin_key_chunked(ctx: Context, inst: Instance, *, shard=(0, 1), window_shard=(0, 1), precompute: int = 1, chunk: int = 1 << 19) -> groth16.ProvingKey:
    """The proving key of make_instance(seed) -- the same bases, point for point -- pinned through the staged builder WITHOUT ever
    holding a vector on the host: every vector is generated on the device in chunks of `chunk` points (ga_gen_bases_at), pulled,
    appended and dropped; a base-range shard generates only the slice it keeps and skips the rest (ga_g16_builder_append with a
    null pointer).  What a rank of a multi-GPU bench run uses: host staging per rank is one chunk, not the 12 GiB key."""
    import ctypes as C
    lib, cid, n, nw = ctx.lib, inst.curve, inst.n, inst.nb_wires
    seed = inst.extra["seed"]
    infA, infB = inst.key["infinityA"], inst.key["infinityB"]
    k, N = int(shard[0]), int(shard[1])
    b = C.c_void_p()
    lib.check(lib.ga_g16_builder_create(ctx.handle, cid, n, nw, k, N, C.byref(b)))
    try:
        for which, (name, group, total, sd) in enumerate(_vector_plan(nw, n, inst.nb_public, infA, infB)):
            lib.check(lib.ga_g16_builder_reserve(b, which, total))
            base, rem = divmod(total, N)                      # the split of ga_g16_builder_reserve (= multigpu.shard_range)
            lo = k * base + min(k, rem)
            cnt = base + (1 if k < rem else 0)
            words = affine_words(cid, group)
            if lo:
                lib.check(lib.ga_g16_builder_append(b, which, None, lo))
            buf = ctx.malloc(max(1, min(chunk, cnt)) * words * 8)
            for c0 in range(lo, lo + cnt, chunk):
                cn = min(chunk, lo + cnt - c0)
                lib.check(lib.ga_gen_bases_at(ctx.handle, cid, group, seed + sd, c0, cn, buf.ptr, None))
                part = buf.to_host((cn, words))
                lib.check(lib.ga_g16_builder_append(b, which, part.ctypes.data, cn))
            buf.free()
            if total - lo - cnt:
                lib.check(lib.ga_g16_builder_append(b, which, None, total - lo - cnt))
        for which, name in enumerate(("alpha1", "beta1", "delta1", "beta2", "delta2")):
            lib.check(lib.ga_g16_builder_set_point(b, which, inst.key[name].ctypes.data))
        lib.check(lib.ga_g16_builder_set_infinity(b, 0, infA.ctypes.data, nw))
        lib.check(lib.ga_g16_builder_set_infinity(b, 1, infB.ctypes.data, nw))
        if int(window_shard[1]) > 1:
            lib.check(lib.ga_g16_builder_set_window_shard(b, int(window_shard[0]), int(window_shard[1])))
    except Exception:
        lib.ga_g16_builder_destroy(b)
        raise
    h = C.c_void_p()
    lib.check(lib.ga_g16_builder_finish(b, int(precompute), C.byref(h)))
    return groth16.ProvingKey.from_handle(ctx, cid, h, nb_wires=nw, domain_cardinality=n, shard=(k, N))


def attach_vector_dlogs(ctx: Context, inst: Instance) -> None:
    """The exponents of the five base vectors of an instance made with with_key=False, want_dlogs=True (the vectors themselves are
    what pin_key_chunked generates shard by shard, same seeds): ga_gen_bases once per vector for its dlog output; the points are
    dropped on the device.  Lets rank 0 of a multi-GPU run check the sharded proof against the closed form (bench.py)."""
    seed = inst.extra["seed"]
    for name, group, count, sd in _vector_plan(inst.nb_wires, inst.n, inst.nb_public, inst.key["infinityA"], inst.key["infinityB"]):
        _, inst.dlogs[name] = _dlogs_only(ctx, inst.curve, group, count, seed + sd)


def _dlogs_only(ctx: Context, cid: int, group: int, count: int, seed: int):
    words = affine_words(cid, group)
    buf = ctx.malloc(max(count, 1) * words * 8)
    dl = ctx.malloc(max(count, 1) * 32)
    try:
        ctx.lib.check(ctx.lib.ga_gen_bases(ctx.handle, cid, group, seed, count, buf.ptr, dl.ptr))
        return None, dl.to_host((count, 4))
    finally:
        buf.free()
        dl.free()


def expected_exponents(inst: Instance, h_bitrev: np.ndarray, dot) -> dict:
    """Discrete logs of Ar, Bs (G2), Krs and of the pre-randomisation sums, from the instance's exponents.
    dot(a_mont, b_canonical) -> int must return sum a_i * b_i mod r with a in Montgomery form and b canonical (the oracle's
    fr_dot, or ga_fr_dot).  h_bitrev: computeH's output (n elements, bit-reversed order, Montgomery)."""
    q = FR_MODULUS[inst.curve]
    W = inst.solution.W
    keepA = inst.key["infinityA"] == 0
    keepB = inst.key["infinityB"] == 0
    sA = dot(np.ascontiguousarray(W[keepA]), inst.dlogs["A"])
    WB = np.ascontiguousarray(W[keepB])
    sB1 = dot(WB, inst.dlogs["B"])
    sB2 = dot(WB, inst.dlogs["B2"])
    sK = dot(np.ascontiguousarray(W[inst.nb_public:]), inst.dlogs["K"])
    sZ = dot(np.ascontiguousarray(h_bitrev[: inst.n - 1]), inst.dlogs["Z"])
    r, s = from_mont(inst.curve, inst.r), from_mont(inst.curve, inst.s)
    d = inst.dlogs
    ar = (d["alpha1"] + sA + r * d["delta1"]) % q
    bs1 = (d["beta1"] + sB1 + s * d["delta1"]) % q
    bs2 = (d["beta2"] + sB2 + s * d["delta2"]) % q
    krs = (sK + sZ + s * ar + r * bs1 - r * s * d["delta1"]) % q
    return {"Ar": ar, "Bs": bs2, "Krs": krs, "partial_A": sA, "partial_B1": sB1, "partial_B2": sB2, "partial_KZ": (sK + sZ) % q}
