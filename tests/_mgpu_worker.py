"""worker for tests/test_multigpu_gloo.py: world_size-2 `gloo` run of both MSM partitionings on the emulation build."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import oracle  # noqa: E402
import pyref  # noqa: E402
from gnark_amd import _lib, multigpu  # noqa: E402
from gnark_amd.device import Context  # noqa: E402
from helpers import BN254, fr_to_arr, jac_to_affine_py  # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lib = _lib.Library(os.path.join(ROOT, "tests", "emu", "libgnark_amd_emu.so"))
    ctx = Context(0, lib=lib)
    c, group, n = BN254, 0, 301          # odd size: ragged shards
    rng = pyref.Xoshiro(77)
    ks = np.array([rng.next() for _ in range(n)], dtype=np.uint64)
    P = oracle.gen_bases(c.cid, group, ks)
    S = fr_to_arr(c, [rng.field(c.r) for _ in range(n)])
    want = jac_to_affine_py(c, group, oracle.msm(c.cid, group, P, S))
    lo, hi = multigpu.shard_range(n, rank, world)
    got_b = multigpu.msm_base_sharded(ctx, c.name, group, P[lo:hi], S[lo:hi], hi - lo, dist)
    got_a = multigpu.msm_window_sharded(ctx, c.name, group, P, S, n, dist)
    assert jac_to_affine_py(c, group, got_b) == want, "base-range sharding mismatch"
    assert jac_to_affine_py(c, group, got_a) == want, "window sharding mismatch"
    from gnark_amd import ecc
    table = ecc.PrecomputedBases(ctx, c.name, group, P)          # whole vector pinned on every rank: windows on the table path
    got_t = multigpu.msm_window_sharded(ctx, c.name, group, table, S, n, dist)
    table.free()
    assert jac_to_affine_py(c, group, got_t) == want, "window sharding on pinned tables mismatch"
    # ---- Groth16 with the key sharded by base-point range: same proof bytes as the oracle -----------------
    from gnark_amd import groth16
    from helpers import pts_to_arr
    cs, wv = pyref.cubic_r1cs(), pyref.cubic_witness(3)
    rng2 = pyref.Xoshiro(2024)
    pk, _, _ = pyref.groth16_setup(c, cs, [rng2.field(c.r) for _ in range(5)])
    r, s = rng2.field(c.r), rng2.field(c.r)
    A, B, Cc = pyref.r1cs_solve(c, cs, wv)
    for precompute in (1, -1):
        dpk = groth16.ProvingKey(
            ctx, c.name, domain_cardinality=pk.n, alpha1=pts_to_arr(c, 0, [pk.alpha1]), beta1=pts_to_arr(c, 0, [pk.beta1]),
            delta1=pts_to_arr(c, 0, [pk.delta1]), A=pts_to_arr(c, 0, pk.A), B=pts_to_arr(c, 0, pk.B), Z=pts_to_arr(c, 0, pk.Z),
            K=pts_to_arr(c, 0, pk.K), beta2=pts_to_arr(c, 1, [pk.beta2]), delta2=pts_to_arr(c, 1, [pk.delta2]),
            B2=pts_to_arr(c, 1, pk.B2), infinityA=pk.infinityA, infinityB=pk.infinityB, precompute=precompute, shard=(rank, world))
        sol = groth16.Solution(fr_to_arr(c, wv), fr_to_arr(c, A), fr_to_arr(c, B), fr_to_arr(c, Cc))
        proof = multigpu.groth16_prove_sharded(dpk, sol, cs.nb_public, fr_to_arr(c, [r]), fr_to_arr(c, [s]), dist)
        proof_rep = multigpu.groth16_prove_sharded(dpk, sol, cs.nb_public, fr_to_arr(c, [r]), fr_to_arr(c, [s]), dist, replicate_h=True)
        dpk.FreeGPUResources()
        assert proof.WriteTo() == pyref.proof_bytes(c, *pyref.groth16_prove(pk, cs, wv, r, s)), "sharded Groth16 proof differs"
        assert proof_rep.WriteTo() == proof.WriteTo(), "replicated-h scheme differs"
    # ---- a larger synthetic instance (2^8 constraints, ragged): every rank holds 1/world of the key ----------------------
    from gnark_amd import synth
    import checkers  # noqa: F401  (oracle/ on sys.path)
    inst = synth.make_instance(ctx, c.name, 8, 0xD157, nb_constraints=250)
    spk = inst.proving_key(ctx, shard=(rank, world), staged_chunk=64)
    try:
        sproof = multigpu.groth16_prove_sharded(spk, inst.solution, inst.nb_public, inst.r, inst.s, dist)
    finally:
        spk.FreeGPUResources()
    wpk = inst.proving_key(ctx, window_shard=(rank, world), precompute=1)      # partition A: whole key per rank, windows shared out
    try:
        wproof = multigpu.groth16_prove_sharded(wpk, inst.solution, inst.nb_public, inst.r, inst.s, dist)
    finally:
        wpk.FreeGPUResources()
    assert np.array_equal(wproof.raw(), sproof.raw()), "window-sharded Groth16 differs from base-range sharded"
    want = oracle.groth16_prove(c.cid, dict(inst.key, n=inst.n), inst.solution.W, inst.solution.A, inst.solution.B, inst.solution.C,
                                inst.nb_public, inst.r, inst.s, nthreads=2)
    assert np.array_equal(sproof.Ar, want[0]) and np.array_equal(sproof.Bs, want[1]) and np.array_equal(sproof.Krs, want[2]), "sharded 2^8"
    # ---- every collective the module uses, with checkable contents (the RCCL self-test of bench.py, here over gloo) -----------
    st = multigpu.collective_selftest(dist)
    assert st["ok"] and st.get("send_recv") is not False, st
    # ---- a rank that fails INSIDE a sharded proof: the fixed collective schedule carries the error to every rank, nobody hangs,
    # and the process group is usable afterwards (the next proof is the right proof)
    spk = inst.proving_key(ctx, shard=(rank, world), staged_chunk=64)
    try:
        for point in ("witness", "h_side", "z"):
            os.environ["GA_MGPU_FAULT"] = "%d:%s" % (world - 1 if point != "h_side" else 0, point)
            try:
                multigpu.groth16_prove_sharded(spk, inst.solution, inst.nb_public, inst.r, inst.s, dist)
                raise AssertionError("the injected fault at %s went unnoticed on rank %d" % (point, rank))
            except multigpu.ShardedProofError as e:
                assert str(world - 1 if point != "h_side" else 0) in str(e)
            finally:
                del os.environ["GA_MGPU_FAULT"]
        again = multigpu.groth16_prove_sharded(spk, inst.solution, inst.nb_public, inst.r, inst.s, dist)
        assert np.array_equal(again.raw(), sproof.raw()), "proof after a failed proof differs"
        ok, why = multigpu.agree(dist, rank != 0, "rank 0 says no")
        assert ok is False and "rank 0: rank 0 says no" in why
        ok, why = multigpu.agree(dist, True)
        assert ok is True and why is None
    finally:
        spk.FreeGPUResources()
    dist.barrier()
    if rank == 0:
        print("MGPU_OK world=%d" % world)
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
