#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X prover backend (driver contract: one JSON line from rank 0).

Workload (BASELINE.json `metric`: "G1 MSM Mscalar-mul/s + Groth16 proofs/s, BN254 2^24 constraints"):
  * a "step" = one BN254 G1 Pippenger MSM over 2^24 (scalar, base) pairs per GPU, bases and scalars resident in
    HBM when the timed region starts (SURVEY 8d config 2/3 shapes: uniform Montgomery scalars, distinct known-dlog
    bases generated on device).  `value` = scalar-muls/s summed over all ranks, in Mscalar-mul/s.
  * with N > 1 ranks the MSM is sharded by base-point range (SURVEY 8e partitioning B, weak scaling: every rank
    owns 2^24 pairs); the exchange step is an RCCL all_gather of one Jacobian partial per rank + a local add.
  * the Groth16 leg (proofs/s at the same size: computeH + 4 G1 MSMs + 1 G2 MSM + host epilogue, key pinned with its window
    tables, solver excluded, C = A o B) is timed separately on rank 0 and reported in the "groth16" object of the same line:
    5 single-caller proofs (every one checked against the closed form from the key's known discrete logs + the polynomial
    identity of h, outside the timed region: "matches_dlog", "check"), then 10 proofs from two host threads on one key (two
    proofs in flight on the context's two lanes: "pipelined"; `--no-pipelined` leaves that leg out of profiler passes).
    With N > 1 ranks the "groth16" object is ONE proof sharded over the N GPUs (strong scaling; gnark_amd/multigpu.py): every rank
    generates only its shard of the key, rank 0 checks the sharded proof against the same closed form afterwards ("matches_dlog");
    its "proof_sha" equals the N = 1 line's.  With 8 ranks (or GA_BENCH_CONFIG4=1) "groth16_bls12_381" is BASELINE config 4.
  * the PLONK leg (BASELINE config 5: kernel work of one BN254 proof at 2^22 gates -- 10 KZG-commit MSMs over a pinned SRS,
    grand product, quotient) is reported in the "plonk" object (N = 1, BN254; `--plonk-log-n 0` disables it).
  * "roofline": dominant kernel (msm_accumulate) vs the 8 TB/s HBM peak using the ALGORITHMIC 96 B per scalar-mul
    (32 B scalar + 64 B affine base); durations come from hipEvents recorded by the library on its own
    stream; "roofline.integer_multiplier" prices the same launches against the measured v_mad_u64_u32 issue rate (the
    binding resource).  "cpu_baseline": the C oracle's Pippenger (oracle/oracle.c, kind "port", one thread per window;
    "threads_used" / "host_cores") on 2^24 points and the oracle's Groth16 prover on a 2^20 sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-n", type=int, default=int(os.environ.get("GA_BENCH_LOGN", "24")))
    ap.add_argument("--groth16-proofs", type=int, default=int(os.environ.get("GA_BENCH_PROOFS", "5")))
    ap.add_argument("--no-check", action="store_true", help="skip the oracle checks of the Groth16 / PLONK legs (outside the timed regions)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the two-caller Groth16 leg (rocprofv3 passes: concurrent proofs stretch per-kernel durations)")
    ap.add_argument("--plonk-log-n", type=int, default=int(os.environ.get("GA_BENCH_PLONK_LOGN", "22")), help="0 disables the PLONK leg")
    ap.add_argument("--curve", default="bn254")
    ap.add_argument("--partition", default=os.environ.get("GA_BENCH_PARTITION", "range"), choices=["range", "window"],
                    help="N > 1 Groth16 leg: key sharded by base-point range (partition B) or by scalar windows (partition A, config 4's wording)")
    return ap.parse_args()


def effective_cores():
    """CPUs this process can really use: the affinity mask capped by the cgroup CPU quota (the GPU boxes show 256 logical CPUs and
    a quota of 16: cpu.max = "1600000 100000")."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    eff = n if quota is None else max(1, min(n, int(quota + 0.5)))
    return eff, n, quota


def stage_stats(records):
    agg = {}
    for name, ms in records:
        a = agg.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += ms
    return {k: {"launches": v[0], "total_ms": round(v[1], 4), "avg_ms": round(v[1] / v[0], 4)} for k, v in agg.items()}


def with_retry(make, what, tries=5, pause=4.0):
    """HBM of a process that has just exited on this GPU (the previous test / bench run) can take a moment to be returned;
    a big allocation that fails is retried a few times before the bench gives up (loudly)."""
    for k in range(tries):
        try:
            return make()
        except Exception as e:
            if k + 1 == tries or not any(t in str(e) for t in ("hipMalloc", "memory", "NOMEM")):
                raise
            sys.stderr.write("bench: %s failed (%s); retrying in %.0f s\n" % (what, str(e)[:120], pause))
            time.sleep(pause)


def synth_groth16(ctx, cid, logn, seed, shard=(0, 1), want_dlogs=True):
    """Synthetic 2^logn-constraint instance: known-dlog key generated on device, pulled to the host once so
    that it goes through the same ga_g16_pk_create upload path a Go caller uses; C = A o B (gnark_amd/synth.py).
    The key is pinned WITH its window tables (precompute = 1, not "auto": auto falls back to plain bases when the tables do not
    fit the free HBM at that moment, and the bench must not silently time a different configuration)."""
    from gnark_amd import synth
    inst = synth.make_instance(ctx, cid, logn, seed, want_dlogs=want_dlogs)
    pre = int(os.environ.get("GA_BENCH_PRECOMPUTE", "1"))
    pk = with_retry(lambda: inst.proving_key(ctx, shard=shard, precompute=pre), "pinning the proving key")
    return inst, pk


def check_groth16(ctx, inst, proof, threads):
    """Outside every timed region: is the timed proof THE proof?  Exponents of Ar, Bs, Krs from the key's known discrete logs by
    O(n) dot products on the CPU oracle, h by the polynomial identity (oracle/checkers.py).  The oracle is the checker here,
    never the thing measured."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import checkers
    import oracle
    import pyref
    from gnark_amd import fft, synth
    c = pyref.BN254 if inst.curve == 0 else pyref.BLS12_381
    sol = inst.solution
    d = fft.Domain(ctx, inst.curve, inst.n)
    try:
        h = d.compute_h(sol.A, sol.B, sol.C)
    finally:
        d.close()
    res = {"method": "known-dlog key: exponents by oracle.fr_dot (CPU), h by A(x)B(x)-C(x)=H(x)(x^n-1) with barycentric A,B,C (oracle/checkers.py)"}
    try:
        checkers.check_compute_h_identity(c, sol.A, sol.B, sol.C, h, inst.n, threads)
        res["h_identity_ok"] = True
    except AssertionError:
        res["h_identity_ok"] = False
    exp = synth.expected_exponents(inst, h, lambda x, y: oracle.fr_dot(inst.curve, x, y))
    pt = lambda group, k: oracle.jac_to_affine(inst.curve, group, oracle.generator_mul(inst.curve, group, k))
    res["matches_dlog"] = bool(np.array_equal(proof.Ar, pt(0, exp["Ar"])) and np.array_equal(proof.Bs, pt(1, exp["Bs"]))
                               and np.array_equal(proof.Krs, pt(0, exp["Krs"])))
    return res


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    # GA_BENCH_EMU=1: dry run of this script's control flow against the CPU emulation build of the library (tests/emu), tiny
    # sizes only -- a development aid for the GPU-less build container, never a measurement (the JSON line says "data": "emulation")
    emu = os.environ.get("GA_BENCH_EMU", "0") == "1"
    ndev = max(1, torch.cuda.device_count())
    device_index = local_rank % ndev       # one rank per GPU on the driver's runs; wraps only in single-GPU smoke runs
    if not emu:
        torch.cuda.set_device(device_index)
    backend = os.environ.get("GA_BENCH_BACKEND", "gloo" if emu else "nccl")   # "gloo" lets a 1-GPU box exercise the N>1 code path
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend)

    import gnark_amd
    from gnark_amd import _lib, ecc
    from gnark_amd.device import curve_id, jac_words
    cid = curve_id(args.curve)
    if emu:
        ctx = gnark_amd.Context(0, lib=_lib.Library(os.path.join(ROOT, "tests", "emu", "libgnark_amd_emu.so")))
    else:
        ctx = gnark_amd.Context(device_index)   # raises when libgnark_amd.so or the GPU is missing: there is no CPU fallback
    lib = ctx.lib
    n = 1 << args.log_n
    words_aff = gnark_amd.device.affine_words(cid, _lib.G1)

    # ---- inputs resident in HBM ---------------------------------------------------------------------------
    bases = ctx.malloc(n * words_aff * 8)
    scalars = ctx.malloc(n * 32)
    base_dlogs = ctx.malloc(n * 32) if not args.no_check else None   # k_i of [k_i]G: the timed result is checked against them
    lib.check(lib.ga_gen_bases(ctx.handle, cid, _lib.G1, 0x5EED0002 + 977 * rank, n, bases.ptr, base_dlogs.ptr if base_dlogs else None))
    lib.check(lib.ga_gen_scalars(ctx.handle, cid, 0x5EED0001 + 977 * rank, n, scalars.ptr))
    # the bases are a pinned key: keep them with their window multiples (ga_msm_table_*, built outside the timed region)
    use_table = os.environ.get("GA_BENCH_TABLE", "1") != "0"
    table = with_retry(lambda: ecc.PrecomputedBases(ctx, cid, _lib.G1, bases, n=n), "building the MSM window table") if use_table else None
    if use_table:
        ti = table.info()
        cbits, nwin = ti["window_bits"], ti["windows"]
    else:
        cbits, nwin = ecc.plan(cid, _lib.G1, n)

    from gnark_amd import multigpu
    dev = torch.device("cuda", device_index) if not emu else None   # with gloo on a GPU box device buffers are staged through the host

    def step():
        # N = 1: plain MSM.  N > 1: partition B (base-point range) -- every rank reduces its own 2^log_n pairs to one
        # Jacobian partial, RCCL all_gather of the partials, local add (gnark_amd/multigpu.py)
        return multigpu.msm_base_sharded(ctx, cid, _lib.G1, table if use_table else bases, scalars, n, dist, dev)

    def fence():
        if not emu:
            torch.cuda.synchronize()
        ctx.sync()
        if world > 1:
            dist.barrier()
        if not emu:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.profile(True)
    ctx.profile_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = step()
    fence()
    elapsed = time.perf_counter() - t0
    stages = stage_stats(ctx.profile_read())
    ctx.profile(False)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    ms_per_step = elapsed * 1e3 / args.steps
    value = world * n * args.steps / elapsed / 1e6
    # is the timed result THE result?  MSM(s, [k_i]G) = [sum s_i k_i]G: the exponent by a dot product on the CPU oracle, the point by
    # the oracle's fixed-base multiplication -- outside the timed region (the oracle is the checker, never the thing timed)
    value_checked = None
    if base_dlogs is not None:
        e, err = None, None
        try:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle
            e = oracle.fr_dot(cid, scalars.to_host((n, 4)), base_dlogs.to_host((n, 4)))
        except Exception as ex:   # a failing checker is reported, it does not hide the measurement
            err = "checker error: " + repr(ex)[:200]
        if world > 1:   # the gathered sum is [sum over ranks of <s, k>]G: every rank contributes its exponent (8 x 32 bits + an ok flag);
            # every rank takes part in the collective whatever happened to its checker
            mine = torch.tensor([((e or 0) >> (32 * i)) & 0xFFFFFFFF for i in range(8)] + [0 if e is None else 1], dtype=torch.int64,
                                device="cuda" if backend == "nccl" else "cpu")
            parts = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            parts = [p.tolist() for p in parts]
            if all(p[8] == 1 for p in parts):
                from gnark_amd import synth
                e = sum(sum(int(v) << (32 * i) for i, v in enumerate(p[:8])) for p in parts) % synth.FR_MODULUS[cid]
            else:
                e, err = None, err or "checker error on another rank"
        if e is not None:
            try:
                want_pt = oracle.jac_to_affine(cid, 0, oracle.generator_mul(cid, 0, e))
                value_checked = bool(np.array_equal(ecc.jac_to_affine(cid, _lib.G1, result), want_pt))
            except Exception as ex:
                value_checked = "checker error: " + repr(ex)[:200]
        else:
            value_checked = err
        base_dlogs.free()

    out = None
    if rank == 0:
        acc = stages.get("msm_accumulate", {"avg_ms": float("nan")})
        alg_bytes = 96.0 * n if cid == 0 else 128.0 * n
        achieved = alg_bytes / (acc["avg_ms"] * 1e-3) / 1e9
        traffic, traffic_source = None, None   # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_latest.json), same shape only
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            ent = pj["kernels"]["msm_accumulate_kernel"].get(args.curve, {}).get(str(args.log_n))
            if ent and world == 1:
                traffic = ent["fetch_bytes"] + ent["write_bytes"]
                traffic_source = "profiles/pmc_latest.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (%s) of this kernel at this shape, not re-measured in this run" % pj.get("source", "round 2")
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "G1 MSM throughput, %s, 2^%d scalar-muls per GPU (Groth16 proofs/s at 2^%d constraints in 'groth16')" % (args.curve.upper(), args.log_n, args.log_n),
            "value": round(value, 3), "value_checked": value_checked, "unit": "Mscalar-mul/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 Montgomery limbs in memory; 29/28-bit limbs, v_mad_u64_u32 (32x32+64) in registers", "data": "emulation" if emu else "synthetic",
            "config": {"workload": "%s G1 Pippenger MSM, 2^%d uniform scalars x distinct known-dlog affine bases per GPU, inputs resident in HBM" % (args.curve.upper(), args.log_n),
                       "curve": args.curve, "window_bits": cbits, "windows": nwin,
                       "precompute": ("[2^(c*w)]P tables for all %d windows, %.1f GiB, one shared bucket set" % (nwin, ti["table_bytes"] / 2**30)) if use_table else "none",
                       "parallelism": "1 GPU" if world == 1 else "base-range sharding x%d, RCCL all_gather of Jacobian partials" % world},
            "roofline": {"bound": "hbm", "kernel": "msm_accumulate29_kernel" if use_table else "msm_accumulate_kernel", "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 6), "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": acc["avg_ms"],
                         "note": "MSM is integer-multiplier bound (SURVEY 8d): ~2.4e4 32-bit MADs per scalar-mul vs 96 B",
                         # "report achieved MAD/s fraction": v_mad_u64_u32 per mixed addition from the shipped ISA
                         # (tools/isa_count.py) x (windows x n) additions per launch, against the issue rate ga_microbench measures
                         "integer_multiplier": (lambda mads: {"v_mad_u64_u32_per_addition": mads, "additions_per_launch": nwin * n,
                                                              "achieved_Tmad_per_s": round(mads * nwin * n / (acc["avg_ms"] * 1e-3) / 1e12, 2),
                                                              "peak_Tmad_per_s": 30.0, "frac": round(mads * nwin * n / (acc["avg_ms"] * 1e-3) / 30e12, 3),
                                                              "peak_source": "ga_microbench v_mad_u64_u32 issue rate (profiles/r01_e_microbench.json)"})(1467 if cid == 0 else 3543)},
            "stages_ms": stages,
        }

    # ---- Groth16 leg (rank 0 only; N=1 semantics) ---------------------------------------------------------------
    if rank == 0 and world == 1 and use_table:
        # the same MSM without precomputed tables (ga_msm on the raw bases), for reference
        ecc.MultiExp(ctx, cid, _lib.G1, bases, scalars, n=n)
        t0 = time.perf_counter()
        for _ in range(2):
            ecc.MultiExp(ctx, cid, _lib.G1, bases, scalars, n=n)
        out["plain_msm_no_tables"] = {"ms_per_msm": round((time.perf_counter() - t0) * 500, 3),
                                      "Mscalar_mul_per_s": round(n / ((time.perf_counter() - t0) / 2) / 1e6, 2)}
        # SURVEY 8d metric (i), second figure: the headline MSM with the SCALARS in host memory (PCIe-inclusive; never `value`)
        s_host = scalars.to_host((n, 4))
        r_host = table.MultiExp(s_host)
        t0 = time.perf_counter()
        for _ in range(3):
            r_host = table.MultiExp(s_host)
        el_h = (time.perf_counter() - t0) / 3
        out["msm_with_scalar_h2d"] = {"ms_per_msm": round(el_h * 1e3, 3), "Mscalar_mul_per_s": round(n / el_h / 1e6, 2),
                                      # (the same group element: since the digits are fused with the first sort pass the order inside a bucket -- and with
                                      # it the Jacobian representative of the sum -- differs from run to run; the affine point does not)
                                      "same_result": bool(np.array_equal(ecc.jac_to_affine(cid, _lib.G1, r_host, lib=ctx.lib),
                                                                         ecc.jac_to_affine(cid, _lib.G1, result, lib=ctx.lib))),
                                      "how": "the timed table MSM with its 2^%d x 32 B of scalars uploaded from pageable host memory inside the call" % args.log_n}
        del s_host
    if table is not None:
        table.free()
        if world > 1:
            bases.free()
            scalars.free()
    if rank == 0 and world == 1 and args.groth16_proofs > 0:   # N = 1 semantics; multi-rank runs time the sharded MSM only
        bases.free()
        scalars.free()
        from gnark_amd import groth16
        t_setup = time.perf_counter()
        threads = os.cpu_count() or 1
        inst, pk = synth_groth16(ctx, cid, args.log_n, 0x5EED0005, want_dlogs=not args.no_check)
        sol, nb_public, r, s = inst.solution, inst.nb_public, inst.r, inst.s
        setup_s = time.perf_counter() - t_setup
        for _ in range(2):   # warm-up: scratch of both lanes of the pair (witness MSMs on lane 0, H side on lane 1)
            groth16.Prove(pk, sol, nb_public, r, s)
        ctx.sync()
        lanes0 = ctx.lane_stats()
        t0 = time.perf_counter()
        for _ in range(args.groth16_proofs):
            proof = groth16.Prove(pk, sol, nb_public, r, s)
        ctx.sync()
        el = time.perf_counter() - t0
        lanes1 = ctx.lane_stats()
        # stage breakdown: a separate pass with the library's stage profiler on.  The profiler serialises a proof on ONE lane (its
        # hipEvent pairs live on the main stream), so these are the kernels' stand-alone durations, not the overlapped schedule timed above
        prof_proofs = 2
        ctx.profile(True)
        ctx.profile_reset()
        ctx.sync()
        tq0 = time.perf_counter()
        for _ in range(prof_proofs):
            groth16.Prove(pk, sol, nb_public, r, s)
        ctx.sync()
        el_prof = time.perf_counter() - tq0
        gst = stage_stats(ctx.profile_read())
        ctx.profile(False)
        # the same proofs from TWO host threads (two goroutines in the Go shim): the second caller proves on the context's second
        # pair of lanes (own streams and scratch) concurrently with the first, so the 2 GiB uploads hide behind the other proof's
        # kernels and the kernels of the two proofs interleave on the device.  Both pairs are warmed first (the second
        # pair's ~10 GB of scratch is allocated on first use), and ga_g16_lane_stats says where the timed proofs ran.
        import threading
        per_thread = max(10, args.groth16_proofs)
        if args.no_pipelined:
            per_thread = 0

        def run_pair(count, sink):
            def prover(k):
                for _ in range(count):
                    sink[k].append(groth16.Prove(pk, sol, nb_public, r, s).raw())
            th = [threading.Thread(target=prover, args=(k,)) for k in range(2)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            ctx.sync()
        pipe_out = [[], []]
        if per_thread:
            run_pair(2, [[], []])   # warm-up of the second lane pair
        pl0 = ctx.lane_stats()
        tp0 = time.perf_counter()
        if per_thread:
            run_pair(per_thread, pipe_out)
        pipe_el = time.perf_counter() - tp0
        pl1 = ctx.lane_stats()
        pipe_same = bool(all(len(po) == per_thread and all(np.array_equal(q, proof.raw()) for q in po) for po in pipe_out))
        pipe_lanes = {k: pl1[k] - pl0[k] for k in ("lanes01_proofs", "lanes23_proofs", "queued_proofs", "split_proofs")}
        pipe_lanes.update({k: pl1[k] for k in ("lanes01_scratch_bytes", "lanes23_scratch_bytes")})
        pk.FreeGPUResources()
        ntt_ms = sum(v["total_ms"] for k, v in gst.items() if k.startswith("ntt_") or k == "h_pointwise") / prof_proofs
        bytes_per_constraint = 992 if cid == 0 else 1184   # 4 G1 + 1 G2 MSM + 7 NTTs
        out["groth16"] = {"proofs_per_s": round(args.groth16_proofs / el, 4), "ms_per_proof": round(el * 1e3 / args.groth16_proofs, 2),
                          "proofs": args.groth16_proofs, "constraints": n, "key_setup_s": round(setup_s, 1),
                          "schedule": {"split_proofs": lanes1["split_proofs"] - lanes0["split_proofs"],
                                       "how": "one caller: witness MSMs (A, B1, B2) on lane 0, uploads of A,B,C + computeH + Z MSM on lane 1 from a helper thread, K MSM on whichever lane is free first (GA_G16_SPLIT=0: everything on lane 0)"},
                          "ms_per_proof_profiled_single_lane": round(el_prof * 1e3 / prof_proofs, 2),
                          "definition": "W,A,B,C in host memory -> Ar,Bs,Krs affine on host; key pinned with window tables (precompute=%s); solver excluded; C = A o B (satisfiable instance)" % os.environ.get("GA_BENCH_PRECOMPUTE", "1"),
                          "algorithmic_bytes": bytes_per_constraint * n, "hbm_frac_whole_proof": round(bytes_per_constraint * n / (el / args.groth16_proofs) / 8e12, 6),
                          "computeH_ms": round(ntt_ms, 3),
                          "computeH_hbm_frac": round(448.0 * n / (ntt_ms * 1e-3) / 8e12, 5) if ntt_ms > 0 else None,
                          "proof_sha": __import__("hashlib").sha256(proof.WriteTo()).hexdigest()[:16],
                          "pipelined": None if per_thread == 0 else {"proofs_per_s": round(2 * per_thread / pipe_el, 4), "ms_per_proof": round(pipe_el * 1e3 / (2 * per_thread), 2),
                                        "proofs": 2 * per_thread, "host_threads": 2, "same_proof_bytes": pipe_same, "lanes": pipe_lanes,
                                        "vs_single_caller": round(pipe_el / (2 * per_thread) / (el / args.groth16_proofs), 4),
                                        "how": "two host threads call ga_g16_prove on one key, both lane pairs warmed first: two proofs in flight on lanes 0/1 and 2/3 of one context; every proof compared with the single-caller proof"},
                          "stages_ms": {k: {"launches": v["launches"], "total_ms": round(v["total_ms"] / prof_proofs, 4), "avg_ms": v["avg_ms"]}
                                        for k, v in gst.items()},
                          "stages_note": "total_ms is per proof, from %d extra proofs with the stage profiler on (single-lane schedule); ms_per_proof above is timed without it" % prof_proofs}
        if not args.no_check:
            t_chk = time.perf_counter()
            try:
                out["groth16"]["check"] = check_groth16(ctx, inst, proof, threads)
            except Exception as e:   # a failing checker must not hide the measurement -- it is reported instead
                out["groth16"]["check"] = {"error": repr(e)[:300], "matches_dlog": None}
            out["groth16"]["check"]["seconds"] = round(time.perf_counter() - t_chk, 1)
            out["groth16"]["matches_dlog"] = out["groth16"]["check"].get("matches_dlog")
        del inst, sol

    # ---- PLONK (BASELINE config 5): kernel work of one BN254 proof at 2^22 gates -- 10 KZG-commit MSMs over a pinned SRS, the
    # grand product and the quotient (computeNumerator + divideByZH) on the device; N = 1, BN254 only
    if rank == 0 and world == 1 and args.plonk_log_n > 0 and cid == 0:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_plonk_kernels
            out["plonk"] = bench_plonk_kernels.run(ctx, args.plonk_log_n, reps=2, reference_count=False)[0]
            if not args.no_check:   # the same device pipeline on a SATISFYING trace, checked by the CPU oracle (outside the timed region)
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import checkers
                import pyref
                t_chk = time.perf_counter()
                try:
                    out["plonk"]["identity_ok"] = bool(checkers.check_plonk_quotient_identity(ctx, pyref.BN254, args.plonk_log_n,
                                                                                              nthreads=os.cpu_count() or 1, pinned=True))
                except AssertionError:
                    out["plonk"]["identity_ok"] = False
                out["plonk"]["identity_check"] = ("h(zeta)(zeta^n-1) == gate + alpha*ordering + alpha^2(Z-1)L1 on a satisfying synthetic trace, "
                                                  "polynomials evaluated by the CPU oracle (oracle/checkers.py), %.1f s" % (time.perf_counter() - t_chk))
        except Exception as e:   # never lose the headline line over the secondary leg
            out["plonk"] = {"error": str(e)[:300]}

    # ---- Groth16 across ranks (N > 1): ONE 2^log_n proof over all GPUs -- strong scaling of BASELINE config 3 / 4 ------------------
    # key sharded by base-point range (or by windows), W uploaded per wire range, computeH's chains on ranks 0..2, h slices scattered
    # over xGMI, one all_gather of the partial sums (gnark_amd/multigpu.py)
    if world > 1 and args.groth16_proofs > 0 and os.environ.get("GA_BENCH_SHARDED_G16", "1") != "0":
        from gnark_amd import groth16, synth
        if table is None:
            bases.free()
            scalars.free()

        def sharded_leg(leg_cid, leg_curve):
            """one 2^log_n proof over the world's GPUs.  Every rank generates ITS shard of the synthetic key on its own device, chunk
            by chunk (synth.pin_key_chunked: no 12 GiB key is staged through host memory), and the solution (the prover's real input)."""
            try:
                check_here = rank == 0 and not args.no_check   # rank 0 checks the sharded proof against the key's known discrete logs
                inst = synth.make_instance(ctx, leg_cid, args.log_n, 0x5EED0005, want_dlogs=check_here, with_key=False)   # same seeds on every rank
                kw = dict(shard=(rank, world)) if args.partition == "range" else dict(window_shard=(rank, world))
                t_pin = time.perf_counter()
                pk = synth.pin_key_chunked(ctx, inst, precompute=1, **kw)
                pin_s = time.perf_counter() - t_pin
                sol, nb_public, r, s = inst.solution, inst.nb_public, inst.r, inst.s
                multigpu.groth16_prove_sharded(pk, sol, nb_public, r, s, dist, dev)   # warm-up
                fence()
                t0 = time.perf_counter()
                for _ in range(args.groth16_proofs):
                    proof = multigpu.groth16_prove_sharded(pk, sol, nb_public, r, s, dist, dev)
                fence()
                el = time.perf_counter() - t0
                rep_ms = None
                if os.environ.get("GA_BENCH_REPLICATE_H", "0") == "1":   # the round-1 scheme, for comparison
                    multigpu.groth16_prove_sharded(pk, sol, nb_public, r, s, dist, dev, replicate_h=True)
                    fence()
                    t0 = time.perf_counter()
                    multigpu.groth16_prove_sharded(pk, sol, nb_public, r, s, dist, dev, replicate_h=True)
                    fence()
                    rep_ms = round((time.perf_counter() - t0) * 1e3, 2)
                lay = groth16.ShardLayout(pk)
                pk.FreeGPUResources()
                tm = torch.tensor([el], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                el = float(tm.item())
                chk = None
                if check_here:   # outside the timed region, after the key has been freed: the closed form of the N = 1 leg
                    try:
                        synth.attach_vector_dlogs(ctx, inst)
                        chk = check_groth16(ctx, inst, proof, os.cpu_count() or 1)
                    except Exception as e:   # a failing CHECKER must not lose the measurement; the line says so
                        chk = {"matches_dlog": None, "checker_error": repr(e)[:300]}
                return {"curve": leg_curve, "proofs_per_s": round(args.groth16_proofs / el, 4), "ms_per_proof": round(el * 1e3 / args.groth16_proofs, 2),
                        "matches_dlog": chk["matches_dlog"] if chk else None, "check": chk,
                        "proofs": args.groth16_proofs, "constraints": n, "scaling": "strong", "partition": args.partition,
                        "mode": ("one proof over %d GPUs: key sharded by base-point range (1/%d of the tables per GPU, each rank generates only its shard), W uploaded per wire range "
                                 "(%d of %d wires on rank 0), A,B,C uploaded 1/N per rank and gathered on the chain owners (N >= 3), computeH chains on ranks 0-2 beside the witness MSMs, h slices scattered, all_gather of 5 partial points" %
                                 (world, world, lay["w_hi"] - lay["w_lo"], lay["nb_wires"])) if args.partition == "range" else
                                ("one proof over %d GPUs: whole key on every GPU, windows of every MSM shared out, h broadcast, all_gather of 5 partial points" % world),
                        "key_pin_s": round(pin_s, 1), "replicate_h_ms_per_proof": rep_ms,
                        "proof_sha": __import__("hashlib").sha256(proof.WriteTo()).hexdigest()[:16]}
            except Exception as e:   # never lose the headline line because of the optional leg
                return {"curve": leg_curve, "error": repr(e)[:300]}
        g16 = sharded_leg(cid, args.curve)
        # BASELINE config 4 (Groth16 BLS12-381 at 2^24 over 8 GPUs) appears in the same line once the node has 8 ranks
        g16_bls = None
        if cid == 0 and (world >= 8 or os.environ.get("GA_BENCH_CONFIG4", "0") == "1"):
            g16_bls = sharded_leg(curve_id("bls12-381"), "bls12-381")
        if rank == 0:
            out["groth16"] = g16
            if g16_bls is not None:
                out["groth16_bls12_381"] = g16_bls

    # ---- CPU baseline: the oracle's Pippenger on a bounded sample (rank 0, N=1) ----------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle
        eff_cores, logical_cpus, quota = effective_cores()
        sample_log = min(args.log_n, 24)
        sn = 1 << sample_log

        def cpu_vs_gpu(logn):
            """the oracle's Pippenger and the library's plain (un-pinned bases: no table) MSM on the same 2^logn inputs"""
            m = 1 << logn
            sb = ctx.malloc(m * words_aff * 8)
            lib.check(lib.ga_gen_bases(ctx.handle, cid, _lib.G1, 0x5EED0002, m, sb.ptr, None))
            P = sb.to_host((m, words_aff))
            ss = ctx.malloc(m * 32)
            lib.check(lib.ga_gen_scalars(ctx.handle, cid, 0x5EED0001, m, ss.ptr))
            S = ss.to_host((m, 4))
            t0 = time.perf_counter()
            ref = oracle.msm(cid, 0, P, S, nthreads=eff_cores)
            cpu = time.perf_counter() - t0
            gpu_res = ecc.MultiExp(ctx, cid, _lib.G1, sb, ss, n=m)   # (also the warm-up of the timed repetitions below)
            reps = 5
            ctx.profile(True)
            ctx.profile_reset()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                ecc.MultiExp(ctx, cid, _lib.G1, sb, ss, n=m)
            ctx.sync()
            gpu = (time.perf_counter() - t0) / reps
            st = stage_stats(ctx.profile_read())
            ctx.profile(False)
            same = bool(np.array_equal(oracle.jac_to_affine(cid, 0, ref), ecc.jac_to_affine(cid, _lib.G1, gpu_res)))
            sb.free()
            ss.free()
            return cpu, gpu, same, st
        cpu_s, _, same, _ = cpu_vs_gpu(sample_log)
        threads_used = min(eff_cores, oracle.msm_windows(cid, sn))   # the port runs one thread per Pippenger window
        out["cpu_baseline"] = {"value": round(sn / cpu_s / 1e6, 4), "unit": "Mscalar-mul/s", "cores": threads_used, "threads_used": threads_used,
                               "effective_cores": eff_cores, "logical_cpus": logical_cpus, "cgroup_cpu_quota": quota, "host_cores": logical_cpus, "kind": "port",
                               "sample": "%s G1 MSM of 2^%d points, oracle/oracle.c Pippenger (one thread per window: %d windows, %d usable CPUs), %.1f s" % (args.curve.upper(), sample_log, oracle.msm_windows(cid, sn), eff_cores, cpu_s),
                               "gpu_result_matches_oracle": same,
                               "note": "a plain-C restatement (64-bit CIOS, no assembly), NOT gnark-crypto: gnark cannot be built here (no Go toolchain); the box shows %d logical CPUs but its cgroup grants %s of them" % (logical_cpus, "all" if quota is None else "%.0f" % quota)}
        # BASELINE config 2: G1 MSM over 2^20 random, UN-PINNED bases (ga_msm: bases converted per call, no table), GPU beside the CPU port
        if args.log_n >= 20:
            try:
                c2_cpu, c2_gpu, c2_same, c2_st = cpu_vs_gpu(20)
                acc2 = c2_st.get("msm_accumulate", {}).get("avg_ms")
                alg2 = (96.0 if cid == 0 else 128.0) * (1 << 20)
                out["config2_msm_2p20_unpinned"] = {
                    "gpu_ms_per_msm": round(c2_gpu * 1e3, 3), "gpu_Mscalar_mul_per_s": round((1 << 20) / c2_gpu / 1e6, 2),
                    "cpu_port_Mscalar_mul_per_s": round((1 << 20) / c2_cpu / 1e6, 4), "cpu_threads": min(eff_cores, oracle.msm_windows(cid, 1 << 20)),
                    "gpu_result_matches_oracle": c2_same,
                    "roofline": {"bound": "hbm", "kernel": "msm_accumulate29_kernel (raw bases: one bucket set per window)", "avg_launch_ms": acc2,
                                 "algorithmic_bytes_per_launch": alg2, "achieved": round(alg2 / (acc2 * 1e-3) / 1e9, 3) if acc2 else None, "peak": 8000.0, "unit": "GB/s",
                                 "frac": round(alg2 / (acc2 * 1e-3) / 8e12, 6) if acc2 else None},
                    "how": "ga_msm on device-resident raw affine bases and Montgomery scalars, 5 timed calls; the CPU port on the same inputs"}
            except Exception as e:
                out["config2_msm_2p20_unpinned"] = {"error": repr(e)[:300]}
        host_cores = eff_cores
        # proofs/s for the same port: the oracle's Groth16 prover (7 FFTs + 4 G1 + 1 G2 MSM) on a bounded 2^20-constraint sample
        if args.groth16_proofs > 0:
            try:
                glog = min(args.log_n, int(os.environ.get("GA_BENCH_CPU_G16_LOGN", "20")))
                from gnark_amd import groth16, synth
                ginst = synth.make_instance(ctx, cid, glog, 0x5EED0020, want_dlogs=False)
                gs = ginst.solution
                t0 = time.perf_counter()
                want = oracle.groth16_prove(cid, dict(ginst.key, n=ginst.n), gs.W, gs.A, gs.B, gs.C, ginst.nb_public, ginst.r, ginst.s, nthreads=host_cores)
                g_s = time.perf_counter() - t0
                gpk = ginst.proving_key(ctx)
                gp = groth16.Prove(gpk, gs, ginst.nb_public, ginst.r, ginst.s)
                gpk.FreeGPUResources()
                g_same = bool(np.array_equal(gp.Ar, want[0]) and np.array_equal(gp.Bs, want[1]) and np.array_equal(gp.Krs, want[2]))
                out["cpu_baseline"]["groth16"] = {"proofs_per_s": round(1.0 / g_s, 4), "constraints": 1 << glog, "kind": "port", "threads_used": host_cores,
                                                  "sample": "oracle/oracle.c Groth16 prover, 2^%d constraints, %d threads, %.1f s" % (glog, host_cores, g_s),
                                                  "gpu_proof_matches_oracle": g_same}
            except Exception as e:
                out["cpu_baseline"]["groth16"] = {"error": repr(e)[:300]}
    if rank == 0:
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
