#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X prover backend (driver contract: ONE compact JSON line from rank 0, < 8 KB, ending in
"summary"; `--detail-file PATH` additionally writes the verbose object with every leg's explanation).

Workload (BASELINE.json `metric`: "G1 MSM Mscalar-mul/s + Groth16 proofs/s, BN254 2^24 constraints, 1/2/4/8 GPU"):
  * a "step" = ONE BN254 G1 Pippenger MSM over 2^24 (scalar, base) pairs, bases (a pinned window table) and scalars resident in HBM
    when the timed region starts (SURVEY 8d config 2/3 shapes: uniform Montgomery scalars, distinct known-dlog bases generated on
    device).  `value` = scalar-muls/s of the whole job in Mscalar-mul/s, checked against [sum s_i k_i]G outside the timed region.
  * N = 1: the whole problem on one GPU.  N > 1: the SAME 2^24-pair problem sharded by base-point range over the N ranks (SURVEY 8e
    partitioning B: rank g owns pairs [g n/N, (g+1) n/N) and contributes one Jacobian point; the exchange step is an all_gather of
    the partials + a local add) -- `"scaling": "strong"`, the fixed 2^24 problem BASELINE quotes at 1/2/4/8 GPUs.  The weak-scaling
    figure (every rank its own 2^24 pairs) is reported beside it as "weak_msm".
  * "groth16": proofs/s at 2^24 constraints (computeH + 4 G1 MSMs + 1 G2 MSM + host epilogue, key pinned with its window tables,
    solver excluded, C = A o B).  N = 1: single caller, two callers on one key ("two_callers"), stage breakdown; every timed proof
    checked against the closed form from the key's known discrete logs + the polynomial identity of h ("matches_dlog").  N > 1: ONE
    proof sharded over the N GPUs (strong scaling; gnark_amd/multigpu.py) in the partition --partition names, the other partition
    beside it ("groth16_window" / "groth16_range"), rank 0 checks each the same way.
  * "groth16_bls12_381" + "msm_bls12_381": BASELINE config 4's curve -- at N = 1 the single-GPU 2^24 proof and the G1 / G2 table MSMs
    with their own roofline figures; at EVERY N > 1 the sharded proof in BOTH partitions: {"window": ..., "range": ...}.
  * "replicas" (N > 1): the throughput figure -- every rank the whole key, independent proofs, aggregate proofs/s.
  * "backend", "world_size", "ranks" (N > 1): the torch.distributed backend really initialised and every rank's device identity
    (name, PCI address, uuid) collected THROUGH that backend.
  * "plonk": BASELINE config 5 (kernel work of one BN254 proof at 2^22 gates: 10 KZG-commit MSMs over a pinned SRS, grand product,
    quotient) with its roofline (N = 1).
  * "roofline": dominant kernel (msm_accumulate) vs the 8 TB/s HBM peak using the ALGORITHMIC 96 B per scalar-mul (32 B scalar + 64 B
    affine base); durations come from hipEvents recorded by the library on its own stream; `bound_actual` and the `int_mad_*`
    scalars price the same launches against the measured v_mad_u64_u32 issue rate (the binding resource); `traffic` is measured IN
    THIS RUN by two child rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; `traffic_source` says so, or names the committed fallback).
  * "cpu_baseline" (rank 0): oracle/msm_fast.c on 2^24 points -- a plain-C restatement of the algorithm gnark-crypto's CPU MultiExp
    publishes (signed digits, batch-affine buckets, tasks on every usable core), NOT gnark-crypto itself; kind "port-batch-affine";
    `value_simple` is oracle/oracle.c's Pippenger (Jacobian buckets, one thread per window) on the same input.  N = 1 adds the
    oracle's Groth16 prover on a 2^20-constraint sample ("groth16_2p20_sample") and config 2; N > 1 carries the MSM figure alone.
  * "nccl_selftest" (N = 1): a one-rank process group over the nccl (= RCCL) backend pushes a sharded proof and an MSM through every
    collective gnark_amd/multigpu.py uses, on device tensors (a 1-GPU box cannot run N > 1, but it can run the RCCL code path).

No rank may enter a collective another rank will not reach: every multi-rank leg is "local step under try -> multigpu.agree() ->
collectives", and a sharded proof walks a fixed collective schedule whatever happens locally (multigpu.ShardedProofError); a rank
that fails makes every rank skip the leg and the line carries the error text.  GA_BENCH_FAIL_RANK / GA_BENCH_FAIL_AT inject faults.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

MADS_PER_ADDITION = {(0, 0): 1467, (1, 0): 3543}            # v_mad_u64_u32 per mixed addition from the shipped ISA (tools/isa_count.py)
MAD_PEAK_T = 34.8                                              # T v_mad_u64_u32 per second the chip sustains on that instruction alone (ga_microbench)
ALG_BYTES = {(0, 0): 96.0, (1, 0): 128.0, (0, 1): 160.0, (1, 1): 224.0}   # scalar + affine base per scalar-mul


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
    ap.add_argument("--no-bls", action="store_true", help="N = 1: leave the BLS12-381 legs (config 4's curve) out")
    ap.add_argument("--no-selftest", action="store_true", help="N = 1: skip the one-rank RCCL self-test")
    ap.add_argument("--only-headline", action="store_true", help="the headline MSM leg alone (what the in-run rocprofv3 counter passes execute)")
    ap.add_argument("--no-pmc", action="store_true", help="N = 1: do not spawn the two rocprofv3 --pmc child passes for roofline.traffic (the committed figure is used)")
    ap.add_argument("--detail-file", default=os.environ.get("GA_BENCH_DETAIL", ""), help="also write the verbose result object (every leg with its prose) to this path; stdout carries the compact line")
    ap.add_argument("--replica-proofs", type=int, default=int(os.environ.get("GA_BENCH_REPLICA_PROOFS", "5")), help="N > 1: proofs per rank in the replica-throughput leg (0 = skip)")
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


def rate(x):
    """a throughput for the JSON line: three decimals, more for the tiny figures of an emulation dry run (2^9 pairs in a second must
    not round to 0)"""
    return round(x, 3) if x >= 1 else float("%.3g" % x)


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


class Run:
    """what every leg needs: ranks, the process group, the library context"""

    def __init__(self, args):
        import torch
        self.args, self.torch = args, torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # GA_BENCH_EMU=1: dry run of this script's control flow against the CPU emulation build of the library (tests/emu), tiny
        # sizes only -- a development aid for the GPU-less build container, never a measurement (the JSON line says "data": "emulation")
        self.emu = os.environ.get("GA_BENCH_EMU", "0") == "1"
        ndev = max(1, torch.cuda.device_count())
        self.device_index = local_rank % ndev       # one rank per GPU on the driver's runs; wraps only in single-GPU smoke runs
        if not self.emu:
            torch.cuda.set_device(self.device_index)
        self.backend = os.environ.get("GA_BENCH_BACKEND", "gloo" if self.emu else "nccl")   # "gloo" lets a 1-GPU box exercise the N>1 code path
        self.dist = None
        if self.world > 1:
            import datetime
            import torch.distributed as dist
            tmo = datetime.timedelta(seconds=int(os.environ.get("GA_BENCH_COLLECTIVE_TIMEOUT_S", "600")))
            if self.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.device_index), timeout=tmo)
            else:
                dist.init_process_group(backend=self.backend, timeout=tmo)
            self.dist = dist
            self.backend = dist.get_backend()   # what was really initialised (the line reports THIS, not what was asked for)
        import gnark_amd
        from gnark_amd import _lib
        if self.emu:
            self.ctx = gnark_amd.Context(0, lib=_lib.Library(os.path.join(ROOT, "tests", "emu", "libgnark_amd_emu.so")))
        else:
            self.ctx = gnark_amd.Context(self.device_index)   # raises when libgnark_amd.so or the GPU is missing: there is no CPU fallback
        self.lib = self.ctx.lib
        self.dev = torch.device("cuda", self.device_index) if not self.emu else None   # with gloo on a GPU box device buffers are staged through the host
        self.coll_dev = self.dev if (self.backend == "nccl" and not self.emu) else None

    def identity(self):
        """this rank's device as the driver can recognise it: name, PCI address, uuid (torch's view of HIP device `device_index`)"""
        if self.emu:
            return "emulation (CPU) pid %d" % os.getpid()
        try:
            p = self.torch.cuda.get_device_properties(self.device_index)
            return "%s pci %04x:%02x:%02x uuid %s" % (p.name, p.pci_domain_id, p.pci_bus_id, p.pci_device_id, str(getattr(p, "uuid", "?"))[:13])
        except Exception as e:
            return "device %d (%s)" % (self.device_index, repr(e)[:60])

    def gather_identities(self):
        """all_gather of every rank's device identity (fixed 96-byte records): did the collective backend really see N ranks on N
        different devices?  Every rank calls it."""
        me = self.identity().encode()[:96].ljust(96, b" ")
        if self.world == 1:
            return [me.decode().rstrip()]
        t = self.torch.tensor(list(me), dtype=self.torch.uint8, device=self.coll_dev or "cpu")
        parts = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        return [bytes(q.cpu().tolist()).decode(errors="replace").rstrip() for q in parts]

    def sum_over_ranks(self, x):
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.coll_dev or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def fence(self):
        if not self.emu:
            self.torch.cuda.synchronize()
        self.ctx.sync()
        if self.world > 1:
            self.dist.barrier()
        if not self.emu:
            self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.world == 1:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.coll_dev or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def agree(self, ok, err=None):
        from gnark_amd import multigpu
        return multigpu.agree(self.dist, ok, err, self.coll_dev)

    def fault(self, point):
        """GA_BENCH_FAIL_RANK=<rank> GA_BENCH_FAIL_AT=<point>: that rank fails there (tests/test_bench_contract.py)"""
        if os.environ.get("GA_BENCH_FAIL_AT") == point and int(os.environ.get("GA_BENCH_FAIL_RANK", "-1")) == self.rank:
            raise RuntimeError("injected fault at '%s' on rank %d (GA_BENCH_FAIL_RANK / GA_BENCH_FAIL_AT)" % (point, self.rank))


def oracle_modules():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import checkers
    import oracle
    import pyref
    return oracle, pyref, checkers


# ---------------------------------------------------------------------------------------------------------------------------------
# MSM legs
# ---------------------------------------------------------------------------------------------------------------------------------
def roofline_object(cid, group, pairs_per_launch, windows, acc, kernel, traffic=None, traffic_source=None):
    """the dominant kernel against the HBM peak with SURVEY 8d's algorithmic bytes (the judged figure: `frac`), and -- same object, flat
    scalars so that they survive the driver's parsing -- against the integer multiplier, the resource that really binds it"""
    alg = ALG_BYTES[(cid, group)] * pairs_per_launch
    ms = acc["avg_ms"] if acc else float("nan")
    achieved = alg / (ms * 1e-3) / 1e9
    r = {"bound": "hbm", "bound_actual": "integer issue (v_mad_u64_u32); HBM is the yardstick SURVEY 8d prescribes, not the limiter", "kernel": kernel,
         "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 6),
         "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms,
         "launches_timed": acc["launches"] if acc else 0}
    mads = MADS_PER_ADDITION.get((cid, group))
    if mads:   # "report achieved MAD/s fraction"
        adds = windows * pairs_per_launch
        r.update({"int_mad_per_addition": mads, "int_additions_per_launch": adds, "int_mad_T_per_s": round(mads * adds / (ms * 1e-3) / 1e12, 2),
                  "int_mad_peak_T_per_s": MAD_PEAK_T, "int_mad_frac": round(mads * adds / (ms * 1e-3) / (MAD_PEAK_T * 1e12), 3),
                  "int_mad_peak_source": "ga_microbench: v_mad_u64_u32 alone, sustained (profiles/r04_d_microbench.json)"})
    return r


def pmc_traffic(kernel_key, curve, log_n):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_latest.json), same kernel and shape only"""
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        ent = pj["kernels"][kernel_key].get(curve, {}).get(str(log_n))
        if ent:
            return (ent["fetch_bytes"] + ent["write_bytes"],
                    "profiles/pmc_latest.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (%s) of this kernel at this shape, not re-measured in this run" % pj.get("source", "round 2"))
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def pmc_traffic_live(args):
    """HBM bytes per launch of the dominant kernel MEASURED IN THIS RUN: two child processes -- `rocprofv3 --pmc FETCH_SIZE
    --kernel-trace` and the same for WRITE_SIZE, separate passes as the profiling guide prescribes (the two counters do not fit one
    pass; --pmc is never combined with anything but --kernel-trace) -- over `bench.py --only-headline --steps 3`.  Returns (bytes per
    launch or None, source text, detail dict).  rocprofv3 reports both counters in KiB per dispatch."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    # (a bench.py that is itself being profiled does not start profilers of its own: the tool library of the outer rocprofv3 is
    # inherited by every child, and counter collection must never meet another tracing mode)
    if any(k.startswith("ROCPROF_") or k == "ROCP_TOOL_LIBRARIES" for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process runs under a profiler: in-run counter passes skipped", {}
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found on this box", {}
    detail, vals = {}, {}
    t0 = time.perf_counter()
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ga_pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
            env.pop(k, None)
        cmd = [exe, "--pmc", ctr, "--kernel-trace", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--only-headline", "--steps", "3",
               "--warmup", "1", "--log-n", str(args.log_n), "--curve", args.curve, "--no-check", "--no-pmc"]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd="/tmp", timeout=int(os.environ.get("GA_BENCH_PMC_TIMEOUT_S", "240")))
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("_results.db")]
            if r.returncode != 0 or not dbs:
                detail[ctr] = "rc %d, %d result files; %s" % (r.returncode, len(dbs), (r.stderr or "")[-160:])
                continue
            db = sqlite3.connect(dbs[0])
            row = db.execute("select count(*), avg(value), avg(duration) from counters_collection where counter_name = ? and kernel_name like "
                             "'%msm_accumulate29_kernel%'", (ctr,)).fetchone()
            db.close()
            if row and row[0]:
                vals[ctr] = float(row[1]) * 1024.0
                detail[ctr] = {"launches": int(row[0]), "bytes_per_launch": round(vals[ctr], 1), "avg_launch_ms_under_rocprof": round(float(row[2]) / 1e6, 3)}
            else:
                detail[ctr] = "no msm_accumulate29_kernel dispatch in the counter collection"
        except Exception as e:
            detail[ctr] = repr(e)[:200]
        finally:
            shutil.rmtree(d, ignore_errors=True)
    detail["seconds"] = round(time.perf_counter() - t0, 1)
    if len(vals) == 2:
        return vals["FETCH_SIZE"] + vals["WRITE_SIZE"], ("this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes (--kernel-trace only) over "
                                                            "bench.py --only-headline --steps 3; KiB x 1024 per dispatch, uncorrected"), detail
    return None, "the in-run rocprofv3 passes failed: " + json.dumps(detail)[:200], detail


def check_msm_result(R, cid, group, scalars_host, dlogs_host, result):
    """is the timed result THE result?  MSM(s, [k_i]G) = [sum s_i k_i]G: the exponent by a dot product on the CPU oracle, the point
    by the oracle's fixed-base multiplication -- outside the timed region (the oracle is the checker, never the thing timed).  With
    N > 1 every rank contributes the exponent of its shard; every rank takes part in the collective whatever happened to its checker."""
    from gnark_amd import ecc, synth
    e, err, oracle = None, None, None
    try:
        oracle = oracle_modules()[0]
        e = oracle.fr_dot(cid, scalars_host, dlogs_host)
    except Exception as ex:   # a failing checker is reported, it does not hide the measurement
        err = "checker error: " + repr(ex)[:200]
    if R.world > 1:
        torch = R.torch
        mine = torch.tensor([((e or 0) >> (32 * i)) & 0xFFFFFFFF for i in range(8)] + [0 if e is None else 1], dtype=torch.int64,
                            device=R.coll_dev or "cpu")
        parts = [torch.zeros_like(mine) for _ in range(R.world)]
        R.dist.all_gather(parts, mine)
        parts = [p.tolist() for p in parts]
        if all(p[8] == 1 for p in parts):
            e = sum(sum(int(v) << (32 * i) for i, v in enumerate(p[:8])) for p in parts) % synth.FR_MODULUS[cid]
        else:
            e, err = None, err or "checker error on another rank"
    if e is None:
        return err
    try:
        want_pt = oracle.jac_to_affine(cid, group, oracle.generator_mul(cid, group, e))
        return bool(np.array_equal(ecc.jac_to_affine(cid, group, result, lib=R.lib), want_pt))
    except Exception as ex:
        return "checker error: " + repr(ex)[:200]


def timed_msm(R, cid, group, n_total, lo, cnt, seed_bases, seed_scalars, steps, warmup, use_table=True, fault_point=None):
    """One MSM problem of n_total pairs (global seeds) of which this rank holds pairs [lo, lo + cnt): generate, pin (table), warm up,
    time `steps` steps between fences, check.  Returns (result dict, keep) -- keep holds the device buffers for follow-up legs (N = 1).
    Multi-rank discipline: local preparation under try, then agree(), then the collectives of the timed loop."""
    from gnark_amd import _lib, ecc, multigpu
    from gnark_amd.device import affine_words
    ctx, lib, args = R.ctx, R.lib, R.args
    words_aff = affine_words(cid, group)
    keep = {"bases": None, "scalars": None, "dlogs": None, "table": None}
    ok, err = True, None
    try:
        if fault_point:
            R.fault(fault_point)
        keep["bases"] = ctx.malloc(max(cnt, 1) * words_aff * 8)
        keep["scalars"] = ctx.malloc(n_total * 32)                 # the whole scalar vector of the problem; a shard reads its range of it
        keep["dlogs"] = ctx.malloc(max(cnt, 1) * 32) if not args.no_check else None   # k_i of [k_i]G: the timed result is checked against them
        lib.check(lib.ga_gen_bases_at(ctx.handle, cid, group, seed_bases, lo, cnt, keep["bases"].ptr, keep["dlogs"].ptr if keep["dlogs"] else None))
        lib.check(lib.ga_gen_scalars(ctx.handle, cid, seed_scalars, n_total, keep["scalars"].ptr))
        if use_table:   # the bases are a pinned key: keep them with their window multiples (ga_msm_table_*, built outside the timed region)
            keep["table"] = with_retry(lambda: ecc.PrecomputedBases(ctx, cid, group, keep["bases"], n=cnt), "building the MSM window table")
    except Exception as e:
        ok, err = False, repr(e)[:300]
    ok, err = R.agree(ok, err)
    if not ok:
        free_keep(keep)
        return {"error": err}, None
    s_ptr = keep["scalars"].ptr + lo * 32
    if use_table:
        ti = keep["table"].info()
        cbits, nwin = ti["window_bits"], ti["windows"]
    else:
        ti = None
        cbits, nwin = ecc.plan(cid, group, cnt)

    def step():
        # N = 1: plain MSM.  N > 1: partition B (base-point range) -- every rank reduces its pairs to one Jacobian partial, RCCL
        # all_gather of the partials, local add (gnark_amd/multigpu.py)
        return multigpu.msm_base_sharded(ctx, cid, group, keep["table"] if use_table else keep["bases"], s_ptr, cnt, R.dist, R.dev)

    for _ in range(warmup):
        step()
    ctx.profile(True)
    ctx.profile_reset()
    R.fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        result = step()
    R.fence()
    elapsed = time.perf_counter() - t0
    stages = stage_stats(ctx.profile_read())
    ctx.profile(False)
    elapsed = R.max_over_ranks(elapsed)
    checked = None
    if keep["dlogs"] is not None:
        from gnark_amd.device import DeviceBuffer
        s_host = DeviceBuffer(ctx, s_ptr, cnt * 32).to_host((cnt, 4))   # (a view of this rank's range of the scalar buffer)
        checked = check_msm_result(R, cid, group, s_host, keep["dlogs"].to_host((cnt, 4)), result)
        keep["dlogs"].free()
        keep["dlogs"] = None
    res = {"ms_per_step": elapsed * 1e3 / steps, "elapsed_s": elapsed, "value_checked": checked, "stages": stages, "result": result,
           "window_bits": cbits, "windows": nwin, "table_info": ti, "pairs_this_rank": cnt}
    return res, keep


def free_keep(keep):
    if not keep:
        return
    for k in ("table", "bases", "scalars", "dlogs"):
        if keep.get(k) is not None:
            keep[k].free()
            keep[k] = None


def headline_leg(R):
    """`value`: N = 1 -- one 2^log_n MSM per step on the GPU; N > 1 -- the SAME problem sharded by base-point range (strong scaling)"""
    from gnark_amd import _lib, multigpu
    from gnark_amd.device import curve_id
    args = R.args
    cid = curve_id(args.curve)
    n = 1 << args.log_n
    lo, hi = multigpu.shard_range(n, R.rank, R.world)
    use_table = os.environ.get("GA_BENCH_TABLE", "1") != "0"
    res, keep = timed_msm(R, cid, _lib.G1, n, lo, hi - lo, 0x5EED0002, 0x5EED0001, args.steps, args.warmup, use_table=use_table, fault_point="headline")
    out = {
        "metric": "G1 MSM throughput, %s, 2^%d scalar-muls %s (Groth16 proofs/s at 2^%d constraints in 'groth16')" % (
            args.curve.upper(), args.log_n, "on one GPU" if R.world == 1 else "sharded over %d GPUs" % R.world, args.log_n),
        "value": None, "value_checked": None, "unit": "Mscalar-mul/s", "n_gpus": R.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u64 Montgomery limbs in memory; 29/28-bit limbs, v_mad_u64_u32 (32x32+64) in registers", "data": "emulation" if R.emu else "synthetic",
        "config": {"workload": "%s G1 Pippenger MSM, ONE problem of 2^%d uniform scalars x distinct known-dlog affine bases%s, inputs resident in HBM" % (
            args.curve.upper(), args.log_n, "" if R.world == 1 else " sharded by base-point range over %d GPUs (2^%d / %d pairs per GPU)" % (R.world, args.log_n, R.world)),
            "curve": args.curve},
    }
    if "error" in res:
        out["error"] = "headline MSM skipped on every rank: " + res["error"]
        return out, None
    ti = res["table_info"]
    out["value"] = rate(n * args.steps / res["elapsed_s"] / 1e6)
    out["ms_per_step"] = round(res["ms_per_step"], 4)
    out["value_checked"] = res["value_checked"]
    out["config"].update({"window_bits": res["window_bits"], "windows": res["windows"],
                          "precompute": ("[2^(c*w)]P tables for all %d windows, %.1f GiB per GPU, one shared bucket set" % (res["windows"], ti["table_bytes"] / 2**30)) if ti else "none",
                          "parallelism": "1 GPU" if R.world == 1 else "base-range sharding x%d, %s all_gather of Jacobian partials" % (
                              R.world, "RCCL (torch.distributed backend nccl)" if R.backend == "nccl" else "torch.distributed backend " + str(R.backend))})
    if R.world > 1:
        out["config"]["backend"] = R.backend
    acc = res["stages"].get("msm_accumulate")
    traffic, tsrc = (None, None)
    if R.world == 1 and not R.emu:
        live = (None, "in-run counter passes switched off (%s)" % ("--only-headline" if args.only_headline else "--no-pmc"), {})
        if not args.no_pmc and not args.only_headline:
            live = pmc_traffic_live(args)
            out["pmc_passes"] = live[2]
        traffic, tsrc = live[0], live[1]
        if traffic is None:   # fall back to the committed figure, and say so
            ctraffic, csrc = pmc_traffic("msm_accumulate_kernel", args.curve, args.log_n)
            if ctraffic is not None:
                traffic, tsrc = ctraffic, "COMMITTED figure, not this run (%s); %s" % (tsrc, csrc)
    out["roofline"] = roofline_object(cid, 0, res["pairs_this_rank"], res["windows"], acc, "msm_accumulate29_kernel" if use_table else "msm_accumulate_kernel", traffic, tsrc)
    if traffic is not None:
        out["roofline"]["traffic_calibration"] = ("FETCH_SIZE + WRITE_SIZE as reported; the guide's x2 for wide coalesced streaming reads does not apply to 64-B "
                                                  "gathers (x2 would exceed full 128-B lines for every gather); this library's 16 B/lane streaming passes read back 1.05x "
                                                  "their bytes (profiles/pmc_latest.json 'calibration')")
        out["roofline"]["traffic_over_algorithmic"] = round(traffic / out["roofline"]["algorithmic_bytes_per_launch"], 2)
    if R.world > 1:
        out["roofline"]["rank"] = 0
        out["roofline"]["note"] = "rank 0's launches over its 2^%d / %d pairs" % (args.log_n, R.world)
    out["stages_ms"] = res["stages"]
    return out, (keep, res)


def single_gpu_msm_extras(R, out, keep, res):
    """N = 1: the same MSM without precomputed tables, and with the scalars in host memory (PCIe-inclusive; never `value`)"""
    from gnark_amd import _lib, ecc
    from gnark_amd.device import curve_id
    ctx, args = R.ctx, R.args
    cid, n = curve_id(args.curve), 1 << args.log_n
    bases, scalars, table = keep["bases"], keep["scalars"], keep["table"]
    if table is None:
        return
    ecc.MultiExp(ctx, cid, _lib.G1, bases, scalars, n=n)
    ctx.profile(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(2):
        ecc.MultiExp(ctx, cid, _lib.G1, bases, scalars, n=n)
    el = (time.perf_counter() - t0) / 2
    st = stage_stats(ctx.profile_read())
    ctx.profile(False)
    out["plain_msm_no_tables"] = {"ms_per_msm": round(el * 1e3, 3), "Mscalar_mul_per_s": round(n / el / 1e6, 2),
                                  "sort_ms": round(sum(v["total_ms"] for k, v in st.items() if k in ("msm_sort", "msm_digits", "msm_tasks", "msm_offsets") or k.startswith("msm_p")) / 2, 3),
                                  "stages_ms": {k: round(v["total_ms"] / 2, 3) for k, v in st.items()}}
    # SURVEY 8d metric (i), second figure: the headline MSM with the SCALARS in host memory
    s_host = scalars.to_host((n, 4))
    r_host = table.MultiExp(s_host)
    t0 = time.perf_counter()
    for _ in range(3):
        r_host = table.MultiExp(s_host)
    el_h = (time.perf_counter() - t0) / 3
    out["msm_with_scalar_h2d"] = {"ms_per_msm": round(el_h * 1e3, 3), "Mscalar_mul_per_s": round(n / el_h / 1e6, 2),
                                  # (the same group element: the order inside a bucket -- and with it the Jacobian representative of the sum --
                                  # differs from run to run since the digits are fused with the first sort pass; the affine point does not)
                                  "same_result": bool(np.array_equal(ecc.jac_to_affine(cid, _lib.G1, r_host, lib=ctx.lib),
                                                                     ecc.jac_to_affine(cid, _lib.G1, res["result"], lib=ctx.lib))),
                                  "how": "the timed table MSM with its 2^%d x 32 B of scalars uploaded from pageable host memory inside the call" % args.log_n}


def weak_msm_leg(R):
    """N > 1: every rank its own 2^log_n pairs (weak scaling; round 3's headline), for comparison with the strong-scaling `value`"""
    from gnark_amd import _lib
    from gnark_amd.device import curve_id
    args = R.args
    cid, n = curve_id(args.curve), 1 << args.log_n
    res, keep = timed_msm(R, cid, _lib.G1, n, 0, n, 0x5EED0002 + 977 * (R.rank + 1), 0x5EED0001 + 977 * (R.rank + 1), args.steps, args.warmup, fault_point="weak_msm")
    free_keep(keep)
    if "error" in res:
        return {"error": "skipped on every rank: " + res["error"]}
    return {"value": rate(R.world * n * args.steps / res["elapsed_s"] / 1e6), "unit": "Mscalar-mul/s", "scaling": "weak",
            "ms_per_step": round(res["ms_per_step"], 4), "value_checked": res["value_checked"], "pairs_per_gpu": n,
            "how": "every rank its own 2^%d pairs (base-range partition of a %d x 2^%d problem), one all_gather of a Jacobian point per step" % (args.log_n, R.world, args.log_n)}


def bls_msm_leg(R):
    """N = 1, BASELINE config 4's curve: the BLS12-381 G1 and G2 table MSMs at 2^log_n with their own roofline objects"""
    from gnark_amd import _lib
    args = R.args
    n = 1 << args.log_n
    out = {}
    for group, name in ((_lib.G1, "g1"), (_lib.G2, "g2")):
        try:
            res, keep = timed_msm(R, 1, group, n, 0, n, 0x5EED0042 + group, 0x5EED0041, 3, 1)
            free_keep(keep)
            if "error" in res:
                out[name] = res
                continue
            acc = res["stages"].get("msm_accumulate")
            traffic, tsrc = pmc_traffic("msm_accumulate_kernel_g%d" % (group + 1), "bls12-381", args.log_n)
            out[name] = {"ms_per_msm": round(res["ms_per_step"], 3), "Mscalar_mul_per_s": round(n / (res["ms_per_step"] * 1e-3) / 1e6, 2),
                         "value_checked": res["value_checked"], "window_bits": res["window_bits"], "windows": res["windows"],
                         "table_GiB": round(res["table_info"]["table_bytes"] / 2**30, 1),
                         "roofline": roofline_object(1, group, n, res["windows"], acc, "msm_accumulate29_kernel<%s<BLS12_381_Fp>>" % ("Fe" if group == 0 else "Fe2"), traffic, tsrc),
                         "stages_ms": {k: v["avg_ms"] for k, v in res["stages"].items()}}
        except Exception as e:   # never lose the headline line over a secondary leg
            out[name] = {"error": repr(e)[:300]}
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# Groth16 legs
# ---------------------------------------------------------------------------------------------------------------------------------
def check_groth16(ctx, inst, proof, threads):
    """Outside every timed region: is the timed proof THE proof?  Exponents of Ar, Bs, Krs from the key's known discrete logs by
    O(n) dot products on the CPU oracle, h by the polynomial identity (oracle/checkers.py).  The oracle is the checker here,
    never the thing measured."""
    oracle, pyref, checkers = oracle_modules()
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


def groth16_single_gpu_leg(R, cid, curve_name):
    """N = 1: W,A,B,C in host memory -> Ar,Bs,Krs affine on host, key pinned with its window tables; one caller, then two callers on
    one key; a profiled pass for the stage table; the proof checked by the key's known discrete logs (outside the timed regions)."""
    import hashlib
    import threading
    from gnark_amd import groth16, synth
    ctx, args = R.ctx, R.args
    n = 1 << args.log_n
    proofs = args.groth16_proofs
    t_setup = time.perf_counter()
    threads = os.cpu_count() or 1
    # synthetic 2^log_n-constraint instance: known-dlog key generated on device, pulled to the host once so that
    # it goes through the same ga_g16_pk_create upload path a Go caller uses; C = A o B.  Pinned WITH its window tables (precompute = 1,
    # not "auto": auto falls back to plain bases when the tables do not fit the free HBM at that moment, and the bench must not
    # silently time a different configuration)
    inst = synth.make_instance(ctx, cid, args.log_n, 0x5EED0005, want_dlogs=not args.no_check)
    pre = int(os.environ.get("GA_BENCH_PRECOMPUTE", "1"))
    pk = with_retry(lambda: inst.proving_key(ctx, precompute=pre), "pinning the proving key")
    sol, nb_public, r, s = inst.solution, inst.nb_public, inst.r, inst.s
    setup_s = time.perf_counter() - t_setup
    warm = int(os.environ.get("GA_BENCH_G16_WARMUP", "2"))
    for _ in range(warm):   # warm-up: scratch of both lanes of the pair (witness MSMs on lane 0, H side on lane 1)
        groth16.Prove(pk, sol, nb_public, r, s)
    ctx.sync()
    lanes0 = ctx.lane_stats()
    t0 = time.perf_counter()
    each = []
    for _ in range(proofs):
        q0 = time.perf_counter()
        proof = groth16.Prove(pk, sol, nb_public, r, s)
        each.append(round((time.perf_counter() - q0) * 1e3, 2))
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
    per_thread = 0 if args.no_pipelined else max(10 if cid == 0 else 5, proofs)

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
    # The drop-in DEFAULT beside the headline (the Go shim's PinToGPU = false, as icicle.go:797-805): the key goes up as plain vectors
    # (precompute = -1: no window tables), ONE proof, the device copy is freed -- per proof.  The first round also pays the allocation
    # of what the plain-vector MSMs need beside the pinned path's scratch; the last round is the figure.
    one_shot = None
    if os.environ.get("GA_BENCH_ONE_SHOT", "1") != "0":
        try:
            rounds = []
            for _ in range(1 if R.emu else 3):
                ctx.sync()
                q0 = time.perf_counter()
                pk1 = inst.proving_key(ctx, precompute=-1)
                q1 = time.perf_counter()
                p1 = groth16.Prove(pk1, sol, nb_public, r, s)
                q2 = time.perf_counter()
                pk1.FreeGPUResources()
                ctx.sync()
                q3 = time.perf_counter()
                rounds.append(((q3 - q0) * 1e3, (q1 - q0) * 1e3, (q2 - q1) * 1e3, (q3 - q2) * 1e3))
            # ... and the same through ga_g16_prove_oneshot: the key uploaded by a helper thread WHILE the proof runs
            fused = []
            for _ in range(1 if R.emu else 3):
                ctx.sync()
                q0 = time.perf_counter()
                p2 = inst.prove_oneshot(ctx)
                ctx.sync()
                fused.append((time.perf_counter() - q0) * 1e3)
            one_shot = {"ms": round(fused[-1], 2), "fused_first_round_ms": round(fused[0], 2), "fused_same_proof_bytes": bool(np.array_equal(p2.raw(), proof.raw())),
                        "sequential_ms": round(rounds[-1][0], 2),
                        "upload_key_ms": round(rounds[-1][1], 2), "prove_ms": round(rounds[-1][2], 2), "free_ms": round(rounds[-1][3], 2),
                        "first_round_ms": round(rounds[0][0], 2), "same_proof_bytes": bool(np.array_equal(p1.raw(), proof.raw())),
                        "how": "ms: ga_g16_prove_oneshot (the key goes up as plain vectors WHILE the proof runs, then is dropped) -- what groth16.Prove of the Go package costs with its default PinToGPU = false; sequential_ms: ga_g16_pk_create(precompute = -1) + ga_g16_prove + ga_g16_pk_destroy one after the other (= upload_key_ms + prove_ms + free_ms); 3 rounds each, the last one quoted"}
        except Exception as e:
            one_shot = {"error": repr(e)[:300]}
    ntt_ms = sum(v["total_ms"] for k, v in gst.items() if k.startswith("ntt_") or k == "h_pointwise") / prof_proofs
    bytes_per_constraint = 992 if cid == 0 else 1184   # 4 G1 + 1 G2 MSM + 7 NTTs
    g = {"curve": curve_name, "proofs_per_s": round(proofs / el, 4), "ms_per_proof": round(el * 1e3 / proofs, 2),
         "proofs": proofs, "constraints": n, "key_setup_s": round(setup_s, 1), "ms_each": each,
         "schedule": {"split_proofs": lanes1["split_proofs"] - lanes0["split_proofs"],
                      "how": "one caller: witness MSMs (A, B1, B2) on lane 0, uploads of A,B,C + computeH + Z MSM on lane 1 from a helper thread, K MSM on whichever lane is free first (GA_G16_SPLIT=0: everything on lane 0)"},
         "ms_per_proof_profiled_single_lane": round(el_prof * 1e3 / prof_proofs, 2),
         "definition": "W,A,B,C in host memory -> Ar,Bs,Krs affine on host; key pinned with window tables (precompute=%s); solver excluded; C = A o B (satisfiable instance)" % os.environ.get("GA_BENCH_PRECOMPUTE", "1"),
         "algorithmic_bytes": bytes_per_constraint * n, "hbm_frac_whole_proof": round(bytes_per_constraint * n / (el / proofs) / 8e12, 6),
         "computeH_ms": round(ntt_ms, 3),
         "computeH_hbm_frac": round(448.0 * n / (ntt_ms * 1e-3) / 8e12, 5) if ntt_ms > 0 else None,
         "proof_sha": hashlib.sha256(proof.WriteTo()).hexdigest()[:16],
         "one_shot_unpinned_ms": one_shot.get("ms") if isinstance(one_shot, dict) else None, "one_shot_unpinned": one_shot,
         "pipelined": None if per_thread == 0 else {"proofs_per_s": round(2 * per_thread / pipe_el, 4), "ms_per_proof": round(pipe_el * 1e3 / (2 * per_thread), 2),
                                                    "proofs": 2 * per_thread, "host_threads": 2, "same_proof_bytes": pipe_same, "lanes": pipe_lanes,
                                                    "vs_single_caller": round(pipe_el / (2 * per_thread) / (el / proofs), 4),
                                                    "how": "two host threads call ga_g16_prove on one key, both lane pairs warmed first: two proofs in flight on lanes 0/1 and 2/3 of one context; every proof compared with the single-caller proof"},
         "stages_ms": {k: {"launches": v["launches"], "total_ms": round(v["total_ms"] / prof_proofs, 4), "avg_ms": v["avg_ms"]} for k, v in gst.items()},
         "stages_note": "total_ms is per proof, from %d extra proofs with the stage profiler on (single-lane schedule); ms_per_proof above is timed without it" % prof_proofs}
    if not args.no_check:
        t_chk = time.perf_counter()
        try:
            g["check"] = check_groth16(ctx, inst, proof, threads)
        except Exception as e:   # a failing checker must not hide the measurement -- it is reported instead
            g["check"] = {"error": repr(e)[:300], "matches_dlog": None}
        g["check"]["seconds"] = round(time.perf_counter() - t_chk, 1)
        g["matches_dlog"] = g["check"].get("matches_dlog")
    return g


def groth16_sharded_leg(R, leg_cid, leg_curve, partition):
    """N > 1: ONE 2^log_n proof over the world's GPUs -- strong scaling of BASELINE config 3 / 4.  Every rank generates ITS shard of
    the synthetic key on its own device, chunk by chunk (synth.pin_key_chunked: no 12 GiB key is staged through host memory), and the
    solution (the prover's real input); W uploaded per wire range, computeH's chains on ranks 0..2, h slices scattered over xGMI, one
    all_gather of the partial sums (gnark_amd/multigpu.py).  Local steps under try, agree(), then collectives."""
    import hashlib
    from gnark_amd import groth16, multigpu, synth
    ctx, args, rank, world = R.ctx, R.args, R.rank, R.world
    n = 1 << args.log_n
    proofs = args.groth16_proofs
    check_here = rank == 0 and not args.no_check   # rank 0 checks the sharded proof against the key's known discrete logs
    pk, inst, ok, err, pin_s = None, None, True, None, None
    try:
        R.fault("pin")
        inst = synth.make_instance(ctx, leg_cid, args.log_n, 0x5EED0005, want_dlogs=check_here, with_key=False)   # same seeds on every rank
        kw = dict(shard=(rank, world)) if partition == "range" else dict(window_shard=(rank, world))
        t_pin = time.perf_counter()
        pk = with_retry(lambda: synth.pin_key_chunked(ctx, inst, precompute=1, **kw), "pinning the key shard")
        pin_s = time.perf_counter() - t_pin
    except Exception as e:
        ok, err = False, repr(e)[:300]
    ok, err = R.agree(ok, err)
    if not ok:
        if pk is not None:
            pk.FreeGPUResources()
        return {"curve": leg_curve, "error": "skipped on every rank, key pinning failed: " + err}
    sol, nb_public, r, s = inst.solution, inst.nb_public, inst.r, inst.s
    try:   # the proof's own collective schedule is failure-proof (multigpu.ShardedProofError is raised on EVERY rank)
        R.fault("warmup_local")
        ok, err = True, None
    except Exception as e:
        ok, err = False, repr(e)[:300]
    ok, err = R.agree(ok, err)
    if ok:
        try:
            multigpu.groth16_prove_sharded(pk, sol, nb_public, r, s, R.dist, R.dev)   # warm-up
        except multigpu.ShardedProofError as e:
            ok, err = False, repr(e)[:300]
        ok, err = R.agree(ok, err)
    if not ok:
        pk.FreeGPUResources()
        return {"curve": leg_curve, "error": "skipped on every rank, warm-up proof failed: " + str(err)}
    proof, el, rep_ms = None, None, None
    try:
        R.fence()
        t0 = time.perf_counter()
        for _ in range(proofs):
            proof = multigpu.groth16_prove_sharded(pk, sol, nb_public, r, s, R.dist, R.dev)
        R.fence()
        el = time.perf_counter() - t0
        if os.environ.get("GA_BENCH_REPLICATE_H", "0") == "1":   # the round-1 scheme, for comparison
            multigpu.groth16_prove_sharded(pk, sol, nb_public, r, s, R.dist, R.dev, replicate_h=True)
            R.fence()
            t0 = time.perf_counter()
            multigpu.groth16_prove_sharded(pk, sol, nb_public, r, s, R.dist, R.dev, replicate_h=True)
            R.fence()
            rep_ms = round((time.perf_counter() - t0) * 1e3, 2)
    except multigpu.ShardedProofError as e:   # raised on every rank at the same point of the schedule: nobody is left in a collective
        lay = groth16.ShardLayout(pk)
        pk.FreeGPUResources()
        return {"curve": leg_curve, "error": "timed proof failed: " + repr(e)[:300]}
    lay = groth16.ShardLayout(pk)
    pk.FreeGPUResources()
    el = R.max_over_ranks(el)
    chk = None
    if check_here:   # outside the timed region, after the key has been freed: the closed form of the N = 1 leg
        try:
            synth.attach_vector_dlogs(ctx, inst)
            chk = check_groth16(ctx, inst, proof, os.cpu_count() or 1)
        except Exception as e:   # a failing CHECKER must not lose the measurement; the line says so
            chk = {"matches_dlog": None, "checker_error": repr(e)[:300]}
    return {"curve": leg_curve, "proofs_per_s": round(proofs / el, 4), "ms_per_proof": round(el * 1e3 / proofs, 2),
            "matches_dlog": chk["matches_dlog"] if chk else None, "check": chk,
            "proofs": proofs, "constraints": n, "scaling": "strong", "partition": partition,
            "mode": ("one proof over %d GPUs: key sharded by base-point range (1/%d of the tables per GPU, each rank generates only its shard), W uploaded per wire range "
                     "(%d of %d wires on rank 0), A,B,C uploaded 1/N per rank and gathered on the chain owners (N >= 3), computeH chains on ranks 0-2 beside the witness MSMs, h slices scattered, all_gather of 5 partial points" %
                     (world, world, lay["w_hi"] - lay["w_lo"], lay["nb_wires"])) if partition == "range" else
                    ("one proof over %d GPUs: whole key on every GPU, windows of every MSM shared out, h broadcast, all_gather of 5 partial points" % world),
            "key_pin_s": round(pin_s, 1), "replicate_h_ms_per_proof": rep_ms,
            "proof_sha": hashlib.sha256(proof.WriteTo()).hexdigest()[:16]}


def replicas_leg(R, cid, curve_name, reference_sha=None):
    """N > 1, the THROUGHPUT figure: every rank pins the WHOLE key with its window tables on its own GPU and
    proves `--replica-proofs` independent proofs, single caller -- what the reference's one-proof-per-device contract
    (icicle.go:77-86,821-823) gives on N devices.  No data-path collective: the ranks meet only at the two fences around the timed
    region and in the all_reduce / all_gather of counts and times after it.  Aggregate proofs/s = proofs of all ranks / max-over-ranks
    time between the fences.  Local steps under try -> agree() -> collectives, like every multi-rank leg."""
    import hashlib
    from gnark_amd import groth16, synth
    ctx, args = R.ctx, R.args
    k = args.replica_proofs
    pk, inst, ok, err, pin_s = None, None, True, None, None
    try:
        R.fault("replica_pin")
        inst = synth.make_instance(ctx, cid, args.log_n, 0x5EED0005, want_dlogs=False, with_key=False)   # same seeds on every rank: same proof
        t_pin = time.perf_counter()
        pk = with_retry(lambda: synth.pin_key_chunked(ctx, inst, precompute=1), "pinning the whole key")
        pin_s = time.perf_counter() - t_pin
        for _ in range(2):
            groth16.Prove(pk, inst.solution, inst.nb_public, inst.r, inst.s)
        ctx.sync()
    except Exception as e:
        ok, err = False, repr(e)[:300]
    ok, err = R.agree(ok, err)
    if not ok:
        if pk is not None:
            pk.FreeGPUResources()
        return {"curve": curve_name, "error": "skipped on every rank: " + str(err)}
    sol, nb_public, r, s = inst.solution, inst.nb_public, inst.r, inst.s
    proof, local_s, err2 = None, float("nan"), None
    R.fence()
    t0 = time.perf_counter()
    try:
        for _ in range(k):
            proof = groth16.Prove(pk, sol, nb_public, r, s)
        ctx.sync()
        local_s = time.perf_counter() - t0
    except Exception as e:   # (no collective inside the loop: a failing rank still reaches the fence below)
        err2 = repr(e)[:300]
    R.fence()
    el = R.max_over_ranks(time.perf_counter() - t0)
    pk.FreeGPUResources()
    done = R.sum_over_ranks(float(k if err2 is None else 0))        # all_reduce of the proof counts
    sha = hashlib.sha256(proof.WriteTo()).hexdigest()[:16] if proof is not None else "-" * 16
    rec = ("%9.3f %s" % (local_s * 1e3 / max(k, 1) if err2 is None else -1.0, sha)).encode()[:32].ljust(32, b" ")
    if R.world > 1:
        t = R.torch.tensor(list(rec), dtype=R.torch.uint8, device=R.coll_dev or "cpu")
        parts = [R.torch.zeros_like(t) for _ in range(R.world)]
        R.dist.all_gather(parts, t)
        recs = [bytes(q.cpu().tolist()).decode().split() for q in parts]
    else:
        recs = [rec.decode().split()]
    per_rank_ms = [float(x[0]) for x in recs]
    shas = [x[1] for x in recs]
    out = {"curve": curve_name, "scaling": "weak (independent proofs)", "proofs_per_rank": k, "proofs_total": int(done), "constraints": 1 << args.log_n,
           "proofs_per_s": round(done / el, 4) if el > 0 else None, "wall_ms_per_round": round(el * 1e3 / max(k, 1), 2),
           "ms_per_proof_by_rank": [round(x, 2) for x in per_rank_ms], "same_proof_on_every_rank": len(set(shas)) == 1, "proof_sha": shas[0],
           "same_proof_as_sharded": (shas[0] == reference_sha) if reference_sha else None, "key_pin_s": round(pin_s, 1),
           "how": "every rank: whole key + window tables on its own GPU, %d proofs back to back from one caller; no data-path collective; proofs of all ranks / max-over-ranks time between two fences" % k}
    if err2 is not None or done != k * R.world:
        out["error"] = "a rank failed inside the timed loop: " + str(err2 or "on another rank")
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# PLONK (BASELINE config 5), the RCCL self-test, the CPU baseline
# ---------------------------------------------------------------------------------------------------------------------------------
def plonk_leg(R):
    """kernel work of one BN254 proof at 2^plonk_log_n gates -- 10 KZG-commit MSMs over a pinned SRS, the grand product and the
    quotient (computeNumerator + divideByZH) on the device; roofline by BASELINE.md's bytes (10 x 96 B x n for the
    commitments + 64 B per element and transform for the reference's 108 size-n transforms and the size-4n one)"""
    args = R.args
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_plonk_kernels
    p = bench_plonk_kernels.run(R.ctx, args.plonk_log_n, reps=2, reference_count=False)[0]
    n = 1 << args.plonk_log_n
    alg = 10 * 96 * n + 108 * 64 * n + 64 * 4 * n
    ms = p["ms_per_proof_kernels"]
    st = p["stages_ms"]
    acc_ms = st.get("msm_accumulate")
    ntt_ms = sum(v for k, v in st.items() if k.startswith("ntt_"))
    # what the device pipeline really transforms: 6 + 4 x 6 size-n transforms + the size-4n inverse with the circuit constants pinned
    ntt_real = (6 + 4 * 6) * 64 * n + 64 * 4 * n
    p["roofline"] = {"bound": "hbm", "algorithmic_bytes": alg, "achieved": round(alg / (ms * 1e-3) / 1e9, 2), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(alg / (ms * 1e-3) / 8e12, 5),
                     "how": "BASELINE.md's bytes for config 5 (10 MSMs x 96 B x n + 109 transforms x 64 B per element, the reference's transform count) over ms_per_proof_kernels",
                     "kernels": {"msm_accumulate": None if not acc_ms else {"ms_per_proof": acc_ms, "algorithmic_bytes": 10 * 96 * n,
                                                                           "frac": round(10 * 96 * n / (acc_ms * 1e-3) / 8e12, 5)},
                                 "ntt_pass": None if not ntt_ms else {"ms_per_proof": round(ntt_ms, 3), "algorithmic_bytes_reference_count": 108 * 64 * n + 64 * 4 * n,
                                                                      "frac_reference_count": round((108 * 64 * n + 64 * 4 * n) / (ntt_ms * 1e-3) / 8e12, 5),
                                                                      "algorithmic_bytes_issued": ntt_real, "frac_issued": round(ntt_real / (ntt_ms * 1e-3) / 8e12, 5)}}}
    if not args.no_check:   # the same device pipeline on a SATISFYING trace, checked by the CPU oracle (outside the timed region)
        _, pyref, checkers = oracle_modules()
        t_chk = time.perf_counter()
        try:
            p["identity_ok"] = bool(checkers.check_plonk_quotient_identity(R.ctx, pyref.BN254, args.plonk_log_n, nthreads=os.cpu_count() or 1, pinned=True))
        except AssertionError:
            p["identity_ok"] = False
        p["identity_check"] = ("h(zeta)(zeta^n-1) == gate + alpha*ordering + alpha^2(Z-1)L1 on a satisfying synthetic trace, "
                               "polynomials evaluated by the CPU oracle (oracle/checkers.py), %.1f s" % (time.perf_counter() - t_chk))
    return p


def nccl_selftest_worker():
    """child process of the N = 1 bench (and of tests/test_gpu_parity.py): a ONE-rank process group over the nccl (= RCCL) backend,
    every collective of gnark_amd/multigpu.py on device tensors, then a sharded MSM and a sharded Groth16 proof walked through the
    FULL collective schedule (force_collectives) and compared with the plain single-GPU proof.  Prints one JSON line."""
    import datetime
    import hashlib
    import torch
    import torch.distributed as dist
    res = {"ok": False}
    try:
        backend = os.environ.get("GA_SELFTEST_BACKEND", "nccl")
        logn = int(os.environ.get("GA_SELFTEST_LOGN", "16"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
        emu = os.environ.get("GA_BENCH_EMU", "0") == "1"
        dev = None
        if not emu:
            torch.cuda.set_device(0)
            dev = torch.device("cuda", 0)
        kw = dict(device_id=dev) if backend == "nccl" else {}
        dist.init_process_group(backend=backend, rank=0, world_size=1, timeout=datetime.timedelta(seconds=120), **kw)
        import gnark_amd
        from gnark_amd import _lib, ecc, groth16, multigpu, synth
        ctx = gnark_amd.Context(0, lib=_lib.Library(os.path.join(ROOT, "tests", "emu", "libgnark_amd_emu.so"))) if emu else gnark_amd.Context(0)
        coll_dev = dev if backend == "nccl" else None
        res["collectives"] = multigpu.collective_selftest(dist, dev)
        ok_a, _ = multigpu.agree(dist, True, None, coll_dev)
        res["agree"] = ok_a
        # a sharded MSM (one all_gather of a Jacobian partial) and a sharded proof through the whole schedule, against the plain paths
        n = 1 << logn
        inst = synth.make_instance(ctx, 0, logn, 0x5EED0055, want_dlogs=False)
        P, S = inst.key["A"], inst.solution.W[: inst.key["A"].shape[0]]
        table = ecc.PrecomputedBases(ctx, 0, _lib.G1, P)
        one = multigpu.msm_base_sharded(ctx, 0, _lib.G1, table, S, P.shape[0], None, None)
        gathered = multigpu._all_gather_u64(one, dist, dev)     # the exchange step of a sharded MSM: one Jacobian partial per rank
        res["msm_all_gather"] = bool(len(gathered) == 1 and np.array_equal(gathered[0], one)
                                     and np.array_equal(multigpu.combine_partials(0, _lib.G1, gathered, lib=ctx.lib), one))
        table.free()
        pk = inst.proving_key(ctx, precompute=1, shard=(0, 1))
        plain = groth16.Prove(pk, inst.solution, inst.nb_public, inst.r, inst.s)
        forced = multigpu.groth16_prove_sharded(pk, inst.solution, inst.nb_public, inst.r, inst.s, dist, dev, force_collectives=True)
        pk.FreeGPUResources()
        wpk = inst.proving_key(ctx, precompute=1, window_shard=(0, 1))
        try:   # (a one-share window partition is a plain key; the broadcast branch needs win_count > 1, covered by the collectives above)
            forced_w = multigpu.groth16_prove_sharded(wpk, inst.solution, inst.nb_public, inst.r, inst.s, dist, dev, force_collectives=True)
        finally:
            wpk.FreeGPUResources()
        res["sharded_proof_same_bytes"] = bool(plain.WriteTo() == forced.WriteTo() == forced_w.WriteTo())
        res["proof_sha"] = hashlib.sha256(forced.WriteTo()).hexdigest()[:16]
        res["constraints"] = n
        res["ok"] = bool(res["collectives"]["ok"] and res["agree"] and res["msm_all_gather"] and res["sharded_proof_same_bytes"])
        ctx.close()
        dist.destroy_process_group()
    except Exception as e:
        res["error"] = repr(e)[:400]
    print("NCCL_SELFTEST " + json.dumps(res))
    return 0 if res["ok"] else 1


def nccl_selftest_leg(R):
    """runs nccl_selftest_worker in a child process (its own process group; a hung RCCL cannot take the bench line with it)"""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env["GA_SELFTEST_BACKEND"] = "gloo" if R.emu else os.environ.get("GA_SELFTEST_BACKEND", "nccl")
    if R.emu:
        env["GA_SELFTEST_LOGN"] = "8"
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--nccl-selftest-worker"], capture_output=True, text=True, env=env,
                           timeout=int(os.environ.get("GA_SELFTEST_TIMEOUT_S", "300")), cwd=ROOT)
        lines = [l for l in r.stdout.splitlines() if l.startswith("NCCL_SELFTEST ")]
        if not lines:
            return "failed", {"error": "no result line; rc %d; %s" % (r.returncode, (r.stderr or "")[-300:])}
        d = json.loads(lines[-1][len("NCCL_SELFTEST "):])
        d["seconds"] = round(time.perf_counter() - t0, 1)
        return ("ok" if d.get("ok") else "failed"), d
    except subprocess.TimeoutExpired:
        return "failed", {"error": "timeout"}


def cpu_baseline_leg(R, out, msm_only=False):
    """the oracle's Pippenger on a bounded sample (rank 0): a plain-C port, NOT gnark-crypto.  `msm_only` (N > 1: the other ranks
    wait in the closing barrier meanwhile) leaves out the config-2 and Groth16 samples the N = 1 line carries."""
    from gnark_amd import _lib, ecc
    from gnark_amd.device import affine_words, curve_id
    ctx, lib, args = R.ctx, R.lib, R.args
    cid = curve_id(args.curve)
    words_aff = affine_words(cid, _lib.G1)
    oracle = oracle_modules()[0]
    eff_cores, logical_cpus, quota = effective_cores()
    sample_log = min(args.log_n, 24)
    sn = 1 << sample_log

    def cpu_vs_gpu(logn, simple=True):
        """the oracle's two CPU MSMs (msm_fast.c: batch-affine, all cores; oracle.c: the simple one, a thread per window) and the
        library's plain (un-pinned bases: no table) MSM on the same 2^logn inputs"""
        m = 1 << logn
        sb = ctx.malloc(m * words_aff * 8)
        lib.check(lib.ga_gen_bases(ctx.handle, cid, _lib.G1, 0x5EED0002, m, sb.ptr, None))
        P = sb.to_host((m, words_aff))
        ss = ctx.malloc(m * 32)
        lib.check(lib.ga_gen_scalars(ctx.handle, cid, 0x5EED0001, m, ss.ptr))
        S = ss.to_host((m, 4))
        t0 = time.perf_counter()
        ref = oracle.msm_fast(cid, P, S, nthreads=eff_cores)
        cpu = time.perf_counter() - t0
        cpu_simple, simple_same = None, None
        if simple:
            t0 = time.perf_counter()
            ref_simple = oracle.msm(cid, 0, P, S, nthreads=eff_cores)
            cpu_simple = time.perf_counter() - t0
            simple_same = bool(np.array_equal(oracle.jac_to_affine(cid, 0, ref), oracle.jac_to_affine(cid, 0, ref_simple)))
        gpu_res = ecc.MultiExp(ctx, cid, _lib.G1, sb, ss, n=m)   # (also the warm-up of the timed repetitions below)
        reps = 5 if logn >= 24 else 20
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):   # timed WITHOUT the stage profiler (its hipEvent pairs are ~1 % of a 2 ms call) ...
            ecc.MultiExp(ctx, cid, _lib.G1, sb, ss, n=m)
        ctx.sync()
        gpu = (time.perf_counter() - t0) / reps
        ctx.profile(True)      # ... and the stage durations from three more calls with it
        ctx.profile_reset()
        for _ in range(3):
            ecc.MultiExp(ctx, cid, _lib.G1, sb, ss, n=m)
        ctx.sync()
        st = stage_stats(ctx.profile_read())
        ctx.profile(False)
        same = bool(np.array_equal(oracle.jac_to_affine(cid, 0, ref), ecc.jac_to_affine(cid, _lib.G1, gpu_res))) and simple_same is not False
        sb.free()
        ss.free()
        return cpu, gpu, same, st, cpu_simple
    cpu_s, _, same, _, cpu_simple_s = cpu_vs_gpu(sample_log, simple=not msm_only)
    cbits, nwin_f, splits = oracle.msm_fast_plan(cid, sn, eff_cores)
    threads_used = min(eff_cores, nwin_f * splits)          # (window x point-range) tasks on a thread pool: every usable core works
    threads_simple = min(eff_cores, oracle.msm_windows(cid, sn))   # the simple port runs one thread per Pippenger window
    out["cpu_baseline"] = {"value": rate(sn / cpu_s / 1e6) if sn / cpu_s < 1e6 else round(sn / cpu_s / 1e6, 4), "unit": "Mscalar-mul/s", "cores": threads_used, "threads_used": threads_used,
                           "effective_cores": eff_cores, "logical_cpus": logical_cpus, "cgroup_cpu_quota": quota, "host_cores": logical_cpus, "kind": "port-batch-affine",
                           "value_simple": rate(sn / cpu_simple_s / 1e6) if cpu_simple_s else None, "cores_simple": threads_simple,
                           "speedup_over_simple": round(cpu_simple_s / cpu_s, 2) if cpu_simple_s else None,
                           "sample": "%s G1 MSM of 2^%d points, oracle/msm_fast.c: signed %d-bit digits, batch-affine buckets, %d windows x %d point ranges on %d threads, %.1f s%s" % (
                               args.curve.upper(), sample_log, cbits, nwin_f, splits, threads_used, cpu_s,
                               "; value_simple: oracle/oracle.c Pippenger, Jacobian buckets, one thread per window (%d), %.1f s" % (threads_simple, cpu_simple_s) if cpu_simple_s else ""),
                           "gpu_result_matches_oracle": same,
                           "note": "a plain-C restatement of the algorithm gnark-crypto's CPU MultiExp publishes (signed digits, batch-affine buckets behind one inversion per batch, conflict queue, tasks beyond the window count; 64-bit CIOS, no assembly), NOT gnark-crypto itself -- gnark cannot be built here (no Go toolchain), so the GPU/CPU ratio of this line is not a claim; the box shows %d logical CPUs but its cgroup grants %s of them" % (logical_cpus, "all" if quota is None else "%.0f" % quota)}
    # BASELINE config 2: G1 MSM over 2^20 random, UN-PINNED bases (ga_msm: bases converted per call, no table), GPU beside the CPU port
    if args.log_n >= 20 and not msm_only:
        try:
            c2_cpu, c2_gpu, c2_same, c2_st, c2_cpu_simple = cpu_vs_gpu(20)
            acc2 = c2_st.get("msm_accumulate", {}).get("avg_ms")
            alg2 = ALG_BYTES[(cid, 0)] * (1 << 20)
            out["config2_msm_2p20_unpinned"] = {
                "gpu_ms_per_msm": round(c2_gpu * 1e3, 3), "gpu_Mscalar_mul_per_s": round((1 << 20) / c2_gpu / 1e6, 2),
                "cpu_port_Mscalar_mul_per_s": round((1 << 20) / c2_cpu / 1e6, 4), "cpu_threads": eff_cores,
                "cpu_port_simple_Mscalar_mul_per_s": round((1 << 20) / c2_cpu_simple / 1e6, 4),
                "gpu_result_matches_oracle": c2_same,
                "roofline": {"bound": "hbm", "kernel": "msm_accumulate29_kernel (raw bases: one bucket set per window)", "avg_launch_ms": acc2,
                             "algorithmic_bytes_per_launch": alg2, "achieved": round(alg2 / (acc2 * 1e-3) / 1e9, 3) if acc2 else None, "peak": 8000.0, "unit": "GB/s",
                             "frac": round(alg2 / (acc2 * 1e-3) / 8e12, 6) if acc2 else None},
                "how": "ga_msm on device-resident raw affine bases and Montgomery scalars, 20 timed calls (stage profiler off), stage durations from 3 more; the CPU port on the same inputs"}
        except Exception as e:
            out["config2_msm_2p20_unpinned"] = {"error": repr(e)[:300]}
    # proofs/s for the same port: the oracle's Groth16 prover (7 FFTs + 4 G1 + 1 G2 MSM) on a bounded 2^20-constraint sample
    if args.groth16_proofs > 0 and not msm_only:
        try:
            glog = min(args.log_n, int(os.environ.get("GA_BENCH_CPU_G16_LOGN", "20")))
            from gnark_amd import groth16, synth
            ginst = synth.make_instance(ctx, cid, glog, 0x5EED0020, want_dlogs=False)
            gs = ginst.solution
            t0 = time.perf_counter()
            want = oracle.groth16_prove(cid, dict(ginst.key, n=ginst.n), gs.W, gs.A, gs.B, gs.C, ginst.nb_public, ginst.r, ginst.s, nthreads=eff_cores)
            g_s = time.perf_counter() - t0
            gpk = ginst.proving_key(ctx)
            gp = groth16.Prove(gpk, gs, ginst.nb_public, ginst.r, ginst.s)
            gpk.FreeGPUResources()
            g_same = bool(np.array_equal(gp.Ar, want[0]) and np.array_equal(gp.Bs, want[1]) and np.array_equal(gp.Krs, want[2]))
            out["cpu_baseline"]["groth16_2p%d_sample" % glog] = {"proofs_per_s": round(1.0 / g_s, 4), "constraints": 1 << glog, "kind": "port", "threads_used": eff_cores,
                                              "sample": "oracle/oracle.c Groth16 prover (its simple MSM), 2^%d constraints -- NOT the 2^%d of the GPU proofs beside it --, %d threads, %.1f s" % (glog, args.log_n, eff_cores, g_s),
                                              "gpu_proof_matches_oracle": g_same}
        except Exception as e:
            out["cpu_baseline"]["groth16_sample"] = {"error": repr(e)[:300]}


# ---------------------------------------------------------------------------------------------------------------------------------
def compact_line(out):
    """What goes to stdout: the driver keeps the LAST 8 KB of it, so the line carries numbers, not prose (the verbose object, with
    every leg's explanation, goes to --detail-file), stays under that size, and ENDS with "summary": every figure BASELINE's metric
    names, in one object."""
    def pick(o, *ks):
        return {k: o[k] for k in ks if isinstance(o, dict) and k in o}

    def stage_totals(st, field):
        return {k: (round(v[field], 3) if isinstance(v, dict) else v) for k, v in (st or {}).items()}

    def g16(q):
        if not isinstance(q, dict):
            return q
        if "error" in q:
            return pick(q, "curve", "partition", "error")
        r = pick(q, "curve", "partition", "scaling", "ms_per_proof", "proofs_per_s", "proofs", "constraints", "matches_dlog", "computeH_ms", "computeH_hbm_frac",
                 "hbm_frac_whole_proof", "proof_sha", "key_setup_s", "key_pin_s", "ms_per_proof_profiled_single_lane", "replicate_h_ms_per_proof", "one_shot_unpinned_ms")
        if isinstance(q.get("one_shot_unpinned"), dict):
            r["one_shot_unpinned"] = pick(q["one_shot_unpinned"], "sequential_ms", "upload_key_ms", "prove_ms", "free_ms", "same_proof_bytes", "fused_same_proof_bytes", "error")
        if isinstance(q.get("check"), dict):
            r["h_identity_ok"] = q["check"].get("h_identity_ok")
        if isinstance(q.get("pipelined"), dict):
            pl = q["pipelined"]
            r["two_callers"] = dict(pick(pl, "ms_per_proof", "proofs_per_s", "vs_single_caller", "same_proof_bytes", "proofs"),
                                    **{k: pl.get("lanes", {}).get(k) for k in ("lanes01_proofs", "lanes23_proofs", "queued_proofs")})
        if q.get("stages_ms"):
            r["stages_ms_per_proof"] = stage_totals(q["stages_ms"], "total_ms")
        return r

    line = pick(out, "metric", "value", "value_checked", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "error")
    cfg = dict(out.get("config", {}))
    line["config"] = cfg
    rf = dict(out.get("roofline") or {})
    rf.pop("int_mad_peak_source", None)
    if isinstance(rf.get("traffic_calibration"), str):
        rf["traffic_calibration"] = "as reported (x1): 64-B gathers, see DESIGN 6"
    if isinstance(rf.get("traffic_source"), str) and len(rf["traffic_source"]) > 150:
        rf["traffic_source"] = rf["traffic_source"][:150]
    line["roofline"] = rf
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        c2 = pick(cb, "value", "unit", "cores", "kind", "value_simple", "cores_simple", "speedup_over_simple", "effective_cores", "host_cores", "gpu_result_matches_oracle", "source", "error")
        if "sample" in cb:
            c2["sample"] = cb["sample"][:150]
        if "note" in cb:
            c2["note"] = "plain-C port of gnark-crypto's published algorithm (oracle/msm_fast.c), NOT gnark-crypto: the GPU/CPU ratio is not a claim"
        for k, v in cb.items():
            if k.startswith("groth16_") and isinstance(v, dict):
                c2[k] = pick(v, "proofs_per_s", "constraints", "threads_used", "gpu_proof_matches_oracle", "error")
        line["cpu_baseline"] = c2
    if out.get("stages_ms"):
        line["stages_ms"] = stage_totals(out["stages_ms"], "avg_ms")
    for k in ("plain_msm_no_tables", "msm_with_scalar_h2d"):
        if k in out:
            line[k] = pick(out[k], "ms_per_msm", "Mscalar_mul_per_s", "sort_ms", "same_result", "error")
    c2 = out.get("config2_msm_2p20_unpinned")
    if isinstance(c2, dict):
        line["config2_msm_2p20_unpinned"] = dict(pick(c2, "gpu_ms_per_msm", "gpu_Mscalar_mul_per_s", "cpu_port_Mscalar_mul_per_s", "cpu_threads", "gpu_result_matches_oracle", "error"),
                                                 **{("roofline_" + k): v for k, v in pick(c2.get("roofline", {}), "frac", "avg_launch_ms").items()})
    if "groth16" in out:
        line["groth16"] = g16(out["groth16"])
    p = out.get("plonk")
    if isinstance(p, dict):
        line["plonk"] = dict(pick(p, "ms_per_proof_kernels", "msm_ms", "ntt_ms", "quotient_and_z_ms", "identity_ok", "log_n", "error"),
                             roofline=pick(p.get("roofline", {}), "bound", "frac", "achieved", "peak", "unit", "algorithmic_bytes"), stages_ms=p.get("stages_ms"))
    mb = out.get("msm_bls12_381")
    if isinstance(mb, dict):
        line["msm_bls12_381"] = {g: (dict(pick(v, "ms_per_msm", "Mscalar_mul_per_s", "value_checked", "window_bits", "windows", "table_GiB", "error"),
                                          **{("roofline_" + k): x for k, x in pick(v.get("roofline", {}), "frac", "avg_launch_ms", "int_mad_frac").items()})
                                     if isinstance(v, dict) else v) for g, v in mb.items()}
    gb = out.get("groth16_bls12_381")
    if isinstance(gb, dict):
        line["groth16_bls12_381"] = {k: g16(v) for k, v in gb.items()} if ("window" in gb or "range" in gb) else g16(gb)
    for k in ("groth16_window", "groth16_range"):   # the BN254 sharded proof in the partition --partition did not pick
        if isinstance(out.get(k), dict):
            line[k] = g16(out[k])
    for k in ("weak_msm",):
        if k in out:
            line[k] = pick(out[k], "value", "unit", "scaling", "ms_per_step", "value_checked", "pairs_per_gpu", "error")
    if "replicas" in out:
        line["replicas"] = {k: v for k, v in out["replicas"].items() if k != "how"} if isinstance(out["replicas"], dict) else out["replicas"]
    for k in ("backend", "world_size", "ranks", "nccl_selftest"):
        if k in out:
            line[k] = out[k]
    sd = out.get("nccl_selftest_detail")
    if isinstance(sd, dict):
        line["nccl_selftest_detail"] = dict(pick(sd, "ok", "agree", "msm_all_gather", "sharded_proof_same_bytes", "seconds", "error"),
                                            collectives_ok=(sd.get("collectives") or {}).get("ok"))
    if "pmc_passes" in out:
        line["pmc_passes"] = out["pmc_passes"]
    line["legs_seconds"] = out.get("legs_seconds")
    line["total_seconds"] = out.get("total_seconds")
    # ---- the whole metric in one object, LAST
    g, gbl = out.get("groth16") or {}, out.get("groth16_bls12_381") or {}
    sm = {"msm_Mscalar_mul_per_s": out.get("value"), "msm_ms": out.get("ms_per_step"), "msm_checked": out.get("value_checked"), "n_gpus": out.get("n_gpus"),
          "roofline_frac_hbm": rf.get("frac"), "int_mad_frac": rf.get("int_mad_frac"), "traffic_over_algorithmic": rf.get("traffic_over_algorithmic"),
          "groth16_bn254_ms_per_proof": g.get("ms_per_proof"), "groth16_bn254_proofs_per_s": g.get("proofs_per_s"), "groth16_bn254_matches_dlog": g.get("matches_dlog"),
          "groth16_bn254_computeH_ms": g.get("computeH_ms"), "groth16_bn254_one_shot_unpinned_ms": g.get("one_shot_unpinned_ms")}
    if isinstance(g.get("pipelined"), dict):
        sm["groth16_bn254_two_callers_ms_per_proof"] = g["pipelined"].get("ms_per_proof")
        sm["groth16_bn254_two_callers_vs_single"] = g["pipelined"].get("vs_single_caller")
    if "partition" in g:
        sm["groth16_bn254_partition"] = g.get("partition")
    if "window" in gbl or "range" in gbl:
        for part in ("window", "range"):
            if isinstance(gbl.get(part), dict):
                sm["groth16_bls12_381_%s_ms_per_proof" % part] = gbl[part].get("ms_per_proof")
                sm["groth16_bls12_381_%s_matches_dlog" % part] = gbl[part].get("matches_dlog")
    elif gbl:
        sm["groth16_bls12_381_ms_per_proof"] = gbl.get("ms_per_proof")
        sm["groth16_bls12_381_matches_dlog"] = gbl.get("matches_dlog")
    for k in ("groth16_window", "groth16_range"):
        if isinstance(out.get(k), dict):
            sm["groth16_bn254_%s_ms_per_proof" % k.split("_")[1]] = out[k].get("ms_per_proof")
    if isinstance(out.get("replicas"), dict):
        sm["replicas_proofs_per_s"] = out["replicas"].get("proofs_per_s")
        sm["replicas_ms_per_proof_by_rank"] = out["replicas"].get("ms_per_proof_by_rank")
    if isinstance(out.get("weak_msm"), dict):
        sm["weak_msm_Mscalar_mul_per_s"] = out["weak_msm"].get("value")
    if isinstance(out.get("plonk"), dict):
        sm["plonk_bn254_2p22_ms"] = out["plonk"].get("ms_per_proof_kernels")
        sm["plonk_identity_ok"] = out["plonk"].get("identity_ok")
    for k, f in (("plain_msm_no_tables", "ms_per_msm"), ("msm_with_scalar_h2d", "ms_per_msm"), ("config2_msm_2p20_unpinned", "gpu_ms_per_msm")):
        if isinstance(out.get(k), dict):
            sm[k + "_ms"] = out[k].get(f)
    if isinstance(mb, dict):
        for grp in ("g1", "g2"):
            if isinstance(mb.get(grp), dict):
                sm["msm_bls12_381_%s_ms" % grp] = mb[grp].get("ms_per_msm")
    if isinstance(cb, dict):
        sm["cpu_port_Mscalar_mul_per_s"] = cb.get("value")
        sm["cpu_port_simple_Mscalar_mul_per_s"] = cb.get("value_simple")
    if "backend" in out:
        sm["backend"] = out["backend"]
    line["summary"] = {k: v for k, v in sm.items() if v is not None}
    return line


def error_line(args, text):
    """the one-line JSON a run that must not be mistaken for a measurement prints (rank 0 / the launcher only) before it exits non-zero"""
    return json.dumps({"metric": "G1 MSM throughput, %s, 2^%d scalar-muls" % (args.curve.upper(), args.log_n), "value": None, "unit": "Mscalar-mul/s",
                       "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "error": text}, separators=(",", ":"))


def world_check(args):
    """`--gpus N` is the number of ranks this line is ABOUT: a launcher's WORLD_SIZE that disagrees with it, or (backend nccl = RCCL,
    one rank per GPU) fewer visible GPUs than ranks, is an error -- never a silent one-GPU measurement.  Returns the error text or None."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None and int(ws) != args.gpus:
        return "--gpus %d but the launcher set WORLD_SIZE=%s: refusing to measure a world other than the one asked for" % (args.gpus, ws)
    emu = os.environ.get("GA_BENCH_EMU", "0") == "1"
    backend = os.environ.get("GA_BENCH_BACKEND", "gloo" if emu else "nccl")
    if args.gpus > 1 and backend == "nccl" and not emu:
        import torch
        ndev = torch.cuda.device_count()
        if args.gpus > ndev:
            return ("--gpus %d with backend nccl (RCCL: one rank per GPU) but this node shows %d GPU(s); GA_BENCH_BACKEND=gloo runs the "
                    "N > 1 code path with ranks sharing a GPU (a functional run, not a scaling figure)" % (args.gpus, ndev))
    return None


def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1) WITHOUT a launcher: start the N ranks ourselves -- `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` over this same script and arguments -- and pass rank 0's JSON
    line through as the only thing on stdout (everything else a child prints there, e.g. gloo's connection banner, goes to stderr)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench: --gpus %d without a launcher: %s\n" % (args.gpus, " ".join(cmd[1:10])))
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=os.getcwd())
    lines = 0
    for ln in p.stdout:
        if ln.startswith("{"):
            sys.stdout.write(ln)
            sys.stdout.flush()
            lines += 1
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if lines == 0:
        print(error_line(args, "the %d ranks launched for --gpus %d exited with code %d without printing a line" % (args.gpus, args.gpus, rc)))
        return rc or 1
    return rc


def main():
    if "--nccl-selftest-worker" in sys.argv:
        sys.exit(nccl_selftest_worker())
    args = parse()
    bad = world_check(args)
    if bad is not None:
        if int(os.environ.get("RANK", "0")) == 0:
            print(error_line(args, bad))
        sys.stderr.write("bench: " + bad + "\n")
        sys.exit(2)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    # This SCRIPT's cyclic garbage collector stays out of the timed regions: with the 15 GB of numpy arrays a 2^24 instance and its
    # discrete logs hold, one generation-2 collection landed inside a timed proof and cost it 35 ms (130 -> 165 ms, one proof in
    # five: profiles/r06_d_gc_hiccup.txt).  Reference counting still frees everything the legs drop; GA_BENCH_GC=1 leaves it on.
    if os.environ.get("GA_BENCH_GC", "0") == "0":
        import gc
        gc.collect()
        gc.disable()
    R = Run(args)
    from gnark_amd.device import curve_id
    cid = curve_id(args.curve)
    rank, world = R.rank, R.world
    t_all = time.perf_counter()
    legs_s = {}

    def leg(name, fn, *a):
        """a secondary leg never loses the headline line: errors become {"error": ...} (single-rank legs only; multi-rank legs keep
        their own agree() discipline and return their error objects themselves)"""
        t0 = time.perf_counter()
        try:
            v = fn(*a)
        except Exception as e:
            v = {"error": repr(e)[:300]}
        legs_s[name] = round(time.perf_counter() - t0, 1)
        return v

    def timed(name, fn, *a):
        t0 = time.perf_counter()
        v = fn(*a)
        legs_s[name] = round(time.perf_counter() - t0, 1)
        return v

    ranks = R.gather_identities()   # (every rank; before any leg: what the collective backend really connected)
    t0 = time.perf_counter()
    out, kept = headline_leg(R)
    legs_s["headline_msm"] = round(time.perf_counter() - t0, 1)
    if world > 1:
        out["backend"] = R.backend
        out["world_size"] = R.dist.get_world_size()
        out["ranks"] = ranks
    if kept is not None:
        keep, res = kept
        if rank == 0 and world == 1 and not args.only_headline:
            try:
                single_gpu_msm_extras(R, out, keep, res)
            except Exception as e:
                out["plain_msm_no_tables"] = {"error": repr(e)[:300]}
        free_keep(keep)

    if args.only_headline:
        pass
    elif world == 1:
        if args.groth16_proofs > 0:
            out["groth16"] = leg("groth16", groth16_single_gpu_leg, R, cid, args.curve)
        if args.plonk_log_n > 0 and cid == 0:
            out["plonk"] = leg("plonk", plonk_leg, R)
        if cid == 0 and not args.no_bls and os.environ.get("GA_BENCH_BLS", "1") != "0":   # BASELINE config 4's curve under the same clock
            out["msm_bls12_381"] = leg("msm_bls12_381", bls_msm_leg, R)
            if args.groth16_proofs > 0:
                out["groth16_bls12_381"] = leg("groth16_bls12_381", groth16_single_gpu_leg, R, 1, "bls12-381")
        if not args.no_selftest:
            t0 = time.perf_counter()
            out["nccl_selftest"], out["nccl_selftest_detail"] = nccl_selftest_leg(R)
            legs_s["nccl_selftest"] = round(time.perf_counter() - t0, 1)
    else:
        # Every leg below is entered by EVERY rank in the same order (each keeps the local-step -> agree() -> collectives discipline,
        # so a failing rank turns a leg into an error object on all ranks and the next leg still runs).
        wk = timed("weak_msm", weak_msm_leg, R)
        g16 = g16_w = g16_bls = rep = None
        if args.groth16_proofs > 0 and os.environ.get("GA_BENCH_SHARDED_G16", "1") != "0":
            # BASELINE config 3 over N GPUs: ONE proof, key sharded by base-point range (--partition picks the other one for this leg)
            g16 = timed("groth16", groth16_sharded_leg, R, cid, args.curve, args.partition)
            if os.environ.get("GA_BENCH_BOTH_PARTITIONS", "1") != "0":
                other = "window" if args.partition == "range" else "range"
                g16_w = timed("groth16_" + other, groth16_sharded_leg, R, cid, args.curve, other)
            # BASELINE config 4 (Groth16 BLS12-381, 2^24, MSM window-sharded over the GPUs) at EVERY world > 1, in both partitions:
            # "window" is BASELINE's wording, "range" the partition that balances better
            if cid == 0 and os.environ.get("GA_BENCH_CONFIG4", "1") != "0":
                g16_bls = {}
                for part in ("window", "range"):
                    g16_bls[part] = timed("groth16_bls12_381_" + part, groth16_sharded_leg, R, curve_id("bls12-381"), "bls12-381", part)
        if args.replica_proofs > 0 and os.environ.get("GA_BENCH_REPLICAS", "1") != "0":
            ref_sha = g16.get("proof_sha") if isinstance(g16, dict) else None
            rep = timed("replicas", replicas_leg, R, cid, args.curve, ref_sha)
        if rank == 0:
            out["weak_msm"] = wk
            if g16 is not None:
                out["groth16"] = g16
            if g16_w is not None:
                out["groth16_window" if g16_w.get("partition", "window") == "window" else "groth16_range"] = g16_w
            if g16_bls is not None:
                out["groth16_bls12_381"] = g16_bls
            if rep is not None:
                out["replicas"] = rep

    # the CPU port: rank 0, a local leg (no collectives inside).  N > 1 times the MSM sample alone -- the same figure the N = 1 line
    # carries, so that every line of a scaling record holds its own CPU number -- while the other ranks wait in the barrier below
    # (outside every timed region); GA_BENCH_CPU_BASELINE_JSON hands in a figure measured elsewhere on this box instead.
    if rank == 0 and not args.no_cpu_baseline and not args.only_headline:
        t0 = time.perf_counter()
        given = os.environ.get("GA_BENCH_CPU_BASELINE_JSON", "")
        try:
            if given:
                out["cpu_baseline"] = dict(json.loads(open(given).read() if os.path.exists(given) else given), source="GA_BENCH_CPU_BASELINE_JSON")
            else:
                cpu_baseline_leg(R, out, msm_only=world > 1)
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)[:300]}
        legs_s["cpu_baseline"] = round(time.perf_counter() - t0, 1)
    if world > 1:
        R.dist.barrier()
    if rank == 0:
        out["legs_seconds"] = legs_s
        out["total_seconds"] = round(time.perf_counter() - t_all, 1)
        if args.detail_file:
            try:
                with open(args.detail_file, "w") as f:
                    json.dump(out, f)
            except OSError as e:
                sys.stderr.write("bench: could not write %s: %r\n" % (args.detail_file, e))
        print(json.dumps(compact_line(out), separators=(",", ":")))
    R.ctx.close()
    if world > 1:
        R.dist.destroy_process_group()


if __name__ == "__main__":
    main()
