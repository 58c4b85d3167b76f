"""bench.py's contract, exercised against the CPU emulation build (GA_BENCH_EMU=1: tiny sizes, the JSON line says "data": "emulation"
-- a dry run of the script's control flow, never a measurement): one JSON line with the fields the driver reads, the self-checks of
the Groth16 / PLONK / BLS12-381 legs, the one-rank collective self-test, and the N > 1 path (two gloo ranks: the strong-scaling
headline -- ONE MSM problem sharded over the ranks --, the weak-scaling figure beside it, the sharded proof), and what happens when a
rank fails: every rank skips the leg and the line carries the error -- in seconds, not after a collective timeout."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def env(emu_lib):
    e = dict(os.environ)
    e["GA_BENCH_EMU"] = "1"
    return e


def test_single_rank_line(env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--log-n", "8", "--steps", "2", "--warmup", "1", "--groth16-proofs", "2",
                        "--plonk-log-n", "6"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["data"] == "emulation" and d["vs_baseline"] is None
    assert d["value"] > 0 and d["value_checked"] is True and d["scaling"] == "strong"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5 and "integer_multiplier" in rf
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["gpu_result_matches_oracle"] is True and cb["host_cores"] >= cb["threads_used"] >= 1
    assert "NOT gnark-crypto" in cb["note"]
    assert cb["groth16"]["gpu_proof_matches_oracle"] is True
    g = d["groth16"]
    assert g["matches_dlog"] is True and g["check"]["h_identity_ok"] is True and g["proofs"] == 2
    assert g["pipelined"]["same_proof_bytes"] is True and g["pipelined"]["host_threads"] == 2
    assert d["plonk"]["identity_ok"] is True
    pr = d["plonk"]["roofline"]                                    # BASELINE config 5 carries its own roofline object
    assert pr["bound"] == "hbm" and pr["algorithmic_bytes"] == (10 * 96 + 108 * 64 + 4 * 64) * 64 and pr["frac"] >= 0 and pr["peak"] == 8000.0
    assert pr["kernels"]["msm_accumulate"]["ms_per_proof"] > 0 and pr["kernels"]["ntt_pass"]["ms_per_proof"] > 0   # (fractions round to 0 under the emulation)
    assert d["msm_with_scalar_h2d"]["same_result"] is True and d["msm_with_scalar_h2d"]["ms_per_msm"] > 0
    # BASELINE config 4's curve under the same clock: the BLS12-381 proof and the G1 / G2 MSMs with their own roofline objects
    gb = d["groth16_bls12_381"]
    assert gb["curve"] == "bls12-381" and gb["matches_dlog"] is True and gb["pipelined"]["same_proof_bytes"] is True
    mb = d["msm_bls12_381"]
    assert mb["g1"]["value_checked"] is True and mb["g2"]["value_checked"] is True
    assert mb["g1"]["roofline"]["algorithmic_bytes_per_launch"] == 128.0 * 256 and mb["g2"]["roofline"]["algorithmic_bytes_per_launch"] == 224.0 * 256
    assert "integer_multiplier" in mb["g1"]["roofline"]
    # the one-rank collective self-test (gloo under the emulation, nccl = RCCL on the GPU box): every collective + a sharded proof
    assert d["nccl_selftest"] == "ok", d.get("nccl_selftest_detail")
    st = d["nccl_selftest_detail"]
    assert st["sharded_proof_same_bytes"] is True and st["msm_all_gather"] is True
    assert all(st["collectives"][k] is True for k in ("all_gather", "gather", "scatter", "broadcast", "all_reduce"))


def _two_ranks(env, port, extra_env=None, timeout=900):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log-n", "9", "--steps", "1", "--warmup", "1", "--groth16-proofs", "1"]
    e = dict(env, **(extra_env or {}))
    import time
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=e, cwd=ROOT, timeout=timeout)
    return r, time.time() - t0


def test_two_ranks_line(env):
    r, _ = _two_ranks(env, 29761)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    # the headline is ONE 2^9-pair MSM sharded over the two ranks (strong scaling: BASELINE quotes a fixed problem at 1/2/4/8 GPUs)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["value_checked"] is True
    assert "sharded" in d["config"]["workload"] and "x2" in d["config"]["parallelism"]
    rf = d["roofline"]
    assert rf["algorithmic_bytes_per_launch"] == 96.0 * 256 and rf["rank"] == 0 and "integer_multiplier" in rf   # rank 0's 2^9 / 2 pairs
    w = d["weak_msm"]
    assert w["scaling"] == "weak" and w["value"] > 0 and w["value_checked"] is True and w["pairs_per_gpu"] == 512
    g = d["groth16"]
    assert "error" not in g, g
    assert g["scaling"] == "strong" and g["constraints"] == 512 and len(g["proof_sha"]) == 16
    assert g["matches_dlog"] is True and g["check"]["h_identity_ok"] is True   # rank 0 checked the sharded proof by the key's known dlogs
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["gpu_result_matches_oracle"] is True   # rank 0 carries it at N > 1 too


@pytest.mark.parametrize("fault,where", [
    ({"GA_BENCH_FAIL_RANK": "1", "GA_BENCH_FAIL_AT": "pin"}, "key pinning failed"),
    ({"GA_BENCH_FAIL_RANK": "1", "GA_BENCH_FAIL_AT": "warmup_local"}, "warm-up proof failed"),
    ({"GA_MGPU_FAULT": "1:witness"}, "warm-up proof failed"),        # inside the proof: the fixed collective schedule carries the error out
    ({"GA_MGPU_FAULT": "0:h_side"}, "warm-up proof failed"),         # on the helper thread of rank 0
    ({"GA_MGPU_FAULT": "1:z"}, "warm-up proof failed"),
])
def test_a_failing_rank_ends_the_leg_on_every_rank(env, fault, where):
    """no rank may enter a collective another rank will not reach: rank 1 (or 0) fails, BOTH ranks skip the sharded-proof leg, the line
    is printed with the error text and the other legs intact -- within seconds, not after the process group's timeout"""
    r, secs = _two_ranks(env, 29771, dict(fault, GA_BENCH_COLLECTIVE_TIMEOUT_S="240"), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert secs < 200, "took %.0f s: a rank was left waiting in a collective" % secs
    d = _line(r.stdout)
    assert d["value"] > 0 and d["value_checked"] is True and d["weak_msm"]["value_checked"] is True
    assert where in d["groth16"]["error"] and "injected fault" in d["groth16"]["error"], d["groth16"]


def test_a_failing_rank_in_the_headline_leg(env):
    r, secs = _two_ranks(env, 29781, {"GA_BENCH_FAIL_RANK": "1", "GA_BENCH_FAIL_AT": "headline", "GA_BENCH_COLLECTIVE_TIMEOUT_S": "240"}, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert secs < 200
    d = _line(r.stdout)
    assert d["value"] is None and "rank 1" in d["error"] and "injected fault" in d["error"]
    assert d["weak_msm"]["value_checked"] is True and d["groth16"]["matches_dlog"] is True   # the other legs still ran
