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


def _check_compact(line_text, d):
    """what every printed line must be: one compact object under the driver's 8 KB tail, the whole metric in "summary" at its end"""
    assert len(line_text) < 8000, len(line_text)
    assert list(d)[-1] == "summary" and len(json.dumps(d["summary"])) < 1500
    return d["summary"]


def test_single_rank_line(env, tmp_path):
    detail = str(tmp_path / "detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--log-n", "8", "--steps", "2", "--warmup", "1", "--groth16-proofs", "2",
                        "--plonk-log-n", "6", "--detail-file", detail], capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _line(r.stdout)
    sm = _check_compact([l for l in r.stdout.splitlines() if l.startswith("{")][0], line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "summary"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["data"] == "emulation" and line["vs_baseline"] is None
    assert line["value"] > 0 and line["value_checked"] is True and line["scaling"] == "strong"
    assert "workload" in line["config"] and "model" not in line["config"]
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5
    # the binding resource sits INSIDE the roofline object as flat scalars (nested objects do not survive the driver's parsing)
    assert "integer issue" in rf["bound_actual"] and rf["int_mad_per_addition"] == 1467 and rf["int_mad_frac"] >= 0 and rf["int_mad_peak_T_per_s"] > 30
    assert "traffic" in rf and "traffic_source" in rf
    cb = line["cpu_baseline"]
    # the CPU figure is the batch-affine port on every usable core; the simple one-thread-per-window port stays beside it
    assert cb["kind"] == "port-batch-affine" and cb["gpu_result_matches_oracle"] is True and cb["host_cores"] >= cb["cores"] >= 1
    assert cb["value"] > 0 and cb["value_simple"] > 0 and cb["speedup_over_simple"] > 0
    assert "NOT gnark-crypto" in cb["note"] and cb["groth16_2p8_sample"]["gpu_proof_matches_oracle"] is True and cb["groth16_2p8_sample"]["constraints"] == 256
    g = line["groth16"]
    assert g["matches_dlog"] is True and g["h_identity_ok"] is True and g["proofs"] == 2 and g["two_callers"]["same_proof_bytes"] is True
    # the drop-in default (key uploaded as plain vectors, one proof, freed) beside the pinned headline
    assert g["one_shot_unpinned_ms"] > 0 and g["one_shot_unpinned"]["same_proof_bytes"] is True and sm["groth16_bn254_one_shot_unpinned_ms"] == g["one_shot_unpinned_ms"]
    assert g["one_shot_unpinned"]["fused_same_proof_bytes"] is True and g["one_shot_unpinned"]["sequential_ms"] > 0
    assert line["plonk"]["identity_ok"] is True and line["plonk"]["roofline"]["bound"] == "hbm"
    assert line["msm_with_scalar_h2d"]["same_result"] is True and line["msm_with_scalar_h2d"]["ms_per_msm"] > 0
    gb = line["groth16_bls12_381"]
    assert gb["curve"] == "bls12-381" and gb["matches_dlog"] is True and gb["two_callers"]["same_proof_bytes"] is True
    mb = line["msm_bls12_381"]
    assert mb["g1"]["value_checked"] is True and mb["g2"]["value_checked"] is True and "roofline_int_mad_frac" in mb["g1"]
    assert line["nccl_selftest"] == "ok", line.get("nccl_selftest_detail")
    # the whole metric in the last object of the line: the BN254 proof figures the round-4 record lost are in the driver's tail
    assert sm["msm_Mscalar_mul_per_s"] == line["value"] and sm["groth16_bn254_ms_per_proof"] == g["ms_per_proof"] and sm["groth16_bn254_matches_dlog"] is True
    for k in ("groth16_bn254_proofs_per_s", "groth16_bn254_two_callers_ms_per_proof", "groth16_bn254_two_callers_vs_single", "groth16_bn254_computeH_ms",
              "groth16_bls12_381_ms_per_proof", "plonk_bn254_2p22_ms", "plain_msm_no_tables_ms", "msm_with_scalar_h2d_ms", "int_mad_frac", "roofline_frac_hbm"):
        assert k in sm, k
    # ---- the verbose object (--detail-file): every leg with its prose and nested objects
    d = json.load(open(detail))
    assert d["value"] == line["value"] and d["groth16"]["check"]["h_identity_ok"] is True and d["groth16"]["pipelined"]["host_threads"] == 2
    pr = d["plonk"]["roofline"]                                    # BASELINE config 5 carries its own roofline object
    assert pr["bound"] == "hbm" and pr["algorithmic_bytes"] == (10 * 96 + 108 * 64 + 4 * 64) * 64 and pr["frac"] >= 0 and pr["peak"] == 8000.0
    assert pr["kernels"]["msm_accumulate"]["ms_per_proof"] > 0 and pr["kernels"]["ntt_pass"]["ms_per_proof"] > 0   # (fractions round to 0 under the emulation)
    mbd = d["msm_bls12_381"]
    assert mbd["g1"]["roofline"]["algorithmic_bytes_per_launch"] == 128.0 * 256 and mbd["g2"]["roofline"]["algorithmic_bytes_per_launch"] == 224.0 * 256
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
    sm = _check_compact([l for l in r.stdout.splitlines() if l.startswith("{")][0], d)
    # the headline is ONE 2^9-pair MSM sharded over the two ranks (strong scaling: BASELINE quotes a fixed problem at 1/2/4/8 GPUs)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["value_checked"] is True
    assert "sharded" in d["config"]["workload"] and "x2" in d["config"]["parallelism"]
    # the truth about the collective: the backend that was really initialised (gloo here -- the line must not say RCCL), the world
    # size the process group reports, and every rank's device identity gathered THROUGH that backend
    assert d["backend"] == "gloo" and d["config"]["backend"] == "gloo" and "gloo" in d["config"]["parallelism"] and "RCCL" not in d["config"]["parallelism"]
    assert d["world_size"] == 2 and len(d["ranks"]) == 2 and all(isinstance(x, str) and x for x in d["ranks"]) and d["ranks"][0] != d["ranks"][1]
    rf = d["roofline"]
    assert rf["algorithmic_bytes_per_launch"] == 96.0 * 256 and rf["rank"] == 0 and "int_mad_frac" in rf   # rank 0's 2^9 / 2 pairs
    w = d["weak_msm"]
    assert w["scaling"] == "weak" and w["value"] > 0 and w["value_checked"] is True and w["pairs_per_gpu"] == 512
    g = d["groth16"]
    assert "error" not in g, g
    assert g["scaling"] == "strong" and g["constraints"] == 512 and len(g["proof_sha"]) == 16 and g["partition"] == "range"
    assert g["matches_dlog"] is True and g["h_identity_ok"] is True   # rank 0 checked the sharded proof by the key's known dlogs
    gw = d["groth16_window"]                                           # the same proof in the other partition: same bytes
    assert gw["partition"] == "window" and gw["matches_dlog"] is True and gw["proof_sha"] == g["proof_sha"]
    # BASELINE config 4 at EVERY world > 1, in both partitions
    gb = d["groth16_bls12_381"]
    assert set(gb) == {"window", "range"}
    for part in ("window", "range"):
        assert gb[part]["curve"] == "bls12-381" and gb[part]["partition"] == part and gb[part]["matches_dlog"] is True, gb[part]
    assert gb["window"]["proof_sha"] == gb["range"]["proof_sha"]
    # the throughput leg: every rank the whole key, independent proofs, counts all_reduced between two fences
    rp = d["replicas"]
    assert "error" not in rp, rp
    assert rp["proofs_total"] == 2 * rp["proofs_per_rank"] and rp["proofs_per_s"] > 0 and len(rp["ms_per_proof_by_rank"]) == 2
    assert rp["same_proof_on_every_rank"] is True and rp["same_proof_as_sharded"] is True
    # every line of a scaling record carries its own CPU figure (rank 0 times the MSM sample while the others wait in the last barrier)
    assert d["cpu_baseline"]["kind"] == "port-batch-affine" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["gpu_result_matches_oracle"] is True
    assert sm["cpu_port_Mscalar_mul_per_s"] == d["cpu_baseline"]["value"]
    for k in ("groth16_bn254_ms_per_proof", "groth16_bn254_window_ms_per_proof", "groth16_bls12_381_window_ms_per_proof", "groth16_bls12_381_range_ms_per_proof",
              "replicas_proofs_per_s", "weak_msm_Mscalar_mul_per_s", "backend"):
        assert k in sm, k


def test_plain_gpus_2_launches_two_ranks(env):
    """the driver's BENCH command shape -- `python bench.py --gpus 2 ...`, no launcher, no WORLD_SIZE -- starts the two ranks itself
    (round 5: --gpus was parsed and never read, so this command measured ONE rank and said n_gpus 1)"""
    e = {k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(GA_BENCH_BOTH_PARTITIONS="0", GA_BENCH_CONFIG4="0", GA_BENCH_REPLICAS="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log-n", "9", "--steps", "1", "--warmup", "1", "--groth16-proofs", "1"],
                       capture_output=True, text=True, env=e, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert [l for l in r.stdout.splitlines() if l.strip()] == [l for l in r.stdout.splitlines() if l.startswith("{")]   # rank 0's line is ALL of stdout
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and len(d["ranks"]) == 2 and d["ranks"][0] != d["ranks"][1]
    assert d["value"] > 0 and d["value_checked"] is True and "sharded" in d["config"]["workload"]
    assert d["groth16"]["matches_dlog"] is True and d["cpu_baseline"]["value"] > 0
    assert "torch.distributed.run" in r.stderr


def test_world_size_mismatch_is_an_error(env):
    """--gpus is the world the line is ABOUT: a launcher world that disagrees is refused, loudly, with a one-line JSON error and a
    non-zero exit code -- never a silent measurement of another world"""
    for gpus, ws in (("2", "1"), ("1", "2"), ("4", "2")):
        e = dict(env, WORLD_SIZE=ws, RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", gpus, "--log-n", "8"], capture_output=True, text=True, env=e, cwd=ROOT, timeout=120)
        assert r.returncode != 0
        d = _line(r.stdout)
        assert d["value"] is None and "WORLD_SIZE=" + ws in d["error"] and d["n_gpus"] == int(gpus)
    # a rank other than 0 exits the same way without printing a line
    e = dict(env, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, env=e, cwd=ROOT, timeout=120)
    assert r.returncode != 0 and r.stdout.strip() == ""


def test_more_ranks_than_gpus_over_rccl_is_an_error():
    """backend nccl (= RCCL) is one rank per GPU: --gpus 2 on a node with fewer GPUs (this container has none) must not fall back
    to ranks sharing a device and call it a 2-GPU figure; GA_BENCH_BACKEND=gloo is the explicit functional mode"""
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GA_BENCH_EMU", "GA_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=e, cwd=ROOT, timeout=300)
    assert r.returncode != 0
    d = _line(r.stdout)
    assert d["value"] is None and d["n_gpus"] == 2 and "GPU(s)" in d["error"] and "GA_BENCH_BACKEND=gloo" in d["error"]


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
    r, secs = _two_ranks(env, 29771, dict(fault, GA_BENCH_COLLECTIVE_TIMEOUT_S="240", GA_BENCH_BOTH_PARTITIONS="0", GA_BENCH_CONFIG4="0", GA_BENCH_REPLICAS="0"), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert secs < 200, "took %.0f s: a rank was left waiting in a collective" % secs
    d = _line(r.stdout)
    assert d["value"] > 0 and d["value_checked"] is True and d["weak_msm"]["value_checked"] is True
    assert where in d["groth16"]["error"] and "injected fault" in d["groth16"]["error"], d["groth16"]


def test_a_failing_rank_in_the_headline_leg(env):
    r, secs = _two_ranks(env, 29781, {"GA_BENCH_FAIL_RANK": "1", "GA_BENCH_FAIL_AT": "headline", "GA_BENCH_COLLECTIVE_TIMEOUT_S": "240", "GA_BENCH_CONFIG4": "0"}, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert secs < 200
    d = _line(r.stdout)
    assert d["value"] is None and "rank 1" in d["error"] and "injected fault" in d["error"]
    assert d["weak_msm"]["value_checked"] is True and d["groth16"]["matches_dlog"] is True   # the other legs still ran


def test_a_failing_rank_in_the_replica_leg(env):
    """the throughput leg keeps the discipline too: rank 1 fails while pinning its replica, every rank skips the leg, the rest of the
    line is intact"""
    r, secs = _two_ranks(env, 29791, {"GA_BENCH_FAIL_RANK": "1", "GA_BENCH_FAIL_AT": "replica_pin", "GA_BENCH_COLLECTIVE_TIMEOUT_S": "240", "GA_BENCH_CONFIG4": "0",
                                      "GA_BENCH_BOTH_PARTITIONS": "0"}, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert secs < 200
    d = _line(r.stdout)
    assert d["value_checked"] is True and d["groth16"]["matches_dlog"] is True
    assert "skipped on every rank" in d["replicas"]["error"] and "injected fault" in d["replicas"]["error"]


def test_counter_passes_are_not_nested_under_a_profiler(monkeypatch):
    """bench.py measures `roofline.traffic` with child rocprofv3 passes -- but not when it is itself being profiled (the outer tool's
    library is inherited by children; --pmc must never meet another tracing mode): it says so and the committed figure is used"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for var, val in (("ROCPROF_OUTPUT_PATH", "/tmp/x"), ("ROCP_TOOL_LIBRARIES", "librocprofiler-sdk-tool.so"), ("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")):
        monkeypatch.setenv(var, val)
        traffic, source, detail = bench.pmc_traffic_live(None)
        assert traffic is None and "under a profiler" in source and detail == {}
        monkeypatch.delenv(var)
