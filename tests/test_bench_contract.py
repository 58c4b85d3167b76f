"""bench.py's contract, exercised against the CPU emulation build (GA_BENCH_EMU=1: tiny sizes, the JSON line says "data": "emulation"
-- a dry run of the script's control flow, never a measurement): one JSON line with the fields the driver reads, the self-checks of
the Groth16 / PLONK legs, and the N > 1 path (two gloo ranks: weak-scaling MSM value + the strong-scaling sharded proof)."""
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
                        "--plonk-log-n", "6"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["data"] == "emulation" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5 and "integer_multiplier" in rf
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["gpu_result_matches_oracle"] is True and cb["host_cores"] >= cb["threads_used"] >= 1
    assert cb["groth16"]["gpu_proof_matches_oracle"] is True
    g = d["groth16"]
    assert g["matches_dlog"] is True and g["check"]["h_identity_ok"] is True and g["proofs"] == 2
    assert g["pipelined"]["same_proof_bytes"] is True and g["pipelined"]["host_threads"] == 2
    assert d["plonk"]["identity_ok"] is True
    assert d["msm_with_scalar_h2d"]["same_result"] is True and d["msm_with_scalar_h2d"]["ms_per_msm"] > 0


def test_two_ranks_line(env):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29761",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log-n", "9", "--steps", "1", "--warmup", "1", "--groth16-proofs", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["value_checked"] is True
    g = d["groth16"]
    assert "error" not in g, g
    assert g["scaling"] == "strong" and g["constraints"] == 512 and len(g["proof_sha"]) == 16
    assert g["matches_dlog"] is True and g["check"]["h_identity_ok"] is True   # rank 0 checked the sharded proof by the key's known dlogs
