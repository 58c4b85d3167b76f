#!/usr/bin/env python3
"""One screen of what a bench.py JSON line says (tools/gpu.sh prints it after every bench step)."""
import json
import sys


def main(path):
    d = None
    for line in open(path):
        if line.startswith("{"):
            d = json.loads(line)
    if d is None:
        print("no JSON line in", path)
        return
    g = lambda o, *ks: {k: o.get(k) for k in ks if isinstance(o, dict) and k in o}
    print("headline", g(d, "value", "value_checked", "ms_per_step", "scaling", "n_gpus", "error"), "roofline", g(d.get("roofline", {}), "frac", "avg_launch_ms", "traffic"))
    print("  stages", {k: v["avg_ms"] for k, v in (d.get("stages_ms") or {}).items()})
    for key in ("groth16", "groth16_bls12_381"):
        q = d.get(key)
        if q:
            print(key, g(q, "ms_per_proof", "proofs_per_s", "matches_dlog", "computeH_ms", "proof_sha", "error", "key_setup_s", "key_pin_s"),
                  "pipelined", g(q.get("pipelined") or {}, "ms_per_proof", "vs_single_caller", "same_proof_bytes", "lanes"))
            if q.get("stages_ms"):
                print("  stages", {k: v["total_ms"] for k, v in q["stages_ms"].items()})
    if d.get("msm_bls12_381"):
        for k, v in d["msm_bls12_381"].items():
            print("msm_bls12_381", k, g(v, "ms_per_msm", "Mscalar_mul_per_s", "value_checked", "error"), "roofline", g(v.get("roofline", {}), "frac", "avg_launch_ms"),
                  "mad", (v.get("roofline", {}).get("integer_multiplier") or {}).get("frac"))
    p = d.get("plonk")
    if p:
        print("plonk", g(p, "ms_per_proof_kernels", "msm_ms", "ntt_ms", "identity_ok", "error"), "roofline", g(p.get("roofline", {}), "frac"), "stages", p.get("stages_ms"))
    for k in ("plain_msm_no_tables", "msm_with_scalar_h2d", "weak_msm", "nccl_selftest", "config2_msm_2p20_unpinned"):
        if k in d:
            v = d[k]
            print(k, v if not isinstance(v, dict) else {a: b for a, b in v.items() if a not in ("how", "roofline", "stages_ms")})
    if d.get("nccl_selftest") != "ok" and "nccl_selftest_detail" in d:
        print("  selftest detail", d["nccl_selftest_detail"])
    print("cpu_baseline", g(d.get("cpu_baseline", {}), "value", "cores", "kind", "error"), "legs_seconds", d.get("legs_seconds"), "total", d.get("total_seconds"))


if __name__ == "__main__":
    main(sys.argv[1])
