#!/usr/bin/env python3
"""One screen of what a bench.py JSON line says (tools/gpu.sh prints it after every bench step): the compact line's "summary" object
and its size -- the driver keeps the last 8 KB of stdout."""
import json
import sys


def main(path):
    d, raw = None, ""
    for line in open(path):
        if line.startswith("{"):
            d, raw = json.loads(line), line
    if d is None:
        print("no JSON line in", path)
        return
    g = lambda o, *ks: {k: o.get(k) for k in ks if isinstance(o, dict) and k in o}
    print("line: %d bytes; last key: %s" % (len(raw), list(d)[-1]))
    print("headline", g(d, "value", "value_checked", "ms_per_step", "scaling", "n_gpus", "error"))
    print("roofline", g(d.get("roofline", {}), "frac", "avg_launch_ms", "traffic", "traffic_over_algorithmic", "int_mad_frac", "bound_actual"))
    print("  traffic_source:", (d.get("roofline", {}) or {}).get("traffic_source"))
    print("  stages", d.get("stages_ms"))
    for k, v in (d.get("summary") or {}).items():
        print("  %-44s %s" % (k, v))
    for key in ("groth16", "groth16_window", "groth16_range", "groth16_bls12_381", "replicas", "weak_msm", "plonk"):
        q = d.get(key)
        if isinstance(q, dict) and (q.get("stages_ms_per_proof") or q.get("stages_ms")):
            print(key, "stages", q.get("stages_ms_per_proof") or q.get("stages_ms"))
        if isinstance(q, dict) and "error" in q:
            print(key, "ERROR", q["error"])
    for k in ("backend", "world_size", "ranks", "nccl_selftest", "pmc_passes"):
        if k in d:
            print(k, d[k])
    print("cpu_baseline", g(d.get("cpu_baseline", {}), "value", "cores", "kind", "error"), "legs_seconds", d.get("legs_seconds"), "total", d.get("total_seconds"))


if __name__ == "__main__":
    main(sys.argv[1])
