#!/usr/bin/env python3
"""Refresh profiles/pmc_latest.json (what bench.py reads for `roofline.traffic`) from the text summaries of a FETCH_SIZE and a
WRITE_SIZE pass (tools/prof_summary.py --pmc; separate rocprofv3 passes with --kernel-trace only, as gpurun requires).

  python tools/pmc_to_json.py bn254 profiles/r03_g_bench24_bn254_pmc_FETCH_SIZE.txt profiles/r03_g_bench24_bn254_pmc_WRITE_SIZE.txt "r03_g ..."
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {   # json key -> (substring of the kernel name, extra json fields)
    "msm_accumulate_kernel": "msm_accumulate29_kernel<ga::Fe<",
    "msm_accumulate_kernel_g2": "msm_accumulate29_kernel<ga::Fe2<",
    "ntt_pass_kernel": "ntt_pass29r4_kernel<",          # natural -> bit-reversed (template argument false)
    "ntt_pass_kernel_to_natural": "ntt_pass29r4_kernel<",
}


def rows(path):
    out = []
    for line in open(path):
        m = re.match(r"\s*(\d+)\s+(\w+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)", line)
        if m:
            out.append((int(m.group(1)), m.group(2), float(m.group(4)), float(m.group(5)), m.group(6)))
    return out


def pick(rs, key):
    sub = KERNELS[key]
    for calls, _ctr, avg, us, name in rs:
        if sub not in name:
            continue
        if key == "ntt_pass_kernel" and "false>" not in name.split("(")[0]:
            continue
        if key == "ntt_pass_kernel_to_natural" and "true>" not in name.split("(")[0]:
            continue
        return calls, avg * 1024.0, us, name.split("(")[0].replace("void ga::", "").replace("ga::", "")
    return None


def main():
    curve, fetch_path, write_path, source = sys.argv[1:5]
    p = os.path.join(ROOT, "profiles", "pmc_latest.json")
    d = json.load(open(p))
    f, w = rows(fetch_path), rows(write_path)
    for key in KERNELS:
        a, b = pick(f, key), pick(w, key)
        if not a or not b:
            continue
        d["kernels"].setdefault(key, {}).setdefault(curve, {})["24"] = {
            "fetch_bytes": round(a[1], 1), "write_bytes": round(b[1], 1), "kernel": a[3], "launches": a[0], "avg_us_under_rocprof": round(a[2], 1)}
    d["source"] = (d.get("source", "") + " | " + source) if curve not in source.split()[0] else source
    json.dump(d, open(p, "w"), indent=1)
    for key in KERNELS:
        print(key, d["kernels"].get(key, {}).get(curve, {}).get("24"))


if __name__ == "__main__":
    main()
