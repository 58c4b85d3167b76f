#!/usr/bin/env python3
"""Per-kernel issue accounting from the SQ counter passes collected by tools/gpu_r2_profiles.sh (text summaries of
tools/prof_summary.py --pmc).  SQ_ACTIVE_INST_ANY + SQ_WAIT_INST_ANY + SQ_WAIT_ANY ~= SQ_WAVE_CYCLES (disjoint, quad-cycles,
/opt/skills/guides/MI355X_MICROARCH.md): a wave is issuing, stalled at the issue port, or parked on s_waitcnt / a barrier.
With N waves per SIMD a saturated VALU gives every wave 1/N of the issue slots.

  python tools/sq_summary.py profiles/r02_d_sq_counters.txt
"""
import collections
import re
import sys


def main(path):
    vals = collections.defaultdict(dict)
    for line in open(path):
        m = re.match(r"\s*(\d+)\s+(SQ_\w+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)", line)
        if not m:
            continue
        calls, ctr, _total, avg, us, name = m.groups()
        k = re.sub(r"^void ga::", "", name)
        k = re.sub(r"\(.*", "", k).replace("ga::", "")
        vals[k][ctr] = float(avg)
        vals[k]["us"] = float(us)
    print("%-52s %9s %8s %8s %8s %10s %12s" % ("kernel", "avg_us", "issue%", "stall%", "parked%", "waves", "VALU/wave"))
    for k, v in sorted(vals.items(), key=lambda kv: -kv[1].get("us", 0)):
        if not all(c in v for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVES")):
            continue
        tot = v["SQ_ACTIVE_INST_ANY"] + v["SQ_WAIT_INST_ANY"] + v["SQ_WAIT_ANY"]
        valu = v.get("SQ_INSTS_VALU", 0.0) / v["SQ_WAVES"]
        print("%-52s %9.1f %8.1f %8.1f %8.1f %10.0f %12.0f" % (k[:52], v["us"], 100 * v["SQ_ACTIVE_INST_ANY"] / tot, 100 * v["SQ_WAIT_INST_ANY"] / tot,
                                                          100 * v["SQ_WAIT_ANY"] / tot, v["SQ_WAVES"], valu))


if __name__ == "__main__":
    main(sys.argv[1])
