#!/usr/bin/env python3
"""Per-kernel issue accounting from the SQ counter passes collected by tools/gpu.sh (step sq) (text summaries of
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
        m = re.match(r"\s*(\d+)\s+((?:SQC?|TA|TCP)_\w+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)", line)
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


    # where the parked and stalled cycles go (tools/gpu.sh sq: passes 3-6), per kernel: LDS, vector memory, instruction cache
    extra = ("SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VMEM_RD", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_VMEM",
             "SQ_INST_LEVEL_VMEM", "SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQ_IFETCH", "TA_TA_BUSY_sum", "TCP_PENDING_STALL_CYCLES_sum",
             "TCP_TCP_TA_DATA_STALL_CYCLES_sum")
    if any(c in v for v in vals.values() for c in extra):
        print()
        print("per kernel, as a share of SQ_WAVE_CYCLES (quad-cycles) unless noted:")
        for k, v in sorted(vals.items(), key=lambda kv: -kv[1].get("us", 0)):
            wc = v.get("SQ_WAVE_CYCLES")
            if not wc:
                continue
            pct = lambda c: ("%5.1f%%" % (100 * v[c] / wc)) if c in v else "   n/a"
            lat = v["SQ_INST_LEVEL_VMEM"] / v["SQ_INSTS_VMEM_RD"] if v.get("SQ_INSTS_VMEM_RD") and "SQ_INST_LEVEL_VMEM" in v else None
            miss = 100 * v["SQC_ICACHE_MISSES"] / v["SQC_ICACHE_REQ"] if v.get("SQC_ICACHE_REQ") and "SQC_ICACHE_MISSES" in v else None
            print("%-52s LDS: wait-issue %s active %s bank-conflict %s | VMEM: active %s inst-cycles %s avg in flight per read %s | I$: req %s miss %s | TA busy %s TCP pending-stall %s"
                  % (k[:52], pct("SQ_WAIT_INST_LDS"), pct("SQ_ACTIVE_INST_LDS"), pct("SQ_LDS_BANK_CONFLICT"), pct("SQ_ACTIVE_INST_VMEM"), pct("SQ_INST_CYCLES_VMEM"),
                     "n/a" if lat is None else "%.0f quad-cycles" % lat, "%.3g" % v["SQC_ICACHE_REQ"] if "SQC_ICACHE_REQ" in v else "n/a",
                     "n/a" if miss is None else "%.1f%%" % miss, "%.3g" % v["TA_TA_BUSY_sum"] if "TA_TA_BUSY_sum" in v else "n/a",
                     "%.3g" % v["TCP_PENDING_STALL_CYCLES_sum"] if "TCP_PENDING_STALL_CYCLES_sum" in v else "n/a"))


if __name__ == "__main__":
    main(sys.argv[1])
