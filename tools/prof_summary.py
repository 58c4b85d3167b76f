#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (ROCm 7.2 default output) into the text tables kept under profiles/.

  python tools/prof_summary.py gpurun_out/prof_stats/bench24_results.db            # kernel stats (== --stats)
  python tools/prof_summary.py --pmc gpurun_out/prof_fetch/b22_results.db            # per-kernel PMC sums
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    m = re.search(r"radix_sort_onesweep_iteration|onesweep_histograms|lookback_scan|radix_sort", name)
    if name.startswith("void rocprim") and m:
        return "rocprim::" + m.group(0) + "<...>"
    return name if len(name) < 150 else name[:147] + "..."


def kernel_stats(path):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
                           "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name "
                           "order by sum(duration) desc"))
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats  ({path})")
    print(f"# total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'calls':>7} {'total_ms':>11} {'avg_us':>11} {'min_us':>10} {'max_us':>10} {'pct':>6} {'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'lds':>6} {'scratch':>8}  name")
    for n, c, s, a, mn, mx, vg, ag, sg, lds, sc in rows:
        print(f"{c:7d} {s / 1e6:11.3f} {a / 1e3:11.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} {100 * s / tot:6.2f} {vg:5d} {ag:5d} {sg:5d} {lds:6d} {sc:8d}  {short(n)}")


def pmc(path):
    db = sqlite3.connect(path)
    rows = list(db.execute("select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection "
                           "group by kernel_name, counter_name order by sum(value) desc"))
    print(f"# rocprofv3 --pmc  ({path}); FETCH_SIZE / WRITE_SIZE are in KiB per dispatch as reported (uncorrected)")
    print(f"{'calls':>7} {'counter':>12} {'sum':>16} {'avg/dispatch':>16} {'avg_us':>10}  name")
    for n, cn, c, s, a, d in rows:
        print(f"{c:7d} {cn:>12} {s:16.1f} {a:16.1f} {d / 1e3:10.2f}  {short(n)}")


if __name__ == "__main__":
    if sys.argv[1] == "--pmc":
        pmc(sys.argv[2])
    else:
        kernel_stats(sys.argv[1])
