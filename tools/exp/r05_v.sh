#!/bin/bash
# round 5, batch V: `python bench.py` (the driver's command, counter passes off: no rocprofv3 inside rocprofv3) under
# rocprofv3 --kernel-trace --stats -- every kernel of every leg (headline, Groth16 on both curves, PLONK, BLS12-381 MSMs) in one table
export TMPDIR=/tmp
OUT=gpurun_out
d=$OUT/stats_tmp_v
rm -rf $d
(time timeout 900 rocprofv3 --kernel-trace --stats -d $d -o k -- python bench.py --no-pmc --no-selftest --detail-file $OUT/r05_v_bench_under_rocprof_detail.json) > $OUT/r05_v_stats.log 2>&1
python tools/prof_summary.py $d/k_results.db > $OUT/r05_v_kernel_stats_full_bench.txt
ls -la $d | head -5
rm -rf $d
head -40 $OUT/r05_v_kernel_stats_full_bench.txt | cut -c1-210
grep '^{' $OUT/r05_v_stats.log | tail -1 > $OUT/r05_v_bench_under_rocprof.json
python tools/bench_digest.py $OUT/r05_v_bench_under_rocprof.json | head -30
tail -4 $OUT/r05_v_stats.log
