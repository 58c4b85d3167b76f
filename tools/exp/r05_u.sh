#!/bin/bash
# round 5, batch U: the headline leg alone, un-profiled and under rocprofv3 --kernel-trace --stats, on ONE box -- every
# msm_accumulate29_kernel launch of the trace is a headline launch, so the trace's average can be compared with the bench line's
# hipEvent average directly (the closing run's r05_z_kernel_stats.txt mixes them with the proof's A / B1 / K / Z launches)
export TMPDIR=/tmp
OUT=gpurun_out
python bench.py --only-headline --steps 20 --warmup 3 > $OUT/r05_u_headline.json 2> $OUT/r05_u_headline.err
python tools/bench_digest.py $OUT/r05_u_headline.json | head -5
d=$OUT/stats_tmp_u
rm -rf $d
timeout 600 rocprofv3 --kernel-trace --stats -d $d -o k -- python bench.py --only-headline --steps 20 --warmup 3 > $OUT/r05_u_stats.log 2>&1
python tools/prof_summary.py $d/k_results.db > $OUT/r05_u_kernel_stats_headline_only.txt
rm -rf $d
grep -E "^#|calls|accumulate29|digits_pass1|p2_|reduce|merge|hist" $OUT/r05_u_kernel_stats_headline_only.txt | cut -c1-200
grep '^{' $OUT/r05_u_stats.log | tail -1 > $OUT/r05_u_headline_under_rocprof.json
python tools/bench_digest.py $OUT/r05_u_headline_under_rocprof.json | head -5
