#!/bin/bash
# Round-3 batch K: which kernels a SMALL MSM (2^16, 2^20; raw and table) spends its time in
OUT=gpurun_out/r3k
mkdir -p $OUT
export TMPDIR=/tmp
for L in 16 20; do
  for M in raw table; do
    timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/st -o k -- python tools/msm_small_trace.py --log-n $L --reps 20 --mode $M > $OUT/msm_${L}_${M}.json 2> $OUT/msm_${L}_${M}.err
    python tools/prof_summary.py $OUT/st/k_results.db 2>/dev/null | grep -v "gen_bases\|msm_table29" | head -34 | cut -c1-60,90-200 > $OUT/msm_${L}_${M}_kernels.txt
    rm -rf $OUT/st
    tail -1 $OUT/msm_${L}_${M}.json | cut -c1-700
    cat $OUT/msm_${L}_${M}_kernels.txt
  done
done
