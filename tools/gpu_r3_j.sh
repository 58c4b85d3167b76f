#!/bin/bash
# Round-3 batch J: the PLONK leg (2^22) under rocprofv3: kernel stats + the two SQ passes for the PLONK-specific kernels
# (plonk_constraints / batch inversion / grand product / coset scaling), which had no counters yet
OUT=gpurun_out/r3j
mkdir -p $OUT
export TMPDIR=/tmp
PL="python tools/bench_plonk_kernels.py"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o k -- $PL > $OUT/plonk_stats.log 2>&1
python tools/prof_summary.py $OUT/stats/k_results.db > $OUT/r03_j_plonk22_kernel_stats.txt 2>/dev/null
rm -rf $OUT/stats
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d $OUT/sq$i -o sq -- $PL > $OUT/sq$i.log 2>&1
  python tools/prof_summary.py --pmc $OUT/sq$i/sq_results.db 2>/dev/null | grep -E "counter|plonk_|fr_batch|fr_chunk|kzg_scale|fr_lincomb|ntt_pass29r4|accumulate29_kernel" | cut -c1-220 >> $OUT/r03_j_plonk_sq_counters.txt
  rm -rf $OUT/sq$i
done
python tools/sq_summary.py $OUT/r03_j_plonk_sq_counters.txt > $OUT/r03_j_plonk_sq_summary.txt 2>&1; cat $OUT/r03_j_plonk_sq_summary.txt
head -30 $OUT/r03_j_plonk22_kernel_stats.txt | cut -c1-200
# small proofs: wall time against the sum of the device stages, and a kernel trace of 2^16 proofs
timeout 300 python tools/small_profile.py --logs 12,14,16,18,20 > $OUT/small_profile.jsonl 2> $OUT/small_profile.err; tail -c 300 $OUT/small_profile.err
cat $OUT/small_profile.jsonl | cut -c1-1200
