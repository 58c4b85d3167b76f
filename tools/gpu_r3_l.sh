#!/bin/bash
# Round-3 batch L: why a 2^16 proof costs what it costs (kernel trace), sweep vs small_profile on the same box
OUT=gpurun_out/r3l
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/size_sweep.py --curve bn254 --logs 16,18,20 --proofs 20 > $OUT/sweep.jsonl 2> $OUT/sweep.err; cat $OUT/sweep.jsonl
timeout 300 python tools/size_sweep.py --curve bn254 --logs 16,18,20 --proofs 20 --precompute 1 > $OUT/sweep_p1.jsonl 2> $OUT/sweep_p1.err; cat $OUT/sweep_p1.jsonl
timeout 300 python tools/size_sweep.py --curve bn254 --logs 16,18,20 --proofs 20 --precompute -1 > $OUT/sweep_m1.jsonl 2> $OUT/sweep_m1.err; cat $OUT/sweep_m1.jsonl
timeout 300 python tools/small_profile.py --logs 16,20 > $OUT/small.jsonl 2> $OUT/small.err; cut -c1-300 $OUT/small.jsonl
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/st -o k -- python tools/size_sweep.py --curve bn254 --logs 16 --proofs 40 > $OUT/trace16.json 2> $OUT/trace16.err
python tools/prof_summary.py $OUT/st/k_results.db 2>/dev/null | grep -v "gen_bases\|msm_table29" | head -44 | cut -c1-60,90-200 > $OUT/trace16_kernels.txt
rm -rf $OUT/st
cat $OUT/trace16_kernels.txt
