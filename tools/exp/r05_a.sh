#!/bin/bash
# round 5, GPU call A: same-box A/B of the sort front end (tools/exp/sort_ab.py) + the 2^20 raw MSM (BASELINE config 2) per setting
OUT=gpurun_out; mkdir -p $OUT; TAG=${TAG:-r05_a}
timeout 900 python tools/exp/sort_ab.py --log-n 24 > $OUT/${TAG}_sort_ab_2p24.txt 2> $OUT/${TAG}_sort_ab.err
tail -3 $OUT/${TAG}_sort_ab.err
for kv in "-" "GA_MSM_FUSE_MIN=0" "GA_MSM_FUSE_MIN=0,GA_MSM_SORT_MODE=1" "GA_MSM_FUSE_MIN=0,GA_MSM_SORT_MODE=7"; do
  envs=""; [ "$kv" != "-" ] && envs=${kv//,/ }
  echo "{\"env\": \"$kv\"}" >> $OUT/${TAG}_msm_2p20_raw.txt
  env $envs timeout 300 python tools/msm_small_trace.py --log-n 20 --reps 30 --mode raw >> $OUT/${TAG}_msm_2p20_raw.txt 2>> $OUT/${TAG}_small.err
done
cat $OUT/${TAG}_sort_ab_2p24.txt | cut -c1-400
cat $OUT/${TAG}_msm_2p20_raw.txt
