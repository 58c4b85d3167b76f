#!/bin/bash
# round 5, GPU call E: the 2^20 raw MSM (BASELINE config 2) over the reduction-group and task-length knobs
OUT=gpurun_out; mkdir -p $OUT; TAG=${TAG:-r05_e}
for kv in "-" "GA_MSM_GROUP=16" "GA_MSM_GROUP=8" "GA_MSM_GROUP=4" "GA_MSM_MIN_SEG=64" "GA_MSM_MIN_SEG=128" "GA_MSM_MIN_SEG=64,GA_MSM_GROUP=8"; do
  envs=""; [ "$kv" != "-" ] && envs=${kv//,/ }
  echo "{\"env\": \"$kv\"}" >> $OUT/${TAG}_msm_2p20_raw_knobs.txt
  env $envs timeout 300 python tools/msm_small_trace.py --log-n 20 --reps 40 --mode raw >> $OUT/${TAG}_msm_2p20_raw_knobs.txt 2>> $OUT/${TAG}_small.err
done
cat $OUT/${TAG}_msm_2p20_raw_knobs.txt
