#!/bin/bash
# round 5, GPU call L: buckets per running-sum group of the window reduction on the 14-limb field (BLS12-381), where the lazy pass runs one wave per SIMD
OUT=gpurun_out; mkdir -p $OUT; TAG=${TAG:-r05_l}
for g in 32 16 8; do
  GA_MSM_GROUP=$g timeout 600 python tools/ab_kernels.py --parts msm --curve bls12-381 --reps 3 --tag group$g >> $OUT/${TAG}_bls_reduce_group.txt 2>> $OUT/${TAG}.err
done
for g in 32 16; do
  GA_MSM_GROUP=$g timeout 600 python tools/ab_kernels.py --parts msm --curve bn254 --reps 3 --tag group$g >> $OUT/${TAG}_bn_reduce_group.txt 2>> $OUT/${TAG}.err
done
cut -c1-700 $OUT/${TAG}_bls_reduce_group.txt $OUT/${TAG}_bn_reduce_group.txt
