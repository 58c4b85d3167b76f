#!/bin/bash
# round 5, GPU call B: from which size does the fused (MSD + XCD) front end beat the library sort?  raw and table MSMs, 2^12 .. 2^22
OUT=gpurun_out; mkdir -p $OUT; TAG=${TAG:-r05_b}
for ln in 12 14 16 18 20 22; do
  timeout 600 python tools/exp/sort_ab.py --log-n $ln --modes 7,1 --shapes raw,table --reps 20 >> $OUT/${TAG}_fuse_min_sweep.txt 2>> $OUT/${TAG}_sweep.err
done
cut -c1-330 $OUT/${TAG}_fuse_min_sweep.txt
