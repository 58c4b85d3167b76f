#!/bin/bash
# round 5, GPU call S: the first sort level as a resident grid that loads the next tile's scalars before writing the current one (new; GA_MSM_P1_GRID
# 256 / 512 / 1024 / 2048) against the one-tile-per-block build (prev), same box
OUT=gpurun_out; mkdir -p $OUT; TAG=${TAG:-r05_s}
GA_LIB_PATH=$PWD/gnark_amd/variants/libgnark_amd_prev.so timeout 600 python tools/exp/sort_ab.py --log-n 24 --modes 3 --library 0 >> $OUT/${TAG}_sort_pipelined_ab.txt 2>> $OUT/${TAG}.err
for g in 256 512 1024 2048 1000000; do
  GA_MSM_P1_GRID=$g timeout 600 python tools/exp/sort_ab.py --log-n 24 --modes 3 --library 0 --extra-env GA_MSM_P1_GRID=$g | sed "s/\"lib\"/\"p1_grid\": $g, \"lib\"/" >> $OUT/${TAG}_sort_pipelined_ab.txt 2>> $OUT/${TAG}.err
done
GA_LIB_PATH=$PWD/gnark_amd/variants/libgnark_amd_prev.so timeout 600 python tools/exp/sort_ab.py --log-n 24 --modes 3 --library 0 >> $OUT/${TAG}_sort_pipelined_ab.txt 2>> $OUT/${TAG}.err
cut -c1-330 $OUT/${TAG}_sort_pipelined_ab.txt
