#!/bin/bash
# round 5, GPU call P: 16-bit key parts between the two sort levels (new) against the closing-run build (prev), same box, two rounds
OUT=gpurun_out; mkdir -p $OUT; TAG=${TAG:-r05_p}
for rep in 1 2; do
  GA_LIB_PATH=$PWD/gnark_amd/variants/libgnark_amd_prev.so timeout 600 python tools/exp/sort_ab.py --log-n 24 --modes 3 --library 0 >> $OUT/${TAG}_sort_u16_keys_ab.txt 2>> $OUT/${TAG}.err
  timeout 600 python tools/exp/sort_ab.py --log-n 24 --modes 3 --library 0 >> $OUT/${TAG}_sort_u16_keys_ab.txt 2>> $OUT/${TAG}.err
done
cut -c1-420 $OUT/${TAG}_sort_u16_keys_ab.txt
