#!/bin/bash
# Round-3 batch M: a 2^27-constraint BN254 proof on one GPU (key 48 GiB without window tables), checked by known dlogs
OUT=gpurun_out/r3m
mkdir -p $OUT
export TMPDIR=/tmp
free -g | head -2 > $OUT/mem.txt; cat /sys/fs/cgroup/memory.max >> $OUT/mem.txt 2>/dev/null
timeout 1200 python tools/size_sweep.py --curve bn254 --logs 27 --check-max 27 --proofs 2 > $OUT/sweep_bn254_27.jsonl 2> $OUT/sweep_bn254_27.err; echo "rc=$?" >> $OUT/sweep_bn254_27.err
cat $OUT/mem.txt $OUT/sweep_bn254_27.jsonl; tail -5 $OUT/sweep_bn254_27.err
