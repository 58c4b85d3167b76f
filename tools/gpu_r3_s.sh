#!/bin/bash
# Round-3 batch S: loading a 2^24-constraint proving key from a file (tmpfs) in gnark's three layouts
OUT=gpurun_out/r3s
mkdir -p $OUT
export TMPDIR=/tmp
df -h /dev/shm /tmp | tee $OUT/df.txt
timeout 1500 python tools/key_io_bench.py --log-n 24 --dir /dev/shm > $OUT/key_io_bn254_24.jsonl 2> $OUT/key_io.err; echo "rc=$?" >> $OUT/key_io.err
cat $OUT/key_io_bn254_24.jsonl; tail -3 $OUT/key_io.err
