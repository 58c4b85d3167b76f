#!/bin/bash
# round 6, batch F: where do the 230-240 ms of pinning a 2^24 BN254 key as plain vectors go (6 GiB over PCIe = 119 ms at the 54 GB/s the
# W upload reaches)?  GA_TRACE_PIN=1 prints the milestones of ga_g16_pk_create; then pin / prove / prove / free, three rounds
export TAG=r06_f
OUT=gpurun_out
GA_TRACE_PIN=1 python tools/exp/one_shot_breakdown.py 24 > $OUT/r06_f_one_shot.jsonl 2> $OUT/r06_f_one_shot.err
cat $OUT/r06_f_one_shot.jsonl; grep "\[pin\]" $OUT/r06_f_one_shot.err | tail -40
