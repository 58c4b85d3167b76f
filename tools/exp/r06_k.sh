#!/bin/bash
# round 6, batch K: pageable uploads take turns (ticket lock, 128 MiB chunks): one-shot timeline again, then the pinned proof with one and
# two callers (must not lose)
export TAG=r06_k
OUT=gpurun_out
bash tools/exp/r06_i.sh 2>&1 | sed -n '/^round/p;/round 2/,$p' | tail -24
tools/gpu.sh g16ab:turns:GA_G16_BATCH_TABLES=1,0
