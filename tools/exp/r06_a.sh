#!/bin/bash
# round 6, batch A: the new paths on the device (batched witness tables, lane-2 fallback, table layout), then the same-box A/B of
# GA_G16_BATCH_TABLES on one pinned 2^24 BN254 key, then the bench line (N = 1) with the new CPU baseline and one-shot figure
export TAG=r06_a
tools/gpu.sh "tests:batched_witness or second_caller or synthetic_2_10 or cubic_bytes or two_callers or known_dlogs or msm_2_14" g16ab:batch:GA_G16_BATCH_TABLES=1,0 bench:--no-pmc
