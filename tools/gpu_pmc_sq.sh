#!/bin/bash
# SQ counters for one kernel family (default: the window reduction) -- separate passes, kernel-trace only
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_SALU" "SQ_IFETCH SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d gpurun_out/prof_sq$i -o sq -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --groth16-proofs 0 > gpurun_out/prof_sq$i.log 2>&1
  python tools/prof_summary.py --pmc gpurun_out/prof_sq$i/sq_results.db 2>/dev/null | grep -E "reduce_groups|accumulate29_kernel|counter" | cut -c1-170
done
