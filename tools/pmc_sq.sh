#!/bin/bash
# Runs ON THE GPU BOX: SQ occupancy / wait / matrix-core-busy counters of the k=7 layer shapes (separate --pmc passes, kernel trace only).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_WAIT_ANY"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_sq$i -o p --output-format csv -- python3 $R/tools/convbench.py --iters 2 --filter "k7 C" > $R/gpurun_out/pmc_sq$i.log 2>&1
done
ls $R/gpurun_out | grep pmc_sq
