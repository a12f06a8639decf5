cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_WAIT_ANY"; do
  tag=$(echo $set | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmcq_$tag -o p --output-format csv -- python3 $R/tools/convbench.py --iters 2 --filter "k7 C" > $R/gpurun_out/pmcq_$tag.log 2>&1
done
ls $R/gpurun_out | grep pmcq
