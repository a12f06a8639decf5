#!/bin/bash
# Runs ON THE GPU BOX: step 1 of tools/profile_round.sh alone (rocprofv3 kernel trace + --stats of the headline bench command) and the
# SQ PMC pass of the same command, for a re-check of the headline class after a kernel change.
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_${TAG}_dac
mkdir -p $OUT
BENCH="python3 $R/bench.py --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats -d $OUT/trace_dac -o p -- $BENCH --steps 3 --warmup 1 > $OUT/trace_dac.log 2>&1
python3 $R/tools/rocpd_summary.py $(find $OUT/trace_dac -name 'p_results.db' | head -1) > $R/gpurun_out/${TAG}_dac_b32.kernel_stats.txt
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  n=${pass%%:*}; c=${pass#*:}
  export NC_LAUNCH_LOG=$OUT/launch_dac_$n.log
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_dac_$n -o p --output-format csv -- $BENCH --steps 2 --warmup 1 > $OUT/pmc_dac_$n.log 2>&1
  unset NC_LAUNCH_LOG
done
cd $R
python tools/pmc_classes.py --key dac44k --out gpurun_out/traffic_${TAG}_dac.json \
    fetch=$(find $OUT/pmc_dac_fetch -name '*counter_collection.csv' | head -1):$OUT/launch_dac_fetch.log \
    write=$(find $OUT/pmc_dac_write -name '*counter_collection.csv' | head -1):$OUT/launch_dac_write.log \
    sq=$(find $OUT/pmc_dac_sq -name '*counter_collection.csv' | head -1):$OUT/launch_dac_sq.log 2>&1 | grep conv_k7
head -12 gpurun_out/${TAG}_dac_b32.kernel_stats.txt | cut -c1-200
rm -rf $OUT
