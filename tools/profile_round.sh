#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the profiles a round commits under profiles/.
#   1. rocprofv3 kernel trace (+ --stats) of the headline bench command            -> <tag>_dac_b32.kernel_stats.txt
#   2. kernel traces of the other BASELINE configs (tools/codecbench.py)           -> <tag>_{encodec48,snac44,snac24}.kernel_stats.txt
#   3. PMC passes (kernel trace only, one counter set per pass: FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU)
#      of the bench command and of the C3 / C5-share commands, each with the engine's launch log (NC_LAUNCH_LOG) so that
#      tools/pmc_classes.py can attribute counters to exactly the launches a kernel class counts       -> traffic.json
# Outputs land under gpurun_out/prof_<tag>/; copy the summaries to profiles/ afterwards.
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python3 $R/bench.py --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats -d $OUT/trace_dac -o p -- $BENCH --steps 3 --warmup 1 > $OUT/trace_dac.log 2>&1
python3 $R/tools/rocpd_summary.py $(find $OUT/trace_dac -name 'p_results.db' | head -1) > $OUT/${TAG}_dac_b32.kernel_stats.txt
for cfg in encodec48 snac44 snac24; do
    rocprofv3 --kernel-trace --stats -d $OUT/trace_$cfg -o p -- python3 $R/tools/codecbench.py --only $cfg --steps 3 --warmup 1 > $OUT/trace_$cfg.log 2>&1
    python3 $R/tools/rocpd_summary.py $(find $OUT/trace_$cfg -name 'p_results.db' | head -1) > $OUT/${TAG}_$cfg.kernel_stats.txt
done
pmc_pass() {   # name, counters, command...
    local name=$1 ctr=$2; shift 2
    export NC_LAUNCH_LOG=$OUT/launch_$name.log
    rocprofv3 --pmc $ctr --kernel-trace -d $OUT/pmc_$name -o p --output-format csv -- "$@" > $OUT/pmc_$name.log 2>&1
    unset NC_LAUNCH_LOG
}
for wl in dac encodec48 snac44; do
    if [ $wl = dac ]; then CMD="$BENCH --steps 2 --warmup 1"; else CMD="python3 $R/tools/codecbench.py --only $wl --steps 2 --warmup 1"; fi
    pmc_pass ${wl}_fetch "FETCH_SIZE" $CMD
    pmc_pass ${wl}_write "WRITE_SIZE" $CMD
    pmc_pass ${wl}_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU" $CMD
done
rm -rf $OUT/trace_*/*/*.db.tmp
find $OUT -name '*.csv' | head -40
ls -la $OUT | head -40
tail -2 $OUT/trace_dac.log | cut -c1-300
