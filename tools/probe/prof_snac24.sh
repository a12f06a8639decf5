#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel trace of the C1 step (SNAC 24 kHz, 1 x 1 s), kernel table + timeline of the last step
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/snac24 -o p -- python3 $R/tools/codecbench.py --only snac24 --steps 4 --warmup 2 > $OUT/snac24.log 2>&1
db=$(find $OUT/snac24 -name 'p_results.db' | head -1)
python3 $R/tools/rocpd_summary.py $db > $OUT/${TAG}_snac24.kernel_stats.txt 2>> $OUT/snac24.log
python3 $R/tools/probe/timeline.py $db 6 > $OUT/${TAG}_snac24.timeline.txt 2>> $OUT/snac24.log
tail -1 $OUT/snac24.log | cut -c1-300
tail -1 $OUT/${TAG}_snac24.timeline.txt
rm -rf $OUT/snac24
