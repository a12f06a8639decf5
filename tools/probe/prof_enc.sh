#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel trace of the C3 step (tools/codecbench.py --only encodec48), kernel table + timeline of the last step
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/encodec48 -o p -- python3 $R/tools/codecbench.py --only encodec48 --steps 4 --warmup 2 > $OUT/encodec48.log 2>&1
db=$(find $OUT/encodec48 -name 'p_results.db' | head -1)
python3 $R/tools/rocpd_summary.py $db > $OUT/${TAG}_encodec48.kernel_stats.txt 2>> $OUT/encodec48.log
python3 $R/tools/probe/timeline.py $db 6 > $OUT/${TAG}_encodec48.timeline.txt 2>> $OUT/encodec48.log
tail -1 $OUT/encodec48.log | cut -c1-300
tail -1 $OUT/${TAG}_encodec48.timeline.txt
rm -rf $OUT/encodec48
