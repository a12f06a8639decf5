#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel trace of the C5 share (SNAC 44.1 kHz, 8 x 5 s), kernel table + timeline of the last step
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/snac44 -o p -- python3 $R/tools/codecbench.py --only snac44 --steps 4 --warmup 2 > $OUT/snac44.log 2>&1
db=$(find $OUT/snac44 -name 'p_results.db' | head -1)
python3 $R/tools/rocpd_summary.py $db > $OUT/${TAG}_snac44.kernel_stats.txt 2>> $OUT/snac44.log
python3 $R/tools/probe/timeline.py $db 6 > $OUT/${TAG}_snac44.timeline.txt 2>> $OUT/snac44.log
tail -1 $OUT/snac44.log | cut -c1-300
tail -1 $OUT/${TAG}_snac44.timeline.txt
rm -rf $OUT/snac44
