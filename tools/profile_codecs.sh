#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel traces of the non-headline configs (tools/codecbench.py: C3 Encodec 48 kHz 16 x 2 s,
# C5 share SNAC 44.1 kHz 8 x 5 s, C1 SNAC 24 kHz 1 x 1 s).  Summaries are written beside the databases; copy them to profiles/.
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
for which in encodec snac; do
    rocprofv3 --kernel-trace --stats -d $OUT/$which -o p -- python3 $R/tools/codecbench.py --only $which --steps 3 --warmup 1 > $OUT/$which.log 2>&1
    db=$(find $OUT/$which -name 'p_results.db' | head -1)
    python3 $R/tools/rocpd_summary.py $db > $OUT/${TAG}_${which}_codecbench.kernel_stats.txt 2>> $OUT/$which.log
    tail -1 $OUT/$which.log | cut -c1-400
done
