# Runs ON THE GPU BOX: kernel trace of the C3 workload (tools/codecbench.py --only encodec48) -> gpurun_out/prof_enc/{enc.kernel_stats.txt,enc.timeline.txt}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_enc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/t -o p -- python3 $R/tools/codecbench.py --only encodec48 --steps 3 --warmup 1 > $OUT/t.log 2>&1
DB=$(find $OUT/t -name 'p_results.db' | head -1)
python3 $R/tools/rocpd_summary.py $DB > $OUT/enc.kernel_stats.txt
python3 $R/tools/probe/timeline.py $DB 4 > $OUT/enc.timeline.txt
tail -3 $OUT/enc.timeline.txt
rm -rf $OUT/t
