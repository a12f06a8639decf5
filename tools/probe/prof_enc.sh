# Runs ON THE GPU BOX: kernel trace of the C3 workload (tools/codecbench.py --only encodec48) -> gpurun_out/prof_enc/enc.kernel_stats.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_enc
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/t -o p -- python3 $R/tools/codecbench.py --only encodec48 --steps 3 --warmup 1 > $OUT/t.log 2>&1
python3 $R/tools/rocpd_summary.py $(find $OUT/t -name 'p_results.db' | head -1) > $OUT/enc.kernel_stats.txt
