R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/r2e
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/fuse -o p -- python3 $R/tools/codecbench.py --only encodec48 --steps 3 --warmup 1 > $OUT/fuse.log 2>&1
python3 $R/tools/rocpd_summary.py $(find $OUT/fuse -name 'p_results.db' | head -1) > $OUT/fuse.kernel_stats.txt
export NC_ENCODEC_NO_FUSE=1
rocprofv3 --kernel-trace --stats -d $OUT/nofuse -o p -- python3 $R/tools/codecbench.py --only encodec48 --steps 3 --warmup 1 > $OUT/nofuse.log 2>&1
python3 $R/tools/rocpd_summary.py $(find $OUT/nofuse -name 'p_results.db' | head -1) > $OUT/nofuse.kernel_stats.txt
tail -1 $OUT/fuse.log | cut -c1-300; tail -1 $OUT/nofuse.log | cut -c1-300
