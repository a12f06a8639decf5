R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/r2s
mkdir -p $OUT
for v in small nosmall; do
  if [ $v = nosmall ]; then export NC_NO_SMALL=1; fi
  python3 $R/tools/codecbench.py --only encodec48 --steps 10 --warmup 3 | tail -1
  rocprofv3 --kernel-trace --stats -d $OUT/$v -o p -- python3 $R/tools/codecbench.py --only encodec48 --steps 3 --warmup 1 > $OUT/$v.log 2>&1
  python3 $R/tools/rocpd_summary.py $(find $OUT/$v -name 'p_results.db' | head -1) > $OUT/$v.kernel_stats.txt
done
