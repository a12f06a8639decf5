#!/bin/bash
# Runs ON THE GPU BOX: per-layer kernel times of the DAC step with the legacy k = 7 instances (NC_NO_XV_K7=1) beside the XV-only ones, from rocprofv3 kernel traces of the same bench command (tools/rocpd_summary.py tables; compare per (kernel, grid) rows).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_xvk7; mkdir -p $OUT
BENCH="python3 $R/bench.py --no-cpu-baseline --no-extra --no-check --steps 6 --warmup 2"
export NC_NO_XV_K7=1      # (the legacy instances)
rocprofv3 --kernel-trace --stats -d $OUT/base -o p -- $BENCH > $OUT/base.log 2>&1
python3 $R/tools/rocpd_summary.py $(find $OUT/base -name 'p_results.db' | head -1) > $R/gpurun_out/xvk7_base.kernel_stats.txt
unset NC_NO_XV_K7
rocprofv3 --kernel-trace --stats -d $OUT/xv -o p -- $BENCH > $OUT/xv.log 2>&1
python3 $R/tools/rocpd_summary.py $(find $OUT/xv -name 'p_results.db' | head -1) > $R/gpurun_out/xvk7_on.kernel_stats.txt
rm -rf $OUT
grep -A22 "per (kernel, grid, LDS)" $R/gpurun_out/xvk7_base.kernel_stats.txt | cut -c1-200
grep -A22 "per (kernel, grid, LDS)" $R/gpurun_out/xvk7_on.kernel_stats.txt | cut -c1-200
