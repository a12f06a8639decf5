#!/bin/bash
# Runs ON THE GPU BOX: the 8-wavefront Euclidean RVQ workgroups (experiments library) against the shipped 4-wavefront form in the C3 step
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in default exp8; do
  OUT=$R/gpurun_out/rvq8_$v; rm -rf $OUT; mkdir -p $OUT
  if [ $v = exp8 ]; then export NC_MI355X_LIB=$R/neuralcodecs_amd/libnc_mi355x_exp.so NC_RVQ_8WAVES=1; fi
  rocprofv3 --kernel-trace --stats -d $OUT/t -o p -- python3 $R/tools/codecbench.py --only encodec48 --steps 6 --warmup 2 > $OUT/log 2>&1
  db=$(find $OUT/t -name 'p_results.db' | head -1)
  echo "== $v"; tail -1 $OUT/log | cut -c1-60; python3 $R/tools/rocpd_summary.py $db 2>/dev/null | grep "grid=.*euclid_rvq" | cut -c1-110
  rm -rf $OUT
done
