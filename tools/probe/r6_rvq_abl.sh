#!/bin/bash
# Runs ON THE GPU BOX: kernel time of the Euclidean RVQ launch in the C3 step under ablation builds of the library (build_abl/lib_rvq_*.so).
# The builds came from tools/probe/mk_abl.sh rvq_<V> nc_encodec.hip -DRVQ_ABL_<V> with TEMPORARY #ifdef RVQ_ABL_NOMFMA / NOEPI / NOUPD / NOE2 blocks in
# euclid_rvq_mfma_kernel (skip the matrix-core instructions / the dist + argmin / the residual update / the |e|^2 chain); the blocks were removed again
# before the commit -- results of such builds are meaningless, only their timing was used (profiles/r06_ab_quantizers.txt).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "" NOMFMA NOEPI NOUPD NOE2; do
  OUT=$R/gpurun_out/rvq_abl_$v; rm -rf $OUT; mkdir -p $OUT
  if [ -n "$v" ]; then export NC_MI355X_LIB=$R/build_abl/lib_rvq_$v.so; else unset NC_MI355X_LIB; fi
  rocprofv3 --kernel-trace --stats -d $OUT/t -o p -- python3 $R/tools/codecbench.py --only encodec48 --steps 6 --warmup 2 > $OUT/log 2>&1
  db=$(find $OUT/t -name 'p_results.db' | head -1)
  echo "== ${v:-default}"; python3 $R/tools/rocpd_summary.py $db 2>/dev/null | grep "grid=.*euclid_rvq" | cut -c1-110
  rm -rf $OUT
done
