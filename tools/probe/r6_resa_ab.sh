#!/bin/bash
# Runs ON THE GPU BOX: res_a_kernel timing under the build_abl/lib_ra_*.so variants (timing probes: some compute wrong results on purpose)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6ab; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in default $(ls $R/build_abl/lib_ra_*.so); do
  n=$(basename $lib .so)
  if [ $lib = default ]; then unset NC_MI355X_LIB; else export NC_MI355X_LIB=$lib; fi
  rm -rf $OUT/tr
  rocprofv3 --kernel-trace --stats -d $OUT/tr -o p -- python3 $R/tools/codecbench.py --only encodec48 --steps 4 --warmup 2 > $OUT/$n.log 2>&1
  db=$(find $OUT/tr -name 'p_results.db' | head -1)
  echo "== $n $(tail -2 $OUT/$n.log | grep -o '"ms": [0-9.]*' | head -1)"
  python3 $R/tools/rocpd_summary.py $db | grep "grid=.*res_a\|grid=3008.*conv1x1_kernel<2\|grid=6016.*conv1x1_kernel<1" | cut -c1-150
done
rm -rf $OUT/tr
