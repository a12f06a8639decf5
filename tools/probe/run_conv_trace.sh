#!/bin/bash
# Runs ON THE GPU BOX: in-kernel phase traces (build_abl/lib_trace*.so, -DNC_CONV_TRACE) of one k = 7 layer and one two-tap layer inside the
# DAC step, with and without the XV staging; prints tools/probe/conv_trace.py's tables.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/conv_trace.txt; : > $OUT
for lib in build_abl/lib_trace*.so; do
  for sel in "7,384,1" "2,384,1" "2,192,1"; do
    for xv in 0 1; do
      [ $xv = 0 ] && export NC_NO_XV=1 || unset NC_NO_XV
      f=gpurun_out/trace_$(basename $lib .so)_${sel//,/_}_xv$xv.bin
      NC_MI355X_LIB=$PWD/$lib NC_CONV_TRACE_FILE=$f NC_CONV_TRACE_SEL=$sel python bench.py --no-cpu-baseline --no-extra --no-check --steps 3 --warmup 2 > /dev/null 2>&1
      echo "== $lib sel=$sel xv=$xv" | tee -a $OUT
      [ -f $f ] && python tools/probe/conv_trace.py $f | tee -a $OUT
    done
  done
done
