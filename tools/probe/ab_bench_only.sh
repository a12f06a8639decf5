#!/bin/bash
# Runs ON THE GPU BOX: headline class table for the shipped library and every build_abl/lib_*.so, interleaved, 2 rounds (NO parity: for
# timing probes whose results are deliberately wrong).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_bench_only.txt; : > $OUT
for rep in 1 2; do
  for setting in NC_DEFAULT=1 "$@" $(ls build_abl/lib_*.so 2>/dev/null | sed "s|^|NC_MI355X_LIB=$PWD/|"); do
    echo "== bench rep $rep $setting" | tee -a $OUT
    env $setting python bench.py --no-cpu-baseline --no-extra --no-check --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k: round(v['ms_per_step'],3) for k, v in d['roofline']['all_classes'].items()})" | tee -a $OUT
  done
done
