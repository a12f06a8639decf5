#!/bin/bash
# Runs ON THE GPU BOX (round 5): A/B of the vectorised window staging (XV) of the two-tap sub-pixel instances on one box.
#   1. parity: op tests + DAC / SNAC suites on the default build and on each build_abl/lib_xvs*.so
#   2. per-layer: DAC's and SNAC 44 kHz's up-convolutions, NC_NO_XV=1 vs default vs variants, 3 rounds (60 / 10 launches each, from idle:
#      compare within the table only)
#   3. the headline bench's class table + the other configurations' step times, interleaved, 2 rounds
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_xv.txt; : > $OUT
SNAC="8,1536,768,16,8,4,576,1 8,768,384,16,8,4,4608,1 8,384,192,6,3,2,36864,1 8,192,96,4,2,1,110592,1"
libs="default $(ls build_abl/lib_xvs*.so 2>/dev/null)"
for l in $libs; do
  [ $l = default ] && unset NC_MI355X_LIB || export NC_MI355X_LIB=$PWD/$l
  echo "== parity $l" | tee -a $OUT
  timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_dac_gpu.py tests/test_snac_gpu.py -m gpu -x -q 2>&1 | tail -2 | tee -a $OUT
done
unset NC_MI355X_LIB
for rep in 1 2 3; do
  echo "== layers rep $rep NC_NO_XV=1" | tee -a $OUT
  NC_NO_XV=1 python tools/convbench.py --filter dec.up --iters 60 2>&1 | grep dec.up | tee -a $OUT
  NC_NO_XV=1 python tools/probe/shapebench.py $SNAC 2>/dev/null | tee -a $OUT
  for l in $libs; do
    [ $l = default ] && unset NC_MI355X_LIB || export NC_MI355X_LIB=$PWD/$l
    echo "== layers rep $rep $l" | tee -a $OUT
    python tools/convbench.py --filter dec.up --iters 60 2>&1 | grep dec.up | tee -a $OUT
    python tools/probe/shapebench.py $SNAC 2>/dev/null | tee -a $OUT
  done
  unset NC_MI355X_LIB
done
for rep in 1 2; do
  for setting in NC_NO_XV=1 NC_DEFAULT=1 $(ls build_abl/lib_xvs*.so 2>/dev/null | sed "s|^|NC_MI355X_LIB=$PWD/|"); do
    echo "== bench rep $rep $setting" | tee -a $OUT
    env $setting python bench.py --no-cpu-baseline --no-extra --no-check --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k: round(v['ms_per_step'],3) for k, v in d['roofline']['all_classes'].items()})" | tee -a $OUT
    env $setting python tools/codecbench.py --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ', {k: v['ms'] for k, v in d.items()})" | tee -a $OUT
  done
done
