#!/bin/bash
# Runs ON THE GPU BOX: parity suites under each setting, then the headline class table + the other configurations' step times, three
# interleaved rounds, one box.   ab_env3.sh "ENV=a" "ENV=b" ...   -> gpurun_out/ab_env3.txt
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_env3.txt; : > $OUT
for setting in "$@"; do
  echo "== parity $setting" | tee -a $OUT
  env $setting timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_dac_gpu.py tests/test_snac_gpu.py -m gpu -x -q 2>&1 | tail -2 | tee -a $OUT
done
for rep in 1 2 3; do
  for setting in "$@"; do
    echo "== bench rep $rep $setting" | tee -a $OUT
    env $setting python bench.py --no-cpu-baseline --no-extra --no-check --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k: round(v['ms_per_step'],3) for k, v in d['roofline']['all_classes'].items()})" | tee -a $OUT
    env $setting python tools/codecbench.py --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ', {k: v['ms'] for k, v in d.items()})" | tee -a $OUT
  done
done
