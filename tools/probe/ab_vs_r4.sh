#!/bin/bash
# Runs ON THE GPU BOX (round 5): same-box A/B of this build against an OLDER build of the library (build_abl/lib_r4.so: check the old commit out
# into a worktree, `make -C neuralcodecs_amd/csrc`, copy its libnc_mi355x.so there; the Python loader skips exports the old library lacks when
# NC_MI355X_LIB is set) and against this build's own fallback switches: parity suites, then the headline class table and the other
# configurations' step times, three interleaved rounds.  Output: gpurun_out/ab_xvk.txt (profiles/r05_ab_xvk_vs_r4.txt is one such run).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_xvk.txt; : > $OUT
for setting in NC_DEFAULT=1 NC_XV_K7=1; do
  echo "== parity $setting" | tee -a $OUT
  env $setting timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_dac_gpu.py tests/test_snac_gpu.py -m gpu -x -q 2>&1 | tail -2 | tee -a $OUT
done
for rep in 1 2 3; do
  for setting in NC_MI355X_LIB=$PWD/build_abl/lib_r4.so NC_NO_XV=1 NC_DEFAULT=1 NC_XV_K7=1; do
    echo "== bench rep $rep $setting" | tee -a $OUT
    env $setting python bench.py --no-cpu-baseline --no-extra --no-check --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k: round(v['ms_per_step'],3) for k, v in d['roofline']['all_classes'].items()})" | tee -a $OUT
    env $setting python tools/codecbench.py --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ', {k: v['ms'] for k, v in d.items()})" | tee -a $OUT
  done
done
