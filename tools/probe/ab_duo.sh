#!/bin/bash
# Runs ON THE GPU BOX (round 5; needs NC_MI355X_LIB=<the EXPERIMENTS=1 library> since the DUO instances moved there): parity suites under
# NC_DUO=1, then the headline class table and two k = 7 layers (tools/convbench.py), default against NC_DUO=1, three interleaved rounds.
# Output: gpurun_out/ab_duo.txt (profiles/r05_ab_duo.txt is the run that rejected the form).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_duo.txt; : > $OUT
echo "== parity NC_DUO=1" | tee -a $OUT
NC_DUO=1 timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_dac_gpu.py tests/test_snac_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee -a $OUT
for rep in 1 2 3; do
  for setting in NC_DEFAULT=1 NC_DUO=1; do
    echo "== bench rep $rep $setting" | tee -a $OUT
    env $setting python bench.py --no-cpu-baseline --no-extra --no-check --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k: round(v['ms_per_step'],3) for k, v in d['roofline']['all_classes'].items()})" | tee -a $OUT
  done
  for setting in NC_DEFAULT=1 NC_DUO=1; do
    echo "== layers rep $rep $setting" | tee -a $OUT
    env $setting python tools/convbench.py --filter "dec.k7 C384" --iters 60 2>&1 | grep "k7" | tee -a $OUT
    env $setting python tools/convbench.py --filter "dec.k7 C768" --iters 60 2>&1 | grep "k7" | tee -a $OUT
  done
done
