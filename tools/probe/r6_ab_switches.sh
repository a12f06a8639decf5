#!/bin/bash
# Runs ON THE GPU BOX: same-box A/B of the round-6 Encodec switches (C3, tools/codecbench.py, alternating)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6ab2; mkdir -p $OUT
for rep in 1 2 3; do
  for v in "default" "NC_NO_DOWN5=1" "NC_NO_DOWN4=1" "NC_NO_UP4=1" "NC_NO_UP2=1" "NC_NO_DOWN2=1" "NC_NO_RES_A=1" "NC_NO_RES_A=1 NC_NO_DOWN2=1 NC_NO_DOWN4=1 NC_NO_DOWN5=1 NC_NO_UP2=1 NC_NO_UP4=1 NC_RMS_TWO_PASS=1 NC_SMALL_K1_COLS=4096"; do
    ms=$(env $( [ "$v" = default ] || echo $v ) python tools/codecbench.py --only encodec48 --steps 30 --warmup 5 2>/dev/null | grep -o '"ms": [0-9.]*' | head -1)
    echo "$rep | $v | $ms" | tee -a $OUT/ab.txt
  done
done
