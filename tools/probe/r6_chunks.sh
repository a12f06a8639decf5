#!/bin/bash
# Runs ON THE GPU BOX: LSTM chunk count x pointwise-GEMM routing (C3, alternating)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "4 1024" "6 1024" "6 512" "8 1024" "8 512" "10 512" "12 512"; do
  set -- $v
  ms=$(NC_LSTM_CHUNKS=$1 NC_SMALL_K1_COLS=$2 python tools/codecbench.py --only encodec48 --steps 30 --warmup 5 2>/dev/null | grep -o '"ms": [0-9.]*' | head -1)
  echo "$rep | NC_LSTM_CHUNKS=$1 NC_SMALL_K1_COLS=$2 | $ms"
done; done
