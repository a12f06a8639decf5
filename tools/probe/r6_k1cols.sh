#!/bin/bash
# Runs ON THE GPU BOX: the chunked LSTM input-projection GEMMs on the short-row kernel (default) vs the pointwise kernel (NC_SMALL_K1_COLS below the chunk width)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 4096 1024 0; do
  ms=$(NC_SMALL_K1_COLS=$v python tools/codecbench.py --only encodec48 --steps 30 --warmup 5 2>/dev/null | grep -o '"ms": [0-9.]*' | head -1)
  echo "$rep | NC_SMALL_K1_COLS=$v | $ms"
done; done
