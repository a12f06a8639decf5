#!/bin/bash
# Runs ON THE GPU BOX: C1 (SNAC 24 kHz, one 1 s clip) with the one-launch residual units forced on / off
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "default" "NC_SNAC_FUSE_MIN_COLS=0" "NC_SNAC_NO_FUSE=1"; do
  ms=$(env $( [ "$v" = default ] || echo $v ) python tools/codecbench.py --only snac24 --steps 200 --warmup 20 2>/dev/null | grep -o '"ms": [0-9.]*' | head -1)
  echo "$rep | $v | $ms"
done; done
