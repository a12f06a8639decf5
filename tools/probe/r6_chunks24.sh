#!/bin/bash
# Runs ON THE GPU BOX: LSTM chunk count on Encodec 24 kHz (16 x 2 s: one column tile) and 48 kHz, alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 4 6 8; do
  NC_LSTM_CHUNKS=$v python tools/codecbench.py --only encodec --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$rep chunks=$v', {k:v['ms'] for k,v in d.items()})"
done; done
