#!/bin/bash
# Runs ON THE GPU BOX: the LDS h tile of the persistent LSTM (default) against the per-wave operand fetch (NC_LSTM_NO_HTILE=1), alternated
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_encodec_gpu.py tests/test_baseline_sizes_gpu.py::test_c3_encodec48k_batch16x2s_vs_oracle_and_batch_invariance -m gpu -q -x 2>&1 | tail -2
for rep in 1 2 3; do
  for v in 0 1; do
    if [ $v = 1 ]; then export NC_LSTM_NO_HTILE=1; else unset NC_LSTM_NO_HTILE; fi
    echo "no_htile=$v $(python tools/codecbench.py --only encodec --steps 30 --warmup 5 --classes 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print({k:(v['ms'], v['classes']['lstm']['ms']) for k,v in d.items()})")"
  done
done
