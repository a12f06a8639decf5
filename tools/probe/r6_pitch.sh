#!/bin/bash
# Runs ON THE GPU BOX: Encodec parity tests, then C3 with / without the aligned trimmed views of the transposed convolutions (alternating)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_encodec_gpu.py tests/test_baseline_sizes_gpu.py::test_c3_encodec48k_batch16x2s_vs_oracle_and_batch_invariance tests/test_baseline_sizes_gpu.py::test_encodec_long_clip_more_than_16_segments -m gpu -q 2>&1 | tail -3
for rep in 1 2 3; do for v in "default" "NC_NO_UP_PITCH=1" "NC_LSTM_CHUNKS=6"; do
  ms=$(env $( [ "$v" = default ] || echo $v ) python tools/codecbench.py --only encodec48 --steps 30 --warmup 5 2>/dev/null | grep -o '"ms": [0-9.]*' | head -1)
  echo "$rep | $v | $ms"
done; done
